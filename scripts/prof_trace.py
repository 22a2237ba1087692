"""Small driver for rocprofv3 counter passes: config-1 primary + shadow rays, a few launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracing_amd as rt
from ray_tracing_amd import scenes

levels = int(sys.argv[1]) if len(sys.argv) > 1 else 9
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
desc = scenes.tetrahedron_scene(levels=levels)
desc.initial_medium = rt.api.MEDIUM_NONE
scene = rt.Scene(desc)
ctx = rt.Context(0)
ctx.upload(scene)
org, dr = scenes.pixel_center_rays(scene.flat.camera, 1920, 1080)
for _ in range(reps):
    h = ctx.trace_closest(org, dr)
    print("closest ms", ctx.last_kernel_ms())
hit = h["obj"] != 0xFFFFFFFF
p = (org[hit] + dr[hit] * h["t"][hit, None]).astype(np.float32)
for lp in ((100.0, 20.0, 100.0), (-3.0, 4.0, 5.0)):
    l = (np.array(lp, np.float32) - p).astype(np.float32)
    l = (l / np.sqrt((l * l).sum(1, dtype=np.float32))[:, None]).astype(np.float32)
    a = ctx.trace_any(p, l)
    print("any ms", ctx.last_kernel_ms(), a["occluded"].mean())
