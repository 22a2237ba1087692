"""Kernel time of rt_trace_closest / rt_trace_any versus batch size (launch floor of the persistent traversal)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
desc = scenes.path_scene(levels=9)
scene = rt.Scene(desc); ctx = rt.Context(0); ctx.upload(scene)
w, h = 1920, 1080
org, dr = scenes.pixel_center_rays(scene.flat.camera, w, h)
rng = np.random.default_rng(1)
for n in (64, 1024, 16384, 262144, 2073600):
    idx = rng.choice(len(org), n, replace=False) if n < len(org) else np.arange(len(org))
    o, d = np.ascontiguousarray(org[idx]), np.ascontiguousarray(dr[idx])
    hits = ctx.trace_closest(o, d, counters=True)
    ms = []
    for _ in range(5):
        ctx.trace_closest(o, d)
        ms.append(ctx.last_kernel_ms())
    ni = hits["n_int"].astype(np.int64) + hits["n_tri"]
    print(f"n={n:8d} kernel ms min {min(ms):.3f} median {sorted(ms)[2]:.3f}   steps/ray mean {ni.mean():.1f} max {ni.max()}")
