"""Driver for rocprofv3 counter passes on the render workload: one 1080p batch of `spp` samples."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 9
desc = scenes.path_scene(levels=levels)
desc.initial_medium = rt.api.MEDIUM_NONE
scene = rt.Scene(desc)
ctx = rt.Context(0)
ctx.upload(scene)
img, st = ctx.render(1920, 1080, spp, 16, 0, 0, flags=rt.api.RENDER_TIME_STAGES)
print(st.as_dict())
