"""Driver for rocprofv3 passes on a render workload: ONE frame.
    prof_render.py SPP [LEVELS] [WORKLOAD=render|config4|config5|config5i] [WIDTH=1920]
render = bench.py's default scene (configs[2]); config4 = the same with the homogeneous medium (configs[3]); config5 = ten
distinct meshes; config5i = ten instances of one mesh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 9
workload = sys.argv[3] if len(sys.argv) > 3 else "render"
width = int(sys.argv[4]) if len(sys.argv) > 4 else 1920
if workload == "render":
    desc = scenes.path_scene(levels=levels)
elif workload == "config4":
    desc = scenes.path_scene(levels=levels, medium=True, sigma_s=0.02)
else:
    desc = scenes.multi_mesh_scene(levels=levels, copies=10, distinct=workload == "config5")
if workload != "config4":
    desc.initial_medium = rt.api.MEDIUM_NONE
scene = rt.Scene(desc)
ctx = rt.Context(0)
ctx.upload(scene)
img, st = ctx.render(width, scenes.image_height(width), spp, 16, 0, 0, flags=rt.api.RENDER_TIME_STAGES)
print(rt.api.build_id(), st.as_dict())
