import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
d = scenes.baby_yoda_scene(os.path.join(ROOT, "tests/golden/baby_yoda.stl"))
sc = rt.Scene(d); ctx = rt.Context(0); ctx.upload(sc)
for fl in (rt.api.RENDER_COUNT_VISITS, rt.api.RENDER_COUNT_VISITS | rt.api.RENDER_NO_TAIL):
    img, st = ctx.render(96, 54, 4, 1, 7, flags=fl)
    print(os.environ.get("RT_TUNE"), fl, "tri", st.tris_tested, st.tris_tested_shadow, "int", st.nodes_visited, st.nodes_visited_shadow)
