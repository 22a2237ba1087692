import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, oracle_binding as ob
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
d = scenes.baby_yoda_scene(os.path.join(ROOT, "tests/golden/baby_yoda.stl"))
sc = rt.Scene(d); ctx = rt.Context(0); ctx.upload(sc); orc = ob.OracleScene(d)
for depth in (1, 2, 3, 16):
  for (w, h, spp) in ((96, 54, 4), (48, 27, 1)):
    ref, ost = orc.render(w, h, spp, depth, 7)
    for fl in (rt.api.RENDER_COUNT_VISITS, rt.api.RENDER_COUNT_VISITS | rt.api.RENDER_NO_TAIL):
        img, st = ctx.render(w, h, spp, depth, 7, flags=fl)
        print(depth, w, fl, "int", st.nodes_visited, st.nodes_visited_shadow, "tri", st.tris_tested, st.tris_tested_shadow, "orc", ost.nodes_visited, ost.tris_tested,
              "rays", st.rays_closest, st.rays_shadow, ost.rays_closest, ost.rays_shadow)
