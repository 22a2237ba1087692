import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, oracle_binding as ob
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
d = scenes.baby_yoda_scene(os.path.join(ROOT, "tests/golden/baby_yoda.stl"))
sc = rt.Scene(d); ctx = rt.Context(0); ctx.upload(sc); orc = ob.OracleScene(d)
depth = 1
for (w, h, spp, mp) in ((96, 54, 1, 0), (96, 54, 2, 0), (96, 54, 4, 0), (96, 54, 4, 96 * 54), (96, 54, 4, 2 * 96 * 54), (192, 108, 1, 0)):
    ref, ost = orc.render(w, h, spp, depth, 7)
    ctx.set_max_paths_in_flight(mp)
    for fl in (rt.api.RENDER_COUNT_VISITS, rt.api.RENDER_COUNT_VISITS | rt.api.RENDER_NO_TAIL):
        for rep in range(2):
            img, st = ctx.render(w, h, spp, depth, 7, flags=fl)
            print(w, spp, mp, fl, "tri", st.tris_tested, st.tris_tested_shadow, "orc", ost.tris_tested, "batches", st.batches)
