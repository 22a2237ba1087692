import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, oracle_binding as ob
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
d = scenes.baby_yoda_scene(os.path.join(ROOT, "tests/golden/baby_yoda.stl"))
sc = rt.Scene(d); ctx = rt.Context(0); ctx.upload(sc); orc = ob.OracleScene(d)
org, dr = scenes.pixel_center_rays(sc.flat.camera, 192, 108)
g = ctx.trace_closest(org, dr, counters=True); o = orc.trace_closest(org, dr)
for k in ("n_int", "n_tri", "prim", "obj"):
    bad = np.flatnonzero(g[k] != o[k]); print(k, len(bad), bad[:5], g[k][bad[:5]], o[k][bad[:5]])
hit = o["obj"] != 0xFFFFFFFF
p = (org + dr * np.where(hit, o["t"], 0)[:, None]).astype(np.float32)[hit]
rng = np.random.default_rng(0); l = rng.normal(size=p.shape).astype(np.float32); l /= np.linalg.norm(l, axis=1, keepdims=True)
ga = ctx.trace_any(p, l, counters=True); oa = orc.trace_any(p, l)
for k in ("n_int", "n_tri", "occluded"):
    bad = np.flatnonzero(ga[k] != oa[k]); print("any", k, len(bad), bad[:5], ga[k][bad[:5]], oa[k][bad[:5]])
nodes = sc.nodes(); print("leaf sizes", np.unique(nodes["num_prim"], return_counts=True))
for fl in (rt.api.RENDER_COUNT_VISITS, rt.api.RENDER_COUNT_VISITS | rt.api.RENDER_NO_TAIL):
    img, st = ctx.render(96, 54, 4, 16, 7, flags=fl); ref, ost = orc.render(96, 54, 4, 16, 7)
    print(fl, st.nodes_visited, st.nodes_visited_shadow, st.tris_tested, st.tris_tested_shadow, ost.nodes_visited, ost.tris_tested)
