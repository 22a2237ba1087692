import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, oracle_binding as ob
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
d = scenes.baby_yoda_scene(os.path.join(ROOT, "tests/golden/baby_yoda.stl"))
sc = rt.Scene(d); ctx = rt.Context(0); ctx.upload(sc); orc = ob.OracleScene(d)
w, h = 96, 54
a, st = ctx.render(w, h, 1, 1, 7, flags=rt.api.RENDER_COUNT_VISITS)
b, st2 = ctx.render(w, h, 1, 1, 7, flags=rt.api.RENDER_COUNT_VISITS | rt.api.RENDER_NO_TAIL)
print(a[..., 0].sum(), b[..., 0].sum(), st.tris_tested, st2.tris_tested)
bad = np.argwhere(a[..., 0] != b[..., 0])
print(bad, a[..., 0][a[..., 0] != b[..., 0]], b[..., 0][a[..., 0] != b[..., 0]])
