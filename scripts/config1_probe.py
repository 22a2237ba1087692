import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, ray_tracing_amd as rt
from ray_tracing_amd import scenes
desc = scenes.volumetric_sphere()
scene = rt.Scene(desc); ctx = rt.Context(0); ctx.upload(scene)
acc = torch.zeros(225, 400, 3, device="cuda")
for _ in range(3):
    t0=time.perf_counter(); _, st = ctx.render(400, 225, 256, 16, 0, 0, out=acc.data_ptr()); dt=time.perf_counter()-t0
_, st = ctx.render(400, 225, 256, 16, 0, 0, flags=rt.api.RENDER_TIME_STAGES, out=acc.data_ptr())
d = st.as_dict()
print("wall %.2f ms" % (dt*1e3), {k: round(v, 2) for k, v in d.items() if k.startswith("ms_")}, d["launches_extend"], d["launches_shadow"], d["launches_tail"], d["paths_tail"], d["iterations"])
