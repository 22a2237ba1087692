import sys, os
sys.path.insert(0, "/root/repo")
import torch, ray_tracing_amd as rt
from ray_tracing_amd import scenes
for name, desc in (("cfg3", scenes.path_scene(levels=9)), ("cfg4", scenes.path_scene(levels=9, medium=True, sigma_s=0.02))):
    scene = rt.Scene(desc); ctx = rt.Context(0); ctx.upload(scene)
    acc = torch.zeros(1080, 1920, 3, device="cuda")
    for _ in range(2):
        _, st = ctx.render(1920, 1080, 64, 16, 0, 0, flags=rt.api.RENDER_TIME_STAGES, out=acc.data_ptr())
    d = st.as_dict()
    print(name, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in d.items() if k.startswith("ms_") or k in ("rays_closest", "rays_shadow", "iterations", "launches_tail", "launches_extend")}, flush=True)
    ctx.close()
