"""A/B of one build switch on config 5 (ten distinct 1 M-triangle meshes: 1.17 GB of geometry, beyond the Infinity Cache)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
desc = scenes.multi_mesh_scene(levels=9, copies=10, distinct=True)
scene = rt.Scene(desc); ctx = rt.Context(0); ctx.upload(scene)
for w in (1920, 3840):
    h = scenes.image_height(w)
    acc = torch.zeros(h, w, 3, device="cuda")
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, st = ctx.render(w, h, 64, 16, 0, 0, out=acc.data_ptr())
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    _, st = ctx.render(w, h, 64, 16, 0, 0, flags=rt.api.RENDER_TIME_STAGES, out=acc.data_ptr())
    d = st.as_dict()
    print(f"config5 distinct {w}x{h} 64 spp: {best*1e3:.1f} ms", {k: round(v, 2) for k, v in d.items() if k.startswith("ms_") and k != "ms_total"}, flush=True)
