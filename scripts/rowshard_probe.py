import sys, os, time
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
desc = scenes.path_scene(levels=9)
scene = rt.Scene(desc); ctx = rt.Context(0); ctx.upload(scene)
for (w, h, spp) in [(1920, 1080, 64), (1920, 1080, 8), (1920, 135, 64), (1920, 270, 64), (1920, 1080, 16), (1920, 540, 64), (1920,1080,32)]:
    acc = torch.zeros(h, w, 3, device="cuda")
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t = time.perf_counter()
        _, st = ctx.render(w, h, spp, 16, 0, 0, out=acc.data_ptr(), flags=rt.api.RENDER_TIME_STAGES)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    print(f"{w}x{h} {spp} spp: {best*1e3:.2f} ms  extend {st.ms_extend:.2f} shade {st.ms_shade:.2f} shadow {st.ms_shadow:.2f} tail/mixed {st.ms_tail:.2f} compact {st.ms_compact:.2f} rays {st.rays_closest+st.rays_shadow}")
