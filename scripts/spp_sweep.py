"""Frame time of the bench scene at 1080p for several sample counts (the per-GPU share of the strong-scaling runs:
64 / N spp): wall time of rt_render (device accumulator), best of 5, and the stage breakdown of one frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ray_tracing_amd as rt
from ray_tracing_amd import scenes
desc = scenes.path_scene(levels=9)
scene = rt.Scene(desc); ctx = rt.Context(0); ctx.upload(scene)
acc = torch.zeros(1080, 1920, 3, device="cuda")
torch.cuda.synchronize()
for spp in [int(a) for a in sys.argv[1:]] or [64, 32, 16, 8]:
    best = 1e9
    for _ in range(6):
        t0 = time.perf_counter()
        _, st = ctx.render(1920, 1080, spp, 16, 0, 0, out=acc.data_ptr())
        best = min(best, (time.perf_counter() - t0) * 1e3)
    _, st = ctx.render(1920, 1080, spp, 16, 0, 0, flags=rt.api.RENDER_TIME_STAGES, out=acc.data_ptr())
    d = st.as_dict()
    print(spp, "spp: wall %.2f ms, device %.2f ms" % (best, st.ms_total),
          {k: round(v, 2) for k, v in d.items() if k.startswith("ms_") and k != "ms_total"},
          "launches", d["launches_extend"], d["launches_shadow"], d["launches_tail"], "paths_tail", d["paths_tail"], flush=True)
