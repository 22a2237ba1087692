"""Where a sparse launch's cycles go (stats variant: scripts/build_variant.sh stats "-DRT_TRACE_STATS=1"): one extend
launch over a small frame's camera rays, cycle counters at the phase boundaries of the traversal loop."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ray_tracing_amd as rt
from ray_tracing_amd import scenes, api
desc = scenes.path_scene(levels=9)
scene = rt.Scene(desc); ctx = rt.Context(0); ctx.upload(scene)
fn = api.lib.rt_debug_trace_stats
buf = (C.c_ulonglong * 40)()
for (w, h) in ((8, 8), (128, 128), (512, 512), (1920, 1080)):
    acc = torch.zeros(h, w, 3, device="cuda")
    fl = api.RENDER_NO_TAIL | api.RENDER_TIME_STAGES
    ctx.render(w, h, 1, 1, 0, 0, out=acc.data_ptr(), flags=fl)
    fn(buf, 1)
    _, st = ctx.render(w, h, 1, 1, 0, 0, out=acc.data_ptr(), flags=fl)
    fn(buf, 0)
    it, run, wait, idle_more, idle_drain, pn, pl, pa = [buf[i] for i in range(8)]
    c = [buf[16 + i] for i in range(5)]
    print(f"{w}x{h} = {w*h} camera rays: extend {st.ms_extend:.3f} ms  iterations {it} lanes/iter {run/max(1,it):.1f}  phases node {pn} leaf {pl} adv {pa}")
    print(f"   wave cycles: sched {c[0]/max(1,it):.0f} per iteration, adv {c[1]/max(1,pa):.0f} per phase, node {c[2]/max(1,pn):.0f} per phase (3 steps), leaf {c[3]/max(1,pl):.0f} per phase; total/iteration {c[4]/max(1,it):.0f}")
