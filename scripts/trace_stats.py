"""Lane-occupancy diagnostics of the persistent traversal loop (needs the `stats` variant library:
scripts/build_variant.sh stats "-DRT_UNIFIED=0 -DRT_TRACE_STATS=1"; RT_AMD_LIB=.../librt_amd_stats.so)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ray_tracing_amd as rt
from ray_tracing_amd import scenes, api
spp = int(os.environ.get("SPP", "64"))
desc = scenes.path_scene(levels=int(os.environ.get("LEVELS", "9")))
scene = rt.Scene(desc); ctx = rt.Context(0); ctx.upload(scene)
w, h = 1920, 1080
acc = torch.zeros(h, w, 3, device="cuda")
fn = api.lib.rt_debug_trace_stats
buf = (C.c_ulonglong * 40)()
ctx.render(w, h, spp, int(os.environ.get("DEPTH", "16")), 0, 0, out=acc.data_ptr())
fn(buf, 1)
_, st = ctx.render(w, h, spp, int(os.environ.get("DEPTH", "16")), 0, 0, out=acc.data_ptr())
fn(buf, 0)
for name, o in (("extend", 0), ("shadow", 8)):
    it, run, wait, idle_more, idle_drain, pn, pl, pa = [buf[o + i] for i in range(8)]
    tot = it * 64
    print(f"{name}: iterations {it}  advance {run/tot:.3f}  wait-phase {wait/tot:.3f}  idle(queue) {idle_more/tot:.3f}  idle(drain) {idle_drain/tot:.3f}"
          f"  phases node {pn/it:.3f} leaf {pl/it:.3f} adv {pa/it:.3f}  lanes/iter {run/it:.1f}")
    c = [buf[16 + o + i] for i in range(5)]
    if c[4]:
        print(f"   wave cycles: scheduler+refill {c[0]/c[4]:.3f}  advance {c[1]/c[4]:.3f}  node {c[2]/c[4]:.3f}  leaf {c[3]/c[4]:.3f}"
              f"   per iteration: {c[4]/it:.0f} cycles (sched {c[0]/it:.0f}, adv {c[1]/max(1,pa):.0f} per adv phase, node {c[2]/max(1,pn):.0f} per node phase, leaf {c[3]/max(1,pl):.0f} per leaf phase)")
    u = [buf[32 + (o // 2) + i] for i in range(4)]
    if u[0]:
        print(f"   node steps {u[0]}: one node for all lanes {u[1]/u[0]:.3f}; lanes on the first lane's node {u[2]/max(1,u[3]):.3f}; lanes per step {u[3]/u[0]:.1f}")
