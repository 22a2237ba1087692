"""Probe: does rendering two half-frames concurrently (two contexts = two streams, two host threads) beat
one 64-spp render?  Drains, launch gaps and the latency-bound late bounces of one half could overlap with
the other half's large launches."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ray_tracing_amd as rt
from ray_tracing_amd import scenes

w, h, spp, depth = 1920, 1080, 64, 16
desc = scenes.path_scene(levels=9)
scene = rt.Scene(desc)
n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ctxs = [rt.Context(0) for _ in range(n_ctx)]
for c in ctxs:
    c.upload(scene)
accs = [torch.zeros(h, w, 3, device="cuda") for _ in range(n_ctx)]
part = spp // n_ctx

def work(i):
    ctxs[i].render(w, h, part, depth, 0, i * part, out=accs[i].data_ptr())

def frame():
    ts = [threading.Thread(target=work, args=(i,)) for i in range(n_ctx)]
    for t in ts: t.start()
    for t in ts: t.join()
    torch.cuda.synchronize()

for _ in range(2): frame()
t0 = time.perf_counter()
for _ in range(5): frame()
dt = (time.perf_counter() - t0) / 5
print(f"{n_ctx} concurrent contexts x {part} spp: {dt*1e3:.1f} ms/frame", flush=True)
