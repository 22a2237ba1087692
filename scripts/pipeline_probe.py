"""Probe: frame THROUGHPUT when whole frames are pipelined over two contexts (two streams, two host threads), each
started a fraction of a frame after the one before, so that the latency-bound late bounces of one frame run beside the large launches
of the next.  Prints ms per frame of one context alone and of the staggered pair."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ray_tracing_amd as rt
from ray_tracing_amd import scenes

w, h, depth = 1920, 1080, 16
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n_ctx = int(sys.argv[3]) if len(sys.argv) > 3 else 2
stagger_ms = float(sys.argv[4]) if len(sys.argv) > 4 else -1.0  # < 0: one frame's time / n_ctx
scene = rt.Scene(scenes.path_scene(levels=9))
ctxs = [rt.Context(0) for _ in range(n_ctx)]
for c in ctxs:
    c.upload(scene)
accs = [torch.zeros(h, w, 3, device="cuda") for _ in range(n_ctx)]


def run(i, n, delay):
    if delay > 0:
        time.sleep(delay)
    for k in range(n):
        ctxs[i].render(w, h, spp, depth, k, 0, out=accs[i].data_ptr())


for i in range(n_ctx):
    run(i, 2, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(0, frames, 0)
torch.cuda.synchronize()
alone = (time.perf_counter() - t0) / frames
if stagger_ms < 0:
    stagger_ms = alone * 1e3 / n_ctx
ts = [threading.Thread(target=run, args=(i, frames, i * stagger_ms * 1e-3)) for i in range(n_ctx)]
t0 = time.perf_counter()
for t in ts:
    t.start()
for t in ts:
    t.join()
torch.cuda.synchronize()
pair = (time.perf_counter() - t0) / (n_ctx * frames)
print(f"{spp} spp: one context {alone*1e3:.2f} ms/frame; {n_ctx} contexts staggered by {stagger_ms:.1f} ms: "
      f"{pair*1e3:.2f} ms/frame over {n_ctx*frames} frames", flush=True)
