"""Soak: repeated uploads and renders through the C ABI; free HBM and frame time must stay put."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
ctx = rt.Context(0)
descs = [scenes.path_scene(levels=7), scenes.primitives_scene(), scenes.path_scene(levels=6, glass_levels=4, medium=True)]
built = [rt.Scene(d) for d in descs]
acc = torch.zeros(540, 960, 3, device="cuda")
ref = {}
t0 = time.time()
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 150):
    k = it % len(built)
    ctx.upload(built[k])
    t = time.perf_counter()
    _, st = ctx.render(960, 540, 8, 16, 3, 0, out=acc.data_ptr())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) * 1e3
    h = int(acc.view(torch.int32).sum(dtype=torch.int64).item())  # checksum of the bits
    if k in ref:
        assert ref[k] == h, (it, k)
    ref[k] = h
    if it % 30 == 0 or it < 3:
        free, total = torch.cuda.mem_get_info()
        print(f"iter {it:4d} scene {k} {ms:7.2f} ms  free HBM {free / 2**30:8.2f} GiB  checksum {h}", flush=True)
free, total = torch.cuda.mem_get_info()
print(f"done in {time.time() - t0:.1f} s, free HBM {free / 2**30:.2f} GiB")
ctx.close()
