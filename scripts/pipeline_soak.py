"""Soak of the frame pipeline: N random frames (sizes from a few hundred to a few million paths, so that launches of every
kind overlap), four in flight, each compared bit for bit with the same frame rendered alone on a separate context.

    python scripts/pipeline_soak.py [N=300] [seed=1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracing_amd as rt
from ray_tracing_amd import scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
scene = rt.Scene(scenes.path_scene(levels=6, glass_levels=4, medium=True, sigma_s=0.3))
pipe = rt.FramePipeline(0, 4)
pipe.upload(scene)
alone = rt.Context(0)
alone.upload(scene)
frames = []
for k in range(n):
    w = int(rng.choice([16, 33, 64, 160, 320, 640, 960, 1920]))
    h = scenes.image_height(w) if w > 16 else 9
    frames.append((w, h, int(rng.integers(1, 9)), int(rng.choice([1, 4, 8, 16])), int(rng.integers(0, 1 << 30)), int(rng.integers(0, 4))))
t0 = time.time()
bad = 0
B = 32  # frames queued at a time
for b in range(0, n, B):
    part = frames[b:b + B]
    outs = [np.full((h, w, 3), np.nan, np.float32) for (w, h, *_r) in part]
    tickets = [pipe.submit(w, h, spp, depth, seed, first, out=o) for (w, h, spp, depth, seed, first), o in zip(part, outs)]
    for (w, h, spp, depth, seed, first), o, t in zip(part, outs, tickets):
        st = pipe.wait(t)
        ref, rst = alone.render(w, h, spp, depth, seed, first)
        if not (np.array_equal(o.view(np.uint32), ref.view(np.uint32)) and st.rays_closest == rst.rays_closest and st.rays_shadow == rst.rays_shadow):
            bad += 1
            print("MISMATCH", (w, h, spp, depth, seed, first), flush=True)
    print(f"{min(b + B, n)} frames, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
pipe.close()
alone.close()
sys.exit(1 if bad else 0)
