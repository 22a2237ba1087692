"""Frame time and Mrays/s of every BASELINE.json config on one GPU (bench.py measures configs[2] only): a lone frame
(best of 3), and frames through a pipeline with two in flight (rt_pipeline_*: throughput over 8 frames)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ray_tracing_amd as rt
from ray_tracing_amd import scenes

ONLY = sys.argv[1] if len(sys.argv) > 1 else ""  # substring of the config names to run (default: all)


def run(name, desc, w, spp, depth=16, reps=2, flight=2, frames=8):
    if ONLY not in name:
        return
    desc = desc()
    t0 = time.time(); scene = rt.Scene(desc); t_build = time.time() - t0
    pipe = rt.FramePipeline(0, flight); pipe.upload(scene)
    ctx = pipe.context(0)
    h = scenes.image_height(w)
    accs = [torch.zeros(h, w, 3, device="cuda") for _ in range(flight)]
    acc = accs[0]
    best = None
    for _ in range(reps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, st = ctx.render(w, h, spp, depth, 0, 0, out=acc.data_ptr())
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    rays = st.rays_closest + st.rays_shadow - st.rays_shadow_skipped  # traced
    def through(n):
        for t in [pipe.submit(w, h, spp, depth, 0, 0, out=accs[k % flight].data_ptr()) for k in range(n)]:
            pipe.wait(t)
    through(flight)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    through(frames)
    torch.cuda.synchronize(); fl = (time.perf_counter() - t0) / frames
    print(f"{name:34s} {scene.flat.n_tris:9d} tris {scene.flat.n_objects:3d} objects  {w}x{h} {spp:4d} spp  {best*1e3:9.1f} ms/frame "
          f"{rays/best/1e6:8.1f} Mrays/s | {flight} frames in flight: {fl*1e3:9.1f} ms/frame {rays/fl/1e6:8.1f} Mrays/s "
          f"({rays/1e6:.1f} M rays, {st.batches} batch(es), lean={st.lean_kernels}, scene build {t_build:.1f} s)", flush=True)
    ctx = None
    pipe.close()

run("config 1 volumetric sphere", lambda: scenes.volumetric_sphere(), 400, 256)
run("config 3 path tracing", lambda: scenes.path_scene(levels=9), 1920, 64)
run("config 4 + homogeneous medium", lambda: scenes.path_scene(levels=9, medium=True, sigma_s=0.02), 1920, 64)
run("config 2 scene, path traced", lambda: scenes.tetrahedron_scene(levels=9), 1920, 64)
run("config 5 instanced (one BVH x10)", lambda: scenes.multi_mesh_scene(levels=9, copies=10), 1920, 64)
run("config 5 ten distinct meshes", lambda: scenes.multi_mesh_scene(levels=9, copies=10, distinct=True), 1920, 64)
run("config 5 instanced, 3840x2160", lambda: scenes.multi_mesh_scene(levels=9, copies=10), 3840, 64, reps=1, frames=4)
run("config 5 distinct, 3840x2160", lambda: scenes.multi_mesh_scene(levels=9, copies=10, distinct=True), 3840, 64, reps=1, frames=4)
run("config 5 distinct, 4K, 256 spp", lambda: scenes.multi_mesh_scene(levels=9, copies=10, distinct=True), 3840, 256, reps=1, frames=4)
