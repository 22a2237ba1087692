"""Frame THROUGHPUT with frames in flight (rt_pipeline_*): ms per frame of one frame at a time against F frames at once.

    python scripts/pipeline_probe.py SPP FRAMES F [comm|reduce]

comm: create the slots' communicators (world size 1) but do not reduce; reduce: reduce every frame onto rank 0 too."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ray_tracing_amd as rt
from ray_tracing_amd import scenes

w, h, depth = 1920, 1080, 16
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n_flight = int(sys.argv[3]) if len(sys.argv) > 3 else 2
mode = sys.argv[4] if len(sys.argv) > 4 else ""
extra = sys.argv[5] if len(sys.argv) > 5 else ""  # torchdist / torchdist-lazy: a torch.distributed NCCL group beside it
if extra.startswith("torchdist"):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    if extra == "torchdist":
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("nccl", rank=0, world_size=1)
    if extra not in ("torchdist-nocoll", "torchdist-after"):
        dist.barrier()
        torch.cuda.synchronize()
pipe = rt.FramePipeline(0, n_flight)
pipe.upload(rt.Scene(scenes.path_scene(levels=9)))
if extra == "torchdist-after":  # bench.py's order: collectives of the torch group once the contexts exist
    x = torch.ones(1, device="cuda")
    dist.all_reduce(x)
    dist.broadcast(x, src=0)
    dist.barrier()
    torch.cuda.synchronize()
if mode in ("comm", "reduce"):
    pipe.comm_create([rt.comm_unique_id() for _ in range(n_flight)], 1, 0)
root = 0 if mode == "reduce" else rt.api.PIPELINE_NO_REDUCE
accs = [torch.zeros(h, w, 3, device="cuda") for _ in range(n_flight)]
torch.cuda.synchronize()


def run(n):
    for t in [pipe.submit(w, h, spp, depth, k, out=accs[k % n_flight].data_ptr(), reduce_root=root) for k in range(n)]:
        pipe.wait(t)


run(2 * n_flight)
ctx = pipe.context(0)
t0 = time.perf_counter()
for k in range(4):
    ctx.render(w, h, spp, depth, k, out=accs[0].data_ptr())
alone = (time.perf_counter() - t0) / 4
t0 = time.perf_counter()
run(frames)
flight = (time.perf_counter() - t0) / frames
print(extra, end=" ")
print(f"{spp} spp: one frame at a time {alone*1e3:.2f} ms/frame; {n_flight} in flight{' + ' + mode if mode else ''}: "
      f"{flight*1e3:.2f} ms/frame over {frames} frames", flush=True)
pipe.close()
