"""N>1 path on CPU: world_size-2 gloo run of the sample-sharding + accumulator-reduce logic
(ray_tracing_amd/distributed.py).  The per-rank renderer here is the oracle (this is a test); on
GPUs bench.py plugs rt_render into the same helpers over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ray_tracing_amd.distributed import sample_shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP, DEPTH, SEED = 40, 22, 5, 6, 4


def test_sample_shard_partitions_exactly():
    for spp in (0, 1, 5, 64, 1000):
        for world in (1, 2, 3, 8):
            parts = [sample_shard(spp, world, r) for r in range(world)]
            assert sum(c for _, c in parts) == spp
            nxt = 0
            for first, count in parts:
                assert first == nxt
                nxt += count
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def _worker(rank, world, port, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_binding as ob
    from ray_tracing_amd import scenes
    from ray_tracing_amd.distributed import render_sharded
    orc = ob.OracleScene(scenes.path_scene(levels=2, glass_levels=1))

    def render_fn(first, count):
        img, _ = orc.render(W, H, count, DEPTH, SEED, first)
        return torch.from_numpy(img)

    acc = render_sharded(render_fn, W, H, SPP, DEPTH, SEED, torch.device("cpu"))
    if rank == 0:
        np.save(out_path, acc.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sample_sharding_reproduces_the_single_rank_frame(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "acc.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    from ray_tracing_amd import scenes
    orc = ob.OracleScene(scenes.path_scene(levels=2, glass_levels=1))
    full, _ = orc.render(W, H, SPP, DEPTH, SEED, 0)
    lo, _ = orc.render(W, H, 3, DEPTH, SEED, 0)   # rank 0 gets samples [0,3), rank 1 [3,5)
    hi, _ = orc.render(W, H, 2, DEPTH, SEED, 3)
    assert np.array_equal(got, lo + hi)           # the reduce is exactly the sum of the two shards
    rmse = np.sqrt(np.mean((got.astype(np.float64) - full) ** 2))
    assert rmse <= 1e-6 * max(1e-30, np.abs(full).mean())  # same paths; only the f32 summation order differs
