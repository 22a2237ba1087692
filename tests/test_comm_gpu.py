"""rt_comm_* / rt_reduce_accum (include/rt_amd.h): the RCCL sum-reduce of the per-GPU radiance accumulators, called
through the C ABI.  World size 1 on a one-GPU box; world size 2 (one process per GPU, the unique id handed over
through a file) when two devices are visible."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_binding as ob
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
from ray_tracing_amd.distributed import sample_shard

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reduce_world_size_1_is_the_identity_and_orders_behind_render(ctx):
    import torch
    desc = scenes.path_scene(levels=3, glass_levels=2)
    ctx.upload(rt.Scene(desc))
    comm = rt.Comm(ctx, rt.comm_unique_id(), 1, 0)
    w, h = 64, 36
    acc = torch.zeros(h, w, 3, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    for root in (0, -1):  # ncclReduce to rank 0, ncclAllReduce
        ctx.render(w, h, 4, 8, 5, out=acc.data_ptr())
        comm.reduce_accum(acc.data_ptr(), w, h, root)
        ref, _ = ob.OracleScene(desc).render(w, h, 4, 8, 5)
        assert np.array_equal(acc.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    with pytest.raises(rt.RtError):  # host memory is refused, not staged
        comm.reduce_accum(np.zeros((h, w, 3), np.float32), w, h, 0)
    with pytest.raises(rt.RtError):
        comm.reduce_accum(acc.data_ptr(), w, h, 3)  # root outside the communicator
    comm.close()


_RANK = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
rank, world, idfile, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
import torch
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
from ray_tracing_amd.distributed import sample_shard
if rank == 0:
    open(idfile + ".tmp", "wb").write(rt.comm_unique_id()); os.rename(idfile + ".tmp", idfile)
deadline = time.time() + 120.0
while not os.path.exists(idfile):  # rank 0 may have died before it wrote the id: give up instead of spinning for ever
    if time.time() > deadline:
        sys.exit(1)
    time.sleep(0.05)
uid = open(idfile, "rb").read()
ctx = rt.Context(rank)
ctx.upload(rt.Scene(scenes.path_scene(levels=3, glass_levels=2)))
comm = rt.Comm(ctx, uid, world, rank)
w, h, spp = 64, 36, 8
first, count = sample_shard(spp, world, rank)
acc = torch.zeros(h, w, 3, dtype=torch.float32, device=f"cuda:{rank}")
torch.cuda.synchronize()
ctx.render(w, h, count, 8, 5, first, out=acc.data_ptr())
comm.reduce_accum(acc.data_ptr(), w, h, 0)
if rank == 0:
    np.save(out, acc.cpu().numpy())
comm.close(); ctx.close()
'''


def test_reduce_world_size_2_sums_the_sample_shards(tmp_path):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    script = tmp_path / "rank.py"
    script.write_text(_RANK)
    idfile, out = str(tmp_path / "id.bin"), str(tmp_path / "sum.npy")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", idfile, out]) for r in range(2)]
    try:
        for p in procs:
            assert p.wait(timeout=300) == 0
    finally:  # a rank that is still alive (its peer died inside the collective init) must not keep holding a GPU
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    got = np.load(out)
    desc = scenes.path_scene(levels=3, glass_levels=2)
    orc = ob.OracleScene(desc)
    parts = [orc.render(64, 36, c, 8, 5, f)[0] for f, c in (sample_shard(8, 2, r) for r in range(2))]
    assert np.array_equal(got.view(np.uint32), (parts[0] + parts[1]).view(np.uint32))  # f32 sum of two shards: exact order
