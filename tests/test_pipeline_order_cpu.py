"""The frame pipeline's multi-rank guarantee on CPU: world_size-2 gloo run of rt_pipeline_* with caller-supplied steps
(rt_pipeline_create_hooked, no device).  Every rank submits the same frames in the same order; each rank's slots finish
their frames in a different, rank-dependent order (the render hook sleeps), and the reduce hook is a gloo all-reduce: the
ranks' collectives pair up frame by frame only if every rank issues them in submission order, one at a time."""
import os
import socket
import sys
import threading
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FRAMES, SLOTS = 24, 3


def _delay(rank, ticket):  # ms: long frames sit in different places on the two ranks
    return (7 * ticket + 13 * rank * (ticket % 5) + (40 if (ticket + 2 * rank) % 7 == 0 else 0)) % 60


def test_hooked_pipeline_runs_hooks_and_reports_their_status():
    import ray_tracing_amd as rt
    seen, order = [], []
    lock = threading.Lock()

    def render(ctx, slot, ticket):
        assert ctx is None  # no device
        time.sleep(0.001 * ((ticket * 5) % 7))
        with lock:
            seen.append((slot, ticket))
        return -1 if ticket == 4 else 0

    def reduce(ctx, slot, ticket):
        with lock:
            order.append(ticket)
        return 0

    pipe = rt.FramePipeline(-1, SLOTS, render=render, reduce=reduce)
    tickets = [pipe.submit_hooked(reduce=(k % 2 == 0)) for k in range(12)]
    for t in tickets:
        if t == 4:
            with pytest.raises(rt.RtError) as e:
                pipe.wait(t)
            assert e.value.code == -1
        else:
            pipe.wait(t)
    assert sorted(seen, key=lambda e: e[1]) == [(k % SLOTS, k) for k in range(12)]  # frame k on slot k mod n
    assert order == [0, 2, 6, 8, 10]  # submission order; the failed frame's reduce is skipped, its turn is not
    with pytest.raises(rt.RtError):
        pipe.submit(8, 8, 1, 1, out=np.zeros((8, 8, 3), np.float32))  # frames of a hooked pipeline go through its hooks
    with pytest.raises(rt.RtError):
        pipe.upload(rt.api.SceneFlat())  # no device
    pipe.close()
    with pytest.raises(rt.RtError):
        rt.FramePipeline(-1, 2)  # a pipeline of rt_render frames needs a device


def _worker(rank, world, port, out_path):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ray_tracing_amd as rt
    frames = [torch.zeros(16) for _ in range(FRAMES)]
    finished = []

    def render(ctx, slot, ticket):
        time.sleep(1e-3 * _delay(rank, ticket))
        frames[ticket] += float((rank + 1) * 1000 + ticket)  # this rank's "accumulator" of frame `ticket`
        finished.append(ticket)
        return 0

    def reduce(ctx, slot, ticket):
        dist.all_reduce(frames[ticket])  # pairs with whatever collective the other rank issues next
        return 0

    os.environ["RT_PIPELINE_PACE"] = "0"
    pipe = rt.FramePipeline(-1, SLOTS, render=render, reduce=reduce)
    tickets = [pipe.submit_hooked(reduce=True) for _ in range(FRAMES)]
    for t in tickets:
        pipe.wait(t)
    pipe.close()
    got = torch.stack(frames)[:, 0].numpy()
    np.save(out_path + f".{rank}.npy", np.stack([got, np.array(finished, np.float32)]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_pair_their_reduces_frame_by_frame(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "frames")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    expect = np.array([1000 + k + 2000 + k for k in range(FRAMES)], np.float32)
    orders = []
    for rank in range(2):
        got, finished = np.load(out + f".{rank}.npy")
        assert np.array_equal(got, expect), rank  # every frame holds the sum of ITS two accumulators
        orders.append(list(finished.astype(int)))
    # the test means something only if frames did finish out of order, and differently on the two ranks
    assert any(o != sorted(o) for o in orders) and orders[0] != orders[1]
