"""rt_pipeline_* (include/rt_amd.h): frames in flight.  Every frame that goes through a pipeline -- whatever runs beside
it on the GPU -- must be the frame a lone rt_render call gives, i.e. the oracle's, bit for bit; statuses and error
texts must come back with the ticket they belong to; the per-frame reduce must come out in submission order."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_binding as ob
import ray_tracing_amd as rt
from ray_tracing_amd import scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _frames():
    # (width, height, spp, depth, seed, spp_first): sizes that take the tail kernel at once and sizes that start as a
    # wavefront, so that launches of every kind overlap
    return [(96, 54, 4, 16, 1, 0), (320, 180, 4, 16, 2, 0), (64, 36, 8, 8, 3, 4), (33, 19, 3, 1, 4, 0),
            (320, 180, 2, 16, 5, 2), (96, 54, 4, 16, 6, 0), (160, 90, 6, 12, 7, 0)]


@pytest.mark.parametrize("n_frames_in_flight", [1, 3])
def test_frames_in_flight_are_the_oracles_frames(n_frames_in_flight):
    desc = scenes.path_scene(levels=5, glass_levels=3, medium=True, sigma_s=0.5)
    pipe = rt.FramePipeline(0, n_frames_in_flight)
    assert pipe.frames_in_flight == n_frames_in_flight
    # up to the GPU's hardware queues (4 by default) every slot gets its own; the third could share one in a process
    # that keeps other streams busy, the second never has to
    assert min(2, n_frames_in_flight) <= pipe.separate_queues <= n_frames_in_flight
    pipe.upload(rt.Scene(desc))
    orc = ob.OracleScene(desc)
    frames = _frames()
    outs = [np.full((h, w, 3), np.nan, np.float32) for (w, h, *_r) in frames]  # host buffers: staged by rt_render
    tickets = [pipe.submit(w, h, spp, depth, seed, first, out=o) for (w, h, spp, depth, seed, first), o in zip(frames, outs)]
    assert tickets == list(range(len(frames)))
    for k in reversed(range(len(frames))):  # waiting order is the caller's business
        st = pipe.wait(tickets[k])
        w, h, spp, depth, seed, first = frames[k]
        ref, ost = orc.render(w, h, spp, depth, seed, first)
        assert np.array_equal(outs[k].view(np.uint32), ref.view(np.uint32)), k
        assert (st.rays_closest, st.rays_shadow, st.paths) == (ost.rays_closest, ost.rays_shadow, w * h * spp)
    # a second round through the same contexts, counting the reference's visits beside plain frames
    t_a = pipe.submit(96, 54, 4, 16, 1, 0, flags=rt.api.RENDER_COUNT_VISITS, out=outs[0])
    t_b = pipe.submit(96, 54, 4, 16, 6, 0, out=outs[5])
    st_a, st_b = pipe.wait(t_a), pipe.wait(t_b)
    ref, ost = orc.render(96, 54, 4, 16, 1, 0)
    assert np.array_equal(outs[0].view(np.uint32), ref.view(np.uint32))
    assert st_a.nodes_visited + st_a.nodes_visited_shadow == ost.nodes_visited
    assert st_b.nodes_visited == 0
    pipe.close()


def test_status_and_error_text_travel_with_the_ticket():
    pipe = rt.FramePipeline(0, 2)
    out = np.zeros((36, 64, 3), np.float32)
    t0 = pipe.submit(64, 36, 2, 4, out=out)  # no scene uploaded yet: the frame fails, not the submit
    with pytest.raises(rt.RtError, match="no scene uploaded") as e:
        pipe.wait(t0)
    assert e.value.code == -5
    pipe.upload(rt.Scene(scenes.tetrahedron_scene(levels=2)))
    good = pipe.submit(64, 36, 2, 4, out=out)
    col = out[:, :1].copy()
    bad = pipe.submit(1, 36, 2, 4, out=col)  # the reference divides by (width - 1)
    with pytest.raises(rt.RtError, match="width and height"):
        pipe.wait(bad)
    assert pipe.wait(good).paths == 64 * 36 * 2
    with pytest.raises(rt.RtError, match="never handed out or has been waited for"):
        pipe.wait(good)
    with pytest.raises(rt.RtError):
        pipe.wait(12345)
    with pytest.raises(rt.RtError, match="rt_pipeline_comm_create"):  # a reduce without communicators
        pipe.submit(64, 36, 2, 4, out=out, reduce_root=0)
    with pytest.raises(rt.RtError):
        pipe.context(2)
    # the borrowed context renders like any other while the pipeline is idle, and outlives nothing
    ctx = pipe.context(1)
    img, st = ctx.render(64, 36, 2, 4)
    assert np.array_equal(img.view(np.uint32), out.view(np.uint32)) and st.paths == 64 * 36 * 2
    # frames still queued at destruction are rendered, not dropped: their buffers are written before close() returns
    late = np.full((36, 64, 3), np.nan, np.float32)
    pipe.submit(64, 36, 2, 4, out=late)
    pipe.close()
    assert np.array_equal(late.view(np.uint32), out.view(np.uint32))


def test_reduce_world_size_1_per_frame_in_submission_order():
    import torch
    desc = scenes.path_scene(levels=3, glass_levels=2)
    pipe = rt.FramePipeline(0, 3)
    pipe.upload(rt.Scene(desc))
    pipe.comm_create([rt.comm_unique_id() for _ in range(3)], 1, 0)
    with pytest.raises(rt.RtError):
        pipe.comm_create([rt.comm_unique_id() for _ in range(3)], 1, 0)  # once
    w, h = 64, 36
    accs = [torch.zeros(h, w, 3, dtype=torch.float32, device="cuda:0") for _ in range(6)]
    torch.cuda.synchronize()
    # a long frame first, short ones behind it: they finish rendering earlier and have to wait for its reduce
    spps = [16, 1, 1, 2, 1, 4]
    tickets = [pipe.submit(w, h, spps[k], 8, 5 + k, out=accs[k].data_ptr(), reduce_root=(0 if k % 2 == 0 else -1))
               for k in range(6)]
    orc = ob.OracleScene(desc)
    for k, t in enumerate(tickets):
        pipe.wait(t)
        ref, _ = orc.render(w, h, spps[k], 8, 5 + k)
        assert np.array_equal(accs[k].cpu().numpy().view(np.uint32), ref.view(np.uint32))
    host = np.zeros((h, w, 3), np.float32)
    bad = pipe.submit(w, h, 1, 8, out=host, reduce_root=0)  # host memory cannot be reduced: the frame reports it ...
    ok = pipe.submit(w, h, 1, 8, 5, out=accs[0].data_ptr(), reduce_root=0)
    with pytest.raises(rt.RtError, match="device memory"):
        pipe.wait(bad)
    pipe.wait(ok)  # ... and the frames behind it still get their turn
    pipe.close()


_RANK = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
rank, world, idfile, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
import torch
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
from ray_tracing_amd.distributed import sample_shard
F = 3
if rank == 0:
    open(idfile + ".tmp", "wb").write(b"".join(rt.comm_unique_id() for _ in range(F))); os.rename(idfile + ".tmp", idfile)
deadline = time.time() + 120.0
while not os.path.exists(idfile):
    if time.time() > deadline:
        sys.exit(1)
    time.sleep(0.05)
blob = open(idfile, "rb").read()
pipe = rt.FramePipeline(rank, F)
pipe.upload(rt.Scene(scenes.path_scene(levels=3, glass_levels=2)))
pipe.comm_create([blob[128 * s:128 * (s + 1)] for s in range(F)], world, rank)
w, h, spp = 64, 36, 8
first, count = sample_shard(spp, world, rank)
frames = 7
accs = [torch.zeros(h, w, 3, dtype=torch.float32, device=f"cuda:{rank}") for _ in range(frames)]
torch.cuda.synchronize()
# the ranks submit at different times (rank 1 pauses after two frames): their contexts finish in different orders, the
# reduces must still pair up frame by frame
tickets = []
for k in range(frames):
    tickets.append(pipe.submit(w, h, count, 8, 5 + k, first, out=accs[k].data_ptr(), reduce_root=0))
    if k == 1 and rank == 1:
        time.sleep(0.5)
for t in tickets:
    pipe.wait(t)
if rank == 0:
    np.save(out, np.stack([a.cpu().numpy() for a in accs]))
pipe.close()
'''


def test_reduce_world_size_2_frames_pair_up(tmp_path):
    import torch
    from ray_tracing_amd.distributed import sample_shard
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    script = tmp_path / "rank.py"
    script.write_text(_RANK)
    idfile, out = str(tmp_path / "id.bin"), str(tmp_path / "sum.npy")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", idfile, out]) for r in range(2)]
    try:
        for p in procs:
            assert p.wait(timeout=300) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    got = np.load(out)
    orc = ob.OracleScene(scenes.path_scene(levels=3, glass_levels=2))
    for k in range(len(got)):
        parts = [orc.render(64, 36, c, 8, 5 + k, f)[0] for f, c in (sample_shard(8, 2, r) for r in range(2))]
        assert np.array_equal(got[k].view(np.uint32), (parts[0] + parts[1]).view(np.uint32)), k


def test_hooked_pipeline_gives_each_hook_its_slots_context():
    """rt_pipeline_create_hooked with a device: the render hook gets the slot's context and may run any calls on it --
    here Engine::ray_colour in batch form (rt_ray_colour) for frames of caller rays, two at a time."""
    desc = scenes.path_scene(levels=4, glass_levels=2)
    scene = rt.Scene(desc)
    cam = scene.flat.camera
    org, dirs = scenes.pixel_center_rays(cam, 48, 27)
    results, errors = {}, []

    def render(ctx_addr, slot, ticket):
        try:
            ctx = rt.Context(_borrowed=(ctx_addr, pipe))
            results[ticket] = (slot, ctx.ray_colour(org, dirs, 2, 8, seed=ticket)[0])
            return 0
        except Exception as e:  # reported through the status; the text is kept for the assertion below
            errors.append(repr(e))
            return -1

    pipe = rt.FramePipeline(0, 2, render=render)
    pipe.upload(scene)
    for t in [pipe.submit_hooked() for _ in range(5)]:
        pipe.wait(t)
    assert not errors, errors
    with pytest.raises(rt.RtError, match="no reduce hook"):
        pipe.submit_hooked(reduce=True)
    ctx = pipe.context(0)
    for ticket, (slot, img) in results.items():
        assert slot == ticket % 2
        ref, _ = ctx.ray_colour(org, dirs, 2, 8, seed=ticket)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    orc = ob.OracleScene(desc)
    ref0 = orc.ray_colour(org, dirs, 2, 8, seed=0)
    ref0 = ref0[0] if isinstance(ref0, tuple) else ref0
    assert np.array_equal(results[0][1].view(np.uint32), np.asarray(ref0, np.float32).reshape(-1, 3).view(np.uint32))
    pipe.close()
