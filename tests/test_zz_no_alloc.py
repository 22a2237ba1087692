"""rt_trace_closest_noalloc (include/rt_amd.h): the allocation-free form of rt_trace_closest, for callers that cannot
call hipMalloc.  Its scratch (work counters + traversal-stack overflow strips) is a static device array; results must
be the bits rt_trace_closest returns -- on two calls in a row, the second with a smaller input (the shared array
keeps the first call's contents)."""
import numpy as np
import pytest

import oracle_binding as ob
import ray_tracing_amd as rt
from ray_tracing_amd import scenes

pytestmark = pytest.mark.gpu
KEYS = ("t", "prim", "obj", "nx", "ny", "nz", "front_face", "n_int", "n_tri")


def _device_batch(torch, org, dr):
    cols = [torch.from_numpy(np.ascontiguousarray(a[:, k])).cuda() for a in (org, dr) for k in range(3)]
    n = len(org)
    out = {k: torch.full((n,), 0x7F if k == "front_face" else -7, dtype=torch.uint8 if k == "front_face" else torch.int32,
                         device="cuda") for k in KEYS}
    return cols, out


def _run(ctx, torch, org, dr, noalloc, counters, t_min=1e-3, t_max=np.inf):
    cols, out = _device_batch(torch, org, dr)
    torch.cuda.synchronize()
    ptrs = {k: v.data_ptr() for k, v in out.items() if counters or k not in ("n_int", "n_tri")}
    fn = ctx.trace_closest_noalloc if noalloc else ctx.trace_closest_device
    fn([c.data_ptr() for c in cols], len(org), ptrs, t_min, t_max)
    return {k: out[k].cpu().numpy() for k in ptrs}


def test_noalloc_matches_rt_trace_closest_twice_second_smaller(ctx):
    import torch
    # 262 144 triangles, tree depth 24: deeper than the lean kernels' 20 LDS stack levels, so the overflow strips
    # of the static scratch are exercised; a x2-scaled copy and a plane beside it
    desc = scenes.SceneDescription()
    red, grey = desc.diffuse((1.0, 0.5, 0.5)), desc.diffuse((0.8, 0.8, 0.8))
    desc.add_mesh(scenes.tetrahedron_mesh(8), red)
    desc.add_mesh(scenes.tetrahedron_mesh(8), red, transform=(scenes.IDENTITY * np.float32(2.0), (1.2, 0.3, 0.0)))
    desc.add_plane((0.0, 0.0, -0.6), (0.0, 0.0, 1.0), grey)
    desc.point_light((100.0, 20.0, 100.0), (60000.0, 60000.0, 60000.0))
    loc = scenes.new_polar(1.6, np.float32(0.3 * np.pi), np.float32(0.1))
    desc.look_at(loc, (0.5, 0.1, 0.0), scenes.ASPECT_RATIO)
    scene = rt.Scene(desc)
    assert max(b.max_depth for b in scene.bvhs()) > 20
    ctx.upload(scene)
    org, dr = scenes.pixel_center_rays(scene.flat.camera, 640, 360)
    rng = np.random.default_rng(3)
    small = rng.choice(len(org), 9001, replace=False)
    for counters in (False, True):
        for o, d, rng_t in ((org, dr, (1e-3, np.inf)), (org[small], dr[small], (1e-3, np.inf)), (org[small[:777]], dr[small[:777]], (0.2, 1.5))):
            want = _run(ctx, torch, o, d, False, counters, *rng_t)
            got = _run(ctx, torch, o, d, True, counters, *rng_t)
            assert (want["obj"] != -1).sum() > 100
            for k in want:
                assert np.array_equal(got[k], want[k]), (counters, len(o), k)
    # the property the entry point exists for: free device memory is the same before and after it (every kernel and
    # buffer it can need exists by now; torch's caching allocator hands the arrays below out of memory it already holds)
    cols, out = _device_batch(torch, org, dr)
    ptrs = {k: v.data_ptr() for k, v in out.items()}
    rays6 = [c.data_ptr() for c in cols]
    ctx.trace_closest_noalloc(rays6, len(org), ptrs)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for n in (len(org), 9001, len(org)):
        ctx.trace_closest_noalloc(rays6, n, ptrs)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0, "rt_trace_closest_noalloc changed the free device memory"
    # ... and directly against the oracle, visit counters included (the child-pair walk counts the reference's visits)
    orc = ob.OracleScene(desc)
    sub = small[:4000]
    got = _run(ctx, torch, org[sub], dr[sub], True, True)
    want = orc.trace_closest(org[sub], dr[sub])
    assert np.array_equal(got["prim"].view(np.uint32), want["prim"]) and np.array_equal(got["obj"].view(np.uint32), want["obj"])
    assert np.array_equal(got["t"].view(np.uint32), want["t"].view(np.uint32))
    assert np.array_equal(got["n_int"].view(np.uint32), want["n_int"]) and np.array_equal(got["n_tri"].view(np.uint32), want["n_tri"])
    # two allocation-free calls back to back, the second smaller, each against the existing entry point
    a1 = _run(ctx, torch, org, dr, True, True)
    a2 = _run(ctx, torch, org[small], dr[small], True, True)
    w1 = _run(ctx, torch, org, dr, False, True)
    w2 = _run(ctx, torch, org[small], dr[small], False, True)
    for k in w1:
        assert np.array_equal(a1[k], w1[k]) and np.array_equal(a2[k], w2[k]), k


def test_noalloc_rejects_what_it_cannot_serve(ctx):
    import torch
    scene = rt.Scene(scenes.tetrahedron_scene(levels=3))
    ctx.upload(scene)
    org, dr = scenes.pixel_center_rays(scene.flat.camera, 32, 18)
    cols, out = _device_batch(torch, org, dr)
    torch.cuda.synchronize()
    host_t = np.empty(len(org), np.float32)
    with pytest.raises(rt.RtError) as e:  # a host array would need a staging allocation
        ctx.trace_closest_noalloc([c.data_ptr() for c in cols], len(org), {"t": host_t.ctypes.data})
    assert e.value.code == -1 and "device memory" in str(e.value)
    with pytest.raises(rt.RtError) as e:  # beyond the documented maximum batch
        ctx.trace_closest_noalloc([c.data_ptr() for c in cols], 0xFFFFFFF1, {"t": out["t"].data_ptr()})
    assert e.value.code == -6
    ctx.trace_closest_noalloc([c.data_ptr() for c in cols], 0, {"t": out["t"].data_ptr()})  # empty batch: nothing to do
