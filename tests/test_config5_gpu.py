"""configs[4] pattern (several mesh objects, each its own BVH instance with a translation,
src/main.rs:149-159) at a size the oracle renders in seconds."""
import numpy as np
import pytest

import oracle_binding as ob
import ray_tracing_amd as rt
from ray_tracing_amd import scenes

pytestmark = pytest.mark.gpu


def test_multi_mesh_scene_matches_oracle(ctx):
    desc = scenes.multi_mesh_scene(levels=4, copies=6)
    scene = rt.Scene(desc)
    ctx.upload(scene)
    orc = ob.OracleScene(desc)
    org, dr = scenes.pixel_center_rays(scene.flat.camera, 192, 108)
    g, o = ctx.trace_closest(org, dr, counters=True), orc.trace_closest(org, dr)
    assert len(np.unique(o["obj"])) >= 7
    for k in ("obj", "prim", "n_int", "n_tri"):
        assert np.array_equal(g[k], o[k])
    assert np.array_equal(g["t"].view(np.uint32), o["t"].view(np.uint32))
    img, st = ctx.render(128, 72, 4, 16, 5)
    ref, ost = orc.render(128, 72, 4, 16, 5)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    assert (st.rays_closest, st.rays_shadow) == (ost.rays_closest, ost.rays_shadow)


def _rotation(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], np.float32)


def test_top_level_cull_changes_nothing(ctx):
    """The flat top level over objects (DScene::wbox, device_trace.h) leaves objects whose world bounds a
    ray misses out of Engine::hit's loop (path_integrator.rs:88-104).  70 objects -- more than the 64 the
    candidate mask covers -- with rotated, scaled and translated meshes, a leaf-root mesh, planes and a
    sphere in between; rays through pixel centres, random rays and rays aimed at the corners of every
    object's world bounds.  The culling kernels (no counters) must give the oracle's bits; so must the
    counting kernels, which do not cull."""
    rng = np.random.default_rng(11)
    desc = scenes.SceneDescription()
    mats = [desc.diffuse((0.8, 0.3, 0.2)), desc.dielectric(1.5), desc.diffuse((0.2, 0.3, 0.8))]
    small, big, single = scenes.tetrahedron_mesh(1), scenes.tetrahedron_mesh(3), scenes.tetrahedron_mesh(0)
    corners = []
    for i in range(70):
        pos = np.array([(i % 10 - 4.5) * 1.1, (i // 10 - 3.0) * 1.1, 0.3 * np.sin(i)], np.float32)
        if i in (5, 66):
            desc.add_plane((0.0, 0.0, -1.0 - 0.01 * i), (0.0, 0.0, 1.0), mats[0])
            continue
        if i == 17:
            desc.add_sphere(tuple(pos), 0.3, mats[2])
            continue
        mesh = single if i % 7 == 3 else (big if i % 3 == 0 else small)
        m = _rotation(rng) * np.float32(rng.uniform(0.3, 0.9))
        tr = None if i % 5 == 0 else (m, tuple(pos))
        desc.add_mesh(mesh, mats[i % 3], transform=tr)
        verts = mesh.arrays()[0]
        lo, hi = verts.min(0), verts.max(0)
        for c in range(8):  # corners of the object-space root box, taken to world space
            v = np.array([hi[a] if c >> a & 1 else lo[a] for a in range(3)], np.float32)
            corners.append(pos + m @ v if tr is not None else v)
    desc.area_light((0.0, 0.0, 30.0), 3.0, (5.0, 5.0, 5.0))
    loc = np.array([0.5, -9.0, 5.0], np.float32)
    desc.set_camera(loc, -loc, scenes.ASPECT_RATIO)
    scene = rt.Scene(desc)
    ctx.upload(scene)
    orc = ob.OracleScene(desc)

    org, dr = scenes.pixel_center_rays(scene.flat.camera, 160, 90)
    n = 6000
    o2 = rng.uniform(-6.0, 6.0, size=(n, 3)).astype(np.float32)
    d2 = rng.normal(size=(n, 3)).astype(np.float32)
    tgt = np.array(corners, np.float32)
    o3 = np.repeat(loc[None, :], len(tgt), 0) + rng.normal(scale=0.5, size=tgt.shape).astype(np.float32)
    d3 = tgt - o3
    org = np.concatenate([org, o2, o3]).astype(np.float32)
    dr = np.concatenate([dr, d2, d3]).astype(np.float32)
    dr = (dr / np.linalg.norm(dr, axis=1, keepdims=True)).astype(np.float32)

    o = orc.trace_closest(org, dr)
    assert len(np.unique(o["obj"])) >= 15 and ((o["obj"] >= 64) & (o["obj"] != rt.api.RT_MISS)).any()
    for counters in (False, True):
        g = ctx.trace_closest(org, dr, counters=counters)
        for k in ("obj", "prim", "front_face"):
            assert np.array_equal(g[k], o[k]), (counters, k)
        for k in ("t", "nx", "ny", "nz"):
            assert np.array_equal(g[k].view(np.uint32), o[k].view(np.uint32)), (counters, k)
    p = (org + dr * np.minimum(o["t"], 20.0)[:, None] * np.float32(0.999)).astype(np.float32)
    dl = rng.normal(size=p.shape).astype(np.float32)
    dl = (dl / np.linalg.norm(dl, axis=1, keepdims=True)).astype(np.float32)
    assert np.array_equal(ctx.trace_any(p, dl)["occluded"], orc.trace_any(p, dl)["occluded"])

    img, st = ctx.render(96, 54, 4, 16, 9)
    ref, ost = orc.render(96, 54, 4, 16, 9)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    assert (st.rays_closest, st.rays_shadow) == (ost.rays_closest, ost.rays_shadow)
