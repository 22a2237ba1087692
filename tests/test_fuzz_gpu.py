"""GPU parity on RANDOM scenes: every surface kind (mesh, sphere, plane, triangle), with and without an affine
transform (rotation x non-uniform scale), every material (Diffuse, Dielectric, Conductor), every light (Point,
Spot, Area), homogeneous and checkered media behind medium interfaces -- drawn from a seed, rendered through
rt_render and through the oracle's Engine::render: radiance sums bit-identical, ray counts and visit counters
equal (tests/test_render_gpu.py::_check, which also runs the non-counting, fetch-counting and no-tail variants).
The fixed scenes of the other tests are the reference's own; these cover the combinations nobody wrote down."""
import os

import numpy as np
import pytest

import ray_tracing_amd as rt
from ray_tracing_amd import scenes
from test_render_gpu import _check

pytestmark = pytest.mark.gpu


def _rotation(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    a, b, c, d = q
    return np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                     [2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b)],
                     [2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d]], np.float32)


def random_scene(seed):
    rng = np.random.default_rng(seed)
    s = scenes.SceneDescription()
    mats = [s.diffuse(tuple(rng.uniform(0.05, 1.0, 3))), s.diffuse(tuple(rng.uniform(0.3, 0.9, 3))),
            s.dielectric(float(rng.choice([1.0, 1.33, 1.5, 2.4]))),
            s.conductor(tuple(rng.uniform(0.2, 2.0, 3)), tuple(rng.uniform(1.0, 4.0, 3)))]
    media = [s.homogeneous_medium(tuple(rng.uniform(0.0, 0.3, 3)), tuple(rng.uniform(0.0, 0.4, 3)),
                                  float(rng.uniform(-0.6, 0.6)), tuple(rng.uniform(0.0, 0.05, 3))),
             s.checkered_medium(float(rng.uniform(0.2, 0.6)), 0.3, 0.2, (0.05, 0.0, 0.0), 0.05, 0.6, (0.0, 0.02, 0.0),
                                float(rng.uniform(-0.3, 0.3)))]
    meshes = [scenes.tetrahedron_mesh(l) for l in (0, 1, 2, 3)]

    def transform(pos):
        k = rng.integers(0, 4)
        if k == 0:
            return None
        m = _rotation(rng) if k >= 2 else np.eye(3, dtype=np.float32)
        sc = rng.uniform(0.4, 1.6, 3) if k == 3 else np.full(3, rng.uniform(0.4, 1.6))
        return ((m * sc[None, :]).astype(np.float32), tuple(np.asarray(pos, np.float32)))

    n_obj = int(rng.integers(2, 8))
    for i in range(n_obj):
        pos = rng.uniform(-2.5, 2.5, 3) * np.array([1.0, 1.0, 0.5])
        mat = int(rng.choice(mats))
        # a medium interface only on closed surfaces with a dielectric skin (what the reference's scenes do)
        medium = (int(rng.choice(media)), None) if mat == mats[2] and rng.random() < 0.6 else None
        kind = rng.integers(0, 4)
        tr = transform(pos)
        at = (0.0, 0.0, 0.0) if tr is not None else tuple(pos)
        if kind == 0:
            mesh = meshes[int(rng.integers(0, 4))]
            if tr is None:  # a mesh is placed by its transform
                tr = (np.eye(3, dtype=np.float32), tuple(np.asarray(pos, np.float32)))
            s.add_mesh(mesh, mat, transform=tr, medium=medium)
        elif kind == 1:
            s.add_sphere(at, float(rng.uniform(0.3, 1.0)), mat, transform=tr, medium=medium)
        elif kind == 2:
            v = rng.uniform(-1.0, 1.0, (3, 3)) + np.asarray(at)[None, :]
            s.add_triangle(tuple(v[0]), tuple(v[1]), tuple(v[2]), int(rng.choice(mats[:2] + mats[3:])), transform=tr)
        else:
            n = rng.normal(size=3) * np.array([0.3, 0.3, 1.0])
            n /= np.linalg.norm(n)
            s.add_plane((0.0, 0.0, float(rng.uniform(-3.0, -1.5))), tuple(n), int(rng.choice(mats[:2])))
    for _ in range(int(rng.integers(1, 4))):
        kind = rng.integers(0, 3)
        pos = rng.uniform(-6.0, 6.0, 3) + np.array([0.0, 0.0, 8.0])
        if kind == 0:
            s.point_light(tuple(pos), tuple(rng.uniform(50.0, 400.0, 3)))
        elif kind == 1:
            aim = -pos + rng.normal(scale=0.5, size=3)
            aim = (aim / np.linalg.norm(aim)).astype(np.float32)
            s.spotlight(tuple(pos), tuple(aim), float(rng.uniform(0.7, 0.97)), tuple(rng.uniform(1.0, 5.0, 3)))
        else:
            s.area_light(tuple(pos * 3.0), float(rng.uniform(1.0, 6.0)), tuple(rng.uniform(1.0, 6.0, 3)))
    loc = rng.uniform(-1.0, 1.0, 3) + np.array([0.0, -7.0, 2.0])
    s.look_at(tuple(loc), tuple(rng.uniform(-0.5, 0.5, 3)), scenes.ASPECT_RATIO)
    return s


@pytest.mark.parametrize("seed", range(int(os.environ.get("RT_FUZZ_SEEDS", "64"))))  # RT_FUZZ_SEEDS=1000: a longer run by hand
def test_random_scene_matches_the_oracle(ctx, seed):
    rng = np.random.default_rng(1000 + seed)
    depth = int(rng.choice([1, 2, 5, 8, 16]))
    spp = int(rng.choice([1, 3, 4]))
    _check(ctx, random_scene(seed), 96, 54, spp, depth, seed=seed, spp_first=int(rng.integers(0, 3)))


@pytest.mark.parametrize("seed", range(100, 108))
def test_random_scene_mid_size_frame(ctx, seed):
    """The same at 320 x 180 x 3 spp (172 800 paths): above the tail kernel's threshold at first, so the random
    materials, media and lights also pass through the wavefront rounds, the mixed launches, the moving compaction
    into the dense regions and then the tail kernel."""
    img, st, _ = _check(ctx, random_scene(seed), 320, 180, 3, 12, seed=seed)
    assert st.launches_extend >= 1 and st.launches_shadow >= 1 and st.launches_tail >= 2  # mixed launch(es) + tail kernel


@pytest.mark.parametrize("seed", range(200, 232))
def test_random_scene_intersector(ctx, seed):
    """rt_trace_closest / rt_trace_any on the random scenes: random rays over random t ranges (finite, infinite, a
    negative t_min), the culling kernels and both counting kernels against the oracle's Engine::hit / hit_bool."""
    import oracle_binding as ob
    rng = np.random.default_rng(seed)
    desc = random_scene(seed)
    scene = rt.Scene(desc)
    ctx.upload(scene)
    orc = ob.OracleScene(desc)
    n = 3000
    org = rng.uniform(-4.0, 4.0, (n, 3)).astype(np.float32)
    tgt = rng.uniform(-2.5, 2.5, (n, 3)).astype(np.float32)
    dr = tgt - org
    dr = (dr / np.linalg.norm(dr, axis=1, keepdims=True)).astype(np.float32)
    dr[: n // 10] *= np.float32(rng.uniform(0.3, 3.0))  # unnormalised directions are taken as they are
    o2, d2 = scenes.pixel_center_rays(scene.flat.camera, 48, 27)
    org, dr = np.concatenate([org, o2]).astype(np.float32), np.concatenate([dr, d2]).astype(np.float32)
    t_min, t_max = [(1e-3, np.inf), (0.0, float(rng.uniform(2.0, 9.0))), (float(rng.uniform(0.5, 3.0)), np.inf),
                    (-5.0, float(rng.uniform(1.0, 6.0)))][seed % 4]
    o = orc.trace_closest(org, dr, t_min, t_max)
    oa = orc.trace_any(org, dr, t_min, t_max)
    for counters, mode in ((False, None), (True, rt.api.COUNT_REFERENCE), (True, rt.api.COUNT_FETCHES)):
        if mode is not None:
            ctx.set_visit_count_mode(mode)
        g = ctx.trace_closest(org, dr, t_min, t_max, counters=counters)
        for k in ("obj", "prim", "front_face"):
            assert np.array_equal(g[k], o[k]), (counters, mode, k)
        for k in ("t", "nx", "ny", "nz"):
            assert np.array_equal(g[k].view(np.uint32), o[k].view(np.uint32)), (counters, mode, k)
        ga = ctx.trace_any(org, dr, t_min, t_max, counters=counters)
        assert np.array_equal(ga["occluded"], oa["occluded"]), (counters, mode)
        if mode == rt.api.COUNT_REFERENCE:
            assert np.array_equal(g["n_int"], o["n_int"]) and np.array_equal(g["n_tri"], o["n_tri"])
            assert np.array_equal(ga["n_int"], oa["n_int"]) and np.array_equal(ga["n_tri"], oa["n_tri"])
    ctx.set_visit_count_mode(rt.api.COUNT_REFERENCE)
