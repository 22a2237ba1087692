"""GPU parity: rt_render (the wavefront pipeline, through the C ABI) against the oracle's
Engine::render on the same scene, size, spp, depth and seed.

North-star bar: per-pixel RMSE within a stated tolerance.  Tolerance used here: RMSE <= 1e-6 of
the mean radiance.  Because both sides share the counter-based RNG, the operation order and the
transcendental functions, the tests additionally require the radiance sums to be BIT-IDENTICAL and
the ray counts (Engine::hit / Engine::hit_bool calls) and traversal visit counters to be equal.
"""
import os

import numpy as np
import pytest

import oracle_binding as ob
import ray_tracing_amd as rt
from ray_tracing_amd import scenes

pytestmark = pytest.mark.gpu

RMSE_TOL = 1e-6


def _check(ctx, desc, w, h, spp, depth, seed=0, spp_first=0, max_paths=None):
    scene = rt.Scene(desc)
    ctx.upload(scene)
    ctx.set_max_paths_in_flight(max_paths or 0)
    orc = ob.OracleScene(desc)
    assert ctx.initial_medium() == orc.initial_medium()
    img, st = ctx.render(w, h, spp, depth, seed, spp_first, flags=rt.api.RENDER_COUNT_VISITS)
    ref, ost = orc.render(w, h, spp, depth, seed, spp_first)
    assert np.isfinite(ref).all()
    rmse = float(np.sqrt(np.mean((img.astype(np.float64) - ref) ** 2)))
    assert rmse <= RMSE_TOL * max(1e-30, float(np.abs(ref).mean())), rmse
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    assert st.rays_closest == ost.rays_closest
    assert st.rays_shadow == ost.rays_shadow
    assert st.nodes_visited + st.nodes_visited_shadow == ost.nodes_visited
    assert st.tris_tested + st.tris_tested_shadow == ost.tris_tested
    assert st.paths == w * h * spp
    assert st.rays_shadow_skipped == 0  # counting the reference's visits: every shadow ray is traced
    # the non-counting kernels give the same image -- without tracing the shadow rays whose light term is exactly zero
    # (rays_shadow stays the reference's count of Engine::hit_bool calls)
    img2, st2 = ctx.render(w, h, spp, depth, seed, spp_first)
    assert np.array_equal(img2.view(np.uint32), ref.view(np.uint32))
    assert (st2.rays_closest, st2.rays_shadow) == (ost.rays_closest, ost.rays_shadow)
    assert st2.rays_shadow_skipped <= st2.rays_shadow
    # the fetch-counting kernels are the timed ones with counters: same image, same skips, no more visits than the reference
    img4, st4 = ctx.render(w, h, spp, depth, seed, spp_first, flags=rt.api.RENDER_COUNT_FETCHES)
    assert np.array_equal(img4.view(np.uint32), ref.view(np.uint32))
    assert st4.rays_shadow_skipped == st2.rays_shadow_skipped
    assert st4.nodes_visited + st4.nodes_visited_shadow <= ost.nodes_visited
    assert st4.tris_tested + st4.tris_tested_shadow <= ost.tris_tested
    # frames this small go to the tail kernel at once (every path alive <= its threshold); the pure
    # wavefront pipeline -- one launch per stage and bounce -- must give the same image and counters
    assert st.launches_tail >= 1 and st.paths_tail > 0
    for fl in (rt.api.RENDER_NO_TAIL | rt.api.RENDER_COUNT_VISITS, rt.api.RENDER_NO_TAIL):
        img3, st3 = ctx.render(w, h, spp, depth, seed, spp_first, flags=fl)
        assert st3.launches_tail == 0 and st3.rays_closest_tail == 0
        assert np.array_equal(img3.view(np.uint32), ref.view(np.uint32))
        assert (st3.rays_closest, st3.rays_shadow) == (ost.rays_closest, ost.rays_shadow)
        assert st3.iterations == st.iterations
    assert st3.launches_extend >= st3.iterations
    ctx.set_max_paths_in_flight(0)
    return img, st, ost


def test_config2_scene_point_light(ctx):
    img, st, _ = _check(ctx, scenes.tetrahedron_scene(levels=4), 96, 54, 4, 16)
    assert img.mean() > 0 and st.rays_shadow > 0


def test_mid_size_frame_takes_wavefront_mixed_and_tail_launches(ctx):
    """230 400 paths: above the tail threshold at first (wavefront rounds), few enough that the shadow rays of
    a bounce share a launch with the rays of the next one (k_mixed), then the tail kernel -- against the
    oracle and against the plain wavefront pipeline (inside _check)."""
    img, st, _ = _check(ctx, scenes.path_scene(levels=5, glass_levels=3), 320, 180, 4, 16, seed=9)
    assert st.launches_extend >= 1 and st.launches_tail >= 2
    assert 0 < st.rays_closest_tail < st.rays_closest


def test_config3_scene_area_light_dielectric(ctx):
    desc = scenes.path_scene(levels=5, glass_levels=3)
    img, st, _ = _check(ctx, desc, 96, 54, 4, 16, seed=3)
    assert img.mean() > 0 and st.iterations > 4
    # the diffuse mesh turns part of its surface away from the light: those shadow rays carry a zero term
    for flags in (0, rt.api.RENDER_NO_TAIL):
        _, st2 = ctx.render(96, 54, 4, 16, 3, flags=flags)
        assert 0 < st2.rays_shadow_skipped < st2.rays_shadow // 4


def test_config4_scene_homogeneous_medium(ctx):
    desc = scenes.path_scene(levels=4, glass_levels=3, medium=True, sigma_s=2.0)
    _check(ctx, desc, 80, 45, 4, 16, seed=11)
    _check(ctx, desc, 64, 36, 2, 5, seed=12)  # odd depth: exercises the u32 depth wrap on scatter (:193)


def test_config1_scene_checkered_medium(ctx):
    img, st, _ = _check(ctx, scenes.volumetric_sphere(), 80, 45, 6, 16, seed=5)
    assert img[..., 0].mean() > img[..., 2].mean()  # red emission of the white cells


def test_primitives_scene_conductor_and_two_lights(ctx):
    _check(ctx, scenes.primitives_scene(), 96, 54, 4, 16, seed=2)


def test_spotlight_and_point_light(ctx):
    """Spotlight::sample (lights.rs:57-75, d2 taken of the already normalised direction) next to a PointLight:
    shading points inside and outside the cone, uniform light pick between the two."""
    img, st, _ = _check(ctx, scenes.spotlight_scene(levels=3), 96, 54, 4, 16, seed=21)
    assert img.mean() > 0 and st.rays_shadow > 0


STL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "baby_yoda.stl")


def test_stl_fed_scene_baby_yoda(ctx):
    """scene_baby_yoda (src/main.rs:214-299) from the reference's own asset (committed as a data fixture): binary STL
    -> read_stl with vertex unification (stl.rs:8-143) -> BVH build -> upload -> HIP path, against the oracle, which
    builds its own tree from the same ingested arrays.  As in the reference (sigma_s = 0), then with scattering and
    the subdivided tetrahedron beside it (the tetrahedron is scaled x48)."""
    img, st, _ = _check(ctx, scenes.baby_yoda_scene(STL), 96, 54, 4, 16, seed=7)
    assert img.mean() > 0 and st.iterations > 4
    _check(ctx, scenes.baby_yoda_scene(STL, sigma_s=0.05, extra_mesh_levels=4), 96, 54, 4, 16, seed=8)


def test_stl_fed_scene_teapot_loop_subdivided(ctx):
    """scene_teapot (src/main.rs:165-211): the reference's other asset, an OPEN mesh, read_stl -> loop_subdivide(.., 1)
    (boundary rules of the Loop scheme) -> BVH -> HIP path against the oracle; also two subdivisions (151 008 triangles,
    SURVEY.md's config-3 alternative)."""
    teapot = os.path.join(os.path.dirname(STL), "utah_teapot.stl")
    desc = scenes.teapot_scene(teapot)
    assert rt.Scene(desc).flat.n_tris == 4 * 9438
    img, st, _ = _check(ctx, desc, 96, 54, 4, 16, seed=18)
    assert img.mean() > 0 and st.rays_shadow > 0
    _check(ctx, scenes.teapot_scene(teapot, divisions=2), 96, 54, 2, 16, seed=19)


def test_batching_and_sample_offsets_do_not_change_the_image(ctx):
    desc = scenes.path_scene(levels=3, glass_levels=2)
    w, h = 64, 36
    a, _, _ = _check(ctx, desc, w, h, 8, 8, seed=9)
    b, _, _ = _check(ctx, desc, w, h, 8, 8, seed=9, max_paths=w * h * 3)  # 3+3+2 samples per batch
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # sample sharding: samples [0,4) + [4,8) rendered separately are the two halves of the same set
    scene = rt.Scene(desc)
    ctx.upload(scene)
    lo, _ = ctx.render(w, h, 4, 8, 9, 0)
    hi, _ = ctx.render(w, h, 4, 8, 9, 4)
    orc = ob.OracleScene(desc)
    rlo, _ = orc.render(w, h, 4, 8, 9, 0)
    rhi, _ = orc.render(w, h, 4, 8, 9, 4)
    assert np.array_equal(lo.view(np.uint32), rlo.view(np.uint32))
    assert np.array_equal(hi.view(np.uint32), rhi.view(np.uint32))
    assert np.allclose(lo + hi, a, rtol=1e-5, atol=1e-6)


def test_edge_cases(ctx):
    desc = scenes.tetrahedron_scene(levels=2)
    scene = rt.Scene(desc)
    ctx.upload(scene)
    img, st = ctx.render(16, 9, 2, 0)  # depth 0: `while depth > 0` never runs
    assert not img.any() and st.rays_closest == 0
    img, st = ctx.render(16, 9, 0, 4)  # zero samples
    assert not img.any()
    with pytest.raises(rt.RtError):
        ctx.render(1, 9, 1, 4)  # the reference divides by (width - 1)
    _check(ctx, desc, 33, 19, 3, 1)  # ragged sizes, depth 1
    # no object at all: every ray misses (k_shade's up-front record reads must stay inside the tables)
    empty = scenes.SceneDescription()
    empty.area_light((0.0, 0.0, 5.0), 1.0, (3.0, 3.0, 3.0))
    empty.set_camera((0.0, -4.0, 1.0), (0.0, 4.0, 0.5), scenes.ASPECT_RATIO)
    _check(ctx, empty, 32, 18, 2, 4)


def test_device_output_buffer(ctx):
    import torch
    desc = scenes.tetrahedron_scene(levels=3)
    scene = rt.Scene(desc)
    ctx.upload(scene)
    out = torch.zeros(36, 64, 3, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    ctx.render(64, 36, 3, 8, 1, out=out.data_ptr())
    ref, _ = ob.OracleScene(desc).render(64, 36, 3, 8, 1)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32))


_LISTED = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np
import oracle_binding as ob
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
ctx = rt.Context(0)
stl = os.path.join(sys.argv[1], "tests", "golden", "baby_yoda.stl")
cases = [(scenes.path_scene(levels=5, glass_levels=3), 96, 54, 4, 16, 3),
         (scenes.path_scene(levels=4, glass_levels=3, medium=True, sigma_s=2.0), 80, 45, 4, 16, 11),
         (scenes.volumetric_sphere(), 80, 45, 6, 16, 5),
         (scenes.primitives_scene(), 96, 54, 4, 16, 2),
         (scenes.spotlight_scene(levels=3), 96, 54, 4, 16, 21),
         (scenes.baby_yoda_scene(stl, sigma_s=0.05, extra_mesh_levels=4), 96, 54, 4, 16, 8),
         (scenes.path_scene(levels=5, glass_levels=3), 320, 180, 4, 16, 9)]
for desc, w, h, spp, depth, seed in cases:
    ctx.upload(rt.Scene(desc))
    orc = ob.OracleScene(desc)
    ref, ost = orc.render(w, h, spp, depth, seed)
    for flags in (0, rt.api.RENDER_NO_TAIL, rt.api.RENDER_COUNT_VISITS | rt.api.RENDER_NO_TAIL):
        img, st = ctx.render(w, h, spp, depth, seed, flags=flags)
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (w, h, seed, flags)
        assert (st.rays_closest, st.rays_shadow) == (ost.rays_closest, ost.rays_shadow)
        if flags & rt.api.RENDER_COUNT_VISITS:
            assert st.nodes_visited + st.nodes_visited_shadow == ost.nodes_visited
print("listed ok")
'''


def test_shading_from_the_list_at_every_bounce(tmp_path):
    """k_classify + list-driven k_shade normally serve bounces of 262 144 paths and more (the full-size tests); here
    RT_LIST_PATHS=1 sends EVERY bounce after the first through them, with and without the tail kernel, on scenes with
    every material, light and medium kind -- in a process of its own, because the threshold is read once."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "listed.py"
    script.write_text(_LISTED)
    env = dict(os.environ, RT_LIST_PATHS="1", RT_TAIL_PATHS="64")
    r = subprocess.run([sys.executable, str(script), root], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "listed ok" in r.stdout, r.stderr[-3000:]
