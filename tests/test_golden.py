"""Committed golden vectors (tests/golden/*.npz, written by tests/golden/make_golden.py with the oracle).

CPU: the oracle still reproduces them bit for bit (pins it against drift; the reference itself holds no
fixtures for traversal or rendering -- SURVEY.md 8c -- and cannot be run here).
GPU: the HIP path reproduces them through the C ABI without the oracle being involved at run time.
"""
import glob
import importlib.util
import os

import numpy as np
import pytest

import oracle_binding as ob
import ray_tracing_amd as rt
from ray_tracing_amd import scenes

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(mg)
NAMES = sorted(mg.CASES)


def _load(name):
    g = np.load(os.path.join(HERE, "golden", name + ".npz"), allow_pickle=False)
    w, h, spp, depth, seed = [int(x) for x in g["params"]]
    return g, w, h, spp, depth, seed


def test_every_case_has_a_fixture():
    have = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(HERE, "golden", "*.npz")))
    assert have == NAMES


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(name):
    desc, w, h, spp, depth, seed = mg.build(name)
    g, gw, gh, gspp, gdepth, gseed = _load(name)
    assert (w, h, spp, depth, seed) == (gw, gh, gspp, gdepth, gseed)
    orc = ob.OracleScene(desc)
    img, st = orc.render(w, h, spp, depth, seed, 0)
    assert np.array_equal(img.view(np.uint32), g["image_bits"])
    assert (st.rays_closest, st.rays_shadow) == (int(g["rays_closest"]), int(g["rays_shadow"]))
    org, dr = scenes.pixel_center_rays(rt.Scene(desc).flat.camera, w, h)
    hits = orc.trace_closest(org, dr)
    assert np.array_equal(hits["t"].view(np.uint32), g["hit_t_bits"])
    assert np.array_equal(hits["prim"], g["hit_prim"]) and np.array_equal(hits["obj"], g["hit_obj"])
    assert np.array_equal(orc.trace_any(org, -dr)["occluded"], g["occluded_reverse"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_reproduces_golden(ctx, name):
    desc, w, h, spp, depth, seed = mg.build(name)
    g = _load(name)[0]
    scene = rt.Scene(desc)
    ctx.upload(scene)
    for flags in (0, rt.api.RENDER_NO_TAIL):
        img, st = ctx.render(w, h, spp, depth, seed, 0, flags=flags)
        assert np.array_equal(img.view(np.uint32), g["image_bits"])          # bit-exact radiance sums
        assert (st.rays_closest, st.rays_shadow) == (int(g["rays_closest"]), int(g["rays_shadow"]))
    org, dr = scenes.pixel_center_rays(scene.flat.camera, w, h)
    hits = ctx.trace_closest(org, dr)
    assert np.array_equal(hits["prim"], g["hit_prim"]) and np.array_equal(hits["obj"], g["hit_obj"])   # ids: bit-exact
    assert np.array_equal(hits["t"].view(np.uint32), g["hit_t_bits"])        # north_star asks 1e-5 relative; we hold 0 ulp
    assert np.array_equal(hits["front_face"], g["hit_front_face"])
    assert np.array_equal(ctx.trace_any(org, -dr)["occluded"], g["occluded_reverse"])
