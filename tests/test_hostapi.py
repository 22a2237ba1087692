"""C++ host mirror of the crate API (ray_tracer.hpp) and the example program built on it.

CPU: the header compiles and links against librt_amd.so (no compute calls).
GPU: the example binary renders the reference's default scene (src/main.rs:25-83) and its PPM is
byte-identical to the Python host path's render of scenes.volumetric_sphere with the same seed.
"""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "examples")


def _build():
    subprocess.run(["make", "-C", EX, "rt_main"], check=True, capture_output=True)
    return os.path.join(EX, "rt_main")


def test_example_compiles_and_links():
    exe = _build()
    assert os.access(exe, os.X_OK)
    out = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "librt_amd.so" in out and "not found" not in out.split("librt_amd.so")[1].split("\n")[0]


@pytest.mark.gpu
def test_example_matches_python_host(ctx, tmp_path):
    from ray_tracing_amd import api, scenes

    exe = os.path.join(EX, "rt_main")
    assert os.path.exists(exe), "examples/rt_main must be built in-tree before the GPU run"
    w, spp, depth, seed = 160, 8, 16, 7
    r = subprocess.run([exe, "volumetric_sphere", str(w), str(spp), str(depth), str(seed)], capture_output=True, timeout=100)
    assert r.returncode == 0, r.stderr.decode()
    h = scenes.image_height(w)
    header = b"P3\n%d %d\n255\n" % (w, h)
    assert r.stdout.startswith(header)
    got = np.array(r.stdout[len(header):].split(), dtype=np.uint8).reshape(h, w, 3)

    scene = api.Scene(scenes.volumetric_sphere())
    ctx.upload(scene)
    acc, _ = ctx.render(w, h, spp, depth, seed)
    want = api.tonemap_u8(acc, spp)
    assert np.array_equal(got, want.reshape(h, w, 3))
