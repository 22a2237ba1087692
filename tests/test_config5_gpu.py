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
