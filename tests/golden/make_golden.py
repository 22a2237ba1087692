"""Generates tests/golden/*.npz with the CPU oracle (oracle/, the C++ restatement of the reference).

The reference itself cannot produce vectors here (Rust toolchain absent) and holds no
fixtures for traversal or rendering, so these files pin the ORACLE against drift (compiler, flags,
later edits) and give the GPU tests expected outputs that do not depend on running the oracle.  The
reference's own known-answer tests are restated inline in tests/test_oracle_golden.py.

    python tests/golden/make_golden.py [name ...]   # rewrites the fixtures (all, or the named ones); commit the result
"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_binding as ob
import ray_tracing_amd as rt
from ray_tracing_amd import scenes

CASES = {  # name -> (scene description factory, width, spp, depth, seed)
    "config1_volumetric_sphere": (lambda: scenes.volumetric_sphere(), 48, 4, 16, 11),
    "config2_tetrahedron_point_light": (lambda: scenes.tetrahedron_scene(levels=3), 64, 2, 16, 12),
    "config3_area_light_dielectric": (lambda: scenes.path_scene(levels=3, glass_levels=2), 64, 4, 16, 13),
    "config4_homogeneous_medium": (lambda: scenes.path_scene(levels=3, glass_levels=2, medium=True, sigma_s=2.0), 48, 4, 16, 14),
    "primitives_conductor": (lambda: scenes.primitives_scene(), 64, 4, 16, 15),
    "stl_baby_yoda_medium": (lambda: scenes.baby_yoda_scene(os.path.join(HERE, "baby_yoda.stl"), sigma_s=0.05, extra_mesh_levels=3), 64, 4, 16, 17),
    "stl_utah_teapot_subdivided": (lambda: scenes.teapot_scene(os.path.join(HERE, "utah_teapot.stl")), 64, 4, 16, 18),
    "spotlight_and_point_light": (lambda: scenes.spotlight_scene(levels=3), 64, 4, 16, 16),
}


def build(name):
    factory, w, spp, depth, seed = CASES[name]
    desc = factory()
    h = scenes.image_height(w)
    return desc, w, h, spp, depth, seed


def main():
    for name in (sys.argv[1:] or CASES):
        desc, w, h, spp, depth, seed = build(name)
        orc = ob.OracleScene(desc)
        img, st = orc.render(w, h, spp, depth, seed, 0)
        scene = rt.Scene(desc)   # host-side only: camera for the ray batch
        org, dr = scenes.pixel_center_rays(scene.flat.camera, w, h)
        hits = orc.trace_closest(org, dr)
        occ = orc.trace_any(org, -dr)
        np.savez_compressed(os.path.join(HERE, name + ".npz"),
                            image_bits=img.view(np.uint32), rays_closest=np.uint64(st.rays_closest),
                            rays_shadow=np.uint64(st.rays_shadow), params=np.array([w, h, spp, depth, seed], np.int64),
                            hit_t_bits=hits["t"].view(np.uint32), hit_prim=hits["prim"], hit_obj=hits["obj"],
                            hit_front_face=hits["front_face"], occluded_reverse=occ["occluded"])
        print(name, img.shape, "rays", st.rays_closest, st.rays_shadow, "hits", int((hits["obj"] != 0xFFFFFFFF).sum()))


if __name__ == "__main__":
    main()
