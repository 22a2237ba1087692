"""Host-side scene preparation (rt_mesh_* / rt_scene_build / tonemap) on the CPU.

The product's BVH builder and the oracle's builder are two independent restatements of
ray_tracer/src/render/bvh.rs:60-241; they must produce the same node array and `indices`
permutation bit for bit.  Subdivision is pinned by the reference's own topology test vectors.
"""
import os
import struct

import numpy as np
import pytest

import oracle_binding as ob
import ray_tracing_amd as rt
from ray_tracing_amd import scenes

REF_MESHES = "/root/reference/src/meshes"


def _same_bvh(scene, orc, obj=0, bvh=0):
    b = orc.bvh(obj)
    n = scene.nodes()
    f = scene.bvhs()[bvh]
    sl = slice(f.node_offset, f.node_offset + f.node_count)
    for k in ("upper", "lower"):
        assert np.array_equal(n[k][sl].view(np.uint32), b[k].view(np.uint32)), k
    assert np.array_equal(n["index"][sl], b["index"])
    assert np.array_equal(n["num_prim"][sl], b["num_prim"])
    assert np.array_equal(scene.prim_ids()[f.tri_offset:f.tri_offset + f.tri_count], b["indices"])
    return f


@pytest.mark.parametrize("levels", [0, 1, 3, 6])
def test_bvh_builder_matches_oracle_builder(levels):
    desc = scenes.tetrahedron_scene(levels=levels)
    f = _same_bvh(rt.Scene(desc), ob.OracleScene(desc))
    assert f.tri_count == 4 ** (levels + 1)
    assert f.node_count <= 2 * f.tri_count - 1  # bvh.rs:62-66


@pytest.mark.parametrize("threads", ["2", "5"])
def test_parallel_bvh_build_is_node_for_node_the_serial_one(threads, monkeypatch):
    """262 144 triangles take the parallel path (upper tree split by all threads,
    subtrees <= 16 384 primitives as tasks, depth-first renumbering); nodes, boxes and the `indices`
    permutation must equal the oracle's serial restatement of bvh.rs:60-241 bit for bit."""
    monkeypatch.setenv("RT_BUILD_THREADS", threads)
    desc = scenes.tetrahedron_scene(levels=8)
    f = _same_bvh(rt.Scene(desc), ob.OracleScene(desc))
    assert f.tri_count == 4 ** 9


def test_parallel_bvh_build_on_soup_with_unsplittable_big_nodes(monkeypatch):
    """40 000 random triangles of which 36 000 are copies of one: an upper-tree node that cannot be split."""
    monkeypatch.setenv("RT_BUILD_THREADS", "4")
    rng = np.random.default_rng(11)
    v = rng.uniform(-1, 1, (3000, 3)).astype(np.float32)
    t = rng.integers(0, 3000, (40000, 3)).astype(np.uint32)
    t[2000:38000] = t[2000]
    mesh = rt.Mesh.from_arrays(v, t)
    desc = rt.SceneDescription()
    desc.add_mesh(mesh, desc.diffuse((1, 1, 1)))
    desc.point_light((0, 0, 5), (1, 1, 1))
    desc.set_camera((0, -4, 0), (0, 1, 0), scenes.ASPECT_RATIO)
    _same_bvh(rt.Scene(desc), ob.OracleScene(desc))


def test_bvh_builder_on_irregular_soup():
    """Random triangle soup with duplicated and degenerate triangles (empty bins, failed partitions)."""
    rng = np.random.default_rng(5)
    v = rng.uniform(-1, 1, (600, 3)).astype(np.float32)
    v[100:140] = v[100]            # coincident vertices -> zero-extent centroid boxes
    t = rng.integers(0, 600, (900, 3)).astype(np.uint32)
    t[50:80] = t[50]               # identical triangles: the SAH cannot separate them
    mesh = rt.Mesh.from_arrays(v, t)
    desc = rt.SceneDescription()
    desc.add_mesh(mesh, desc.diffuse((1, 1, 1)))
    desc.point_light((0, 0, 5), (1, 1, 1))
    desc.set_camera((0, -4, 0), (0, 1, 0), scenes.ASPECT_RATIO)
    scene = rt.Scene(desc)
    f = _same_bvh(scene, ob.OracleScene(desc))
    leaves = scene.nodes()["num_prim"][f.node_offset:f.node_offset + f.node_count]
    assert leaves.max() > 1        # a multi-primitive leaf exists (no max leaf size in the reference)
    assert leaves.sum() == 900


def test_loop_subdivision_reference_topology_vector():
    """subdivision_surface.rs:419-447 `subdivision`: one triangle split once."""
    m = rt.Mesh.from_arrays(np.zeros((3, 3), np.float32), [[0, 1, 2]])
    v, t = m.loop_subdivide(1).arrays()
    assert len(v) == 6
    assert t.tolist() == [[0, 3, 5], [3, 1, 4], [5, 4, 2], [3, 4, 5]]


def test_loop_subdivision_boundary_fan():
    """The mesh of subdivision_surface.rs:399-416 (`vertex_neighbours`): 4 boundary vertices + 1 interior."""
    v = np.array([[1, 0, 0], [0, 1, 0], [-1, 0, 0], [0, -1, 0], [0, 0, 1]], np.float32)
    t = [[0, 1, 4], [1, 2, 4], [2, 3, 4], [3, 0, 4]]
    sv, st = rt.Mesh.from_arrays(v, t).loop_subdivide(1).arrays()
    assert len(st) == 16 and len(sv) == 5 + 8
    # boundary rule (:306-311, valence 3 -> weight 3): corner (1,0,0) with boundary neighbours 1 and 3
    # first subdivide(): (6*p + n0 + n1)/8, then limit: (3*p' + n0' + n1')/5 -- just check symmetry and finiteness
    assert np.isfinite(sv).all()
    assert np.allclose(sv[0][[0]], -sv[2][[0]], atol=1e-6)


def test_tetrahedron_counts_and_closedness():
    m = scenes.tetrahedron_mesh(4)
    nv, nt = m.counts()
    assert nt == 4 ** 5 and nv == nt // 2 + 2  # V = F/2 + 2 for a closed genus-0 surface
    v, t = m.arrays()
    edges = {}
    for a, b, c in t.tolist():
        for e in ((a, b), (b, c), (c, a)):
            edges[e] = edges.get(e, 0) + 1
    assert all(cnt == 1 and (e[1], e[0]) in edges for e, cnt in edges.items())  # consistent orientation


def _write_stl(path, tris, trailing=b""):
    with open(path, "wb") as f:
        f.write(b"x" * 80 + struct.pack("<I", len(tris)))
        for tri in tris:
            f.write(struct.pack("<3f", 0, 0, 0))
            for p in tri:
                f.write(struct.pack("<3f", *p))
            f.write(b"\0\0")
        f.write(trailing)


def test_stl_reader_unifies_vertices_on_bit_patterns(tmp_path):
    tris = [[(0, 0, 0), (1, 0, 0), (0, 1, 0)], [(1, 0, 0), (1, 1, 0), (0, 1, 0)], [(0.0, 0, 0), (-0.0, 0, 0), (0, 0, 1)]]
    p = str(tmp_path / "a.stl")
    _write_stl(p, tris)
    v, t = rt.Mesh.read_stl(p).arrays()
    assert t.tolist() == [[0, 1, 2], [1, 3, 2], [0, 4, 5]]  # -0.0 has a different bit pattern than 0.0 (stl.rs:110-112)
    assert len(v) == 6
    _write_stl(p, tris, trailing=b"junk")
    with pytest.raises(rt.RtError) as e:  # stl.rs:89-99
        rt.Mesh.read_stl(p)
    assert e.value.code == -4
    with pytest.raises(rt.RtError):
        rt.Mesh.read_stl(str(tmp_path / "missing.stl"))
    with open(p, "wb") as f:
        f.write(b"x" * 80 + struct.pack("<I", 5) + b"\0" * 60)
    with pytest.raises(rt.RtError):
        rt.Mesh.read_stl(p)
    # a header that announces 2^32 - 1 triangles must be refused from the file's size, before anything is reserved
    with open(p, "wb") as f:
        f.write(b"x" * 80 + struct.pack("<I", 0xFFFFFFFF) + b"\0" * 100)
    with pytest.raises(rt.RtError) as e:
        rt.Mesh.read_stl(p)
    assert e.value.code == -4 and "truncated" in str(e.value)


def test_distinct_meshes_get_their_own_trees():
    """configs[4] as the reference builds it (src/main.rs:149-159): one mesh and one BVH per object."""
    scene = rt.Scene(scenes.multi_mesh_scene(levels=3, copies=4, distinct=True))
    b = scene.bvhs()
    assert scene.flat.n_bvhs == 4 and scene.flat.n_tris == 4 * 256
    nodes = scene.nodes()
    roots = [nodes["upper"][x.node_offset].tolist() for x in b]
    assert len({tuple(r) for r in roots}) == 4  # rotated / stretched copies: different bounds, different trees


@pytest.mark.skipif(not os.path.isdir(REF_MESHES), reason="reference assets not mounted (GPU box)")
def test_stl_reader_on_the_reference_assets():
    """Counts recorded in SURVEY.md section 2 for src/meshes/*.stl (read in place, never copied)."""
    assert rt.Mesh.read_stl(os.path.join(REF_MESHES, "utah_teapot.stl")).counts() == (4719, 9438)
    assert rt.Mesh.read_stl(os.path.join(REF_MESHES, "baby_yoda.stl")).counts() == (1957, 4120)


def test_scene_build_object_new_and_plane_new():
    desc = scenes.path_scene(levels=2, glass_levels=1, medium=True)
    scene = rt.Scene(desc)
    flat = scene.flat
    assert flat.n_objects == 3 and flat.n_bvhs == 2
    glass = flat.objects[1]
    assert glass.has_transform == 1
    s = np.float32(0.8)
    inv = np.float32(1.0) / (s * (s * s - np.float32(0)) )  # det of diag(s): matrix.rs:49-54 order
    assert abs(glass.inv_matrix[0] - 1.25) < 1e-6 and glass.inv_matrix[1] == 0.0
    assert abs(glass.inv_translation[1] - 0.55 / 0.8) < 1e-6
    plane = flat.objects[2]
    assert list(plane.surf[:4]) == [0.0, 0.0, 1.0, np.float32(0.32)]  # normal, d = -p.n
    assert glass.medium_inside == 0 and glass.medium_outside == -1
    with pytest.raises(rt.RtError):  # singular transform: Matrix::inverse -> None -> unwrap panics
        d2 = rt.SceneDescription()
        d2.add_sphere((0, 0, 0), 1.0, d2.diffuse((1, 1, 1)), transform=(np.zeros((3, 3), np.float32), (0, 0, 0)))
        d2.set_camera((0, -4, 0), (0, 1, 0), 1.0)
        rt.Scene(d2)


def test_instanced_meshes_share_one_bvh():
    scene = rt.Scene(scenes.multi_mesh_scene(levels=2, copies=4))
    assert scene.flat.n_bvhs == 1 and scene.flat.n_objects == 5
    assert [scene.flat.objects[i].bvh for i in range(1, 5)] == [0, 0, 0, 0]


def test_tonemap_write_pixel_semantics(tmp_path):
    """path_integrator.rs:19-27 and :309-312."""
    acc = np.zeros((2, 2, 3), np.float32)
    acc[0, 0] = [4.0, 1.0, 0.25]      # /spp(4) -> 1, .25, .0625 -> sqrt 1, .5, .25 -> clamp .999 -> 255,128,64
    acc[0, 1] = [np.nan, 1.0, 1.0]    # NaN guard -> (255,255,0)/4 -> sqrt(63.75) clamps -> 255,255,0
    acc[1, 0] = [-1.0, 0.0, 1e-12]    # sqrt(-x) = NaN -> `as u8` = 0
    acc[1, 1] = [0.04, 0.36, 16.0]
    img = rt.api.tonemap_u8(acc, 4)
    assert img[1, 0].tolist() == [255, 128, 64]   # rows are written top to bottom = j descending
    assert img[1, 1].tolist() == [255, 255, 0]
    assert img[0, 0].tolist() == [0, 0, 0]
    assert img[0, 1].tolist() == [int(256 * 0.1), int(256 * 0.3), 255]
    p = str(tmp_path / "o.ppm")
    rt.api.write_ppm(p, img)
    lines = open(p).read().split("\n")
    assert lines[:3] == ["P3", "2 2", "255"] and lines[3] == "0 0 0" and len(lines) == 8


def test_camera_new_matches_oracle_rays():
    desc = scenes.path_scene(levels=1, glass_levels=1)
    scene = rt.Scene(desc)
    orc = ob.OracleScene(desc)
    import ctypes as C
    u = np.array([0.0, 0.5, 1.0, 0.25], np.float32)
    v = np.array([0.0, 0.5, 1.0, 0.75], np.float32)
    outs = [np.empty(4, np.float32) for _ in range(6)]
    ob.lib.orc_camera_rays(C.c_void_p(orc.handle), C.c_void_p(u.ctypes.data), C.c_void_p(v.ctypes.data), C.c_size_t(4),
                           *[C.c_void_p(o.ctypes.data) for o in outs])
    cam = scene.flat.camera
    o = np.array(cam.origin[:], np.float32)
    for k in range(4):
        d = (np.array(cam.lower_left_corner[:], np.float32) + u[k] * np.array(cam.horizontal[:], np.float32)
             + v[k] * np.array(cam.vertical[:], np.float32) - o).astype(np.float32)
        d = d / np.linalg.norm(d)
        assert np.allclose([outs[3][k], outs[4][k], outs[5][k]], d, atol=2e-7)
        assert [outs[0][k], outs[1][k], outs[2][k]] == o.tolist()
