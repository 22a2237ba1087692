"""Pins the oracle (oracle/) against every known-answer test the reference holds for the hot path
(SURVEY.md 8c).  Each case restates the inputs and expected outputs of a `#[test]` in the reference
(file:line cited); values are data, not code.  What the reference does NOT pin (traverse_bvh, the
SAH builder, triangle_intersection, ray_colour, BSDFs, lights, media) is "parity unpinned" there and
is covered by analytic known-answer tests added below.
"""
import ctypes as C
import math

import numpy as np
import pytest

import oracle_binding as ob

lib = ob.lib
f3, f9 = ob.f3, ob.f9


def aabb_hit(a, b, o, d, tmin, tmax):
    t = C.c_float()
    r = lib.orc_aabb_hit(f3(a), f3(b), f3(o), f3(d), C.c_float(tmin), C.c_float(tmax), C.byref(t))
    return t.value if r else None


def sphere_hit(c, r, o, d, tmin, tmax):
    t, n, ff = C.c_float(), (C.c_float * 3)(), C.c_int()
    ok = lib.orc_sphere_hit(f3(c), C.c_float(r), f3(o), f3(d), C.c_float(tmin), C.c_float(tmax), C.byref(t), n, C.byref(ff))
    return (t.value, np.array(n[:]), bool(ff.value)) if ok else None


def vec_op(op, a, b=(0, 0, 0), s=0.0):
    out = (C.c_float * 3)()
    lib.orc_vec_op(op, f3(a), f3(b), C.c_float(s), out)
    return list(out)


# ---- geometry/primitives.rs:324-377
def test_hit_sphere():  # :330-337
    s3 = np.float32(math.sqrt(3.0))
    d = [float(np.float32(1.0) * (np.float32(1.0) / np.float32(math.sqrt(3.0))))] * 3
    t, n, ff = sphere_hit((1, 1, 1), 1.0, (0, 0, 0), d, 0.0, math.inf)
    assert abs(float(s3) - 1.0 - t) < 1e-6
    assert abs(np.linalg.norm(n) - 1.0) < 1e-6
    assert ff


def test_hit_sphere_tangent():  # :340-347
    r = float(np.float32(1.0) / np.float32(math.sqrt(2.0)))
    t, n, _ = sphere_hit((1, 1, 1), 1.0, (0, 0, 0), (r, r, 0.0), 0.0, math.inf)
    assert abs(math.sqrt(2.0) - t) < 1e-6
    assert abs(np.linalg.norm(n) - 1.0) < 1e-6


def test_miss_sphere():  # :350-355
    assert sphere_hit((1, 1, 1), 1.0, (0, 0, 0), (0, 0, 1), 0.0, math.inf) is None


def test_hit_bounding_box():  # :358-366 -- exact Some(1.0) / None, un-normalised direction
    assert aabb_hit((1, 0, 0), (2, 2, 2), (0, 0, 0), (1, 1, 1), 0.0, 10.0) == 1.0
    assert aabb_hit((1, 0, 0), (2, 2, 2), (0, 0, 0), (-1, -1, -1), 0.0, 10.0) is None


def test_combine_bounding_boxes():  # :369-376
    up, lo = (C.c_float * 3)(), (C.c_float * 3)()
    lib.orc_aabb_combine(f3((1, 0, 0)), f3((2, 2, 2)), f3((0, 0, 0)), f3((3, 0.5, 0.5)), up, lo)
    assert list(up) == [3.0, 2.0, 2.0] and list(lo) == [0.0, 0.0, 0.0]


# ---- core/object.rs:111-152
def test_intersect_object():
    def obj_hit(has_t, m, tr):
        t = C.c_float()
        ok = lib.orc_object_sphere_hit(f3((0, 0, 0)), C.c_float(0.5), has_t, f9(m), f3(tr), f3((10, 0, 0)), f3((-1, 0, 0)),
                                       C.c_float(0.0), C.c_float(100.0), C.byref(t))
        return t.value if ok else None
    ident = np.eye(3)
    assert obj_hit(0, ident, (0, 0, 0)) == 9.5
    assert obj_hit(1, ident, (0, 0, 0)) == 9.5
    rot = (C.c_float * 9)()
    lib.orc_matrix_rotation(f3((1, 1, 1)), C.c_float(1.0), rot)
    assert obj_hit(1, list(rot), (0, 0, 0)) is not None
    assert obj_hit(1, list(rot), (0, 0, 1)) is None


# ---- math/utils.rs:49-59
def test_tangent_space():
    z = vec_op(0, (1, 2, 3))
    t1, t2 = (C.c_float * 3)(), (C.c_float * 3)()
    lib.orc_tangent_space(f3(z), t1, t2)
    vecs = [np.array(t1[:]), np.array(t2[:]), np.array(z)]
    for i in range(3):
        j = (i + 1) % 3
        assert abs(np.dot(vecs[i], vecs[i]) - 1.0) < 1e-6
        assert abs(np.dot(vecs[i], vecs[j])) < 1e-6


# ---- math/vec3d/vec3d.rs:196-233
def test_vec3d_normalise_dot_cross():
    u = vec_op(0, (1, 2, 3))
    for got, want in zip(u, (0.2672612419124243, 0.5345224838248487, 0.8017837257372731)):
        assert abs(got - want) < 1e-6
    assert lib.orc_vec_dot(f3((1, 2, 3)), f3((1, -1, 1))) == 2.0
    assert vec_op(1, (1, 2, 3), (1, -1, 1)) == [5.0, 2.0, -3.0]
    assert vec_op(1, (1, -1, 1), (1, 2, 3)) == [-5.0, -2.0, 3.0]


# ---- math/vec3d/ops.rs:386-451 (operator algebra on exactly representable values)
def test_vec3d_operators():
    a, b = (1.0, 2.0, 3.0), (4.0, 5.0, 6.0)
    assert vec_op(2, a, b) == [5.0, 7.0, 9.0]
    assert vec_op(3, a, b) == [-3.0, -3.0, -3.0]
    assert vec_op(4, a) == [-1.0, -2.0, -3.0]
    assert vec_op(5, a, s=2.0) == [2.0, 4.0, 6.0]
    assert vec_op(6, a, s=2.0) == [2.0, 4.0, 6.0]
    assert vec_op(7, a, s=2.0) == [0.5, 1.0, 1.5]
    assert vec_op(8, a, s=2.0) == [0.5, 1.0, 1.5]
    # Div multiplies by the reciprocal (ops.rs:5-11), DivAssign divides (ops.rs:37-43): they differ in the last bit
    x = (np.float32(1.0), np.float32(5.0), np.float32(7.0))
    div = vec_op(7, x, s=3.0)
    diva = vec_op(8, x, s=3.0)
    r = np.float32(1.0) / np.float32(3.0)
    assert div == [float(np.float32(v) * r) for v in x]
    assert diva == [float(np.float32(v) / np.float32(3.0)) for v in x]
    assert div != diva


# ---- math/matrix.rs:148-165
def test_matrix_inverse():
    m = [1, 2, 3, 0, 4, 5, 0, 0, 6]
    inv, prod = (C.c_float * 9)(), (C.c_float * 9)()
    assert lib.orc_matrix_inverse(f9(m), inv)
    lib.orc_matrix_mul(f9(m), inv, prod)
    # the reference's own bound (matrix.rs:159): |M * M^-1 - I| < 1e-9 per element
    for i in range(3):
        for j in range(3):
            assert abs(prod[3 * i + j] - (1.0 if i == j else 0.0)) < 1e-9
    assert not lib.orc_matrix_inverse(f9([1, 2, 3, 2, 4, 6, 0, 0, 1]), inv)  # singular -> None


# ---- added known-answer tests for what the reference leaves unpinned
def test_triangle_intersection_known_answers():  # primitives.rs:138-186
    def tri(o, d, tmin=0.0, tmax=math.inf):
        t, n, ff = C.c_float(), (C.c_float * 3)(), C.c_int()
        ok = lib.orc_triangle_hit(f3((0, 0, 0)), f3((1, 0, 0)), f3((0, 1, 0)), f3(o), f3(d), C.c_float(tmin), C.c_float(tmax),
                                  C.byref(t), n, C.byref(ff))
        return (t.value, list(n), bool(ff.value)) if ok else None
    t, n, ff = tri((0.25, 0.25, 1.0), (0, 0, -1))
    assert t == 1.0 and n == [0.0, 0.0, 1.0]
    assert ff is True             # det = e1 . (d x e2) = +1: ray arrives on the side the CCW normal points to (:176)
    t, n, ff = tri((0.25, 0.25, -2.0), (0, 0, 1))
    assert t == 2.0 and n == [0.0, 0.0, 1.0] and ff is False  # from behind: the geometric normal is NOT flipped
    assert tri((0.75, 0.75, 1.0), (0, 0, -1)) is None      # u + v > 1
    assert tri((0.25, 0.25, 1.0), (1, 0, 0)) is None       # parallel: |det| < EPSILON
    assert tri((0.25, 0.25, 1.0), (0, 0, -1), 0.0, 1.0) is None  # t < t_max is strict
    assert tri((0.0, 0.0, 1.0), (0, 0, -1)) is not None    # u = v = 0 on the corner is inside


def test_plane_front_face_sense():  # primitives.rs:229-247: front_face = denom > 0
    t, n, ff = C.c_float(), (C.c_float * 3)(), C.c_int()
    assert lib.orc_plane_hit(f3((0, 0, -0.6)), f3((0, 0, 2)), f3((0, 0, 1)), f3((0, 0, -1)), C.c_float(1e-3), C.c_float(math.inf),
                             C.byref(t), n, C.byref(ff))
    assert abs(t.value - 1.6) < 1e-6 and list(n) == [0.0, 0.0, 1.0] and ff.value == 0


def test_zero_thickness_box_is_missed():  # primitives.rs:316: t_max <= t_min -> None
    assert aabb_hit((0, 0, 1), (1, 1, 1), (0.5, 0.5, 0), (0, 0, 1), 0.0, 10.0) is None


def test_philox_known_answers():
    """Philox4x32-10 known-answer vectors (Random123 kat_vectors)."""
    def ph(ctr, key):
        out = (C.c_uint32 * 4)()
        lib.orc_philox((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
        return [hex(x) for x in out]
    assert ph([0, 0, 0, 0], [0, 0]) == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert ph([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    assert ph([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_rng_f32_grid_and_range():
    vals = np.array([lib.orc_rng_f32(1234, p, 3, 1, d) for p in range(64) for d in range(8)], np.float32)
    assert (vals >= 0).all() and (vals < 1).all()
    assert np.array_equal(vals * 2 ** 23, np.round(vals * 2 ** 23))  # on the 2^-23 grid
    assert 0.4 < vals.mean() < 0.6


def test_transcendentals_are_faithful():
    """pmath.h sin/cos/exp/ln: within 1 ulp of libm (float64 reference) over the ranges the render loop uses."""
    rng = np.random.default_rng(0)
    def ulp_err(fn, xs, ref):
        got = np.array([lib.orc_pm(fn, float(x)) for x in xs], np.float32)
        want = ref(xs.astype(np.float64))
        spacing = np.spacing(np.abs(want).astype(np.float32)).astype(np.float64)
        return np.max(np.abs(got.astype(np.float64) - want) / spacing)
    phi = rng.uniform(0, 2 * math.pi, 4000).astype(np.float32)
    assert ulp_err(0, phi, np.sin) < 1.0
    assert ulp_err(1, phi, np.cos) < 1.0
    x = -rng.uniform(0, 90, 4000).astype(np.float32)
    assert ulp_err(2, x, np.exp) < 1.0
    y = (1.0 - (rng.integers(0, 2 ** 23, 4000) * 2.0 ** -23)).astype(np.float32)
    assert ulp_err(3, y, np.log) < 1.0
    assert lib.orc_pm(3, 1.0) == 0.0 and lib.orc_pm(2, 0.0) == 1.0
    assert lib.orc_pm(2, -200.0) == 0.0
