// Host-side f32 vector algebra for scene preparation.
//
// Every operator keeps the rounding behaviour of the reference's math module, because the BVH
// builder's output (and so the device traversal order) depends on it:
//   Vec / f32   -> multiply by the reciprocal        (math/vec3d/ops.rs:5-11)
//   dot / cross -> plain sums of products, left to right, no FMA (math/vec3d/vec3d.rs:118-130)
//   max / min   -> IEEE maxNum/minNum (NaN ignored), like f32::max/min
// Build with -ffp-contract=off.
#pragma once
#include <cmath>
#include <cstdint>

namespace rth {

struct V3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};

inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 fill(float v) { return V3{v, v, v}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline float recip(float x) { return 1.0f / x; }
inline V3 operator/(V3 a, float s) { return a * recip(s); }  // ops.rs:5-11
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float norm_squared(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
inline float norm_recip(V3 a) { return recip(std::sqrt(norm_squared(a))); }  // vec3d.rs:75-77
inline V3 vmax(V3 a, V3 b) { return {fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)}; }
inline V3 vmin(V3 a, V3 b) { return {fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)}; }

// vec3d.rs:81-85.  The reference asserts the reciprocal norm is finite; callers check `ok`.
inline V3 normalise(V3 a, bool* ok) {
    float r = norm_recip(a);
    if (ok) *ok = std::isfinite(r);
    return a * r;
}

struct Box {  // AxisAlignedBoundingBox (primitives.rs:250-253): field order upper, lower
    V3 upper, lower;
};
inline Box box_empty() { return {fill(-INFINITY), fill(INFINITY)}; }  // primitives.rs:264-269
inline Box box_grow(Box b, V3 p) { return {vmax(b.upper, p), vmin(b.lower, p)}; }
inline Box box_combine(const Box& a, const Box& b) {
    return {vmax(a.upper, b.upper), vmin(a.lower, b.lower)};
}
inline float box_half_area(const Box& b) {  // primitives.rs:282-285
    V3 w = b.upper - b.lower;
    return w.x * w.y + w.y * w.z + w.z * w.x;
}

struct M3 {  // math/matrix.rs:4-6, row-major
    float d[3][3];
};
inline V3 mat_mul(const M3& m, V3 v) {  // matrix.rs:28-35
    return {m.d[0][0] * v.x + m.d[0][1] * v.y + m.d[0][2] * v.z,
            m.d[1][0] * v.x + m.d[1][1] * v.y + m.d[1][2] * v.z,
            m.d[2][0] * v.x + m.d[2][1] * v.y + m.d[2][2] * v.z};
}
inline float mat_det(const M3& s) {  // matrix.rs:49-54
    const float(*m)[3] = s.d;
    return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) -
           m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
           m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
}
inline bool mat_inverse(const M3& s, M3* out) {  // matrix.rs:56-84
    float det = mat_det(s);
    if (std::fabs(det) < 1e-9f) return false;
    float inv_det = recip(det);
    const float(*m)[3] = s.d;
    M3 r = {{{(m[1][1] * m[2][2] - m[1][2] * m[2][1]) * inv_det,
              (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * inv_det,
              (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * inv_det},
             {(m[1][2] * m[2][0] - m[1][0] * m[2][2]) * inv_det,
              (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * inv_det,
              (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * inv_det},
             {(m[1][0] * m[2][1] - m[1][1] * m[2][0]) * inv_det,
              (m[0][1] * m[2][0] - m[0][0] * m[2][1]) * inv_det,
              (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * inv_det}}};
    *out = r;
    return true;
}

}  // namespace rth
