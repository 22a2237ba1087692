// Device-side f32 algebra, the four transcendental functions and the counter-based RNG.
//
// Everything here is written so that a gfx950 lane produces the bit pattern an IEEE-754 CPU
// produces for the reference's expression: one rounding per written operation, evaluated in the
// written order.  The translation unit MUST be compiled with -ffp-contract=off (hipcc contracts
// a*b+c into v_fma_f32 by default) and with correctly rounded f32 divide and sqrt
// (-fhip-fp32-correctly-rounded-divide-sqrt, hipcc's default).
//
// Reference semantics restated here (file:line in ray_tracer/src):
//   Vec3D / f32 multiplies by the reciprocal        math/vec3d/ops.rs:5-11
//   Vec3D /= f32 divides per component              math/vec3d/ops.rs:37-43
//   dot / cross are plain left-to-right sums        math/vec3d/vec3d.rs:118-130
//   f32::max/min ignore a NaN operand               (v_max_f32/v_min_f32 in IEEE mode do too)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_DEV __device__ __forceinline__

namespace rtd {

constexpr float F32_EPSILON = 1.1920928955078125e-7f;
constexpr float PI_F = 3.14159265358979323846f;
constexpr float PI_RECIP = 1.0f / PI_F;  // bsdf.rs:8

struct V3 {
    float x, y, z;
};

RT_DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
RT_DEV V3 fill(float v) { return V3{v, v, v}; }
RT_DEV float recip(float x) { return 1.0f / x; }
RT_DEV V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_DEV V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_DEV V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
RT_DEV V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
RT_DEV V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
RT_DEV V3 operator/(V3 a, float s) { return a * recip(s); }
RT_DEV V3 div_assign(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
RT_DEV V3 mul_elemwise(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
RT_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
RT_DEV float norm_squared(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
RT_DEV float norm_recip(V3 a) { return recip(sqrtf(norm_squared(a))); }
RT_DEV V3 normalise(V3 a) { return a * norm_recip(a); }  // vec3d.rs:81-85 (no assert: NaN propagates)
RT_DEV float max_elem(V3 a) { return fmaxf(fmaxf(a.x, a.y), a.z); }
RT_DEV float safe_sqrt(float v) { return v <= 0.0f ? 0.0f : sqrtf(v); }  // utils.rs:7-13
RT_DEV float signum(float x) { return x != x ? x : copysignf(1.0f, x); }
RT_DEV float comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

RT_DEV void tangent_space(V3 v, V3* t1, V3* t2) {  // utils.rs:17-26
    float s = signum(v.z);
    float a = -recip(s + v.z);
    float b = v.x * v.y * a;
    *t1 = v3(1.0f + s * v.x * v.x * a, s * b, -s * v.x);
    *t2 = v3(b, s + v.y * v.y * a, -v.y);
}
RT_DEV V3 to_shading_coords(V3 dir, V3 normal) {  // bsdf.rs:21-28
    V3 a, b;
    tangent_space(normal, &a, &b);
    return v3(dot(dir, a), dot(dir, b), dot(dir, normal));
}
RT_DEV V3 to_world_coords(V3 dir, V3 normal) {  // bsdf.rs:30-33
    V3 a, b;
    tangent_space(normal, &a, &b);
    return dir.x * a + dir.y * b + dir.z * normal;
}

// 3x3 row-major (matrix.rs:28-46)
template <class P>  // P: pointer to 9 floats in any address space
RT_DEV V3 mat_mul(P m, V3 v) {
    return {m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z,
            m[6] * v.x + m[7] * v.y + m[8] * v.z};
}
template <class P>
RT_DEV V3 mat_transpose_mul(P m, V3 v) {
    return {m[0] * v.x + m[3] * v.y + m[6] * v.z, m[1] * v.x + m[4] * v.y + m[7] * v.z,
            m[2] * v.x + m[5] * v.y + m[8] * v.z};
}

// ---------------------------------------------------------------- cos / sin / exp / ln
// Rust leaves the precision of f32::cos/sin/exp/ln unspecified; parity needs the host checker and
// the device to evaluate the same function.  Specification: the fixed binary64 operation sequence
// below (plain * + - /, round-to-nearest-even rint), rounded once to binary32.
RT_DEV void sincos_core(float xf, double* s, double* c) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_HI = 1.57079632673412561417e+00;
    const double PIO2_LO = 6.07710050650619224932e-11;
    double x = (double)xf;
    double k = rint(x * TWO_OVER_PI);
    double r = (x - k * PIO2_HI) - k * PIO2_LO;
    double z = r * r;
    double ps = -1.66666666666666324348e-01 +
                z * (8.33333333332248946124e-03 +
                     z * (-1.98412698298579493134e-04 +
                          z * (2.75573137070700676789e-06 +
                               z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
    double sr = r + r * z * ps;
    double pc = 4.16666666666666019037e-02 +
                z * (-1.38888888888741095749e-03 +
                     z * (2.48015872894767294178e-05 +
                          z * (-2.75573143513906633035e-07 +
                               z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
    double cr = (1.0 - 0.5 * z) + z * z * pc;
    long long q = (long long)k & 3;
    if (q == 0) { *s = sr; *c = cr; }
    else if (q == 1) { *s = cr; *c = -sr; }
    else if (q == 2) { *s = -sr; *c = -cr; }
    else { *s = -cr; *c = sr; }
}
RT_DEV void sincosf_(float x, float* s, float* c) {
    double ds, dc;
    sincos_core(x, &ds, &dc);
    *s = (float)ds;
    *c = (float)dc;
}
RT_DEV float expf_(float xf) {
    if (xf != xf) return xf;
    if (xf > 88.8f) return __builtin_inff();
    if (xf < -104.0f) return 0.0f;
    const double LOG2E = 1.44269504088896338700e+00;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double x = (double)xf;
    double k = rint(x * LOG2E);
    double r = (x - k * LN2_HI) - k * LN2_LO;
    double p = 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    long long ki = (long long)k;
    double scale = __longlong_as_double((ki + 1023) << 52);
    return (float)(p * scale);
}
RT_DEV float logf_(float yf) {
    if (yf != yf || yf < 0.0f) return __builtin_nanf("");
    if (yf == 0.0f) return -__builtin_inff();
    if (yf == __builtin_inff()) return yf;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double d = (double)yf;
    unsigned long long bits = (unsigned long long)__double_as_longlong(d);
    long long e = (long long)((bits >> 52) & 0x7FF) - 1023;
    bits = (bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double m = __longlong_as_double((long long)bits);
    if (m > 1.41421356237309514547) {
        m = m * 0.5;
        e += 1;
    }
    double s = (m - 1.0) / (m + 1.0);
    double z = s * s;
    double p = 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    double lnm = 2.0 * s * p;
    double ed = (double)e;
    return (float)(ed * LN2_HI + (lnm + ed * LN2_LO));
}

// ---------------------------------------------------------------- counter-based RNG
// Philox4x32-10; key = 64-bit seed, counter = (pixel, global sample, bounce, block); draw d of a bounce is
// word d % 4 of block d / 4 (oracle/rng.h).  Replaces the single sequential fastrand::Rng of the
// reference (path_integrator.rs:286).
RT_DEV void philox_block(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 multiply per product (v_mad_u64_u32) instead of a mul_hi and a mul_lo: integer
        // multiplies are quarter rate and a block has 20 products
        const unsigned long long p0 = (unsigned long long)M0 * c0, p1 = (unsigned long long)M1 * c2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += W0;
        k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct Rng {  // used from draw 0 of a bounce onwards, in order
    uint32_t k0, k1, pixel, sample, bounce, dim;
    uint32_t w[4];
    RT_DEV uint32_t u32() {
        const uint32_t i = dim & 3u;
        if (i == 0u) philox_block(pixel, sample, bounce, dim >> 2, k0, k1, w);
        ++dim;
        return i == 0u ? w[0] : i == 1u ? w[1] : i == 2u ? w[2] : w[3];
    }
    RT_DEV float f32() { return (float)(u32() >> 9) * 1.1920928955078125e-7f; }  // [0,1) on 2^-23
    RT_DEV uint32_t usize(uint32_t n) { return __umulhi(u32(), n); }              // rng.usize(0..n)
};

}  // namespace rtd
