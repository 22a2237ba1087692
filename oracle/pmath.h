// ORACLE -- test infrastructure only (see vec3d.h).
//
// The four transcendental functions the render loop calls: cos and sin (utils.rs:34, lights.rs:146,
// medium.rs:214), ln (utils.rs:41) and exp (medium.rs:78).  Rust documents the precision of
// f32::cos/sin/ln/exp as unspecified and platform dependent, so no bit pattern of the reference is
// pinned here; what matters for parity is that the CPU restatement and the device evaluate the
// SAME function.  They are therefore specified as: evaluate in binary64 with the fixed operation
// sequence below (plain * + - /, rint; no FMA contraction), then round once to binary32.  The
// binary64 result is within ~1e-16 of the true value, so the rounded result is the correctly
// rounded f32 in all but a vanishing fraction of inputs (checked against libm in the tests).
// The device code restates the same sequence (ray_tracing_amd/csrc/device_math.h).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {
namespace pm {

inline void sincos_core(float xf, double* s, double* c) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_HI = 1.57079632673412561417e+00;  // 33 bits of pi/2
    const double PIO2_LO = 6.07710050650619224932e-11;  // pi/2 - PIO2_HI
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
    switch (q) {
        case 0: *s = sr; *c = cr; break;
        case 1: *s = cr; *c = -sr; break;
        case 2: *s = -sr; *c = -cr; break;
        default: *s = -cr; *c = sr; break;
    }
}
// |x| <= 2^20 is the supported range (the render loop passes phi in [0, 2*pi)).
inline float cosf_(float x) { double s, c; sincos_core(x, &s, &c); return (float)c; }
inline float sinf_(float x) { double s, c; sincos_core(x, &s, &c); return (float)s; }

inline float expf_(float xf) {
    if (std::isnan(xf)) return xf;
    if (xf > 88.8f) return INFINITY;
    if (xf < -104.0f) return 0.0f;
    const double LOG2E = 1.44269504088896338700e+00;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double x = (double)xf;
    double k = rint(x * LOG2E);
    double r = (x - k * LN2_HI) - k * LN2_LO;
    // Taylor series to r^13/13! in Horner form
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
    int64_t ki = (int64_t)k;  // in [-151, 129]
    uint64_t bits = (uint64_t)(ki + 1023) << 52;
    double scale;
    memcpy(&scale, &bits, 8);
    return (float)(p * scale);
}

inline float logf_(float yf) {
    if (std::isnan(yf) || yf < 0.0f) return NAN;
    if (yf == 0.0f) return -INFINITY;
    if (std::isinf(yf)) return yf;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double d = (double)yf;  // exact, and normal in binary64 even for an f32 subnormal
    uint64_t bits;
    memcpy(&bits, &d, 8);
    int64_t e = (int64_t)((bits >> 52) & 0x7FF) - 1023;
    bits = (bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double m;
    memcpy(&m, &bits, 8);  // [1, 2)
    if (m > 1.41421356237309514547) {
        m = m * 0.5;
        e += 1;
    }
    double s = (m - 1.0) / (m + 1.0);
    double z = s * s;
    // ln(m) = 2*atanh(s) = 2s*(1 + z/3 + z^2/5 + ... + z^10/21)
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

}  // namespace pm
}  // namespace orc
