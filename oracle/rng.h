// ORACLE -- test infrastructure only (see vec3d.h).
//
// The reference draws from one `fastrand::Rng` (crate fastrand = "1.9.0", ray_tracer/Cargo.toml:9;
// source not in the tree, no Cargo.lock) seeded non-deterministically by `Rng::new()`
// (path_integrator.rs:286), so its stream cannot be reproduced: parity is UNPINNED at this
// boundary by construction.  Both sides therefore use the counter-based generator the north-star
// prescribes, stated here and mirrored on the device:
//
//   Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11)
//   key     = (seed low 32, seed high 32)
//   counter = (pixel = j*width+i, global sample index, bounce, block index within the bounce)
//   value   = draw d of a bounce is output word d % 4 of block d / 4 (all four words of a block are used:
//             the block function is the expensive part of shading on the device -- 40 quarter-rate
//             integer multiplies -- and a bounce makes about seven draws)
//
// bounce 0 holds the two camera-jitter draws of Engine::render (path_integrator.rs:303-304);
// iteration k >= 1 of the `while depth > 0` loop of ray_colour (:150) is bounce k.
//   f32():  (word >> 9) * 2^-23, i.e. [0,1) on the 2^-23 grid (what fastrand's f32() yields too)
//   usize(0..n): (word * n) >> 32
#pragma once
#include <cstdint>

namespace orc {

inline void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0;
        uint64_t p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0;
        k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct Rng {
    uint32_t key[2];
    uint32_t pixel, sample, bounce, dim;
    uint32_t block[4], have;  // cached block and its index (0xFFFFFFFF: none)
    Rng(uint64_t seed, uint32_t pixel_, uint32_t sample_)
        : key{(uint32_t)seed, (uint32_t)(seed >> 32)}, pixel(pixel_), sample(sample_), bounce(0), dim(0),
          block{0, 0, 0, 0}, have(0xFFFFFFFFu) {}
    void set_bounce(uint32_t b) { bounce = b; dim = 0; have = 0xFFFFFFFFu; }
    uint32_t u32() {
        const uint32_t b = dim >> 2, w = dim & 3u;
        if (have != b) {
            uint32_t ctr[4] = {pixel, sample, bounce, b};
            philox4x32_10(ctr, key, block);
            have = b;
        }
        ++dim;
        return block[w];
    }
    float f32() { return (float)(u32() >> 9) * 1.1920928955078125e-7f; }
    uint32_t usize(uint32_t n) { return (uint32_t)(((uint64_t)u32() * n) >> 32); }
};

}  // namespace orc
