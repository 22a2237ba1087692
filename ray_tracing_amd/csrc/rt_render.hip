// rt_render: Engine::render / Engine::ray_colour (path_integrator.rs:138-318) as a wavefront
// pipeline on gfx950:
//
//   generate        the counts of bounce 0; the first bounce's state is a function of the slot (camera origin, jittered
//                   Camera::ray, throughput 1), recomputed where it is needed -- stored, one lane per (pixel, sample), only
//                   for rt_ray_colour's caller rays and for frames the tail kernel takes from the start
//   extend          one lane per live path: Engine::hit (closest hit through the BVH)
//   shade           one lane per live path: one iteration of the `while depth > 0` loop --
//                   light hit + MIS, medium majorant step, BSDF sample, NEE set-up, throughput,
//                   medium switch, Russian roulette -- AND the stream compaction: a block counts its
//                   survivors and shadow rays (wave64 ballots), reserves room for them with one atomic
//                   and writes their state straight to consecutive positions of the next bounce's region
//   publish         one wave: turns the reserved totals into the next bounce's counts
//   shadow-connect  one lane per shadow ray: Engine::hit_bool, adds the pending NEE term
//   resolve         per pixel, adds the samples of the batch IN SAMPLE ORDER into the frame sum
//
// A path's identity is its ORIGINAL slot (= pixel * samples + sample): it names the pixel, the sample -- hence the
// random numbers -- and the radiance record, and it travels with the path's state from region to region.  Radiance
// additions happen in the reference's order (NEE of iteration k before emission of iteration k+1), and every random
// draw is addressed by (pixel, sample, bounce, draw), so the sum written for a pixel is bit-identical to the
// sequential CPU loop whatever order the compaction leaves the paths in.
#include <cmath>
#include <cstring>

#include "context.h"
#include "device_shade.h"
#include "device_trace.h"

using namespace rtd;

namespace {

constexpr float T_MIN = 1e-3f;  // path_integrator.rs:15
constexpr int SHADE_BLOCK = 256;

constexpr uint32_t QF_ALIVE = 1u, QF_SHADOW = 2u, QF_SKIPPED = 4u;  // QF_SKIPPED: a shadow ray the reference issues and nobody needs

// Positions.  The paths of a bounce live in NSEG segments of one region: segment s holds its paths at positions
// [s * seg_cap, s * seg_cap + count[s]), seg_cap = ceil(batch / NSEG).  The survivors of segment s are written to
// segment s of the OTHER region by k_shade (densely from the segment's first position, in the order the blocks'
// reservations arrive), the shadow rays of the bounce to segment s of the shadow records.  A segment is a band of the
// image (positions start out as pixel-major slots), it is what the workgroups of the trace kernels draw their rays
// from (device_trace.h: one work counter per segment), and it is what keeps the reservations off a single address:
// a block reserves with ONE 64-bit atomic (survivors | shadow rays << 32) on its segment's accumulator, and
// consecutive blocks belong to different segments (device-scope atomics on one address complete at the memory side,
// about one per 15 ns: 518 400 blocks on one address would be 8 ms, on 64 they are 0.12 ms in parallel).
constexpr uint32_t NSEG = WORK_MAX_SEG;
static_assert(NSEG == 64, "k_publish is one wave, one lane per segment");
constexpr uint32_t ACC_STRIDE = 16;  // u64 between accumulators: one 128-byte line each
constexpr uint32_t LIST_STRIDE = 32; // u32 between the list counters: one 128-byte line each
// A bounce of n paths uses ns = 2^k <= NSEG segments of stride NSEG / ns * seg_cap (segments_for() on the host: what
// a launch of n rays wants as work segments, 64 from 8 Mi rays up, one below 256 Ki) -- a small launch over 128 nearly
// empty segments spends its time on their counters -- and since paths only die, ns only shrinks from bounce to bounce: a
// segment of the next bounce is the union of consecutive segments of this one, so its stride always holds them.
// counts block (u32), two of them, ping-pong over the bounces: [0, NSEG) shadow rays per segment of the bounce BEFORE
// this one, [NSEG, 2 NSEG) paths alive per segment at this bounce, the totals, and at CNT_MIXED the 2 ns segments of a
// mixed launch in its order (shadow rays of the ns segments, then the paths of the ns segments)
// ... and three lists of the segments that HAVE items (their number, then their indices): the home segments of the trace
// launches' waves (device_trace.h, live_segs) for a launch over the paths, over the shadow rays, over both (mixed order)
constexpr uint32_t CNT_SHADOW = 0, CNT_ALIVE = NSEG, CNT_TOT_ALIVE = 2 * NSEG, CNT_TOT_SHADOW = 2 * NSEG + 1, CNT_MIXED = 2 * NSEG + 4,
                   CNT_LIVE_ALIVE = 4 * NSEG + 4, CNT_LIVE_SHADOW = CNT_LIVE_ALIVE + NSEG + 1, CNT_LIVE_MIXED = CNT_LIVE_SHADOW + NSEG + 1,
                   CNT_U32 = CNT_LIVE_MIXED + 2 * NSEG + 1 + 1;

struct PathArrays {      // the arrays one bounce works on (all indexed by POSITION unless noted)
    // 16-byte records; the w lanes carry the scalar path state, so a bounce moves 3 records per path
    // in and 3 out (the bounce number is the wavefront iteration: a kernel argument, not path state)
    float4* ray_o;       // origin.xyz, bits: current medium (int) << 1 | previous bounce was specular
    float4* ray_d;       // direction.xyz, prev_pdf
    float4* beta;        // throughput.xyz, bits: depth (u32; wraps like the reference's, path_integrator.rs:193)
    uint32_t* oid;       // the path's original slot; null at the first bounce, where position = slot
    float4 *nray_o, *nray_d, *nbeta;  // the same three records in the region the survivors go to ...
    uint32_t* noid;                   // ... with their original slot
    float4* rad;         // radiance.xyz, -   indexed by ORIGINAL SLOT (k_resolve sums it in sample order)
    float4* hit_a;       // t, normal.xyz (normal of analytic surfaces only)
    uint2* hit_b;        // obj | front_face<<31 (RT_MISS = no hit), triangle slot of a mesh hit (normal: shade_path)
    float4* sh_o;        // shadow ray: origin.xyz (the vertex just shaded), bits: original slot
    float4* sh_d;        // shadow direction.xyz, -
    float4* sh_c;        // pending NEE term.xyz, apply flag
    const uint32_t* cnt; // counts block of this bounce (CNT_*)
    unsigned long long* acc;  // [NSEG * ACC_STRIDE] reservations of this bounce: survivors | shadow rays << 32 per segment
    uint32_t* acc_skipped;    // shadow rays of this bounce that nobody needs (a total)
    uint32_t ns, seg_stride;          // segments of this bounce's paths (and of the shadow records traced with them) and their stride
    uint32_t ns_out, out_stride;      // segments the survivors and the shadow rays of this bounce go to
    uint32_t n_positions;             // NSEG * seg_cap = ns * seg_stride: items of a mixed launch below it are shadow records
    uint32_t* work;      // WORK_BUF_U32 (device_trace.h), zeroed by k_publish of every bounce: closest-hit launches (and the
                         // tail kernel) use the first NSEG counters, shadow launches the second NSEG, a mixed launch all of
                         // them (its 2 NSEG segments in order) -- each counter serves one launch per bounce
    const uint32_t* list;  // k_shade: the positions (inside their segment) of the paths that need shading, per segment, dense from the
                           // segment's first position (k_classify); null = every position below the segment's count
    uint32_t* list_cnt;    // [NSEG * LIST_STRIDE] entries of `list` per segment; zeroed by k_publish
    unsigned long long* tail;   // k_tail totals (see k_tail)
    unsigned long long* visit; // [0] n_int extend, [1] n_tri extend, [2] n_int shadow, [3] n_tri shadow
};

struct RenderParams {
    uint32_t width, height, npix;
    uint32_t s_count;       // samples of this batch
    uint32_t pixel_major;   // slot = pixel * s_count + sample_in_batch (1) or sample_in_batch * npix + pixel (0), see slot_pixel()
    uint32_t sample_first;  // global index of sample_in_batch 0
    uint32_t depth;
    uint32_t k0, k1;        // Philox key = seed
    uint32_t bounce;        // k_shade: RNG bounce index of this wavefront iteration (= iteration + 1)
    uint32_t skip_zero_nee;   // do not trace shadow rays whose light term is exactly zero (not when counting the reference's visits)
    uint32_t implicit_first;  // the first bounce's state is a function of the slot alone: origin, throughput, depth and
                              // medium are the same for every path (camera origin, 1, rp.depth, initial medium), the
                              // direction is Camera::ray of the slot's jitter draws (camera_ray_dir), the radiance is 0.
                              // k_generate then stores nothing: the first extend and shade recompute the direction
                              // (one Philox block and a normalisation against 16 B written and 32 B read per path) and
                              // the first extend stores the radiance records' zeros
};
// Which (pixel, sample in batch) a path's original slot names.  Pixel-major: the 64 lanes of a wave start with up to
// 64 samples of ONE pixel -- rays a fraction of a pixel apart, which walk the same nodes down to the leaves (one
// cache line per fetch instruction instead of up to 64) -- and a contiguous piece of the queue is a contiguous
// piece of the image, so the workgroups an XCD draws from its home segments keep that part of the tree in its L2.
// Sample-major (the first version) puts 64 neighbouring pixels of one sample index in a wave: at 1080p over a
// million triangles those are 64 different leaves.
RT_DEV void slot_pixel(const RenderParams& rp, uint32_t id, uint32_t* pixel, uint32_t* s_local) {
    if (rp.pixel_major) {  // uniform
        const uint32_t p = id / rp.s_count;
        *pixel = p;
        *s_local = id - p * rp.s_count;
    } else {
        const uint32_t sl = id / rp.npix;
        *s_local = sl;
        *pixel = id - sl * rp.npix;
    }
}

RT_DEV float pack_medium(int medium, bool prev_specular) {
    return __uint_as_float(((uint32_t)medium << 1) | (prev_specular ? 1u : 0u));
}

// ---------------------------------------------------------------- generate
// Camera::ray (camera.rs:60-63) of the jittered (u, v) of path_integrator.rs:303-305 for (pixel, sample in batch): the two
// draws of RNG bounce 0.  One definition for k_generate and for the kernels that recompute it (RenderParams::implicit_first).
RT_DEV V3 camera_ray_dir(const DScene& sc, const RenderParams& rp, uint32_t pixel, uint32_t s_local) {
    const uint32_t i = pixel % rp.width, j = pixel / rp.width;
    Rng rng{rp.k0, rp.k1, pixel, rp.sample_first + s_local, 0u, 0u};
    const float u = ((float)i + rng.f32()) / (float)(rp.width - 1);
    const float v = ((float)j + rng.f32()) / (float)(rp.height - 1);
    const V3 origin = ld3(sc.camera.origin);
    const V3 dir = ld3(sc.camera.lower_left_corner) + u * ld3(sc.camera.horizontal) + v * ld3(sc.camera.vertical) - origin;
    return normalise(dir);
}

// Position = slot: every segment but the last starts full.  Block 0 also writes the counts of bounce 0; with
// RenderParams::implicit_first that is all there is to do (one block is launched).
__global__ __launch_bounds__(256) void k_generate(DScene sc, PathArrays pa, RenderParams rp, uint32_t n_paths, rt_rays ur,
                                                  uint32_t* cnt0) {
    uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < NSEG) {  // bounce 0: every path is alive, no shadow rays pending
        const unsigned long long first = (unsigned long long)threadIdx.x * pa.seg_stride;
        cnt0[CNT_ALIVE + threadIdx.x] = (threadIdx.x >= pa.ns || first >= n_paths) ? 0u : (uint32_t)min((unsigned long long)pa.seg_stride, n_paths - first);
        cnt0[CNT_SHADOW + threadIdx.x] = 0u;
        cnt0[CNT_LIVE_ALIVE + 1u + threadIdx.x] = threadIdx.x;  // the segments with paths are the first ones
        if (threadIdx.x == 0) {
            cnt0[CNT_TOT_ALIVE] = n_paths;
            cnt0[CNT_TOT_SHADOW] = 0u;
            const unsigned long long full = ((unsigned long long)n_paths + pa.seg_stride - 1u) / pa.seg_stride;
            cnt0[CNT_LIVE_ALIVE] = (uint32_t)min(full, (unsigned long long)pa.ns);
            cnt0[CNT_LIVE_SHADOW] = cnt0[CNT_LIVE_MIXED] = 0u;
        }
    }
    if (slot >= n_paths || rp.implicit_first) return;
    uint32_t pixel, s_local;
    slot_pixel(rp, slot, &pixel, &s_local);
    V3 origin, dir;
    if (ur.ox) {  // uniform: rt_ray_colour -- "pixel" k is the caller's ray k, taken as it is (no jitter draws)
        origin = v3(ur.ox[pixel], ur.oy[pixel], ur.oz[pixel]);
        dir = v3(ur.dx[pixel], ur.dy[pixel], ur.dz[pixel]);
    } else {
        origin = ld3(sc.camera.origin);
        dir = camera_ray_dir(sc, rp, pixel, s_local);
    }
    // :139-148: beta = 1, radiance = 0, prev_pdf = 1, prev_specular = true, current medium = initial
    pa.ray_o[slot] = make_float4(origin.x, origin.y, origin.z, pack_medium(sc.initial_medium, true));
    pa.beta[slot] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(rp.depth));
    pa.ray_d[slot] = make_float4(dir.x, dir.y, dir.z, 1.0f);
    pa.rad[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// ---------------------------------------------------------------- extend
RT_DEV void add_visits(unsigned long long* dst, unsigned long long a, unsigned long long b) {
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off);
        b += __shfl_down(b, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&dst[0], a);
        atomicAdd(&dst[1], b);
    }
}

// Light record l for a wave-uniform l: read through the constant address space (scalar loads into SGPRs, see
// rtd::CObject) instead of one vector round trip per field.
RT_DEV rt_light load_light_uniform(const rt_light* lights, uint32_t l) {
    static_assert(sizeof(rt_light) == 44, "rt_light is copied as 11 dwords");
    const __attribute__((address_space(4))) uint32_t* p = (const __attribute__((address_space(4))) uint32_t*)(lights + l);
    uint32_t w[11];
    for (int i = 0; i < 11; ++i) w[i] = p[i];
    rt_light out;
    __builtin_memcpy(&out, w, sizeof out);
    return out;
}

// A closest-hit ray of a path that met nothing (:155-173): hit_light over the whole ray decides whether the path has
// anything left to do -- add the light's radiance -- or is simply over; the trace kernel has the ray in registers, so it
// says so in the second word of the miss record (k_classify reads that word instead of the ray).
RT_DEV void mark_escaped(const DScene& sc, const PathArrays& pa, uint32_t k, const V3& o, const V3& d) {
    uint32_t lit = 0u;
    for (uint32_t l = 0; l < sc.n_lights; ++l) {
        float t;
        if (light_hit(load_light_uniform(sc.lights, l), o, d, T_MIN, __builtin_inff(), &t)) lit = 1u;
    }
    pa.hit_b[k] = make_uint2(RT_MISS, lit);
}

// FIRST: 1 = always the first bounce of an implicit frame, 0 = never, -1 = decided at run time (`first`).  The lean kernel has
// one instantiation for each: with both ray sources in one kernel the loop loses registers to scalar spills.
template <int FIRST = -1>
struct ExtendIOT {  // ray i = the path at position i; hits go to the position's hit record
    PathArrays pa;
    unsigned long long n_int, n_tri;
    bool first;       // first bounce with implicit state (RenderParams::implicit_first): Camera::ray of the slot, nothing is loaded
    const DScene& sc;
    const RenderParams& rp;
    RT_DEV uint32_t load(uint32_t i, V3* o, V3* d) const {  // returns the position: the ray's tag from here on
        if (FIRST > 0 || (FIRST < 0 && first)) {  // uniform
            uint32_t pixel, s_local;
            slot_pixel(rp, i, &pixel, &s_local);
            *o = ld3(sc.camera.origin);
            *d = camera_ray_dir(sc, rp, pixel, s_local);
            // :140: the path's radiance starts at 0.  Stored from here: this launch waits for node fetches, not for its
            // stores (1 TB/s on the memory side), whereas the first shade is bound by the bytes it moves (+0.6 ms there)
            pa.rad[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        } else {
            float4 o4 = pa.ray_o[i], d4 = pa.ray_d[i];
            *o = v3(o4.x, o4.y, o4.z);
            *d = v3(d4.x, d4.y, d4.z);
        }
        return i;
    }
    // a closer hit than any before it on this ray (Engine::hit's `closest = Some(..)`, :92-94)
    RT_DEV void hit(uint32_t i, float t, V3 n, uint32_t meta, uint32_t prim) const {
        pa.hit_a[i] = make_float4(t, n.x, n.y, n.z);
        pa.hit_b[i] = make_uint2(meta, prim);
    }
    // a closer hit on a mesh: the normal is left to shade_path, which rebuilds it from triangle `tri`
    static constexpr bool DEFER_NORMAL = true;
    RT_DEV void hit_mesh(uint32_t i, float t, uint32_t meta, uint32_t tri) const {
        ((float*)&pa.hit_a[i])[0] = t;
        pa.hit_b[i] = make_uint2(meta, tri);
    }
    RT_DEV void escaped(uint32_t i, const V3& o, const V3& d) const {
        if (FIRST > 0) pa.hit_b[i] = make_uint2(RT_MISS, 1u);  // the first shade looks at every path (no list): nothing to find out
        else mark_escaped(sc, pa, i, o, d);
    }
    RT_DEV void emit_closest(uint32_t, bool, const Visits& vis) {
        n_int += vis.n_int;
        n_tri += vis.n_tri;
    }
    RT_DEV void emit_any(uint32_t, bool, const Visits&) {}
};
typedef ExtendIOT<-1> ExtendIO;

// launch-wide tuning of a render trace kernel: n_seg segments of seg_cap positions, counts in device memory
RT_DEV TraceTune render_tune(const TuneBase& b, uint32_t n_total, uint32_t n_seg, uint32_t seg_cap) {
    TraceTune t;
    t.chunk = b.chunk ? b.chunk : (n_total >= (16u << 20) ? 128u : 64u);
    t.refill_min = b.refill_min;
    t.w_adv = b.w_adv;
    t.w_node = b.w_node;
    t.w_leaf = b.w_leaf;
    t.n_seg = n_seg;
    t.seg_len = seg_cap;
    return t;
}

template <int COUNT>
__global__ RT_TRACE_ATTR void k_extend(DScene sc, PathArrays pa, RenderParams rp, uint32_t first, TuneBase base, uint32_t* overflow,
                                       uint32_t overflow_stride) {
    __shared__ uint32_t lds_stack[RT_STACK_LDS * RT_BLOCK];
    const uint32_t gtid = blockIdx.x * RT_BLOCK + threadIdx.x;
    const uint32_t n = pa.cnt[CNT_TOT_ALIVE];
    const TraceTune tune = render_tune(base, n, pa.ns, pa.seg_stride);
    ExtendIO io{pa, 0ull, 0ull, first != 0u, sc, rp};
    trace_loop<TRACE_CLOSEST, COUNT, RT_STACK_LDS, RT_BLOCK>(sc, io, n, pa.work, tune, T_MIN, __builtin_inff(),
                                                             lds_stack + threadIdx.x, overflow + gtid, overflow_stride,
                                                             pa.cnt + CNT_ALIVE, 0u, pa.cnt + CNT_LIVE_ALIVE);
    if (COUNT) add_visits(&pa.visit[0], io.n_int, io.n_tri);
}

// The same kernel for scenes whose objects are meshes and planes (rtd::SceneFeatures): fewer registers, more waves.
#ifndef RT_WAVES_EU_LEAN_FIRST
#define RT_WAVES_EU_LEAN_FIRST RT_WAVES_EU_LEAN
#endif
template <int FIRST>
__global__ __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(FIRST > 0 ? RT_WAVES_EU_LEAN_FIRST : RT_WAVES_EU_LEAN))) void k_extend_lean(DScene sc, PathArrays pa, RenderParams rp, TuneBase base, uint32_t* overflow,
                                                 uint32_t overflow_stride) {
    __shared__ uint32_t lds_stack[RT_STACK_LEAN * RT_BLOCK];
    const uint32_t gtid = blockIdx.x * RT_BLOCK + threadIdx.x;
    const uint32_t n = pa.cnt[CNT_TOT_ALIVE];
    const TraceTune tune = render_tune(base, n, pa.ns, pa.seg_stride);
    ExtendIOT<FIRST> io{pa, 0ull, 0ull, FIRST != 0, sc, rp};
    trace_loop<TRACE_CLOSEST, false, RT_STACK_LEAN, RT_BLOCK, RT_LEAN_FEAT, RT_LEAN_OVERFLOW>(
        sc, io, n, pa.work, tune, T_MIN, __builtin_inff(), lds_stack + threadIdx.x, overflow + gtid, overflow_stride, pa.cnt + CNT_ALIVE, 0u, pa.cnt + CNT_LIVE_ALIVE);
}

// ---------------------------------------------------------------- shadow-connect
// The NEE term of an unoccluded shadow ray joins the path's radiance (path_integrator.rs:234-236).  Term and
// radiance are requested together and the sum is stored either way (unchanged bits when p_light was 0): the
// wave waits for one round trip here, not for two in a row.
RT_DEV void add_pending_nee(const PathArrays& pa, uint32_t i, uint32_t id) {  // shadow record i of the path with original slot id
    const float4 c = pa.sh_c[i];
    float4 r = pa.rad[id];
    const bool add = c.w != 0.0f;  // p_light > 0
    r.x = add ? r.x + c.x : r.x;
    r.y = add ? r.y + c.y : r.y;
    r.z = add ? r.z + c.z : r.z;
    pa.rad[id] = r;
}

struct ShadowIO {  // ray i = shadow record i; an unoccluded ray adds its pending NEE term
    PathArrays pa;
    unsigned long long n_int, n_tri;
    RT_DEV uint32_t load(uint32_t i, V3* o, V3* d) const {
        float4 o4 = pa.sh_o[i], d4 = pa.sh_d[i];
        *o = v3(o4.x, o4.y, o4.z);
        *d = v3(d4.x, d4.y, d4.z);
        return i;
    }
    static constexpr bool DEFER_NORMAL = false;
    RT_DEV void hit(uint32_t, float, V3, uint32_t, uint32_t) const {}
    RT_DEV void escaped(uint32_t, const V3&, const V3&) const {}
    RT_DEV void emit_closest(uint32_t, bool, const Visits&) {}
    RT_DEV void emit_any(uint32_t i, bool occluded, const Visits& vis) {
        if (!occluded) add_pending_nee(pa, i, __float_as_uint(pa.sh_o[i].w));
        n_int += vis.n_int;
        n_tri += vis.n_tri;
    }
};

// path_integrator.rs:219: the shadow ray is tested over T_MIN..T_MAX, not up to the light.  pa.cnt is the counts block
// k_publish wrote AFTER the bounce that produced the rays: its CNT_SHADOW half.
template <int COUNT>
__global__ RT_TRACE_ATTR_ANY void k_shadow(DScene sc, PathArrays pa, TuneBase base, uint32_t* overflow, uint32_t overflow_stride) {
    __shared__ uint32_t lds_stack[RT_STACK_ANY * RT_BLOCK];
    const uint32_t gtid = blockIdx.x * RT_BLOCK + threadIdx.x;
    const uint32_t n = pa.cnt[CNT_TOT_SHADOW];
    const TraceTune tune = render_tune(base, n, pa.ns, pa.seg_stride);
    ShadowIO io{pa, 0ull, 0ull};
    trace_loop<TRACE_ANY, COUNT, RT_STACK_ANY, RT_BLOCK>(sc, io, n, pa.work, tune, T_MIN, __builtin_inff(),
                                                         lds_stack + threadIdx.x, overflow + gtid, overflow_stride, pa.cnt + CNT_SHADOW, NSEG, pa.cnt + CNT_LIVE_SHADOW);
    if (COUNT) add_visits(&pa.visit[2], io.n_int, io.n_tri);
}

__global__ RT_TRACE_ATTR_LEAN_ANY void k_shadow_lean(DScene sc, PathArrays pa, TuneBase base, uint32_t* overflow, uint32_t overflow_stride) {
    __shared__ uint32_t lds_stack[RT_STACK_LEAN_ANY * RT_BLOCK];
    const uint32_t gtid = blockIdx.x * RT_BLOCK + threadIdx.x;
    const uint32_t n = pa.cnt[CNT_TOT_SHADOW];
    const TraceTune tune = render_tune(base, n, pa.ns, pa.seg_stride);
    ShadowIO io{pa, 0ull, 0ull};
    trace_loop<TRACE_ANY, false, RT_STACK_LEAN_ANY, RT_BLOCK, RT_LEAN_FEAT, RT_LEAN_OVERFLOW>(
        sc, io, n, pa.work, tune, T_MIN, __builtin_inff(), lds_stack + threadIdx.x, overflow + gtid, overflow_stride, pa.cnt + CNT_SHADOW, NSEG, pa.cnt + CNT_LIVE_SHADOW);
}

// ---------------------------------------------------------------- shade
RT_DEV float mis_weight(float pa, float pb) {  // path_integrator.rs:30-39, power 2
    float a = pa * pa;
    float b = pb * pb;
    if (a + b == 0.0f) return 0.0f;
    return a / (a + b);
}

// What an iteration leaves behind besides radiance: the state of the path's next ray (QF_ALIVE) and its shadow ray
// (QF_SHADOW).  The caller decides where they go (k_shade: compacted into the next region; k_tail: in place).
struct ShadeOut {
    float4 o4, d4, b4;  // next ray: origin | medium bits, direction | prev_pdf, throughput | depth bits (o4 also when only QF_SHADOW: the vertex)
    float4 sh_d, sh_c;  // shadow direction, pending NEE term | apply flag
    uint32_t id;        // the path's original slot
};

// One iteration of the loop body of ray_colour for the path at position `slot`.  Returns QF_* flags.
RT_DEV uint32_t shade_path(const DScene& sc, const PathArrays& pa, const RenderParams& rp, uint32_t slot,
                           uint32_t bounce, ShadeOut& out) {
    // Everything the slot names is requested at once: nothing below depends on anything but `slot`, and a
    // wave that waits for each record in turn spends its time in memory round trips (4 waves per SIMD).
    float4 d4, o4, b4;
    const uint2 hb = pa.hit_b[slot];
    float4 ha = pa.hit_a[slot];
    if (rp.implicit_first) {  // uniform: the state k_generate left implicit (:139-148), position = slot
        uint32_t px, sl;
        slot_pixel(rp, slot, &px, &sl);
        const V3 cd = camera_ray_dir(sc, rp, px, sl);
        d4 = make_float4(cd.x, cd.y, cd.z, 1.0f);
        o4 = make_float4(sc.camera.origin[0], sc.camera.origin[1], sc.camera.origin[2], pack_medium(sc.initial_medium, true));
        b4 = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(rp.depth));
    } else {
        d4 = pa.ray_d[slot];
        o4 = pa.ray_o[slot];
        b4 = pa.beta[slot];
    }
    // ... and what the hit names (object record, triangle of a mesh hit) as soon as the hit is here, ahead of the
    // light tests; a miss reads record 0 / triangle 0 (always mapped) and never looks at them
    const bool have_hit = hb.x != RT_MISS;
    const DObject& ob = sc.objects[have_hit ? (hb.x & 0x7FFFFFFFu) : 0u];
    const uint32_t ob_kind = ob.kind, ob_has_transform = ob.has_transform, ob_material = ob.material;
    const uint32_t ob_interface = ob.has_medium_interface;
    const int32_t ob_inside = ob.medium_inside, ob_outside = ob.medium_outside;
    const uint32_t tri = have_hit ? hb.y : 0u;
    const float4 q1 = ld16(sc.tris, tri * 48u + 16u), q2 = ld16(sc.tris, tri * 48u + 32u);
    // a copy: requested here, used after the light tests (a miss reads material 0 and never looks at it; the
    // tables are at least 256 B each, rt_upload_scene, so record 0 is mapped even in an empty scene)
    const rt_material mat = sc.materials[have_hit ? ob_material : 0u];

    V3 ro = v3(o4.x, o4.y, o4.z), rd = v3(d4.x, d4.y, d4.z), beta = v3(b4.x, b4.y, b4.z);
    float prev_pdf = d4.w;
    uint32_t depth = __float_as_uint(b4.w);
    const uint32_t mbits = __float_as_uint(o4.w);
    int medium = (int)mbits >> 1;
    bool prev_specular = (mbits & 1u) != 0;

    depth -= 1;  // `while depth > 0 { depth -= 1;` (:150-151)
    const uint32_t id = pa.oid ? pa.oid[slot] : slot;  // the path's original slot: pixel, sample, radiance record
    out.id = id;
    uint32_t pixel, s_local;
    slot_pixel(rp, id, &pixel, &s_local);
    Rng rng{rp.k0, rp.k1, pixel, rp.sample_first + s_local, bounce, 0u};

    if (!have_hit) ha = make_float4(__builtin_inff(), 0.0f, 0.0f, 0.0f);
    float t_hit = ha.x;  // T_MAX for a miss (:155)
    V3 normal = v3(ha.y, ha.z, ha.w);

    // hit_light (:107-121) limited to t_hit
    {
        int light = -1;
        float t_max = t_hit;
        for (uint32_t l = 0; l < sc.n_lights; ++l) {
            float t;
            const rt_light lt = load_light_uniform(sc.lights, l);
            if (light_hit(lt, ro, rd, T_MIN, t_max, &t)) {
                t_max = t;
                light = (int)l;
            }
        }
        if (light >= 0) {
            V3 emitted = ld3(sc.lights[light].intensity);
            float4 r = pa.rad[id];
            V3 add = mul_elemwise(beta, emitted);
            if (!prev_specular) {
                float p_light = (1.0f / (float)sc.n_lights) * light_pdf(sc.lights[light], ro, rd);  // :162
                float w = mis_weight(prev_pdf, p_light);
                add = w * add;
            }
            r.x += add.x;
            r.y += add.y;
            r.z += add.z;
            pa.rad[id] = r;
            return 0;  // light terminates the path (:166)
        }
    }
    if (!have_hit) return 0;  // escaped (:170-173)

    // medium step (:177-199)
    if (medium >= 0) {
        float transmittance;
        MediumInteraction mi;
        int k = medium_sample_interaction(sc.media[medium], ro, rd, t_hit, rng, &transmittance, &mi);
        if (k == MI_NULL) {
            beta = beta * transmittance;
        } else if (k == MI_ABSORPTION) {
            float4 r = pa.rad[id];
            V3 add = mul_elemwise(beta, mi.emission);
            r.x += add.x;
            r.y += add.y;
            r.z += add.z;
            pa.rad[id] = r;
            return 0;
        } else {
            depth -= 1;  // u32, wraps at 0 exactly like the release-mode reference (:193)
            beta = beta * (transmittance * mi.f / mi.pdf);
            out.o4 = make_float4(mi.origin.x, mi.origin.y, mi.origin.z, o4.w);
            out.d4 = make_float4(mi.dir.x, mi.dir.y, mi.dir.z, prev_pdf);
            out.b4 = make_float4(beta.x, beta.y, beta.z, __uint_as_float(depth));
            return depth > 0 ? QF_ALIVE : 0;  // `continue`
        }
    }

    bool front_face = (hb.x >> 31) != 0;
    if (ob_kind == RT_SURF_MESH) {  // hit_mesh left the normal to us: primitives.rs:175, then object.rs:60-63
        normal = normalise(cross(v3(q1.x, q1.y, q1.z), v3(q2.x, q2.y, q2.z)));
        if (ob_has_transform) normal = normalise(mat_transpose_mul(ob.inv_m, normal));
    }
    V3 interaction_point = ro + rd * t_hit;
    V3 wo_local = to_shading_coords(-rd, normal);

    BsdfSample bs = bsdf_sample(mat, wo_local, rng);
    uint32_t flags = 0;

    if (!bs.is_specular) {  // NEE (:210-238); the visibility test is deferred to shadow-connect
        uint32_t light_index = rng.usize(sc.n_lights);  // sample_lights_uniform (:124-131)
        float select_light_pdf = 1.0f / (float)sc.n_lights;
        const rt_light nee_light = sc.n_lights == 1u ? load_light_uniform(sc.lights, 0u) : sc.lights[light_index];
        LightSample ls = light_sample(nee_light, interaction_point, rng);
        V3 wi_local = to_shading_coords(ls.direction, normal);
        float p_light = ls.pdf * select_light_pdf;
        V3 bsdf_val = bsdf_eval(mat, wi_local, wo_local);
        float cosv = fabsf(wi_local.z);
        V3 contribution = mul_elemwise(ls.spectrum, bsdf_val * cosv);
        float p_bsdf = bsdf_pdf(mat, wi_local, wo_local);
        float w = mis_weight(p_light, p_bsdf);
        V3 add = (w * mul_elemwise(beta, contribution)) / p_light;
        // The reference traces the shadow ray first and evaluates the term afterwards (:219-236).  With the term in
        // hand, a term that cannot change a bit of the radiance needs no ray: p_light = 0 (never added, :230) or
        // exactly (+-0, +-0, +-0) (r + 0 = r; the radiance starts at +0 and sums of non-negative-zero terms never
        // reach -0; a NaN term is not zero and is traced).  4.4 % of the bench frame's shadow rays -- the light
        // behind the surface, Diffuse::eval = 0 -- and its longest ones: they start on the far side of the mesh and
        // run through it (frame 58.6 -> 53.9 ms, same bits).
        const bool useless = rp.skip_zero_nee && (!(p_light > 0.0f) || (add.x == 0.0f && add.y == 0.0f && add.z == 0.0f));
        if (!useless) {
            out.sh_d = make_float4(ls.direction.x, ls.direction.y, ls.direction.z, 0.0f);
            out.sh_c = make_float4(add.x, add.y, add.z, p_light > 0.0f ? 1.0f : 0.0f);
            flags |= QF_SHADOW;
        } else {
            flags |= QF_SKIPPED;
        }
    }

    // throughput and next ray (:241-250)
    V3 bsdf_val_cos = bs.spectrum * fabsf(bs.dir_in.z);
    beta = mul_elemwise(beta, bsdf_val_cos) / bs.pdf;
    prev_pdf = bs.pdf;
    V3 nd = to_world_coords(bs.dir_in, normal);

    if (bs.is_transmission && ob_interface)  // :252-261
        medium = front_face ? ob_inside : ob_outside;

    // Russian roulette (:263-272)
    float q = 1.0f - max_elem(beta);
    const bool survives = !(rng.f32() < q);
    // the vertex is the origin of the next ray AND of the pending shadow ray
    out.o4 = make_float4(interaction_point.x, interaction_point.y, interaction_point.z, pack_medium(medium, bs.is_specular));
    if (!survives) return flags;
    beta = div_assign(beta, 1.0f - q);
    out.d4 = make_float4(nd.x, nd.y, nd.z, prev_pdf);
    out.b4 = make_float4(beta.x, beta.y, beta.z, __uint_as_float(depth));
    if (depth > 0) flags |= QF_ALIVE;
    return flags;
}

// Which paths of a bounce have anything to shade.  From the second bounce on most rays leave the scene (four in five
// after the first diffuse bounce of the bench frame) and nineteen in twenty of those pass every light: shade_path reads
// 76 B of state for each to find that it has nothing to do, and the paths that did hit something are scattered thinly
// over the waves -- a wave pays the whole shading sequence for a dozen busy lanes (second k_shade of the bench frame:
// 4.3 ms, of which 2.4 for reading the escaped paths and 1.9 for 21 M hits that cost 0.8 when they are dense).  This
// kernel lists the positions that need shade_path -- a hit, or a light on the escaped ray (hit_light over the whole ray,
// :107-121; `any` light is all that matters here: the trace kernels test it while they have the ray, mark_escaped, and
// say so in the miss record, so 8 B per path are all this kernel reads) -- per segment, in position order up to the order the blocks'
// reservations arrive in; k_shade then runs on the list.  A path that is not listed is one for which shade_path
// returns 0 without touching anything.
constexpr uint32_t CLASSIFY_TILE = SHADE_BLOCK * 4;  // four consecutive positions per thread: 32 B of miss / hit records
__global__ __launch_bounds__(SHADE_BLOCK) void k_classify(PathArrays pa, uint32_t* list) {
    __shared__ uint32_t wave_cnt[SHADE_BLOCK / 64];
    __shared__ uint32_t base;
    const uint32_t seg = blockIdx.x % pa.ns, first = (blockIdx.x / pa.ns) * CLASSIFY_TILE;
    const uint32_t n = pa.cnt[CNT_ALIVE + seg];
    if (first >= n) return;
    const uint32_t seg0 = seg * pa.seg_stride, p0 = first + threadIdx.x * 4u;
    uint32_t bits = 0u;  // bit j: position p0 + j needs shading -- a hit, or an escaped ray with a light on it (mark_escaped)
    if (p0 + 3u < n) {   // (segments start at multiples of the shade block: the four records are 32-byte aligned)
        const uint4* h = reinterpret_cast<const uint4*>(pa.hit_b + seg0 + p0);
        const uint4 a = h[0], b = h[1];
        bits = ((a.x != RT_MISS || a.y != 0u) ? 1u : 0u) | ((a.z != RT_MISS || a.w != 0u) ? 2u : 0u) |
               ((b.x != RT_MISS || b.y != 0u) ? 4u : 0u) | ((b.z != RT_MISS || b.w != 0u) ? 8u : 0u);
    } else {
        for (uint32_t q = 0; q < 4u && p0 + q < n; ++q) {
            const uint2 hb = pa.hit_b[seg0 + p0 + q];
            if (hb.x != RT_MISS || hb.y != 0u) bits |= 1u << q;
        }
    }
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    uint32_t before = 0u, total = 0u;  // listed positions in the lanes below this one / in the wave
    for (uint32_t q = 0; q < 4u; ++q) {
        const unsigned long long m = __ballot((bits >> q) & 1u);
        before += (uint32_t)__popcll(m & below);
        total += (uint32_t)__popcll(m);
    }
    if (lane == 0) wave_cnt[wave] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < SHADE_BLOCK / 64; ++w) t += wave_cnt[w];
        base = t ? atomicAdd(&pa.list_cnt[seg * LIST_STRIDE], t) : 0u;
    }
    __syncthreads();
    if (bits) {
        uint32_t pos = seg0 + base + before;
        for (uint32_t w = 0; w < wave; ++w) pos += wave_cnt[w];
        for (uint32_t q = 0; q < 4u; ++q)
            if ((bits >> q) & 1u) list[pos++] = p0 + q;
    }
}

// Block b works on segment b % ns (consecutive blocks on different segments: see NSEG above), on the 256 positions
// from (b / ns) * 256 of that segment (of the segment's list, when the bounce has one).  After shading, the block's survivors and shadow rays are counted (ballots),
// the block reserves room for both in its segment of the next region / of the shadow records with one atomic, and
// every lane writes its records to its place: block offset + waves before + rank inside the wave's ballot.  Inside a
// block the order of the paths is kept, between blocks it is the order the reservations arrive in -- close to the
// launch order, so what was coherent stays so, and no bit of the image depends on it.
// LISTED: the grid is ns x (tiles per segment, capped at RT_SHADE_TILES) and a block takes every (gridDim.x / ns)-th tile of its
// segment's list.  The grid can only be sized from what the list is drawn from; one block per tile starts a million blocks
// to find four in five of them beyond the list's end, each holding a wave slot for the microsecond its count takes to
// arrive.  The dense launches keep one tile per block: the same loop around their tile costs them 13 % (registers live
// across it, a third barrier).
template <bool LISTED>
__global__ __launch_bounds__(SHADE_BLOCK) void k_shade(DScene sc, PathArrays pa, RenderParams rp) {
    __shared__ uint32_t wave_cnt[3][SHADE_BLOCK / 64];
    __shared__ uint32_t base[2];
    const uint32_t seg = blockIdx.x % pa.ns, tile_stride = LISTED ? (gridDim.x / pa.ns) * SHADE_BLOCK : 0xFFFFFF00u;
    const uint32_t n = LISTED ? pa.list_cnt[seg * LIST_STRIDE] : pa.cnt[CNT_ALIVE + seg];
    const uint32_t seg0 = seg * pa.seg_stride;
    const uint32_t oseg = seg / (pa.ns / pa.ns_out), oseg0 = oseg * pa.out_stride;  // where this segment's survivors go
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (uint32_t first = (blockIdx.x / pa.ns) * SHADE_BLOCK; first < n; first += tile_stride) {  // uniform per block; one pass unless LISTED
        const uint32_t local = first + threadIdx.x;
        uint32_t flags = 0;
        ShadeOut out;
        if (local < n) flags = shade_path(sc, pa, rp, seg0 + (LISTED ? pa.list[seg0 + local] : local), rp.bounce, out);
        const unsigned long long alive = __ballot((flags & QF_ALIVE) != 0);
        const unsigned long long shadow = __ballot((flags & QF_SHADOW) != 0);
        const unsigned long long skipped = __ballot((flags & QF_SKIPPED) != 0);
        if (lane == 0) {
            wave_cnt[0][wave] = (uint32_t)__popcll(alive);
            wave_cnt[1][wave] = (uint32_t)__popcll(shadow);
            wave_cnt[2][wave] = (uint32_t)__popcll(skipped);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t t[3] = {0, 0, 0};
            for (int w = 0; w < SHADE_BLOCK / 64; ++w)
                for (int q = 0; q < 3; ++q) t[q] += wave_cnt[q][w];
            unsigned long long r = 0ull;
            if (t[0] | t[1]) r = atomicAdd(&pa.acc[oseg * ACC_STRIDE], (unsigned long long)t[0] | ((unsigned long long)t[1] << 32));
            base[0] = (uint32_t)r;
            base[1] = (uint32_t)(r >> 32);
            if (t[2]) atomicAdd(pa.acc_skipped, t[2]);
        }
        __syncthreads();
        if (flags & QF_ALIVE) {
            uint32_t pos = oseg0 + base[0] + (uint32_t)__popcll(alive & below);
            for (uint32_t w = 0; w < wave; ++w) pos += wave_cnt[0][w];
            pa.nray_o[pos] = out.o4;
            pa.nray_d[pos] = out.d4;
            pa.nbeta[pos] = out.b4;
            pa.noid[pos] = out.id;
        }
        if (flags & QF_SHADOW) {
            uint32_t pos = oseg0 + base[1] + (uint32_t)__popcll(shadow & below);
            for (uint32_t w = 0; w < wave; ++w) pos += wave_cnt[1][w];
            pa.sh_o[pos] = make_float4(out.o4.x, out.o4.y, out.o4.z, __uint_as_float(out.id));
            pa.sh_d[pos] = out.sh_d;
            pa.sh_c[pos] = out.sh_c;
        }
        if (!LISTED) break;
        __syncthreads();  // the next tile reuses wave_cnt and base
    }
}

// One wave after every k_shade: the reservations become the counts of the next bounce (CNT_* layout) -- for the kernels
// that follow in the stream and, through pinned memory, for the host (totals, fullest segment) -- and the accumulators
// and the work counters of the trace launches start again at zero.
__global__ __launch_bounds__(64) void k_publish(PathArrays pa, uint32_t* cnt_out, uint32_t* counts_host) {
    const uint32_t s = threadIdx.x;  // NSEG == 64 == one wave
    const unsigned long long r = pa.acc[s * ACC_STRIDE];
    pa.acc[s * ACC_STRIDE] = 0ull;
    const uint32_t a = (uint32_t)r, sh = (uint32_t)(r >> 32);  // zero beyond ns_out: nobody reserved there
    cnt_out[CNT_ALIVE + s] = a;
    cnt_out[CNT_SHADOW + s] = sh;
    if (s < pa.ns_out) {
        cnt_out[CNT_MIXED + s] = sh;
        cnt_out[CNT_MIXED + pa.ns_out + s] = a;
    }
    {  // the segments with items, in order: of the paths, of the shadow rays, of both in a mixed launch's order
        const unsigned long long ma = __ballot(a != 0u), ms = __ballot(sh != 0u);
        const uint32_t ra = __builtin_amdgcn_mbcnt_hi((uint32_t)(ma >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ma, 0u));
        const uint32_t rs = __builtin_amdgcn_mbcnt_hi((uint32_t)(ms >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ms, 0u));
        const uint32_t na = (uint32_t)__popcll(ma), nsh = (uint32_t)__popcll(ms);
        if (a != 0u) {
            cnt_out[CNT_LIVE_ALIVE + 1u + ra] = s;
            cnt_out[CNT_LIVE_MIXED + 1u + nsh + ra] = pa.ns_out + s;
        }
        if (sh != 0u) {
            cnt_out[CNT_LIVE_SHADOW + 1u + rs] = s;
            cnt_out[CNT_LIVE_MIXED + 1u + rs] = s;
        }
        if (s == 0) {
            cnt_out[CNT_LIVE_ALIVE] = na;
            cnt_out[CNT_LIVE_SHADOW] = nsh;
            cnt_out[CNT_LIVE_MIXED] = na + nsh;
        }
    }
    uint32_t ta = a, ts = sh, ma = a;
    for (int off = 32; off > 0; off >>= 1) {
        ta += __shfl_down(ta, off);
        ts += __shfl_down(ts, off);
        ma = max(ma, (uint32_t)__shfl_down(ma, off));
    }
    for (uint32_t i = s; i < WORK_BUF_U32; i += 64u) pa.work[i] = 0u;
    pa.list_cnt[s * LIST_STRIDE] = 0u;
    if (s == 0) {
        const uint32_t sk = *pa.acc_skipped;
        *pa.acc_skipped = 0u;
        cnt_out[CNT_TOT_ALIVE] = ta;
        cnt_out[CNT_TOT_SHADOW] = ts;
        counts_host[0] = ta;  // paths alive after this bounce
        counts_host[1] = ts;  // its shadow rays
        counts_host[2] = sk;  // shadow rays not needed
        counts_host[3] = ma;  // paths in the fullest segment
        __threadfence_system();
    }
}

// ---------------------------------------------------------------- tail
// A frame of a few thousand paths spends its time in launches and host round trips (5 launches and a
// sync per bounce).  When no more than RT_TAIL_PATHS (default 131 072) paths are alive, k_tail runs each
// of them to its end inside ONE persistent launch: a lane traces the path's ray, shades the vertex when
// the ray finishes (the same shade_path), traces the shadow ray if one was produced, adds the pending
// light term if it is unoccluded, and goes on with the next bounce.  Per path the order of operations
// -- and of the float additions into its radiance -- is the wavefront's, so the image is bit-identical
// (tests/test_render_gpu.py renders every scene both ways).
// It is NOT a cure for the tail of a big frame: there every bounce is bounded by its longest ray
// (hundreds of dependent fetches, ~0.35 ms) in either scheme, and with shading inside the traversal loop
// each iteration costs more -- 219 000 paths took 4.0 ms here against 4.3 ms as five wavefront rounds
// (profiles/r01/tuning.md), hence the low threshold.
struct TailIO {
    PathArrays pa;      // the region the paths are in; their state is updated IN PLACE (nothing is compacted here)
    const DScene* sc;
    const RenderParams* rp;
    uint32_t n_closest, n_shadow, n_skipped, iters, max_iters;  // per lane
    unsigned long long n_int, n_tri;                   // closest-hit visits (COUNT)
    unsigned long long n_int_sh, n_tri_sh;             // shadow visits (COUNT)
    bool alive_after;                                  // the path goes on after its pending shadow ray
    uint32_t id;                                       // original slot of the lane's path
    RT_DEV uint32_t load(uint32_t i, V3* o, V3* d) {
        float4 o4 = pa.ray_o[i], d4 = pa.ray_d[i];
        *o = v3(o4.x, o4.y, o4.z);
        *d = v3(d4.x, d4.y, d4.z);
        n_closest += 1;
        iters = 1;
        if (iters > max_iters) max_iters = iters;
        return i;
    }
    static constexpr bool DEFER_NORMAL = true;
    RT_DEV void hit_mesh(uint32_t i, float t, uint32_t meta, uint32_t tri) const {
        ((float*)&pa.hit_a[i])[0] = t;
        pa.hit_b[i] = make_uint2(meta, tri);
    }
    static constexpr bool CONTINUES = true;  // next() carries the lane on to the path's shadow ray / next bounce
    RT_DEV bool first_is_any(uint32_t) const { return false; }
    RT_DEV void hit(uint32_t i, float t, V3 n, uint32_t meta, uint32_t prim) const {
        pa.hit_a[i] = make_float4(t, n.x, n.y, n.z);
        pa.hit_b[i] = make_uint2(meta, prim);
    }
    RT_DEV void escaped(uint32_t, const V3&, const V3&) const {}  // (next() writes the miss record: the lane shades it itself)
    RT_DEV void emit_closest(uint32_t, bool, const Visits&) {}
    RT_DEV void emit_any(uint32_t, bool, const Visits&) {}
    RT_DEV bool next_bounce(uint32_t i, V3* o, V3* d, bool* any) {
        float4 o4 = pa.ray_o[i], d4 = pa.ray_d[i];
        *o = v3(o4.x, o4.y, o4.z);
        *d = v3(d4.x, d4.y, d4.z);
        *any = false;
        n_closest += 1;
        iters += 1;
        if (iters > max_iters) max_iters = iters;
        return true;
    }
    RT_DEV bool next(uint32_t i, bool was_any, bool got, bool occluded, const Visits& vis, V3* o, V3* d, bool* any) {
        if (!was_any) {
            n_int += vis.n_int;
            n_tri += vis.n_tri;
            if (!got) pa.hit_b[i] = make_uint2(RT_MISS, RT_MISS);
            ShadeOut out;
            const uint32_t flags = shade_path(*sc, pa, *rp, i, rp->bounce + iters - 1u, out);
            id = out.id;
            alive_after = (flags & QF_ALIVE) != 0;
            if (flags & QF_SKIPPED) n_skipped += 1;
            if (flags & QF_ALIVE) {
                pa.ray_o[i] = out.o4;
                pa.ray_d[i] = out.d4;
                pa.beta[i] = out.b4;
            }
            if (flags & QF_SHADOW) {
                pa.sh_c[i] = out.sh_c;
                *o = v3(out.o4.x, out.o4.y, out.o4.z);
                *d = v3(out.sh_d.x, out.sh_d.y, out.sh_d.z);
                *any = true;
                n_shadow += 1;
                return true;
            }
        } else {
            n_int_sh += vis.n_int;
            n_tri_sh += vis.n_tri;
            if (!occluded) add_pending_nee(pa, i, id);
        }
        return alive_after ? next_bounce(i, o, d, any) : false;
    }
};

RT_DEV unsigned long long wave_sum_u64(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

// pa.tail[0..7]: closest rays, shadow rays, n_int, n_tri (closest), n_int, n_tri (shadow), max iterations, shadow rays not needed
template <int COUNT>
__global__ __launch_bounds__(RT_BLOCK) void k_tail(DScene sc, PathArrays pa, RenderParams rp, TuneBase base, uint32_t* overflow,
                                                   uint32_t overflow_stride) {
    __shared__ uint32_t lds_stack[RT_STACK_LDS * RT_BLOCK];
    const uint32_t gtid = blockIdx.x * RT_BLOCK + threadIdx.x;
    const uint32_t n = pa.cnt[CNT_TOT_ALIVE];
    const TraceTune tune = render_tune(base, n, pa.ns, pa.seg_stride);
    TailIO io{pa, &sc, &rp, 0u, 0u, 0u, 0u, 0u, 0ull, 0ull, 0ull, 0ull, false, 0u};
    trace_loop<TRACE_PATH, COUNT, RT_STACK_LDS, RT_BLOCK>(sc, io, n, pa.work, tune, T_MIN, __builtin_inff(),
                                                          lds_stack + threadIdx.x, overflow + gtid, overflow_stride, pa.cnt + CNT_ALIVE, 0u, pa.cnt + CNT_LIVE_ALIVE);
    unsigned long long v[6] = {io.n_closest, io.n_shadow, io.n_int, io.n_tri, io.n_int_sh, io.n_tri_sh};
    uint32_t mx = io.max_iters;
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_down(mx, off));
    for (int k = 0; k < 6; ++k) v[k] = wave_sum_u64(v[k]);
    const unsigned long long sk = wave_sum_u64(io.n_skipped);
    if ((threadIdx.x & 63) == 0) {
        for (int k = 0; k < 6; ++k)
            if (v[k]) atomicAdd(&pa.tail[k], v[k]);
        atomicMax(&pa.tail[6], (unsigned long long)mx);
        if (sk) atomicAdd(&pa.tail[7], sk);
    }
}

// ---------------------------------------------------------------- mixed (every bounce but the largest)
// A traversal launch ends with a drain -- waves finishing their last, longest rays (~0.35 ms) while the
// rest of the chip idles -- and a bounce has two launches: the shadow rays of bounce k, then the rays of
// bounce k+1.  Neither depends on the other (one adds to the radiance, the other writes hit records), so up to
// RT_MIX_PATHS rays (default 128 Mi; per ray the mixed loop costs a little more than the dedicated kernels, per launch
// it saves one drain: 1080p frame 68.4 / 66.1 / 68.4 ms at 4 Mi / 64-128 Mi / 256 Mi) they share ONE launch of 2 NSEG
// segments: segment s < NSEG is segment s of the shadow records, segment NSEG + s is segment s of the region the
// survivors went to.  k_shade of bounce k+1 still runs after both, so the order of additions into a path's radiance is
// unchanged.  `pa` is the view of bounce k+1 (its region, the counts k_publish wrote after bounce k).
struct MixedIO {
    PathArrays pa;
    uint32_t n_shadow_items;  // NSEG * seg_cap: items below it are shadow records, item i at or above it is position i - n_shadow_items
    unsigned long long n_int, n_tri, n_int_sh, n_tri_sh;
    const DScene* sc;
    static constexpr bool CONTINUES = false;  // a finished ray is finished: the lane takes a new one
    RT_DEV bool first_is_any(uint32_t i) const { return i < n_shadow_items; }
    RT_DEV uint32_t load(uint32_t i, V3* o, V3* d) const {  // the tag is the item
        const bool sh = i < n_shadow_items;
        const uint32_t k = sh ? i : i - n_shadow_items;
        float4 o4, d4;
        if (sh) {
            o4 = pa.sh_o[k];
            d4 = pa.sh_d[k];
        } else {
            o4 = pa.ray_o[k];
            d4 = pa.ray_d[k];
        }
        *o = v3(o4.x, o4.y, o4.z);
        *d = v3(d4.x, d4.y, d4.z);
        return i;
    }
    static constexpr bool DEFER_NORMAL = true;
    RT_DEV void hit_mesh(uint32_t i, float t, uint32_t meta, uint32_t tri) const {
        const uint32_t k = i - n_shadow_items;
        ((float*)&pa.hit_a[k])[0] = t;
        pa.hit_b[k] = make_uint2(meta, tri);
    }
    RT_DEV void hit(uint32_t i, float t, V3 n, uint32_t meta, uint32_t prim) const {
        const uint32_t k = i - n_shadow_items;
        pa.hit_a[k] = make_float4(t, n.x, n.y, n.z);
        pa.hit_b[k] = make_uint2(meta, prim);
    }
    RT_DEV void escaped(uint32_t i, const V3& o, const V3& d) const { mark_escaped(*sc, pa, i - n_shadow_items, o, d); }
    RT_DEV void emit_closest(uint32_t, bool, const Visits&) {}
    RT_DEV void emit_any(uint32_t, bool, const Visits&) {}
    RT_DEV bool next(uint32_t i, bool was_any, bool got, bool occluded, const Visits& vis, V3*, V3*, bool*) {
        if (was_any) {
            n_int_sh += vis.n_int;
            n_tri_sh += vis.n_tri;
            if (!occluded) add_pending_nee(pa, i, __float_as_uint(pa.sh_o[i].w));
        } else {
            n_int += vis.n_int;
            n_tri += vis.n_tri;
        }
        return false;
    }
};

template <int COUNT>
__global__ RT_TRACE_ATTR void k_mixed(DScene sc, PathArrays pa, TuneBase base, uint32_t* overflow, uint32_t overflow_stride) {
    __shared__ uint32_t lds_stack[RT_STACK_LDS * RT_BLOCK];
    const uint32_t gtid = blockIdx.x * RT_BLOCK + threadIdx.x;
    const uint32_t n = pa.cnt[CNT_TOT_SHADOW] + pa.cnt[CNT_TOT_ALIVE];  // shadow rays of this bounce, then the rays of the next
    const TraceTune tune = render_tune(base, n, 2u * pa.ns, pa.seg_stride);
    MixedIO io{pa, pa.n_positions, 0ull, 0ull, 0ull, 0ull, &sc};
    trace_loop<TRACE_PATH, COUNT, RT_STACK_LDS, RT_BLOCK>(sc, io, n, pa.work, tune, T_MIN, __builtin_inff(),
                                                          lds_stack + threadIdx.x, overflow + gtid, overflow_stride, pa.cnt + CNT_MIXED, 0u, pa.cnt + CNT_LIVE_MIXED);
    if (COUNT) {  // accounted with the tail kernel's share: rays traced outside k_extend / k_shadow
        add_visits(&pa.tail[2], io.n_int, io.n_tri);
        add_visits(&pa.tail[4], io.n_int_sh, io.n_tri_sh);
    }
}

__global__ RT_TRACE_ATTR_LEAN void k_mixed_lean(DScene sc, PathArrays pa, TuneBase base, uint32_t* overflow, uint32_t overflow_stride) {
    __shared__ uint32_t lds_stack[RT_STACK_LEAN * RT_BLOCK];
    const uint32_t gtid = blockIdx.x * RT_BLOCK + threadIdx.x;
    const uint32_t n = pa.cnt[CNT_TOT_SHADOW] + pa.cnt[CNT_TOT_ALIVE];
    const TraceTune tune = render_tune(base, n, 2u * pa.ns, pa.seg_stride);
    MixedIO io{pa, pa.n_positions, 0ull, 0ull, 0ull, 0ull, &sc};
    trace_loop<TRACE_PATH, false, RT_STACK_LEAN, RT_BLOCK, RT_LEAN_FEAT, RT_LEAN_OVERFLOW>(
        sc, io, n, pa.work, tune, T_MIN, __builtin_inff(), lds_stack + threadIdx.x, overflow + gtid, overflow_stride, pa.cnt + CNT_MIXED, 0u, pa.cnt + CNT_LIVE_MIXED);
}

// ---------------------------------------------------------------- resolve
// colour += ray_colour(...) over the samples of the batch, in sample order (:302-308).
__global__ __launch_bounds__(256) void k_resolve(PathArrays pa, float* rgb_sum, uint32_t npix, uint32_t s_count,
                                                 int first_batch, uint32_t pixel_major) {
    uint32_t pixel = blockIdx.x * 256u + threadIdx.x;
    if (pixel >= npix) return;
    float cx = 0.0f, cy = 0.0f, cz = 0.0f;
    if (!first_batch) {
        cx = rgb_sum[3 * (size_t)pixel];
        cy = rgb_sum[3 * (size_t)pixel + 1];
        cz = rgb_sum[3 * (size_t)pixel + 2];
    }
    for (uint32_t s = 0; s < s_count; ++s) {
        float4 r = pa.rad[pixel_major ? (size_t)pixel * s_count + s : (size_t)s * npix + pixel];
        cx += r.x;
        cy += r.y;
        cz += r.z;
    }
    rgb_sum[3 * (size_t)pixel] = cx;
    rgb_sum[3 * (size_t)pixel + 1] = cy;
    rgb_sum[3 * (size_t)pixel + 2] = cz;
}

// The same sum for the pixel-major slot order: a pixel's samples are s_count CONSECUTIVE radiance records, so one
// thread per pixel would read with a stride of s_count * 16 bytes (measured: 2.3 ms instead of 0.37 ms at 1080p / 64
// spp).  A block takes 64 pixels: all 256 threads copy tiles of 16 samples -- runs of 256 contiguous bytes -- into
// LDS, then one thread per pixel adds its 16 samples in sample order.
constexpr uint32_t RESOLVE_PIXELS = 64, RESOLVE_TILE = 16;
__global__ __launch_bounds__(256) void k_resolve_pixel_major(PathArrays pa, float* rgb_sum, uint32_t npix, uint32_t s_count,
                                                             int first_batch) {
    __shared__ float4 tile[RESOLVE_PIXELS][RESOLVE_TILE + 1];
    const uint32_t p0 = blockIdx.x * RESOLVE_PIXELS;
    const uint32_t pixel = p0 + threadIdx.x;  // threads 0..63 own a pixel
    const bool owner = threadIdx.x < RESOLVE_PIXELS && pixel < npix;
    float cx = 0.0f, cy = 0.0f, cz = 0.0f;
    if (owner && !first_batch) {
        cx = rgb_sum[3 * (size_t)pixel];
        cy = rgb_sum[3 * (size_t)pixel + 1];
        cz = rgb_sum[3 * (size_t)pixel + 2];
    }
    for (uint32_t s0 = 0; s0 < s_count; s0 += RESOLVE_TILE) {
        const uint32_t ts = min(RESOLVE_TILE, s_count - s0);
        for (uint32_t e = threadIdx.x; e < RESOLVE_PIXELS * RESOLVE_TILE; e += 256u) {
            const uint32_t pl = e / RESOLVE_TILE, sl = e % RESOLVE_TILE;
            if (sl < ts && p0 + pl < npix) tile[pl][sl] = pa.rad[(size_t)(p0 + pl) * s_count + s0 + sl];
        }
        __syncthreads();
        if (owner)
            for (uint32_t sl = 0; sl < ts; ++sl) {
                const float4 r = tile[threadIdx.x][sl];
                cx += r.x;
                cy += r.y;
                cz += r.z;
            }
        __syncthreads();
    }
    if (owner) {
        rgb_sum[3 * (size_t)pixel] = cx;
        rgb_sum[3 * (size_t)pixel + 1] = cy;
        rgb_sum[3 * (size_t)pixel + 2] = cz;
    }
}

bool is_device_ptr(const void* p) {
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeDevice;
}

}  // namespace

// The wavefront renderer behind rt_render (user_rays == nullptr: camera paths of a width x height image) and
// rt_ray_colour (user_rays: device pointers to `width` caller rays, height 1; "pixel" k is ray k).
int rt_render_impl(rt_ctx* ctx, uint32_t width, uint32_t height, const rt_rays* user_rays, uint32_t spp_first,
                   uint32_t spp_count, uint32_t depth, uint64_t seed, uint32_t flags, float* rgb_sum, rt_stats* stats) {
    const char* const who = user_rays ? "rt_ray_colour" : "rt_render";  // the entry point the caller knows
    if (!ctx || !rgb_sum) {
        rth::set_error("%s: null argument", who);
        return RT_ERR_INVALID;
    }
    if (!ctx->has_scene) {
        rth::set_error("%s: no scene uploaded (call rt_upload_scene first)", who);
        return RT_ERR_STATE;
    }
    if (!user_rays && (width < 2 || height < 2)) {  // the reference divides by (w-1), (h-1) (:303-304)
        rth::set_error("%s: width and height must be >= 2", who);
        return RT_ERR_INVALID;
    }
    const rt_rays no_rays = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const rt_rays gen_rays = user_rays ? *user_rays : no_rays;
    const uint64_t npix64 = (uint64_t)width * height;
    if (npix64 > 0x7FFFFFFFull) {
        rth::set_error("%s: image too large", who);
        return RT_ERR_LIMIT;
    }
    const uint32_t npix = (uint32_t)npix64;
    if (ctx->scene.n_lights == 0 && ctx->scene.n_objects > 0) {
        // sample_lights_uniform would index an empty Vec (panic) on the first non-specular hit
        rth::set_error("%s: scene has no lights (the reference panics in sample_lights_uniform)", who);
        return RT_ERR_INVALID;
    }
    RT_HIP(hipSetDevice(ctx->device));
    hipStream_t stream = ctx->stream;

    // ---- batch size: samples in flight at once.  Every wavefront iteration ends in a tail of
    // half-empty waves and a host round trip, so the batch is made as large as memory allows
    // (measured at 1080p/32 spp: 8 Mi paths 145 ms, 64 Mi paths 71 ms): by default up to 40 % of
    // the free HBM, 196 B per path slot (two regions of 52 B, hit 24 B, shadow record 48 B, radiance 16 B, list entry 4 B).
    uint64_t max_paths = ctx->max_paths_in_flight;
    if (max_paths == 0) {
        size_t free_b = 0, total_b = 0;
        RT_HIP(hipMemGetInfo(&free_b, &total_b));
        free_b += ctx->path_state.bytes;  // our own previous allocation is reusable
        max_paths = (uint64_t)((double)free_b * 0.4 / 196.0);
        if (max_paths < ((uint64_t)1 << 20)) max_paths = (uint64_t)1 << 20;
    }
    uint32_t s_batch = (uint32_t)std::max<uint64_t>(1, max_paths / npix);
    if (s_batch > spp_count) s_batch = spp_count ? spp_count : 1;
    constexpr uint64_t MAX_BATCH = 0x7F000000ull;  // 2 NSEG segments of a mixed launch stay below 2^32 items
    while ((uint64_t)s_batch * npix > MAX_BATCH) --s_batch;
    const uint64_t P = (uint64_t)s_batch * npix;
    // positions: NSEG segments of seg_cap (a multiple of the shade block) each
    const uint32_t seg_cap = (uint32_t)(((P + NSEG - 1) / NSEG + SHADE_BLOCK - 1) / SHADE_BLOCK * SHADE_BLOCK);
    const uint64_t Ppos = (uint64_t)seg_cap * NSEG;  // >= P

    // ---- carve the path state out of one allocation
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    struct RegionOff {
        size_t ray_o, ray_d, beta, oid;
    } ro[2];
    for (int r = 0; r < 2; ++r)
        ro[r].ray_o = carve(Ppos * 16), ro[r].ray_d = carve(Ppos * 16), ro[r].beta = carve(Ppos * 16), ro[r].oid = carve(Ppos * 4);
    const size_t o_hit_a = carve(Ppos * 16), o_hit_b = carve(Ppos * 8), o_sh_o = carve(Ppos * 16), o_sh_d = carve(Ppos * 16),
                 o_sh_c = carve(Ppos * 16), o_rad = carve(P * 16);
    // small state, zeroed together at the start of every batch: accumulators, skipped total, work counters
    const size_t o_acc = carve((size_t)NSEG * ACC_STRIDE * 8), o_skip = carve(256), o_work = carve(rtd::WORK_BUF_U32 * 4),
                 o_lcnt = carve((size_t)NSEG * LIST_STRIDE * 4);
    const size_t zero_bytes = off - o_acc;
    const size_t o_list = carve(Ppos * 4);
    const size_t o_cnt = carve(2 * CNT_U32 * 4), o_vis = carve(256), o_tail = carve(256);
    int rc = ctx->path_state.ensure(off);
    if (rc != RT_OK) return rc;
    if ((rc = rt_ctx_ensure_readback(ctx)) != RT_OK) return rc;
    char* base = (char*)ctx->path_state.p;
    uint32_t* const counts = (uint32_t*)(base + o_cnt);
    // the arrays bounce k works on: its paths in region k & 1, the survivors go to the other one
    // segments per bounce (see NSEG above): what a launch of that many rays wants; never growing over the bounces.
    // ns_of[k + 1] has to be known when bounce k is shaded, i.e. from bounds the host has then.
    std::vector<uint32_t> ns_of;
    auto segments_for = [](uint64_t n) {
        uint32_t ns = 1;
        while (ns < NSEG && (uint64_t)ns * 2u * 131072u <= n) ns *= 2;
        return ns;
    };
    auto view = [&](size_t k) {
        PathArrays q;
        const RegionOff &in = ro[k & 1], &out = ro[(k + 1) & 1];
        q.ray_o = (float4*)(base + in.ray_o);
        q.ray_d = (float4*)(base + in.ray_d);
        q.beta = (float4*)(base + in.beta);
        q.oid = k == 0 ? nullptr : (uint32_t*)(base + in.oid);  // first bounce: position = slot
        q.nray_o = (float4*)(base + out.ray_o);
        q.nray_d = (float4*)(base + out.ray_d);
        q.nbeta = (float4*)(base + out.beta);
        q.noid = (uint32_t*)(base + out.oid);
        q.rad = (float4*)(base + o_rad);
        q.hit_a = (float4*)(base + o_hit_a);
        q.hit_b = (uint2*)(base + o_hit_b);
        q.sh_o = (float4*)(base + o_sh_o);
        q.sh_d = (float4*)(base + o_sh_d);
        q.sh_c = (float4*)(base + o_sh_c);
        q.cnt = counts + CNT_U32 * (k & 1);
        q.acc = (unsigned long long*)(base + o_acc);
        q.acc_skipped = (uint32_t*)(base + o_skip);
        q.n_positions = (uint32_t)Ppos;
        q.ns = ns_of[k];
        q.seg_stride = seg_cap * (NSEG / q.ns);
        q.ns_out = ns_of[k + 1];
        q.out_stride = seg_cap * (NSEG / q.ns_out);
        q.work = (uint32_t*)(base + o_work);
        q.list = nullptr;
        q.list_cnt = (uint32_t*)(base + o_lcnt);
        q.visit = (unsigned long long*)(base + o_vis);
        q.tail = (unsigned long long*)(base + o_tail);
        return q;
    };
    ns_of.assign(2, NSEG);
    const PathArrays pa = view(0);  // what does not depend on the bounce: rad, work, visit, tail
    RT_HIP(hipMemsetAsync(pa.visit, 0, 256, stream));
    RT_HIP(hipMemsetAsync(pa.tail, 0, 256, stream));

    // ---- output: device accumulator (the caller's buffer if it is device memory)
    float* d_sum = rgb_sum;
    const size_t sum_bytes = (size_t)npix * 3 * sizeof(float);
    const bool host_out = !is_device_ptr(rgb_sum);
    if (host_out) {
        if ((rc = ctx->sample_sum.ensure(sum_bytes)) != RT_OK) return rc;
        d_sum = (float*)ctx->sample_sum.p;
    }

    const int trace_grid_max = RT_TRACE_GRID;
    if ((rc = rt_ctx_ensure_overflow(ctx, (size_t)trace_grid_max * RT_BLOCK)) != RT_OK) return rc;
    const uint32_t ostride = (uint32_t)trace_grid_max * RT_BLOCK;
    uint32_t* overflow = (uint32_t*)ctx->overflow.p;

    // visit counters: the reference's visits (equal to the oracle's) or the fetches the timed kernels perform
    const int count = (flags & RT_RENDER_COUNT_VISITS) ? rtd::COUNT_REFERENCE : (flags & RT_RENDER_COUNT_FETCHES) ? rtd::COUNT_FETCHES : rtd::COUNT_NONE;
    const bool use_tail = (flags & RT_RENDER_NO_TAIL) == 0;
    static const uint64_t mix_paths = [] {  // RT_MIX_PATHS: rays (shadow + next bounce) up to which they share a launch
        const char* e = getenv("RT_MIX_PATHS");
        return e ? (uint64_t)strtoull(e, nullptr, 10) : (uint64_t)134217728;
    }();
    const bool use_mix = use_tail && mix_paths > 0;
    uint64_t mixed_closest = 0, mixed_shadow = 0;
    // scenes without analytic sphere / triangle objects (meshes, planes, transforms) take the lean kernels
    const bool lean = rt_scene_is_lean(ctx) && getenv("RT_NO_LEAN") == nullptr;
    static const uint32_t tail_paths = [] {  // RT_TAIL_PATHS overrides the switch-over point (experiments)
        const char* e = getenv("RT_TAIL_PATHS");
        return e ? (uint32_t)strtoul(e, nullptr, 10) : 131072u;  // profiles/r03/tuning.md section 4
    }();
    static const uint32_t list_paths = [] {  // RT_LIST_PATHS: bounces after the first with at least this many paths are shaded from a list (0: never)
        const char* e = getenv("RT_LIST_PATHS");
        return e ? (uint32_t)strtoul(e, nullptr, 10) : (1u << 18);  // profiles/r03/tuning.md section 11
    }();
    static const uint32_t shade_tiles = [] {  // RT_SHADE_TILES: blocks per segment of a k_shade launch, each taking every that-many-th tile
        const char* e = getenv("RT_SHADE_TILES");
        return e && atoi(e) > 0 ? (uint32_t)atoi(e) : 256u;
    }();
    static const uint32_t async_below = [] {  // RT_ASYNC_BELOW: bounces with at most this many paths are enqueued
        const char* e = getenv("RT_ASYNC_BELOW");  // without waiting for their exact counts (0: always wait)
        return e ? (uint32_t)strtoul(e, nullptr, 10) : (16u << 20);
    }();
    const bool timing = (flags & RT_RENDER_TIME_STAGES) != 0;
    // HIP events on the library's stream around every launch of a stage, summed per stage after the frame's last
    // synchronisation (the host runs ahead of the device, so a span cannot be read back where it is recorded).
    // k_extend (stage 1, the kernel the roofline figure is about) and the mixed / tail launches (stage 5) are always
    // timed; the other stages only with RT_RENDER_TIME_STAGES.  The event pairs are cached in the context.
    constexpr int N_STAGES = 6;  // generate, extend, shade, shadow, publish ("compact"), mixed + tail
    float stage_ms[N_STAGES] = {0, 0, 0, 0, 0, 0};
    struct Span {
        int stage;
        size_t pair;
    };
    std::vector<Span> spans;
    size_t pairs_used = 0;
    auto timed = [&](int k) { return timing || k == 1 || k == 5; };
    auto t_begin = [&](int k) -> int {
        if (!timed(k)) return RT_OK;
        if (pairs_used == ctx->ev_pool.size() / 2) {  // the pool only ever holds complete pairs
            hipEvent_t e0 = nullptr, e1 = nullptr;
            RT_HIP(hipEventCreate(&e0));
            const hipError_t e = hipEventCreate(&e1);
            if (e != hipSuccess) {
                (void)hipEventDestroy(e0);
                RT_HIP(e);
            }
            ctx->ev_pool.push_back(e0);
            ctx->ev_pool.push_back(e1);
        }
        spans.push_back({k, pairs_used});
        RT_HIP(hipEventRecord(ctx->ev_pool[2 * pairs_used], stream));
        return RT_OK;
    };
    auto t_end = [&](int k) -> int {
        if (!timed(k)) return RT_OK;
        RT_HIP(hipEventRecord(ctx->ev_pool[2 * pairs_used + 1], stream));
        ++pairs_used;
        return RT_OK;
    };
    auto t_collect = [&]() -> int {  // after the stream has been synchronised
        for (const Span& sp : spans) {
            float ms = 0.0f;
            RT_HIP(hipEventElapsedTime(&ms, ctx->ev_pool[2 * sp.pair], ctx->ev_pool[2 * sp.pair + 1]));
            stage_ms[sp.stage] += ms;
        }
        spans.clear();
        pairs_used = 0;
        return RT_OK;
    };

    rt_stats st;
    memset(&st, 0, sizeof st);
    RT_HIP(hipEventRecord(ctx->ev[0], stream));

    RenderParams rp;
    rp.width = width;
    rp.height = height;
    rp.npix = npix;
    rp.pixel_major = getenv("RT_SAMPLE_MAJOR") == nullptr ? 1u : 0u;
    rp.skip_zero_nee = (count != rtd::COUNT_REFERENCE && getenv("RT_TRACE_ZERO_NEE") == nullptr) ? 1u : 0u;
    rp.depth = depth;
    rp.k0 = (uint32_t)seed;
    rp.k1 = (uint32_t)(seed >> 32);
    const TuneBase tune = rt_tune_base();

    // kernel selection by counting mode / scene features, one place each
    auto launch_extend = [&](int grid, const PathArrays& q, uint32_t first) {
        if (count == rtd::COUNT_REFERENCE)
            hipLaunchKernelGGL(k_extend<rtd::COUNT_REFERENCE>, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, q, rp, first, tune, overflow, ostride);
        else if (count == rtd::COUNT_FETCHES)
            hipLaunchKernelGGL(k_extend<rtd::COUNT_FETCHES>, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, q, rp, first, tune, overflow, ostride);
        else if (lean && first)
            hipLaunchKernelGGL(k_extend_lean<1>, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, q, rp, tune, overflow, ostride);
        else if (lean)
            hipLaunchKernelGGL(k_extend_lean<0>, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, q, rp, tune, overflow, ostride);
        else
            hipLaunchKernelGGL(k_extend<rtd::COUNT_NONE>, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, q, rp, first, tune, overflow, ostride);
    };
    auto launch_shadow = [&](int grid, const PathArrays& q) {
        if (count == rtd::COUNT_REFERENCE)
            hipLaunchKernelGGL(k_shadow<rtd::COUNT_REFERENCE>, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, q, tune, overflow, ostride);
        else if (count == rtd::COUNT_FETCHES)
            hipLaunchKernelGGL(k_shadow<rtd::COUNT_FETCHES>, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, q, tune, overflow, ostride);
        else if (lean)
            hipLaunchKernelGGL(k_shadow_lean, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, q, tune, overflow, ostride);
        else
            hipLaunchKernelGGL(k_shadow<rtd::COUNT_NONE>, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, q, tune, overflow, ostride);
    };
    auto launch_mixed = [&](int grid, const PathArrays& q) {
        if (count == rtd::COUNT_REFERENCE)
            hipLaunchKernelGGL(k_mixed<rtd::COUNT_REFERENCE>, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, q, tune, overflow, ostride);
        else if (count == rtd::COUNT_FETCHES)
            hipLaunchKernelGGL(k_mixed<rtd::COUNT_FETCHES>, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, q, tune, overflow, ostride);
        else if (lean)
            hipLaunchKernelGGL(k_mixed_lean, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, q, tune, overflow, ostride);
        else
            hipLaunchKernelGGL(k_mixed<rtd::COUNT_NONE>, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, q, tune, overflow, ostride);
    };
    auto launch_tail = [&](int grid, const PathArrays& q) {
        if (count == rtd::COUNT_REFERENCE)
            hipLaunchKernelGGL(k_tail<rtd::COUNT_REFERENCE>, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, q, rp, tune, overflow, ostride);
        else if (count == rtd::COUNT_FETCHES)
            hipLaunchKernelGGL(k_tail<rtd::COUNT_FETCHES>, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, q, rp, tune, overflow, ostride);
        else
            hipLaunchKernelGGL(k_tail<rtd::COUNT_NONE>, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, q, rp, tune, overflow, ostride);
    };

    // What the host remembers of a bounce: which launches it enqueued, and the counts k_publish wrote for it
    // (read from pinned memory once the event recorded behind that kernel has completed).
    struct Bounce {
        bool extend = false, mixed = false, shadow = false, got = false;
        uint32_t n_next = 0, n_shadow = 0, n_skipped = 0;  // paths alive after the bounce, its shadow rays traced / not needed
        uint32_t seg_max = 0;                              // survivors in the fullest segment
    };
    std::vector<Bounce> hist;
    auto wait_counts = [&](size_t k) -> int {
        if (hist[k].got) return RT_OK;
        RT_HIP(hipEventSynchronize(ctx->rb_ev[k % RT_READBACK_RING]));
        const volatile uint32_t* h = ctx->rb_host + 4 * (k % RT_READBACK_RING);
        hist[k].n_next = h[0];
        hist[k].n_shadow = h[1];
        hist[k].n_skipped = h[2];
        hist[k].seg_max = h[3];
        hist[k].got = true;
        return RT_OK;
    };

    if (spp_count == 0) RT_HIP(hipMemsetAsync(d_sum, 0, sum_bytes, stream));
    bool first_batch = true;
    for (uint32_t s0 = 0; s0 < spp_count; s0 += s_batch) {
        const uint32_t s_count = std::min(s_batch, spp_count - s0);
        const uint32_t n_paths = s_count * npix;
        rp.sample_first = spp_first + s0;
        rp.s_count = s_count;
        // first-bounce state implicit unless the tail kernel takes the batch from its first bounce (it reads
        // every path's state from memory)
        const bool implicit = !user_rays && depth > 0 && !(use_tail && n_paths <= tail_paths) && getenv("RT_NO_IMPLICIT") == nullptr;
        rp.implicit_first = implicit ? 1u : 0u;
        rp.bounce = 0u;
        // ---- the wavefront loop.  n_ub is an UPPER BOUND of the paths alive at bounce k (seg_ub: in its fullest
        // segment): exact while bounces are big (the host waits for the counts of bounce k before it enqueues bounce
        // k + 1: a round trip of tens of microseconds against launches of milliseconds, and exact counts pick the launch
        // shapes), one bounce old once they are small (n_ub <= async_below): then the host enqueues bounce k + 1 while
        // the device still works on bounce k, with grids sized from the counts of bounce k -- paths only die, in every
        // segment, so they are bounds -- and every kernel takes its true counts from device memory.  That removes the
        // host round trip and the idle gap behind it from every late bounce (0.1 ms each, profiles/r02/tuning.md).
        hist.clear();
        uint32_t n_ub = depth > 0 ? n_paths : 0;
        ns_of.assign(2, segments_for(n_paths));
        uint32_t seg_ub = std::min<uint64_t>(n_ub, (uint64_t)seg_cap * (NSEG / ns_of[0]));
        RT_HIP(hipMemsetAsync(base + o_acc, 0, zero_bytes, stream));  // accumulators, skipped total, work counters
        if ((rc = t_begin(0)) != RT_OK) return rc;
        hipLaunchKernelGGL(k_generate, dim3(implicit ? 1u : (n_paths + 255) / 256), dim3(256), 0, stream, ctx->scene, view(0), rp, n_paths, gen_rays, counts);
        if ((rc = t_end(0)) != RT_OK) return rc;
        st.paths += n_paths;
        st.batches += 1;

        size_t k = 0;
        uint32_t tail_iters_total = 0;
        bool extend_done = false;  // the rays of this bounce were traced by the previous bounce's mixed launch
        while (n_ub > 0) {
            if (k >= (size_t)1 << 20) {
                rth::set_error("%s: wavefront did not drain after 2^20 iterations", who);
                return RT_ERR_LIMIT;
            }
            // the survivors of this bounce are at most n_ub: that picks the next bounce's segments
            if (ns_of.size() < k + 2) ns_of.resize(k + 2);
            ns_of[k + 1] = std::min(ns_of[k], segments_for(n_ub));
            const PathArrays pr = view(k);
            rp.bounce = (uint32_t)k + 1;  // iteration k of `while depth > 0` draws from RNG bounce k+1 (bounce 0: camera jitter)
            const uint32_t first_implicit = (implicit && k == 0) ? 1u : 0u;
            rp.implicit_first = first_implicit;
            if (use_tail && n_ub <= tail_paths) {  // few paths left: one launch takes them to their end
                RT_HIP(hipMemsetAsync(pa.tail + 6, 0, 8, stream));
                if ((rc = t_begin(5)) != RT_OK) return rc;
                launch_tail(rt_trace_grid(n_ub), pr);
                if ((rc = t_end(5)) != RT_OK) return rc;
                unsigned long long tail_iters = 0;
                RT_HIP(hipMemcpyAsync(&tail_iters, pa.tail + 6, 8, hipMemcpyDeviceToHost, stream));
                RT_HIP(hipStreamSynchronize(stream));
                if (k > 0 && (rc = wait_counts(k - 1)) != RT_OK) return rc;
                const uint32_t n_exact = k > 0 ? hist[k - 1].n_next : n_paths;
                if (n_exact > 0) {
                    st.paths_tail += n_exact;
                    st.launches_tail += 1;
                }
                tail_iters_total = (uint32_t)tail_iters;
                break;
            }
            hist.emplace_back();
            if (!extend_done) {
                if ((rc = t_begin(1)) != RT_OK) return rc;
                launch_extend(rt_trace_grid(n_ub), pr, first_implicit);
                if ((rc = t_end(1)) != RT_OK) return rc;
                hist[k].extend = true;
            }
            extend_done = false;

            // shade + compaction: NSEG interleaved segments, grid sized from the fullest one
            const uint32_t seg_blocks = (seg_ub + SHADE_BLOCK - 1) / SHADE_BLOCK;
            if ((rc = t_begin(2)) != RT_OK) return rc;
            PathArrays ps = pr;
            if (k > 0 && list_paths > 0 && n_ub >= list_paths) {  // (the first bounce's paths nearly all hit something)
                uint32_t* const list = (uint32_t*)(base + o_list);
                hipLaunchKernelGGL(k_classify, dim3((seg_ub + CLASSIFY_TILE - 1) / CLASSIFY_TILE * pr.ns), dim3(SHADE_BLOCK), 0, stream, pr, list);
                ps.list = list;
            }
            if (ps.list) hipLaunchKernelGGL(k_shade<true>, dim3(std::min(seg_blocks, shade_tiles) * pr.ns), dim3(SHADE_BLOCK), 0, stream, ctx->scene, ps, rp);
            else hipLaunchKernelGGL(k_shade<false>, dim3(seg_blocks * pr.ns), dim3(SHADE_BLOCK), 0, stream, ctx->scene, ps, rp);
            if ((rc = t_end(2)) != RT_OK) return rc;
            if (ns_of.size() < k + 3) ns_of.resize(k + 3, ns_of[k + 1]);
            ns_of[k + 2] = ns_of[k + 1];  // placeholder until bounce k + 1 decides it (view(k + 1) only uses it for its own shade)
            const PathArrays pn = view(k + 1);  // the next bounce: the other region, the counts k_publish writes now
            if ((rc = t_begin(4)) != RT_OK) return rc;
            hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, stream, pr, (uint32_t*)pn.cnt, ctx->rb_host + 4 * (k % RT_READBACK_RING));
            if ((rc = t_end(4)) != RT_OK) return rc;
            RT_HIP(hipEventRecord(ctx->rb_ev[k % RT_READBACK_RING], stream));

            // bounds of the next bounce's rays and of this bounce's shadow rays
            uint32_t nn_ub, ns_ub, ng_ub;
            if (n_ub > async_below) {  // big bounce: wait for its exact counts
                if ((rc = wait_counts(k)) != RT_OK) return rc;
                nn_ub = hist[k].n_next;
                ns_ub = hist[k].n_shadow;
                ng_ub = hist[k].seg_max;
                if (nn_ub > n_ub || ns_ub > n_ub || ng_ub > nn_ub) {
                    rth::set_error("%s: compaction produced impossible counts", who);
                    return RT_ERR_HIP;
                }
            } else {  // small bounce: the counts of the bounce BEFORE it have arrived meanwhile (or will, without idling the device)
                if (k > 0 && (rc = wait_counts(k - 1)) != RT_OK) return rc;
                nn_ub = ns_ub = k > 0 ? std::min(n_ub, hist[k - 1].n_next) : n_ub;
                // (the fullest of the next bounce's segments holds at most the survivors of the ns / ns_out segments it merges)
                ng_ub = (uint32_t)std::min<uint64_t>(nn_ub, (uint64_t)(k > 0 ? std::min(seg_ub, hist[k - 1].seg_max) : seg_ub) * (ns_of[k] / ns_of[k + 1]));
            }

            const bool tail_next = use_tail && nn_ub <= tail_paths;  // the tail kernel traces its own first ray
            if (use_mix && ns_ub > 0 && nn_ub > 0 && !tail_next && (uint64_t)ns_ub + nn_ub <= mix_paths) {
                // the shadow rays of this bounce and the rays of the next one in one launch
                if ((rc = t_begin(5)) != RT_OK) return rc;
                launch_mixed(rt_trace_grid((size_t)ns_ub + nn_ub), pn);
                if ((rc = t_end(5)) != RT_OK) return rc;
                hist[k].mixed = true;
                extend_done = true;
            } else if (ns_ub > 0) {
                if ((rc = t_begin(3)) != RT_OK) return rc;
                launch_shadow(rt_trace_grid(ns_ub), pn);
                if ((rc = t_end(3)) != RT_OK) return rc;
                hist[k].shadow = true;
            }
            n_ub = nn_ub;
            seg_ub = ng_ub;
            ++k;
        }
        // ---- what was traced, from the counts of every bounce (all arrived by now or after this wait)
        uint32_t iter = 0;
        for (size_t b = 0; b < hist.size(); ++b) {
            if ((rc = wait_counts(b)) != RT_OK) return rc;
            const uint32_t n_b = b > 0 ? hist[b - 1].n_next : n_paths;  // paths alive at bounce b
            if (hist[b].n_next > n_b || hist[b].n_shadow > n_b) {
                rth::set_error("%s: compaction produced impossible counts", who);
                return RT_ERR_HIP;
            }
            if (n_b == 0) continue;  // a bounce enqueued on a stale bound after the last path had died
            iter += 1;
            st.rays_shadow += hist[b].n_skipped;  // issued by the reference's algorithm, traced by nobody
            st.rays_shadow_skipped += hist[b].n_skipped;
            if (hist[b].extend) {
                st.rays_closest += n_b;
                st.launches_extend += 1;
            }
            if (hist[b].mixed) {
                st.rays_shadow += hist[b].n_shadow;
                st.rays_closest += hist[b].n_next;
                mixed_closest += hist[b].n_next;
                mixed_shadow += hist[b].n_shadow;
                st.launches_tail += 1;
            } else if (hist[b].shadow && hist[b].n_shadow > 0) {
                st.rays_shadow += hist[b].n_shadow;
                st.launches_shadow += 1;
            }
        }
        iter += tail_iters_total;
        if (iter > st.iterations) st.iterations = iter;
        if (rp.pixel_major)
            hipLaunchKernelGGL(k_resolve_pixel_major, dim3((npix + RESOLVE_PIXELS - 1) / RESOLVE_PIXELS), dim3(256), 0, stream, pa,
                               d_sum, npix, s_count, first_batch ? 1 : 0);
        else
            hipLaunchKernelGGL(k_resolve, dim3((npix + 255) / 256), dim3(256), 0, stream, pa, d_sum, npix, s_count,
                               first_batch ? 1 : 0, 0u);
        first_batch = false;
    }
    RT_HIP(hipGetLastError());
    RT_HIP(hipEventRecord(ctx->ev[1], stream));
    unsigned long long visit[4] = {0, 0, 0, 0}, tail[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (count) RT_HIP(hipMemcpyAsync(visit, pa.visit, sizeof visit, hipMemcpyDeviceToHost, stream));
    if (st.launches_tail) RT_HIP(hipMemcpyAsync(tail, pa.tail, sizeof tail, hipMemcpyDeviceToHost, stream));  // tail and mixed launches
    if (host_out) RT_HIP(hipMemcpyAsync(rgb_sum, d_sum, sum_bytes, hipMemcpyDeviceToHost, stream));
    RT_HIP(hipStreamSynchronize(stream));
    if ((rc = t_collect()) != RT_OK) return rc;
    RT_HIP(hipEventElapsedTime(&st.ms_total, ctx->ev[0], ctx->ev[1]));
    st.rays_closest_tail = tail[0] + mixed_closest;
    st.rays_shadow_tail = tail[1] + mixed_shadow;
    st.nodes_visited_tail = tail[2];
    st.tris_tested_tail = tail[3];
    st.rays_closest += tail[0];
    st.rays_shadow += tail[1] + tail[7];
    st.rays_shadow_skipped += tail[7];
    st.nodes_visited = visit[0] + tail[2];
    st.tris_tested = visit[1] + tail[3];
    st.nodes_visited_shadow = visit[2] + tail[4];
    st.tris_tested_shadow = visit[3] + tail[5];
    st.ms_tail = stage_ms[5];
    st.lean_kernels = lean ? 1u : 0u;
    st.nodes_visited_shadow_tail = tail[4];
    st.tris_tested_shadow_tail = tail[5];
    st.ms_generate = stage_ms[0];
    st.ms_extend = stage_ms[1];
    st.ms_shade = stage_ms[2];
    st.ms_shadow = stage_ms[3];
    st.ms_compact = stage_ms[4];
    if (stats) *stats = st;
    return RT_OK;
}

extern "C" int rt_render(rt_ctx* ctx, uint32_t width, uint32_t height, uint32_t spp_first, uint32_t spp_count,
                         uint32_t depth, uint64_t seed, uint32_t flags, float* rgb_sum, rt_stats* stats) try {
    return rt_render_impl(ctx, width, height, nullptr, spp_first, spp_count, depth, seed, flags, rgb_sum, stats);
} catch (...) {
    return rth::on_exception("rt_render");
}

#if RT_TRACE_STATS
extern "C" int rt_debug_occupancy(int* out4) {  // resident workgroups per CU: extend, shadow, extend_lean, shade
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&out4[0], k_extend<rtd::COUNT_NONE>, RT_BLOCK, 0) != hipSuccess) return -2;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&out4[1], k_shadow<rtd::COUNT_NONE>, RT_BLOCK, 0) != hipSuccess) return -2;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&out4[2], k_extend_lean<0>, RT_BLOCK, 0) != hipSuccess) return -2;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&out4[3], k_shade<false>, SHADE_BLOCK, 0) != hipSuccess) return -2;
    return 0;
}
extern "C" int rt_debug_trace_stats(unsigned long long* out32, int reset) {
    if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(rtd::g_trace_stats), 40 * sizeof(unsigned long long)) != hipSuccess) return -2;
    if (reset) {
        unsigned long long z[40] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(rtd::g_trace_stats), z, sizeof z) != hipSuccess) return -2;
    }
    return 0;
}
#endif
