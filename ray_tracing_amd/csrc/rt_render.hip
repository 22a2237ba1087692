// rt_render: Engine::render / Engine::ray_colour (path_integrator.rs:138-318) as a wavefront
// pipeline on gfx950:
//
//   generate        one lane per (pixel, sample): jittered Camera::ray, initial path state
//   extend          one lane per live path: Engine::hit (closest hit through the BVH)
//   shade           one lane per live path: one iteration of the `while depth > 0` loop --
//                   light hit + MIS, medium majorant step, BSDF sample, NEE set-up, throughput,
//                   medium switch, Russian roulette; leaves alive / shadow flags
//   compact         wave64 ballot + prefix sums: stable stream compaction of the live-path queue
//                   and of the shadow-ray queue (block totals come out of shade)
//   shadow-connect  one lane per shadow ray: Engine::hit_bool, adds the pending NEE term
//   resolve         per pixel, adds the samples of the batch IN SAMPLE ORDER into the frame sum
//
// A path keeps one slot (= sample_in_batch * pixels + pixel) for its whole life; queues hold slot
// ids.  Radiance additions happen in the reference's order (NEE of iteration k before emission of
// iteration k+1), and every random draw is addressed by (pixel, sample, bounce, draw), so the sum
// written for a pixel is bit-identical to the sequential CPU loop.
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

struct PathArrays {      // SoA over slots
    // 16-byte records; the w lanes carry the scalar path state, so a bounce moves 3 records per path
    // in and 3 out (the bounce number is the wavefront iteration: a kernel argument, not path state)
    float4* ray_o;       // origin.xyz, bits: current medium (int) << 1 | previous bounce was specular
    float4* ray_d;       // direction.xyz, prev_pdf
    float4* beta;        // throughput.xyz, bits: depth (u32; wraps like the reference's, path_integrator.rs:193)
    float4* rad;         // radiance.xyz, -   ALWAYS indexed by the path's original slot (k_resolve sums it in sample order)
    uint32_t* oid;       // original slot (= sample_in_batch * pixels + pixel: names the pixel, the sample and the rad
                         // record) of the path in each slot of a DENSE region; null in the first region, where slot = id
    float4* hit_a;       // t, normal.xyz (normal of analytic surfaces only)
    uint2* hit_b;        // obj | front_face<<31 (RT_MISS = no hit), triangle slot of a mesh hit (normal: shade_path)
    float4* sh_d;        // shadow direction.xyz, - (the shadow ray starts at ray_o: the vertex just shaded)
    float4* sh_c;        // pending NEE term.xyz, apply flag
    uint32_t* queue[2];  // live-path queues (ping-pong)
    uint32_t* shadow_q;
    uint32_t* qflags;    // per queue position: QF_* of the path there
    uint32_t* block_cnt; // [2][blocks]: alive, shadow totals per shade block
    uint32_t* block_off; // [2][blocks]: exclusive scan inside the block's scan group
    uint32_t* group_tot; // [2][groups]: totals of each group of SCAN_GROUP blocks
    uint32_t* group_off; // [2][groups]: exclusive scan of the group totals
    uint32_t* counts;    // [2][4], ping-pong over the bounces: [0] live paths of the bounce, [1] shadow rays of the bounce
                         // before it.  Every kernel reads its n from here: the host enqueues a bounce before it knows how
                         // many paths survived the previous one (rt_render below).
    uint32_t* counts_host;  // the same two numbers per bounce, written by k_compact_scan_top into pinned host memory
    uint32_t max_blocks, max_groups;  // strides of block_cnt / block_off and group_tot / group_off
    uint32_t* work;      // extend work counters [0, WORK_U32), shadow work counters [WORK_U32, 2*WORK_U32); zeroed by
                         // k_compact_scan_top of every bounce
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
    uint32_t identity_queue;  // queue position i holds slot i (first bounce; every bounce after a moving compaction): no look-up
    uint32_t implicit_first;  // the first bounce's origin, throughput, depth and medium are the same for every
                              // path (camera origin, 1, rp.depth, initial medium): k_generate does not store
                              // them and the first extend / shade do not load them (64 B per path less traffic)
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
__global__ __launch_bounds__(256) void k_generate(DScene sc, PathArrays pa, RenderParams rp, uint32_t n_paths, rt_rays ur) {
    uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (slot >= n_paths) return;
    if (slot == 0) {  // bounce 0: every path is alive, no shadow rays pending
        pa.counts[0] = n_paths;
        pa.counts[1] = 0u;
    }
    uint32_t pixel, s_local;
    slot_pixel(rp, slot, &pixel, &s_local);
    V3 origin, dir;
    if (ur.ox) {  // uniform: rt_ray_colour -- "pixel" k is the caller's ray k, taken as it is (no jitter draws)
        origin = v3(ur.ox[pixel], ur.oy[pixel], ur.oz[pixel]);
        dir = v3(ur.dx[pixel], ur.dy[pixel], ur.dz[pixel]);
    } else {
        uint32_t i = pixel % rp.width, j = pixel / rp.width;
        Rng rng{rp.k0, rp.k1, pixel, rp.sample_first + s_local, 0u, 0u};
        // path_integrator.rs:303-305
        float u = ((float)i + rng.f32()) / (float)(rp.width - 1);
        float v = ((float)j + rng.f32()) / (float)(rp.height - 1);
        origin = ld3(sc.camera.origin);
        dir = ld3(sc.camera.lower_left_corner) + u * ld3(sc.camera.horizontal) + v * ld3(sc.camera.vertical) - origin;
        dir = normalise(dir);  // camera.rs:60-63
    }
    // :139-148: beta = 1, radiance = 0, prev_pdf = 1, prev_specular = true, current medium = initial
    if (!rp.implicit_first) {
        pa.ray_o[slot] = make_float4(origin.x, origin.y, origin.z, pack_medium(sc.initial_medium, true));
        pa.beta[slot] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(rp.depth));
    }
    pa.ray_d[slot] = make_float4(dir.x, dir.y, dir.z, 1.0f);
    pa.rad[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    // no queue entry: the first bounce's queue is the identity and nobody looks it up (RenderParams::identity_queue)
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

struct ExtendIO {  // rays come from the live-path queue, hits go to the path's slot
    PathArrays pa;
    const uint32_t* queue;
    unsigned long long n_int, n_tri;
    bool first;       // first bounce with implicit state (RenderParams::implicit_first): every ray starts at the camera
    bool identity;    // queue position = slot (first bounce, and after a moving compaction)
    float cam[3];     // ... whose origin travels by value (a pointer into the kernel arguments turns the load of the
                      // origin into a flat load of a selected address)
    RT_DEV uint32_t load(uint32_t i, V3* o, V3* d) const {  // returns the path's slot: the ray's tag from here on
        const uint32_t slot = identity ? i : queue[i];
        float4 d4 = pa.ray_d[slot];
        if (first) {  // uniform
            *o = v3(cam[0], cam[1], cam[2]);
        } else {
            float4 o4 = pa.ray_o[slot];
            *o = v3(o4.x, o4.y, o4.z);
        }
        *d = v3(d4.x, d4.y, d4.z);
        return slot;
    }
    // a closer hit than any before it on this ray (Engine::hit's `closest = Some(..)`, :92-94)
    RT_DEV void hit(uint32_t slot, float t, V3 n, uint32_t meta, uint32_t prim) const {
        pa.hit_a[slot] = make_float4(t, n.x, n.y, n.z);
        pa.hit_b[slot] = make_uint2(meta, prim);
    }
    // a closer hit on a mesh: the normal is left to shade_path, which rebuilds it from triangle `tri`
    static constexpr bool DEFER_NORMAL = true;
    RT_DEV void hit_mesh(uint32_t slot, float t, uint32_t meta, uint32_t tri) const {
        ((float*)&pa.hit_a[slot])[0] = t;
        pa.hit_b[slot] = make_uint2(meta, tri);
    }
    RT_DEV void emit_closest(uint32_t slot, bool got, const Visits& vis) {
        if (!got) pa.hit_b[slot] = make_uint2(RT_MISS, RT_MISS);
        n_int += vis.n_int;
        n_tri += vis.n_tri;
    }
    RT_DEV void emit_any(uint32_t, bool, const Visits&) {}
};

template <int COUNT>
__global__ RT_TRACE_ATTR void k_extend(DScene sc, PathArrays pa, const uint32_t* queue, const uint32_t* counts, uint32_t first,
                                                     uint32_t* work_counter, TuneBase base, uint32_t* overflow,
                                                     uint32_t overflow_stride) {
    __shared__ uint32_t lds_stack[RT_STACK_LDS * RT_BLOCK];
    const uint32_t gtid = blockIdx.x * RT_BLOCK + threadIdx.x;
    const uint32_t n = counts[0];
    const TraceTune tune = make_tune(base, n, gridDim.x);
    ExtendIO io{pa, queue, 0ull, 0ull, (first & 1u) != 0u, first != 0u, {sc.camera.origin[0], sc.camera.origin[1], sc.camera.origin[2]}};
    trace_loop<TRACE_CLOSEST, COUNT, RT_STACK_LDS, RT_BLOCK>(sc, io, n, work_counter, tune, T_MIN, __builtin_inff(),
                                                             lds_stack + threadIdx.x, overflow + gtid, overflow_stride);
    if (COUNT) add_visits(&pa.visit[0], io.n_int, io.n_tri);
}

// The same kernel for scenes whose objects are meshes and planes (rtd::SceneFeatures): fewer registers, more waves.
__global__ RT_TRACE_ATTR_LEAN void k_extend_lean(DScene sc, PathArrays pa, const uint32_t* queue, const uint32_t* counts, uint32_t first,
                                                 uint32_t* work_counter, TuneBase base, uint32_t* overflow,
                                                 uint32_t overflow_stride) {
    __shared__ uint32_t lds_stack[RT_STACK_LEAN * RT_BLOCK];
    const uint32_t gtid = blockIdx.x * RT_BLOCK + threadIdx.x;
    const uint32_t n = counts[0];
    const TraceTune tune = make_tune(base, n, gridDim.x);
    ExtendIO io{pa, queue, 0ull, 0ull, (first & 1u) != 0u, first != 0u, {sc.camera.origin[0], sc.camera.origin[1], sc.camera.origin[2]}};
    trace_loop<TRACE_CLOSEST, false, RT_STACK_LEAN, RT_BLOCK, RT_LEAN_FEAT, RT_LEAN_OVERFLOW>(sc, io, n, work_counter, tune, T_MIN, __builtin_inff(),
                                                                  lds_stack + threadIdx.x, overflow + gtid, overflow_stride);
}

// ---------------------------------------------------------------- shadow-connect
// The NEE term of an unoccluded shadow ray joins the path's radiance (path_integrator.rs:234-236).  Term and
// radiance are requested together and the sum is stored either way (unchanged bits when p_light was 0): the
// wave waits for one round trip here, not for two in a row.
RT_DEV void add_pending_nee(const PathArrays& pa, uint32_t slot) {
    const float4 c = pa.sh_c[slot];
    const uint32_t id = pa.oid ? pa.oid[slot] : slot;  // uniform: dense region or the first one
    float4 r = pa.rad[id];
    const bool add = c.w != 0.0f;  // p_light > 0
    r.x = add ? r.x + c.x : r.x;
    r.y = add ? r.y + c.y : r.y;
    r.z = add ? r.z + c.z : r.z;
    pa.rad[id] = r;
}

struct ShadowIO {  // rays come from the shadow queue; an unoccluded ray adds its pending NEE term
    PathArrays pa;
    unsigned long long n_int, n_tri;
    RT_DEV uint32_t load(uint32_t i, V3* o, V3* d) const {
        const uint32_t slot = pa.shadow_q[i];
        float4 o4 = pa.ray_o[slot], d4 = pa.sh_d[slot];
        *o = v3(o4.x, o4.y, o4.z);
        *d = v3(d4.x, d4.y, d4.z);
        return slot;
    }
    static constexpr bool DEFER_NORMAL = false;
    RT_DEV void hit(uint32_t, float, V3, uint32_t, uint32_t) const {}
    RT_DEV void emit_closest(uint32_t, bool, const Visits&) {}
    RT_DEV void emit_any(uint32_t slot, bool occluded, const Visits& vis) {
        if (!occluded) {
            add_pending_nee(pa, slot);
        }
        n_int += vis.n_int;
        n_tri += vis.n_tri;
    }
};

// path_integrator.rs:219: the shadow ray is tested over T_MIN..T_MAX, not up to the light.
template <int COUNT>
__global__ RT_TRACE_ATTR_ANY void k_shadow(DScene sc, PathArrays pa, const uint32_t* counts, uint32_t* work_counter,
                                                     TuneBase base, uint32_t* overflow, uint32_t overflow_stride) {
    __shared__ uint32_t lds_stack[RT_STACK_ANY * RT_BLOCK];
    const uint32_t gtid = blockIdx.x * RT_BLOCK + threadIdx.x;
    const uint32_t n = counts[1];
    const TraceTune tune = make_tune(base, n, gridDim.x);
    ShadowIO io{pa, 0ull, 0ull};
    trace_loop<TRACE_ANY, COUNT, RT_STACK_ANY, RT_BLOCK>(sc, io, n, work_counter, tune, T_MIN, __builtin_inff(),
                                                         lds_stack + threadIdx.x, overflow + gtid, overflow_stride);
    if (COUNT) add_visits(&pa.visit[2], io.n_int, io.n_tri);
}

__global__ RT_TRACE_ATTR_LEAN_ANY void k_shadow_lean(DScene sc, PathArrays pa, const uint32_t* counts, uint32_t* work_counter,
                                                     TuneBase base, uint32_t* overflow, uint32_t overflow_stride) {
    __shared__ uint32_t lds_stack[RT_STACK_LEAN_ANY * RT_BLOCK];
    const uint32_t gtid = blockIdx.x * RT_BLOCK + threadIdx.x;
    const uint32_t n = counts[1];
    const TraceTune tune = make_tune(base, n, gridDim.x);
    ShadowIO io{pa, 0ull, 0ull};
    trace_loop<TRACE_ANY, false, RT_STACK_LEAN_ANY, RT_BLOCK, RT_LEAN_FEAT, RT_LEAN_OVERFLOW>(sc, io, n, work_counter, tune, T_MIN, __builtin_inff(),
                                                              lds_stack + threadIdx.x, overflow + gtid, overflow_stride);
}

// ---------------------------------------------------------------- shade
RT_DEV float mis_weight(float pa, float pb) {  // path_integrator.rs:30-39, power 2
    float a = pa * pa;
    float b = pb * pb;
    if (a + b == 0.0f) return 0.0f;
    return a / (a + b);
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

// One iteration of the loop body of ray_colour for the path in `slot`.  Returns QF_* flags.
RT_DEV uint32_t shade_path(const DScene& sc, const PathArrays& pa, const RenderParams& rp, uint32_t slot,
                           uint32_t bounce) {
    // Everything the slot names is requested at once: nothing below depends on anything but `slot`, and a
    // wave that waits for each record in turn spends its time in memory round trips (4 waves per SIMD).
    float4 d4 = pa.ray_d[slot], o4, b4;
    const uint2 hb = pa.hit_b[slot];
    float4 ha = pa.hit_a[slot];
    if (rp.implicit_first) {  // uniform: the state k_generate left implicit (:139-148)
        o4 = make_float4(sc.camera.origin[0], sc.camera.origin[1], sc.camera.origin[2], pack_medium(sc.initial_medium, true));
        b4 = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(rp.depth));
    } else {
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
            pa.ray_o[slot] = make_float4(mi.origin.x, mi.origin.y, mi.origin.z, o4.w);
            pa.ray_d[slot] = make_float4(mi.dir.x, mi.dir.y, mi.dir.z, prev_pdf);
            pa.beta[slot] = make_float4(beta.x, beta.y, beta.z, __uint_as_float(depth));
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
            pa.sh_d[slot] = make_float4(ls.direction.x, ls.direction.y, ls.direction.z, 0.0f);
            pa.sh_c[slot] = make_float4(add.x, add.y, add.z, p_light > 0.0f ? 1.0f : 0.0f);
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
    // the vertex is the origin of the next ray AND of the pending shadow ray: written if either exists
    if (survives || (flags & QF_SHADOW))
        pa.ray_o[slot] = make_float4(interaction_point.x, interaction_point.y, interaction_point.z,
                                     pack_medium(medium, bs.is_specular));
    if (!survives) return flags;
    beta = div_assign(beta, 1.0f - q);
    pa.ray_d[slot] = make_float4(nd.x, nd.y, nd.z, prev_pdf);
    pa.beta[slot] = make_float4(beta.x, beta.y, beta.z, __uint_as_float(depth));
    if (depth > 0) flags |= QF_ALIVE;
    return flags;
}

__global__ __launch_bounds__(SHADE_BLOCK) void k_shade(DScene sc, PathArrays pa, RenderParams rp,
                                                       const uint32_t* queue, const uint32_t* counts) {
    __shared__ uint32_t wave_cnt[3][SHADE_BLOCK / 64];
    const uint32_t n = counts[0];
    if (blockIdx.x * SHADE_BLOCK >= n) return;  // the grid is sized from an upper bound of n (uniform per block)
    uint32_t i = blockIdx.x * SHADE_BLOCK + threadIdx.x;
    uint32_t flags = 0;
    if (i < n) {
        // the first bounce's queue is the identity (k_generate): no look-up
        flags = shade_path(sc, pa, rp, rp.identity_queue ? i : queue[i], rp.bounce);
        pa.qflags[i] = flags;
    }
    // block totals for the compaction scan (wave ballots)
    unsigned long long alive = __ballot((flags & QF_ALIVE) != 0);
    unsigned long long shadow = __ballot((flags & QF_SHADOW) != 0);
    unsigned long long skipped = __ballot((flags & QF_SKIPPED) != 0);
    uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        wave_cnt[0][wave] = (uint32_t)__popcll(alive);
        wave_cnt[1][wave] = (uint32_t)__popcll(shadow);
        wave_cnt[2][wave] = (uint32_t)__popcll(skipped);
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        uint32_t s = 0;
        for (int w = 0; w < SHADE_BLOCK / 64; ++w) s += wave_cnt[threadIdx.x][w];
        pa.block_cnt[threadIdx.x * pa.max_blocks + blockIdx.x] = s;
    }
}

// ---------------------------------------------------------------- tail
// A frame of a few thousand paths spends its time in launches and host round trips (5 launches and a
// sync per bounce).  When no more than RT_TAIL_PATHS (default 32 768) paths are alive, k_tail runs each
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
    PathArrays pa;
    const uint32_t* queue;
    const DScene* sc;
    const RenderParams* rp;
    uint32_t n_closest, n_shadow, n_skipped, iters, max_iters;  // per lane
    unsigned long long n_int, n_tri;                   // closest-hit visits (COUNT)
    unsigned long long n_int_sh, n_tri_sh;             // shadow visits (COUNT)
    bool alive_after;                                  // the path goes on after its pending shadow ray
    bool identity;                                     // queue position = slot
    RT_DEV uint32_t load(uint32_t i, V3* o, V3* d) {
        const uint32_t slot = identity ? i : queue[i];
        float4 o4 = pa.ray_o[slot], d4 = pa.ray_d[slot];
        *o = v3(o4.x, o4.y, o4.z);
        *d = v3(d4.x, d4.y, d4.z);
        n_closest += 1;
        iters = 1;
        if (iters > max_iters) max_iters = iters;
        return slot;
    }
    static constexpr bool DEFER_NORMAL = true;
    RT_DEV void hit_mesh(uint32_t slot, float t, uint32_t meta, uint32_t tri) const {
        ((float*)&pa.hit_a[slot])[0] = t;
        pa.hit_b[slot] = make_uint2(meta, tri);
    }
    static constexpr bool CONTINUES = true;  // next() carries the lane on to the path's shadow ray / next bounce
    RT_DEV bool first_is_any(uint32_t) const { return false; }
    RT_DEV void hit(uint32_t slot, float t, V3 n, uint32_t meta, uint32_t prim) const {
        pa.hit_a[slot] = make_float4(t, n.x, n.y, n.z);
        pa.hit_b[slot] = make_uint2(meta, prim);
    }
    RT_DEV void emit_closest(uint32_t, bool, const Visits&) {}
    RT_DEV void emit_any(uint32_t, bool, const Visits&) {}
    RT_DEV bool next_bounce(uint32_t slot, V3* o, V3* d, bool* any) {
        float4 o4 = pa.ray_o[slot], d4 = pa.ray_d[slot];
        *o = v3(o4.x, o4.y, o4.z);
        *d = v3(d4.x, d4.y, d4.z);
        *any = false;
        n_closest += 1;
        iters += 1;
        if (iters > max_iters) max_iters = iters;
        return true;
    }
    RT_DEV bool next(uint32_t slot, bool was_any, bool got, bool occluded, const Visits& vis, V3* o, V3* d, bool* any) {
        if (!was_any) {
            n_int += vis.n_int;
            n_tri += vis.n_tri;
            if (!got) pa.hit_b[slot] = make_uint2(RT_MISS, RT_MISS);
            const uint32_t flags = shade_path(*sc, pa, *rp, slot, rp->bounce + iters - 1u);
            alive_after = (flags & QF_ALIVE) != 0;
            if (flags & QF_SKIPPED) n_skipped += 1;
            if (flags & QF_SHADOW) {
                float4 o4 = pa.ray_o[slot], d4 = pa.sh_d[slot];
                *o = v3(o4.x, o4.y, o4.z);
                *d = v3(d4.x, d4.y, d4.z);
                *any = true;
                n_shadow += 1;
                return true;
            }
        } else {
            n_int_sh += vis.n_int;
            n_tri_sh += vis.n_tri;
            if (!occluded) {
                add_pending_nee(pa, slot);
            }
        }
        return alive_after ? next_bounce(slot, o, d, any) : false;
    }
};

RT_DEV unsigned long long wave_sum_u64(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

// pa.tail[0..7]: closest rays, shadow rays, n_int, n_tri (closest), n_int, n_tri (shadow), max iterations, shadow rays not needed
template <int COUNT>
__global__ __launch_bounds__(RT_BLOCK) void k_tail(DScene sc, PathArrays pa, RenderParams rp, const uint32_t* queue,
                                                   const uint32_t* counts, uint32_t* work_counter, TuneBase base,
                                                   uint32_t* overflow, uint32_t overflow_stride) {
    __shared__ uint32_t lds_stack[RT_STACK_LDS * RT_BLOCK];
    const uint32_t gtid = blockIdx.x * RT_BLOCK + threadIdx.x;
    const uint32_t n = counts[0];
    const TraceTune tune = make_tune(base, n, gridDim.x);
    TailIO io{pa, queue, &sc, &rp, 0u, 0u, 0u, 0u, 0u, 0ull, 0ull, 0ull, 0ull, false, rp.identity_queue != 0u};
    trace_loop<TRACE_PATH, COUNT, RT_STACK_LDS, RT_BLOCK>(sc, io, n, work_counter, tune, T_MIN, __builtin_inff(),
                                                          lds_stack + threadIdx.x, overflow + gtid, overflow_stride);
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
// bounce k+1.  Neither depends on the other (one adds to the radiance, the other writes hit records; both
// only read the vertex in ray_o), so up to RT_MIX_PATHS rays (default 128 Mi; per ray the mixed loop costs
// a little more than the dedicated kernels, per launch it saves one drain: 1080p frame 68.4 / 66.1 / 68.4 ms
// at 4 Mi / 64-128 Mi / 256 Mi) they share ONE launch: items [0, n_shadow) are the shadow
// queue, items [n_shadow, n_shadow + n_next) the live-path queue.  k_shade of bounce k+1 still runs after
// both, so the order of additions into a path's radiance is unchanged.
struct MixedIO {
    PathArrays ps;          // region the shadow rays of this bounce live in
    PathArrays pa;          // region of the next bounce's rays (the same one unless the compaction moved the paths)
    const uint32_t* queue;  // live paths of the NEXT bounce
    uint32_t n_shadow;
    bool identity;          // queue position = slot in `pa`
    unsigned long long n_int, n_tri, n_int_sh, n_tri_sh;
    static constexpr bool CONTINUES = false;  // a finished ray is finished: the lane takes a new one
    RT_DEV bool first_is_any(uint32_t i) const { return i < n_shadow; }
    RT_DEV uint32_t load(uint32_t i, V3* o, V3* d) const {
        const bool sh = i < n_shadow;
        uint32_t slot;
        float4 o4, d4;
        if (sh) {
            slot = ps.shadow_q[i];
            o4 = ps.ray_o[slot];
            d4 = ps.sh_d[slot];
        } else {
            slot = identity ? i - n_shadow : queue[i - n_shadow];
            o4 = pa.ray_o[slot];
            d4 = pa.ray_d[slot];
        }
        *o = v3(o4.x, o4.y, o4.z);
        *d = v3(d4.x, d4.y, d4.z);
        return slot;
    }
    static constexpr bool DEFER_NORMAL = true;
    RT_DEV void hit_mesh(uint32_t slot, float t, uint32_t meta, uint32_t tri) const {
        ((float*)&pa.hit_a[slot])[0] = t;
        pa.hit_b[slot] = make_uint2(meta, tri);
    }
    RT_DEV void hit(uint32_t slot, float t, V3 n, uint32_t meta, uint32_t prim) const {
        pa.hit_a[slot] = make_float4(t, n.x, n.y, n.z);
        pa.hit_b[slot] = make_uint2(meta, prim);
    }
    RT_DEV void emit_closest(uint32_t, bool, const Visits&) {}
    RT_DEV void emit_any(uint32_t, bool, const Visits&) {}
    RT_DEV bool next(uint32_t slot, bool was_any, bool got, bool occluded, const Visits& vis, V3*, V3*, bool*) {
        if (was_any) {
            n_int_sh += vis.n_int;
            n_tri_sh += vis.n_tri;
            if (!occluded) {
                add_pending_nee(ps, slot);
            }
        } else {
            n_int += vis.n_int;
            n_tri += vis.n_tri;
            if (!got) pa.hit_b[slot] = make_uint2(RT_MISS, RT_MISS);
        }
        return false;
    }
};

template <int COUNT>
__global__ RT_TRACE_ATTR void k_mixed(DScene sc, PathArrays ps, PathArrays pa, const uint32_t* queue, const uint32_t* counts,
                                      uint32_t identity, uint32_t* work_counter, TuneBase base, uint32_t* overflow,
                                      uint32_t overflow_stride) {
    __shared__ uint32_t lds_stack[RT_STACK_LDS * RT_BLOCK];
    const uint32_t gtid = blockIdx.x * RT_BLOCK + threadIdx.x;
    const uint32_t n_shadow = counts[1], n = n_shadow + counts[0];  // shadow rays of this bounce, then the rays of the next
    const TraceTune tune = make_tune(base, n, gridDim.x);
    MixedIO io{ps, pa, queue, n_shadow, identity != 0u, 0ull, 0ull, 0ull, 0ull};
    trace_loop<TRACE_PATH, COUNT, RT_STACK_LDS, RT_BLOCK>(sc, io, n, work_counter, tune, T_MIN, __builtin_inff(),
                                                          lds_stack + threadIdx.x, overflow + gtid, overflow_stride);
    if (COUNT) {  // accounted with the tail kernel's share: rays traced outside k_extend / k_shadow
        add_visits(&pa.tail[2], io.n_int, io.n_tri);
        add_visits(&pa.tail[4], io.n_int_sh, io.n_tri_sh);
    }
}

__global__ RT_TRACE_ATTR_LEAN void k_mixed_lean(DScene sc, PathArrays ps, PathArrays pa, const uint32_t* queue, const uint32_t* counts,
                                                uint32_t identity, uint32_t* work_counter, TuneBase base, uint32_t* overflow,
                                                uint32_t overflow_stride) {
    __shared__ uint32_t lds_stack[RT_STACK_LEAN * RT_BLOCK];
    const uint32_t gtid = blockIdx.x * RT_BLOCK + threadIdx.x;
    const uint32_t n_shadow = counts[1], n = n_shadow + counts[0];
    const TraceTune tune = make_tune(base, n, gridDim.x);
    MixedIO io{ps, pa, queue, n_shadow, identity != 0u, 0ull, 0ull, 0ull, 0ull};
    trace_loop<TRACE_PATH, false, RT_STACK_LEAN, RT_BLOCK, RT_LEAN_FEAT, RT_LEAN_OVERFLOW>(sc, io, n, work_counter, tune, T_MIN,
                                                                         __builtin_inff(), lds_stack + threadIdx.x,
                                                                         overflow + gtid, overflow_stride);
}

// ---------------------------------------------------------------- compact
// Exclusive scan of the per-block totals in two levels, both queues (alive, shadow) at once:
//   k_compact_scan_groups: one workgroup per 1024 consecutive shade blocks -- coalesced read, wave
//       shuffle scan, LDS scan of the 16 wave totals; writes offsets local to the group + the group total
//   k_compact_scan_top:    one workgroup scans the group totals (<= 8192 groups: 2^31 slots / 256 / 1024)
// The scatter adds group offset + local offset.  (The first version scanned everything in ONE
// workgroup with a serial strip per thread: 2.6 ms for the 518 400 blocks of a 64-spp 1080p batch.)
constexpr uint32_t SCAN_GROUP = 1024;

RT_DEV uint32_t block_excl_scan_1024(uint32_t v, uint32_t* wave_sum /*[16]*/, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = v;
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t y = __shfl_up(x, off);
        if (lane >= (uint32_t)off) x += y;
    }
    if (lane == 63) wave_sum[wave] = x;
    __syncthreads();
    if (wave == 0) {
        uint32_t w = lane < 16 ? wave_sum[lane] : 0u;
        uint32_t t = w;
        for (int off = 1; off < 16; off <<= 1) {
            uint32_t y = __shfl_up(t, off);
            if (lane >= (uint32_t)off) t += y;
        }
        if (lane < 16) wave_sum[lane] = t - w;  // exclusive over waves
        if (lane == 15) *total = t;
    }
    __syncthreads();
    return x - v + wave_sum[wave];
}

__global__ __launch_bounds__(1024) void k_compact_scan_groups(PathArrays pa, const uint32_t* counts) {
    __shared__ uint32_t wave_sum[3][16];
    __shared__ uint32_t total[3];
    const uint32_t n = counts[0];
    const uint32_t n_blocks = (n + SHADE_BLOCK - 1) / SHADE_BLOCK;
    if (blockIdx.x * SCAN_GROUP >= n_blocks) return;  // grid sized from an upper bound of n
    const uint32_t b = blockIdx.x * SCAN_GROUP + threadIdx.x;
    for (int q = 0; q < 3; ++q) {  // alive, shadow, skipped (a total only: nothing is scattered for it)
        const uint32_t v = b < n_blocks ? pa.block_cnt[q * pa.max_blocks + b] : 0u;
        const uint32_t e = block_excl_scan_1024(v, wave_sum[q], &total[q]);
        if (q < 2 && b < n_blocks) pa.block_off[q * pa.max_blocks + b] = e;
    }
    if (threadIdx.x < 3) pa.group_tot[threadIdx.x * pa.max_groups + blockIdx.x] = total[threadIdx.x];
}

// One workgroup: scans the group totals, publishes the counts of the next bounce -- to device memory for the kernels
// that follow in the stream, and to pinned host memory for the host, which reads them a bounce later -- and zeroes
// the work counters of the trace launches that follow.
__global__ __launch_bounds__(1024) void k_compact_scan_top(PathArrays pa, const uint32_t* counts, uint32_t* counts_out,
                                                           uint32_t* counts_host) {
    __shared__ uint32_t wave_sum[3][16];
    __shared__ uint32_t total[3];
    const uint32_t n = counts[0];
    const uint32_t n_blocks = (n + SHADE_BLOCK - 1) / SHADE_BLOCK;
    const uint32_t n_groups = (n_blocks + SCAN_GROUP - 1) / SCAN_GROUP;
    const uint32_t per = (n_groups + 1023u) / 1024u;  // <= 8
    const uint32_t begin = threadIdx.x * per, end = min(begin + per, n_groups);
    for (int q = 0; q < 3; ++q) {
        uint32_t s = 0;
        for (uint32_t g = begin; g < end; ++g) s += pa.group_tot[q * pa.max_groups + g];
        uint32_t run = block_excl_scan_1024(s, wave_sum[q], &total[q]);
        for (uint32_t g = begin; q < 2 && g < end; ++g) {
            const uint32_t c = pa.group_tot[q * pa.max_groups + g];
            pa.group_off[q * pa.max_groups + g] = run;
            run += c;
        }
    }
    for (uint32_t i = threadIdx.x; i < 2u * WORK_U32; i += 1024u) pa.work[i] = 0u;
    if (threadIdx.x < 3) {
        counts_out[threadIdx.x] = total[threadIdx.x];  // [0] paths alive after this bounce, [1] its shadow rays, [2] shadow rays not needed
        counts_host[threadIdx.x] = total[threadIdx.x];
        __threadfence_system();
    }
}

// Stable scatter: position = block offset + waves before + rank inside the wave ballot.
// move != 0: a MOVING compaction -- the surviving paths' state (origin, direction, throughput and their w lanes) goes
// to slot `position` of the dense region `dst` together with the path's original slot, and the next bounce's queue is
// the identity.  Late bounces keep a fifth of the slots alive or less: every 16-byte record a kernel touched pulled a
// 64-byte sector for it (k_shade: 106 ps per path at the third bounce against 38 ps at the second, profiles/r02).
__global__ __launch_bounds__(SHADE_BLOCK) void k_compact_scatter(PathArrays pa, const uint32_t* queue, uint32_t* next_queue,
                                                                 const uint32_t* counts, uint32_t identity, PathArrays dst,
                                                                 uint32_t move) {
    __shared__ uint32_t wave_cnt[2][SHADE_BLOCK / 64];
    const uint32_t n = counts[0];
    if (blockIdx.x * SHADE_BLOCK >= n) return;
    uint32_t i = blockIdx.x * SHADE_BLOCK + threadIdx.x;
    uint32_t flags = i < n ? pa.qflags[i] : 0u;
    uint32_t slot = i < n ? (identity ? i : queue[i]) : 0u;  // the first bounce's queue is the identity: no look-up
    unsigned long long alive = __ballot((flags & QF_ALIVE) != 0);
    unsigned long long shadow = __ballot((flags & QF_SHADOW) != 0);
    uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        wave_cnt[0][wave] = (uint32_t)__popcll(alive);
        wave_cnt[1][wave] = (uint32_t)__popcll(shadow);
    }
    __syncthreads();
    unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    if (flags & QF_ALIVE) {
        uint32_t pos = pa.group_off[blockIdx.x / SCAN_GROUP] + pa.block_off[blockIdx.x];
        for (uint32_t w = 0; w < wave; ++w) pos += wave_cnt[0][w];
        pos += (uint32_t)__popcll(alive & below);
        if (move) {  // uniform
            const float4 o4 = pa.ray_o[slot], d4 = pa.ray_d[slot], b4 = pa.beta[slot];
            dst.ray_o[pos] = o4;
            dst.ray_d[pos] = d4;
            dst.beta[pos] = b4;
            dst.oid[pos] = pa.oid ? pa.oid[slot] : slot;
        } else {
            next_queue[pos] = slot;
        }
    }
    if (flags & QF_SHADOW) {
        uint32_t pos = pa.group_off[pa.max_groups + blockIdx.x / SCAN_GROUP] + pa.block_off[pa.max_blocks + blockIdx.x];
        for (uint32_t w = 0; w < wave; ++w) pos += wave_cnt[1][w];
        pa.shadow_q[pos + (uint32_t)__popcll(shadow & below)] = slot;
    }
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
    if (!ctx || !rgb_sum) {
        rth::set_error("rt_render: null argument");
        return RT_ERR_INVALID;
    }
    if (!ctx->has_scene) {
        rth::set_error("rt_render: no scene uploaded (call rt_upload_scene first)");
        return RT_ERR_STATE;
    }
    if (!user_rays && (width < 2 || height < 2)) {  // the reference divides by (w-1), (h-1) (:303-304)
        rth::set_error("rt_render: width and height must be >= 2");
        return RT_ERR_INVALID;
    }
    const rt_rays no_rays = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const rt_rays gen_rays = user_rays ? *user_rays : no_rays;
    const uint64_t npix64 = (uint64_t)width * height;
    if (npix64 > 0x7FFFFFFFull) {
        rth::set_error("rt_render: image too large");
        return RT_ERR_LIMIT;
    }
    const uint32_t npix = (uint32_t)npix64;
    if (ctx->scene.n_lights == 0 && ctx->scene.n_objects > 0) {
        // sample_lights_uniform would index an empty Vec (panic) on the first non-specular hit
        rth::set_error("rt_render: scene has no lights (the reference panics in sample_lights_uniform)");
        return RT_ERR_INVALID;
    }
    RT_HIP(hipSetDevice(ctx->device));
    hipStream_t stream = ctx->stream;

    // ---- batch size: samples in flight at once.  Every wavefront iteration ends in a tail of
    // half-empty waves and a host round trip, so the batch is made as large as memory allows
    // (measured at 1080p/32 spp: 8 Mi paths 145 ms, 64 Mi paths 71 ms): by default up to 40 % of
    // the free HBM, ~212 B per path slot (148 B + two dense regions of a quarter of the slots, 128 B each).
    uint64_t max_paths = ctx->max_paths_in_flight;
    if (max_paths == 0) {
        size_t free_b = 0, total_b = 0;
        RT_HIP(hipMemGetInfo(&free_b, &total_b));
        free_b += ctx->path_state.bytes;  // our own previous allocation is reusable
        max_paths = (uint64_t)((double)free_b * 0.4 / 212.0);
        if (max_paths < ((uint64_t)1 << 20)) max_paths = (uint64_t)1 << 20;
    }
    uint32_t s_batch = (uint32_t)std::max<uint64_t>(1, max_paths / npix);
    if (s_batch > spp_count) s_batch = spp_count ? spp_count : 1;
    while ((uint64_t)s_batch * npix > 0x7FFFFFFFull) --s_batch;
    const uint64_t P = (uint64_t)s_batch * npix;
    const uint32_t max_blocks = (uint32_t)((P + SHADE_BLOCK - 1) / SHADE_BLOCK);

    // ---- carve the path state out of one allocation
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    const uint32_t max_groups = max_blocks / SCAN_GROUP + 1;
    // Regions: 0 = one slot per (sample, pixel) of the batch; 1 and 2 = dense regions of a quarter of the slots, the
    // targets of the moving compaction (ping-pong).  rad, the scan scratch, counts and work counters are shared.
    const uint64_t Rcap = std::max<uint64_t>(P / 4, 1);
    struct RegionOff {
        size_t ray_o, ray_d, beta, hit_a, hit_b, sh_d, sh_c, q0, q1, sq, qf, oid;
    } ro[3];
    for (int r = 0; r < 3; ++r) {
        const uint64_t cap = r == 0 ? P : Rcap;
        ro[r].ray_o = carve(cap * 16), ro[r].ray_d = carve(cap * 16), ro[r].beta = carve(cap * 16), ro[r].hit_a = carve(cap * 16);
        ro[r].hit_b = carve(cap * 8), ro[r].sh_d = carve(cap * 16), ro[r].sh_c = carve(cap * 16), ro[r].q0 = carve(cap * 4);
        ro[r].q1 = carve(cap * 4), ro[r].sq = carve(cap * 4), ro[r].qf = carve(cap * 4), ro[r].oid = r == 0 ? 0 : carve(cap * 4);
    }
    size_t o_rad = carve(P * 16), o_bc = carve((size_t)max_blocks * 12), o_bo = carve((size_t)max_blocks * 8),
           o_gt = carve((size_t)max_groups * 12), o_go = carve((size_t)max_groups * 8), o_cnt = carve(256), o_vis = carve(256),
           o_tail = carve(256), o_work = carve(2 * rtd::WORK_U32 * 4);
    int rc = ctx->path_state.ensure(off);
    if (rc != RT_OK) return rc;
    if ((rc = rt_ctx_ensure_readback(ctx)) != RT_OK) return rc;
    char* base = (char*)ctx->path_state.p;
    PathArrays reg[3];
    for (int r = 0; r < 3; ++r) {
        PathArrays& q = reg[r];
        q.ray_o = (float4*)(base + ro[r].ray_o);
        q.ray_d = (float4*)(base + ro[r].ray_d);
        q.beta = (float4*)(base + ro[r].beta);
        q.rad = (float4*)(base + o_rad);
        q.oid = r == 0 ? nullptr : (uint32_t*)(base + ro[r].oid);
        q.hit_a = (float4*)(base + ro[r].hit_a);
        q.hit_b = (uint2*)(base + ro[r].hit_b);
        q.sh_d = (float4*)(base + ro[r].sh_d);
        q.sh_c = (float4*)(base + ro[r].sh_c);
        q.queue[0] = (uint32_t*)(base + ro[r].q0);
        q.queue[1] = (uint32_t*)(base + ro[r].q1);
        q.shadow_q = (uint32_t*)(base + ro[r].sq);
        q.qflags = (uint32_t*)(base + ro[r].qf);
        q.block_cnt = (uint32_t*)(base + o_bc);
        q.block_off = (uint32_t*)(base + o_bo);
        q.group_tot = (uint32_t*)(base + o_gt);
        q.group_off = (uint32_t*)(base + o_go);
        q.counts = (uint32_t*)(base + o_cnt);
        q.counts_host = ctx->rb_host;
        q.max_blocks = max_blocks;
        q.max_groups = max_groups;
        q.work = (uint32_t*)(base + o_work);
        q.visit = (unsigned long long*)(base + o_vis);
        q.tail = (unsigned long long*)(base + o_tail);
    }
    const PathArrays& pa = reg[0];  // what is not per region: rad, counts, work, visit, tail
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
    // scenes without analytic sphere / triangle objects (meshes, planes, transforms) take the lean kernels
    static const uint64_t mix_paths = [] {  // RT_MIX_PATHS: rays (shadow + next bounce) up to which they share a launch
        const char* e = getenv("RT_MIX_PATHS");
        return e ? (uint64_t)strtoull(e, nullptr, 10) : (uint64_t)134217728;
    }();
    const bool use_mix = use_tail && mix_paths > 0;
    uint64_t mixed_closest = 0, mixed_shadow = 0;
    const bool lean = rt_scene_is_lean(ctx) && getenv("RT_NO_LEAN") == nullptr;
    static const uint32_t tail_paths = [] {  // RT_TAIL_PATHS overrides the switch-over point (experiments)
        const char* e = getenv("RT_TAIL_PATHS");
        return e ? (uint32_t)strtoul(e, nullptr, 10) : 32768u;
    }();
    static const uint32_t async_below = [] {  // RT_ASYNC_BELOW: bounces with at most this many paths are enqueued
        const char* e = getenv("RT_ASYNC_BELOW");  // without waiting for their exact counts (0: always wait)
        return e ? (uint32_t)strtoul(e, nullptr, 10) : (16u << 20);
    }();
    const bool use_move = getenv("RT_NO_MOVE") == nullptr;  // moving compaction (k_compact_scatter)
    const bool timing = (flags & RT_RENDER_TIME_STAGES) != 0;
    // HIP events on the library's stream around every launch of a stage, summed per stage after the frame's last
    // synchronisation (the host runs ahead of the device, so a span cannot be read back where it is recorded).
    // k_extend (stage 1, the kernel the roofline figure is about) and the mixed / tail launches (stage 5) are always
    // timed; the other stages only with RT_RENDER_TIME_STAGES.  The event pairs are cached in the context.
    constexpr int N_STAGES = 6;  // generate, extend, shade, shadow, compact, tail
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
        if (pairs_used == ctx->ev_pool.size() / 2) {
            hipEvent_t e0 = nullptr, e1 = nullptr;
            RT_HIP(hipEventCreate(&e0));
            ctx->ev_pool.push_back(e0);
            hipError_t e = hipEventCreate(&e1);
            ctx->ev_pool.push_back(e1);  // a null second event is destroyed safely by rt_destroy
            RT_HIP(e);
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
    uint32_t* const work_ext = pa.work;                   // closest-hit launches and the tail kernel
    uint32_t* const work_sh = pa.work + rtd::WORK_U32;    // shadow and mixed launches

    // kernel selection by counting mode / scene features, one place each
    auto launch_extend = [&](int grid, const PathArrays& pa, const uint32_t* queue, const uint32_t* cnt, uint32_t first) {
        if (count == rtd::COUNT_REFERENCE)
            hipLaunchKernelGGL(k_extend<rtd::COUNT_REFERENCE>, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, pa, queue, cnt, first, work_ext, tune, overflow, ostride);
        else if (count == rtd::COUNT_FETCHES)
            hipLaunchKernelGGL(k_extend<rtd::COUNT_FETCHES>, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, pa, queue, cnt, first, work_ext, tune, overflow, ostride);
        else if (lean)
            hipLaunchKernelGGL(k_extend_lean, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, pa, queue, cnt, first, work_ext, tune, overflow, ostride);
        else
            hipLaunchKernelGGL(k_extend<rtd::COUNT_NONE>, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, pa, queue, cnt, first, work_ext, tune, overflow, ostride);
    };
    auto launch_shadow = [&](int grid, const PathArrays& pa, const uint32_t* cnt) {
        if (count == rtd::COUNT_REFERENCE)
            hipLaunchKernelGGL(k_shadow<rtd::COUNT_REFERENCE>, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, pa, cnt, work_sh, tune, overflow, ostride);
        else if (count == rtd::COUNT_FETCHES)
            hipLaunchKernelGGL(k_shadow<rtd::COUNT_FETCHES>, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, pa, cnt, work_sh, tune, overflow, ostride);
        else if (lean)
            hipLaunchKernelGGL(k_shadow_lean, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, pa, cnt, work_sh, tune, overflow, ostride);
        else
            hipLaunchKernelGGL(k_shadow<rtd::COUNT_NONE>, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, pa, cnt, work_sh, tune, overflow, ostride);
    };
    auto launch_mixed = [&](int grid, const PathArrays& ps, const PathArrays& pa, const uint32_t* queue_next, uint32_t identity, const uint32_t* cnt) {
        if (count == rtd::COUNT_REFERENCE)
            hipLaunchKernelGGL(k_mixed<rtd::COUNT_REFERENCE>, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, ps, pa, queue_next, cnt, identity, work_sh, tune, overflow, ostride);
        else if (count == rtd::COUNT_FETCHES)
            hipLaunchKernelGGL(k_mixed<rtd::COUNT_FETCHES>, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, ps, pa, queue_next, cnt, identity, work_sh, tune, overflow, ostride);
        else if (lean)
            hipLaunchKernelGGL(k_mixed_lean, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, ps, pa, queue_next, cnt, identity, work_sh, tune, overflow, ostride);
        else
            hipLaunchKernelGGL(k_mixed<rtd::COUNT_NONE>, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, ps, pa, queue_next, cnt, identity, work_sh, tune, overflow, ostride);
    };
    auto launch_tail = [&](int grid, const PathArrays& pa, const uint32_t* queue, const uint32_t* cnt) {
        if (count == rtd::COUNT_REFERENCE)
            hipLaunchKernelGGL(k_tail<rtd::COUNT_REFERENCE>, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, pa, rp, queue, cnt, work_ext, tune, overflow, ostride);
        else if (count == rtd::COUNT_FETCHES)
            hipLaunchKernelGGL(k_tail<rtd::COUNT_FETCHES>, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, pa, rp, queue, cnt, work_ext, tune, overflow, ostride);
        else
            hipLaunchKernelGGL(k_tail<rtd::COUNT_NONE>, dim3(grid), dim3(RT_BLOCK), 0, stream, ctx->scene, pa, rp, queue, cnt, work_ext, tune, overflow, ostride);
    };

    // What the host remembers of a bounce: which launches it enqueued, and the counts k_compact_scan_top published
    // for it (read from pinned memory once the event recorded behind that kernel has completed).
    struct Bounce {
        bool extend = false, mixed = false, shadow = false, got = false;
        uint32_t n_next = 0, n_shadow = 0, n_skipped = 0;  // paths alive after the bounce, its shadow rays traced / not needed
    };
    std::vector<Bounce> hist;
    auto wait_counts = [&](size_t k) -> int {
        if (hist[k].got) return RT_OK;
        RT_HIP(hipEventSynchronize(ctx->rb_ev[k % RT_READBACK_RING]));
        const volatile uint32_t* h = ctx->rb_host + 4 * (k % RT_READBACK_RING);
        hist[k].n_next = h[0];
        hist[k].n_shadow = h[1];
        hist[k].n_skipped = h[2];
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
        rp.identity_queue = 1u;
        rp.bounce = 0u;
        RT_HIP(hipMemsetAsync(pa.work, 0, 2 * rtd::WORK_U32 * 4, stream));
        if ((rc = t_begin(0)) != RT_OK) return rc;
        hipLaunchKernelGGL(k_generate, dim3((n_paths + 255) / 256), dim3(256), 0, stream, ctx->scene, reg[0], rp, n_paths, gen_rays);
        if ((rc = t_end(0)) != RT_OK) return rc;
        st.paths += n_paths;
        st.batches += 1;

        // ---- the wavefront loop.  n_ub is an UPPER BOUND of the paths alive at bounce k: exact while bounces are big
        // (the host waits for the counts of bounce k before it enqueues bounce k + 1: a round trip of tens of
        // microseconds against launches of milliseconds, and exact counts pick the launch shapes), one bounce old once
        // they are small (n_ub <= async_below): then the host enqueues bounce k + 1 while the device still works on
        // bounce k, with grids sized from the count of bounce k -- paths only die, so it is a bound -- and every
        // kernel takes its true n from device memory.  That removes the host round trip and the idle gap behind it
        // from every late bounce (0.1 ms each, profiles/r02/tuning.md).
        hist.clear();
        uint32_t n_ub = depth > 0 ? n_paths : 0;
        int rcur = 0;               // region the live paths are in
        int cur = 0;                // which of the region's two queues holds them ...
        bool ident = true;          // ... unless queue position = slot (first bounce, after a moving compaction)
        size_t k = 0;
        uint32_t tail_iters_total = 0;
        bool extend_done = false;  // the rays of this bounce were traced by the previous bounce's mixed launch
        while (n_ub > 0) {
            if (k >= (size_t)1 << 20) {
                rth::set_error("rt_render: wavefront did not drain after 2^20 iterations");
                return RT_ERR_LIMIT;
            }
            const PathArrays& pr = reg[rcur];
            const uint32_t* cnt_in = pa.counts + 4 * (k & 1);
            uint32_t* cnt_out = pa.counts + 4 * ((k + 1) & 1);
            rp.bounce = (uint32_t)k + 1;  // iteration k of `while depth > 0` draws from RNG bounce k+1 (bounce 0: camera jitter)
            const uint32_t first_implicit = (implicit && k == 0) ? 1u : 0u;
            rp.implicit_first = first_implicit;
            rp.identity_queue = ident ? 1u : 0u;
            if (use_tail && n_ub <= tail_paths) {  // few paths left: one launch takes them to their end
                RT_HIP(hipMemsetAsync(pa.tail + 6, 0, 8, stream));
                if ((rc = t_begin(5)) != RT_OK) return rc;
                launch_tail(rt_trace_grid(n_ub), pr, pr.queue[cur], cnt_in);
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
                launch_extend(rt_trace_grid(n_ub), pr, pr.queue[cur], cnt_in, first_implicit ? 1u : (ident ? 2u : 0u));
                if ((rc = t_end(1)) != RT_OK) return rc;
                hist[k].extend = true;
            }
            extend_done = false;

            const uint32_t n_blocks = (n_ub + SHADE_BLOCK - 1) / SHADE_BLOCK;
            if ((rc = t_begin(2)) != RT_OK) return rc;
            hipLaunchKernelGGL(k_shade, dim3(n_blocks), dim3(SHADE_BLOCK), 0, stream, ctx->scene, pr, rp, pr.queue[cur], cnt_in);
            if ((rc = t_end(2)) != RT_OK) return rc;

            if ((rc = t_begin(4)) != RT_OK) return rc;
            const uint32_t n_groups = (n_blocks + SCAN_GROUP - 1) / SCAN_GROUP;
            hipLaunchKernelGGL(k_compact_scan_groups, dim3(n_groups), dim3(1024), 0, stream, pr, cnt_in);
            hipLaunchKernelGGL(k_compact_scan_top, dim3(1), dim3(1024), 0, stream, pr, cnt_in, cnt_out,
                               ctx->rb_host + 4 * (k % RT_READBACK_RING));
            if ((rc = t_end(4)) != RT_OK) return rc;
            RT_HIP(hipEventRecord(ctx->rb_ev[k % RT_READBACK_RING], stream));

            // bounds of the next bounce's rays and of this bounce's shadow rays
            uint32_t nn_ub, ns_ub;
            if (n_ub > async_below) {  // big bounce: wait for its exact counts
                if ((rc = wait_counts(k)) != RT_OK) return rc;
                nn_ub = hist[k].n_next;
                ns_ub = hist[k].n_shadow;
                if (nn_ub > n_ub || ns_ub > n_ub) {
                    rth::set_error("rt_render: compaction produced impossible counts");
                    return RT_ERR_HIP;
                }
            } else {  // small bounce: the counts of the bounce BEFORE it have arrived meanwhile (or will, without idling the device)
                if (k > 0 && (rc = wait_counts(k - 1)) != RT_OK) return rc;
                nn_ub = ns_ub = k > 0 ? std::min(n_ub, hist[k - 1].n_next) : n_ub;
            }
            // moving compaction: the survivors' state goes to a dense region as soon as one can hold them
            const bool move = use_move && nn_ub > 0 && nn_ub <= Rcap;
            const int rnext = move ? (rcur == 1 ? 2 : 1) : rcur;
            const PathArrays& pn = reg[rnext];
            if ((rc = t_begin(4)) != RT_OK) return rc;
            hipLaunchKernelGGL(k_compact_scatter, dim3(n_blocks), dim3(SHADE_BLOCK), 0, stream, pr, pr.queue[cur],
                               pr.queue[cur ^ 1], cnt_in, ident ? 1u : 0u, pn, move ? 1u : 0u);
            if ((rc = t_end(4)) != RT_OK) return rc;
            const uint32_t* queue_next = move ? pn.queue[0] : pr.queue[cur ^ 1];

            const bool tail_next = use_tail && nn_ub <= tail_paths;  // the tail kernel traces its own first ray
            if (use_mix && ns_ub > 0 && nn_ub > 0 && !tail_next && (uint64_t)ns_ub + nn_ub <= mix_paths) {
                // the shadow rays of this bounce and the rays of the next one in one launch
                if ((rc = t_begin(5)) != RT_OK) return rc;
                launch_mixed(rt_trace_grid((size_t)ns_ub + nn_ub), pr, pn, queue_next, move ? 1u : 0u, cnt_out);
                if ((rc = t_end(5)) != RT_OK) return rc;
                hist[k].mixed = true;
                extend_done = true;
            } else if (ns_ub > 0) {
                if ((rc = t_begin(3)) != RT_OK) return rc;
                launch_shadow(rt_trace_grid(ns_ub), pr, cnt_out);
                if ((rc = t_end(3)) != RT_OK) return rc;
                hist[k].shadow = true;
            }
            n_ub = nn_ub;
            rcur = rnext;
            if (move) {
                cur = 0;
                ident = true;
            } else {
                cur ^= 1;
                ident = false;
            }
            ++k;
        }
        // ---- what was traced, from the counts of every bounce (all arrived by now or after this wait)
        uint32_t iter = 0;
        for (size_t b = 0; b < hist.size(); ++b) {
            if ((rc = wait_counts(b)) != RT_OK) return rc;
            const uint32_t n_b = b > 0 ? hist[b - 1].n_next : n_paths;  // paths alive at bounce b
            if (hist[b].n_next > n_b || hist[b].n_shadow > n_b) {
                rth::set_error("rt_render: compaction produced impossible counts");
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
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&out4[2], k_extend_lean, RT_BLOCK, 0) != hipSuccess) return -2;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&out4[3], k_shade, SHADE_BLOCK, 0) != hipSuccess) return -2;
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
