// Persistent-wave ray traversal: the hot loop behind rt_trace_closest / rt_trace_any / k_extend /
// k_shadow.
//
// Why this shape (PMC profile of the first version, profiles/r01/pmc_trace_v0): with one static
// strip of rays per lane the grid averaged 1.2 resident waves per SIMD (rays over background
// finish at once, waves over the mesh run long) and a wave stalled on its slowest ray.  Here
//   * a launch is a fixed number of waves that PULL rays from a global counter, so the chip stays
//     full until the queue is empty;
//   * each lane is a small state machine (advance-object / internal node / leaf) and an idle lane
//     is refilled with a new ray as soon as >= refill_min lanes of its wave are idle, so one long
//     ray no longer parks 63 lanes;
//   * per ray the arithmetic and the visiting order are still exactly those of
//     Engine::hit -> Object::hit -> traverse_bvh (path_integrator.rs:88-104, object.rs:32-87,
//     bvh.rs:243-300): the schedule changes, the result (and the visit counters) do not.
#pragma once
#include "device_scene.h"


namespace rtd {

#if RT_TRACE_STATS  // diagnostics build only (scripts/build_variant.sh stats -DRT_TRACE_STATS=1)
static __device__ unsigned long long g_trace_stats[32];
#define RT_STAT(i, v) s_stat[i] += (v)
#else
#define RT_STAT(i, v)
#endif

// Launch-time tuning of the wave scheduler (uniform kernel argument).
constexpr uint32_t WORK_MAX_SEG = 64;     // work counters per launch
constexpr uint32_t WORK_STRIDE = 32;      // u32 between counters: one 128-byte line each
constexpr uint32_t WORK_U32 = WORK_MAX_SEG * WORK_STRIDE;
struct TraceTune {
    uint32_t chunk;       // rays a wave takes from a work counter per atomic
    uint32_t n_seg;       // the queue is split into n_seg contiguous segments, one counter each
    uint32_t seg_len;     // rays per segment (multiple of chunk)
    uint32_t refill_min;  // idle lanes that trigger a refill
    uint32_t w_adv, w_node, w_leaf;  // phase weights: each loop iteration runs the ONE phase with the largest
                                     // (lanes waiting for it) x weight
};

constexpr uint32_t RAY_NONE = 0xFFFFFFFFu;
constexpr uint32_t REF_ADVANCE = 0xFFFFFFFFu;  // "current object finished": never a valid leaf ref

// 64-byte child pair / 48-byte triangle fetches with a 32-bit byte offset from a uniform base
// (global_load_dwordx4 v, v_off, s[base]); rt_upload_scene bounds both arrays to < 4 GiB.
RT_DEV float4 ld16(const void* base, uint32_t byte_off) {
    return *reinterpret_cast<const float4*>(static_cast<const char*>(base) + byte_off);
}

#ifndef RT_COOP
#define RT_COOP 0
#endif
// Quad-cooperative record fetch (RT_COOP=1; compiled out by default -- see the end of this comment).
// Measured on MI355X (scripts/microbench/gather64.hip): a wave in
// which every lane reads its own 64-byte record with 4 x global_load_dwordx4 is limited by the
// vector L1's divergent-access rate (one lane-line per clock: 99 G records/s chip-wide however small
// the table), and the traversal ran at 76 G records/s -- against THAT ceiling, not HBM's.  When the 4
// lanes of a quad read the 4 x 16 bytes of ONE record per instruction, each instruction touches 16
// lines instead of 64 and the same gather runs at 230-300 G records/s.  The pieces are handed to
// their owner through a per-lane 80-byte LDS slot (80, not 64: the slots of the 8 lanes served by
// one LDS pass then fall on disjoint banks).
// In the traversal loop the cooperative fetch costs the same whether 10 or 60 lanes need a record,
// while the per-lane fetch is charged per active lane; at the ~45-50 % of lanes that advance per
// iteration here the two came out within 5 % of each other (extend 54.8 ms cooperative vs 52.5 ms
// per-lane, profiles/r01/tuning.md), so the simpler per-lane path is the default.
typedef float f4v __attribute__((ext_vector_type(4)));  // native vector: assignable through an address-space pointer
typedef __attribute__((address_space(3))) f4v lds_f4;
RT_DEV f4v to_v(float4 a) { return f4v{a.x, a.y, a.z, a.w}; }
RT_DEV float4 to_f4(f4v a) { return make_float4(a.x, a.y, a.z, a.w); }
constexpr uint32_t STAGE_F4 = 5;  // float4 per lane slot
template <int M>
RT_DEV uint32_t quad_bcast(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, M * 0x55, 0xf, 0xf, true);
}
// every lane of the wave must call this (uniform control flow); off = this lane's record offset
RT_DEV void coop_fetch64(const void* base, uint32_t off, lds_f4* stage_wave, uint32_t lane, float4* r0, float4* r1,
                         float4* r2, float4* r3) {
    const uint32_t j16 = (lane & 3u) << 4, q = lane & ~3u, j = lane & 3u;
    const float4 p0 = ld16(base, quad_bcast<0>(off) + j16), p1 = ld16(base, quad_bcast<1>(off) + j16),
                 p2 = ld16(base, quad_bcast<2>(off) + j16), p3 = ld16(base, quad_bcast<3>(off) + j16);
    stage_wave[(q + 0u) * STAGE_F4 + j] = to_v(p0);
    stage_wave[(q + 1u) * STAGE_F4 + j] = to_v(p1);
    stage_wave[(q + 2u) * STAGE_F4 + j] = to_v(p2);
    stage_wave[(q + 3u) * STAGE_F4 + j] = to_v(p3);
    __builtin_amdgcn_wave_barrier();  // LDS operations of one wave complete in order; this only pins the code order
    *r0 = to_f4(stage_wave[lane * STAGE_F4 + 0u]);
    *r1 = to_f4(stage_wave[lane * STAGE_F4 + 1u]);
    *r2 = to_f4(stage_wave[lane * STAGE_F4 + 2u]);
    *r3 = to_f4(stage_wave[lane * STAGE_F4 + 3u]);
    __builtin_amdgcn_wave_barrier();
}

template <bool ANY, bool COUNT, int STACK_LDS, int BLOCK, class IO>
RT_DEV void trace_loop(const DScene& sc, IO& io, uint32_t n, uint32_t* work_counter, const TraceTune tune, float t_min,
                       float t_max_in,
                       uint32_t* lds, uint32_t* overflow, uint32_t overflow_stride, float4* lds_stage = nullptr) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64u - lane));
    Stack<STACK_LDS, BLOCK> st;
    st.set_lds(lds);
    st.overflow = overflow;
    st.overflow_stride = overflow_stride;
    st.sp = 0;
    // lds_stage: RT_BLOCK * STAGE_F4 float4 of LDS; this wave's 64 slots
#if RT_COOP
    lds_f4* const stage_wave = (lds_f4*)lds_stage + (threadIdx.x & ~63u) * STAGE_F4;
#endif

    const uint32_t refill_min = tune.refill_min, chunk = tune.chunk;
    uint32_t ray = RAY_NONE;
    bool more = true;
    uint32_t pool_next = 0, pool_end = 0;  // wave-uniform
    uint32_t seg = blockIdx.x % tune.n_seg, seg_tried = 0;  // wave-uniform
    V3 wo = v3(0, 0, 0), wd = wo;                     // world-space ray
    V3 o = wo, d = wo, inv = wo;                      // object-space ray of the current object
    float nr = 1.0f;                                  // |M^-1 d|^-1 of the current object
    uint32_t obj = 0;
    uint32_t cur = REF_ADVANCE;
    float tmax = 0.0f;                                // running t_max inside the current object
    float best_t = 0.0f;                              // Engine::hit's shrinking t_max (world)
    bool got = false;                                 // a hit has been recorded through io.hit()
    uint32_t hit_tri = 0;
    bool hit_ff = false, obj_hit = false, fresh = false;
    Visits vis = {0, 0};
#if RT_TRACE_STATS
    unsigned long long s_stat[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

    for (;;) {
        // ---------------------------------------------------------------- refill idle lanes
        // The wave owns a pool [pool_next, pool_end) of consecutive ray indices; idle lanes draw
        // from it, and it is restocked `chunk` rays at a time from the launch's work counters.
        if (more) {
            const bool idle = ray == RAY_NONE;
            const unsigned long long m = __ballot(idle);
            const uint32_t n_idle = (uint32_t)__popcll(m);
            if (n_idle >= refill_min || (n_idle > 0 && __ballot(!idle) == 0ull)) {
                // A wave draws `chunk` consecutive rays at a time from the counter of its current
                // segment (home segment = block index mod n_seg; the next one when that is drained).
                // One counter for the whole launch was the bottleneck of the first version: device-
                // scope atomics on one address complete at the memory side, about one per 15 ns, and
                // a frame needs 4.4 M of them at chunk 64 (profiles/r01/tuning.md).
                while (pool_next == pool_end && more) {
                    uint32_t b = 0;
                    if (lane == 0) b = atomicAdd(work_counter + seg * WORK_STRIDE, chunk);
                    b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
                    const uint32_t seg_begin = seg * tune.seg_len;
                    if (b < tune.seg_len && seg_begin + b < n) {
                        pool_next = seg_begin + b;
                        pool_end = min(min(pool_next + chunk, seg_begin + tune.seg_len), n);
                    } else {
                        seg = seg + 1u == tune.n_seg ? 0u : seg + 1u;
                        if (++seg_tried == tune.n_seg) more = false;
                    }
                }
                if (more) {
                    const uint32_t take = min(n_idle, pool_end - pool_next);
                    const uint32_t rank = (uint32_t)__popcll(m & lt_mask);
                    if (idle && rank < take) {
                        ray = pool_next + rank;
                        io.load(ray, &wo, &wd);
                        best_t = t_max_in;
                        got = false;
                        cur = REF_ADVANCE;
                        obj_hit = false;
                        fresh = true;
                        st.sp = 0;
                        if (COUNT) vis = {0, 0};
                    }
                    pool_next += take;
                }
            }
        }
        // ---------------------------------------------------------------- pick ONE phase
        // PMC of the run-every-phase version: 24 % VALU lane utilisation (a phase ran whenever a
        // single lane needed it).  Lanes now wait in their state until their phase is the best use
        // of the next issue slots; only that phase runs in this iteration.
        const bool adv = ray != RAY_NONE && cur == REF_ADVANCE;
        const bool isnode = ray != RAY_NONE && (cur & LEAF_BIT) == 0u;
        const bool isleaf = ray != RAY_NONE && (cur & LEAF_BIT) != 0u && cur != REF_ADVANCE;
        const uint32_t n_adv = (uint32_t)__popcll(__ballot(adv)), n_node = (uint32_t)__popcll(__ballot(isnode)),
                       n_leaf = (uint32_t)__popcll(__ballot(isleaf));
        if ((n_adv | n_node | n_leaf) == 0u) break;
#if RT_COOP
        // Cooperative mode: ONE 64-byte fetch per lane per iteration serves node lanes (child pair)
        // and leaf lanes (triangle) alike -- its cost does not depend on how many lanes use it --
        // so every lane inside a tree moves on each traversal iteration.  The triangle arithmetic
        // (the expensive half) is deferred until enough leaf lanes have gathered; a waiting leaf
        // lane simply receives its triangle again.
        const bool run_trav = (n_node + n_leaf) * tune.w_node >= n_adv * tune.w_adv && (n_node + n_leaf) > 0u;
        const bool run_adv = !run_trav;
        const bool run_tri = run_trav && n_leaf * tune.w_leaf >= n_node;
#else
        const uint32_t s_adv = n_adv * tune.w_adv, s_node = n_node * tune.w_node, s_leaf = n_leaf * tune.w_leaf;
        const bool run_node = s_node >= s_leaf && s_node >= s_adv;
        const bool run_leaf = !run_node && s_leaf >= s_adv;
        const bool run_adv = !run_node && !run_leaf;
#endif
#if RT_TRACE_STATS
        if (lane == 0) {
#if RT_COOP
            const uint32_t run = run_trav ? n_node + (run_tri ? n_leaf : 0u) : n_adv;
            const bool run_node = run_trav, run_leaf = run_tri;
#else
            const uint32_t run = run_node ? n_node : run_leaf ? n_leaf : n_adv;
#endif
            RT_STAT(0, 1);                         // iterations
            RT_STAT(1, run);                       // lanes that advance
            RT_STAT(2, n_adv + n_node + n_leaf - run);  // lanes waiting for another phase
            RT_STAT(more ? 3 : 4, 64u - n_adv - n_node - n_leaf);  // idle lanes (queue not empty / draining)
            RT_STAT(5, run_node ? 1 : 0);
            RT_STAT(6, run_leaf ? 1 : 0);
            RT_STAT(7, run_adv ? 1 : 0);
        }
#endif

        // ---------------------------------------------------------------- A: next object
        if (run_adv) {
            if (adv) {
                if (obj_hit) {  // close the mesh traversal that just ended (Object::hit tail, object.rs:56-65)
                    const DObject& ob = sc.objects[obj];
                    float4 q0 = ld16(sc.tris, hit_tri * 48u), q1 = ld16(sc.tris, hit_tri * 48u + 16u),
                           q2 = ld16(sc.tris, hit_tri * 48u + 32u);
                    V3 nrm = normalise(cross(v3(q1.x, q1.y, q1.z), v3(q2.x, q2.y, q2.z)));  // primitives.rs:175
                    float t = tmax;
                    if (ob.has_transform) {
                        t = t * nr;
                        nrm = normalise(mat_transpose_mul(ob.inv_m, nrm));
                    }
                    best_t = t;
                    got = true;
                    io.hit(ray, t, nrm, obj | (hit_ff ? 0x80000000u : 0u), __float_as_uint(q0.w));
                    obj_hit = false;
                }
                for (;;) {  // Engine::hit's loop over objects (path_integrator.rs:90-95)
                    obj = fresh ? 0u : obj + 1u;
                    fresh = false;
                    if (obj >= sc.n_objects) {
                        if (ANY) io.emit_any(ray, false, vis);
                        else io.emit_closest(ray, got, vis);
                        ray = RAY_NONE;
                        break;
                    }
                    const DObject& ob = sc.objects[obj];
                    o = wo;
                    d = wd;
                    if (ob.has_transform) {  // object.rs:39-45
                        o = mat_mul(ob.inv_m, wo) + v3(ob.inv_t[0], ob.inv_t[1], ob.inv_t[2]);
                        d = mat_mul(ob.inv_m, wd);
                        nr = norm_recip(d);
                        d = d * nr;
                    }
                    const float t_lim = ANY ? t_max_in : best_t;
                    if (ob.kind == RT_SURF_MESH) {
                        inv = v3(recip(d.x), recip(d.y), recip(d.z));
                        // Skip a mesh whose ROOT box the ray misses.  The reference never tests the
                        // root (bvh.rs:252) and would fetch its child pair and miss both children;
                        // that outcome is certain: a child box lies inside the root box bound for
                        // bound (both are min/max over subsets of the same vertices) and every step
                        // of AABB::hit is monotone in the bounds, so a ray that fails the root test
                        // fails both child tests with the same arithmetic.  No triangle is tested in
                        // either case: the result is identical, one node visit and two phase changes
                        // cheaper.  Not applied when counting visits (those stay the reference's) or
                        // when the root is itself a leaf.
                        if (!COUNT && (ob.root_ref & LEAF_BIT) == 0u) {
                            float t_entry;
                            if (!aabb_hit(&ob.surf[0], &ob.surf[3], o, inv, t_min, t_lim, &t_entry)) continue;
                        }
                        tmax = t_lim;
                        cur = ob.root_ref;
                        st.sp = 0;
                        break;
                    }
                    float t = 0.0f;
                    V3 nrm = wo;
                    uint32_t ff = 0;
                    bool h = false;
                    if (ob.kind == RT_SURF_SPHERE) {
                        h = sphere_hit(v3(ob.surf[0], ob.surf[1], ob.surf[2]), ob.surf[3], o, d, t_min, t_lim, &t, &nrm, &ff);
                    } else if (ob.kind == RT_SURF_PLANE) {
                        nrm = v3(ob.surf[0], ob.surf[1], ob.surf[2]);
                        h = plane_hit(nrm, ob.surf[3], o, d, t_min, t_lim, &t, &ff);
                    } else if (ob.kind == RT_SURF_TRIANGLE) {
                        V3 e1 = v3(ob.surf[3], ob.surf[4], ob.surf[5]), e2 = v3(ob.surf[6], ob.surf[7], ob.surf[8]);
                        float det;
                        h = tri_hit(v3(ob.surf[0], ob.surf[1], ob.surf[2]), e1, e2, o, d, t_min, t_lim, &t, &det);
                        nrm = normalise(cross(e1, e2));
                        ff = det > 0.0f;
                    }
                    if (h) {
                        if (ANY) {
                            io.emit_any(ray, true, vis);
                            ray = RAY_NONE;
                            break;
                        }
                        if (ob.has_transform) {
                            t = t * nr;
                            nrm = normalise(mat_transpose_mul(ob.inv_m, nrm));
                        }
                        best_t = t;
                        got = true;
                        io.hit(ray, t, nrm, obj | (ff << 31), 0u);
                    }
                }
            }
        }

#if RT_COOP
        // ---------------------------------------------------------------- B+C: traversal step
        if (run_trav) {
            const uint32_t k = cur & ~LEAF_BIT;
            const uint32_t off = isnode ? cur << 6 : isleaf ? sc.tri_off + k * 48u : 0u;
            float4 a0, a1, b0, b1;
            coop_fetch64(sc.pairs, off, stage_wave, lane, &a0, &a1, &b0, &b1);
            if (isnode) {
                if (COUNT) vis.n_int += 1;
                float lo1[3] = {a0.x, a0.y, a0.z}, hi1[3] = {a0.w, a1.x, a1.y};
                float lo2[3] = {b0.x, b0.y, b0.z}, hi2[3] = {b0.w, b1.x, b1.y};
                const uint32_t ref1 = __float_as_uint(a1.z), ref2 = __float_as_uint(b1.z);
                float t1, t2;
                const bool h1 = aabb_hit(lo1, hi1, o, inv, t_min, tmax, &t1);
                const bool h2 = aabb_hit(lo2, hi2, o, inv, t_min, tmax, &t2);
                if (h1 && h2) {  // bvh.rs:287-295: far pushed, near next; near = child1 unless t1 > t2
                    const bool swap = t1 > t2;
                    st.push(swap ? ref1 : ref2);
                    cur = swap ? ref2 : ref1;
                } else if (h1) {
                    cur = ref1;
                } else if (h2) {
                    cur = ref2;
                } else {
                    cur = st.sp == 0 ? REF_ADVANCE : st.pop();
                }
            }
            if (run_tri) {
                if (isleaf) {  // one primitive of the leaf per step, in `indices` order (bvh.rs:256-266)
                    if (COUNT) vis.n_tri += 1;
                    float t, det;
                    bool done = false;
                    if (tri_hit(v3(a0.x, a0.y, a0.z), v3(a1.x, a1.y, a1.z), v3(b0.x, b0.y, b0.z), o, d, t_min, tmax, &t,
                                &det)) {
                        if (ANY) {
                            io.emit_any(ray, true, vis);
                            ray = RAY_NONE;
                            done = true;
                        } else {
                            tmax = t;
                            hit_tri = k;
                            hit_ff = det > 0.0f;
                            obj_hit = true;
                        }
                    }
                    if (!done) {
                        if (__float_as_uint(a1.w)) cur = st.sp == 0 ? REF_ADVANCE : st.pop();
                        else cur = LEAF_BIT | (k + 1u);
                    }
                }
            }
        }
#else
        // ---------------------------------------------------------------- B: internal node
        if (run_node) {
            if (isnode) {
                const uint32_t off = cur << 6;
                float4 a0 = ld16(sc.pairs, off), a1 = ld16(sc.pairs, off + 16u), b0 = ld16(sc.pairs, off + 32u),
                       b1 = ld16(sc.pairs, off + 48u);
                if (COUNT) vis.n_int += 1;
                float lo1[3] = {a0.x, a0.y, a0.z}, hi1[3] = {a0.w, a1.x, a1.y};
                float lo2[3] = {b0.x, b0.y, b0.z}, hi2[3] = {b0.w, b1.x, b1.y};
                const uint32_t ref1 = __float_as_uint(a1.z), ref2 = __float_as_uint(b1.z);
                float t1, t2;
                const bool h1 = aabb_hit(lo1, hi1, o, inv, t_min, tmax, &t1);
                const bool h2 = aabb_hit(lo2, hi2, o, inv, t_min, tmax, &t2);
                if (h1 && h2) {  // bvh.rs:287-295: far pushed, near next; near = child1 unless t1 > t2
                    const bool swap = t1 > t2;
                    st.push(swap ? ref1 : ref2);
                    cur = swap ? ref2 : ref1;
                } else if (h1) {
                    cur = ref1;
                } else if (h2) {
                    cur = ref2;
                } else {
                    cur = st.sp == 0 ? REF_ADVANCE : st.pop();
                }
            }
        }

        // ---------------------------------------------------------------- C: leaf
        if (run_leaf) {
            if (isleaf) {
                uint32_t k = cur & ~LEAF_BIT;
                uint32_t last;
                bool done = false;
                do {  // primitives in `indices` order (bvh.rs:256-266)
                    float4 q0 = ld16(sc.tris, k * 48u), q1 = ld16(sc.tris, k * 48u + 16u), q2 = ld16(sc.tris, k * 48u + 32u);
                    if (COUNT) vis.n_tri += 1;
                    float t, det;
                    if (tri_hit(v3(q0.x, q0.y, q0.z), v3(q1.x, q1.y, q1.z), v3(q2.x, q2.y, q2.z), o, d, t_min, tmax, &t,
                                &det)) {
                        if (ANY) {
                            io.emit_any(ray, true, vis);
                            ray = RAY_NONE;
                            done = true;
                            break;
                        }
                        tmax = t;
                        hit_tri = k;
                        hit_ff = det > 0.0f;
                        obj_hit = true;
                    }
                    last = __float_as_uint(q1.w);
                    ++k;
                } while (!last);
                if (!done) cur = st.sp == 0 ? REF_ADVANCE : st.pop();
            }
        }
#endif
    }
#if RT_TRACE_STATS
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_trace_stats[(ANY ? 8 : 0) + i], s_stat[i]);
#endif
}

}  // namespace rtd
