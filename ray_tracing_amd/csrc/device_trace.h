// Persistent-wave ray traversal: the hot loop behind rt_trace_closest / rt_trace_any / k_extend /
// k_shadow / k_tail.
//
// Why this shape (PMC profile of the first version, profiles/r01/pmc_trace_v0): with one static
// strip of rays per lane the grid averaged 1.2 resident waves per SIMD (rays over background
// finish at once, waves over the mesh run long) and a wave stalled on its slowest ray.  Here
//   * a launch is a fixed number of waves that PULL rays from work counters, so the chip stays
//     full until the queue is empty;
//   * each lane is a small state machine (advance-object / internal node / leaf) and an idle lane
//     is refilled with a new ray as soon as >= refill_min (24) lanes of its wave are idle, so one long
//     ray no longer parks 63 lanes;
//   * per ray the arithmetic and the visiting order are still exactly those of
//     Engine::hit -> Object::hit -> traverse_bvh (path_integrator.rs:88-104, object.rs:32-87,
//     bvh.rs:243-300): the schedule changes, the result (and the visit counters) do not.
//
// Tried and dropped (profiles/r01/tuning.md): one fetch shape serving node and leaf lanes in the same
// iteration (every step then pays box AND triangle arithmetic: -14 %), and a quad-cooperative 64-byte
// fetch transposed through LDS (3x the gather rate in scripts/microbench/gather64.hip, but its cost
// is per iteration rather than per active lane: +-5 % here).
#pragma once
#include "device_scene.h"

namespace rtd {

#if RT_TRACE_STATS  // diagnostics build only (scripts/build_variant.sh stats -DRT_TRACE_STATS=1)
static __device__ unsigned long long g_trace_stats[40];
#define RT_STAT(i, v) s_stat[i] += (v)
#else
#define RT_STAT(i, v)
#endif

// Launch-time tuning of the wave scheduler (uniform kernel argument).
constexpr uint32_t WORK_MAX_SEG = 64;     // work counters per ray kind and launch (a mixed launch has two kinds: 128 segments)
constexpr uint32_t WORK_STRIDE = 32;      // u32 between counters: one 128-byte line each
constexpr uint32_t WORK_U32 = WORK_MAX_SEG * WORK_STRIDE;
constexpr uint32_t WORK_BUF_U32 = 2 * WORK_U32 + 32;  // a launch's work buffer: 128 counters + the two drained masks (own line)
struct TraceTune {
    uint32_t chunk;       // rays a wave takes from a work counter per atomic
    uint32_t n_seg;       // the queue is split into n_seg segments, one counter each (<= 2 * WORK_MAX_SEG)
    uint32_t seg_len;     // rays per segment (multiple of chunk); with per-segment counts (trace_loop's seg_counts): the
                          // distance between the first items of consecutive segments
    uint32_t refill_min;  // idle lanes that trigger a refill
    uint32_t w_adv, w_node, w_leaf;  // phase weights: each loop iteration runs the ONE phase with the largest
                                     // (lanes waiting for it) x weight
};

// The launch-independent part of TraceTune (host: rt_tune_base(); RT_TUNE overrides it) and the rule that sizes the
// rest from the number of rays -- evaluated on the host for rt_trace_*, whose n is an argument, and at the top of
// rt_render's trace kernels, whose n lives in device memory (rt_render.hip: the host enqueues a bounce before it
// knows how many paths survived the previous one).
struct TuneBase {
    uint32_t chunk;  // 0 = by size: 128 rays per atomic on frame-sized launches, 64 on small ones (finer tail balance)
    uint32_t refill_min, w_adv, w_node, w_leaf, max_seg;
};
__host__ __device__ inline TraceTune make_tune(const TuneBase& b, uint32_t n, uint32_t grid) {
    TraceTune t;
    t.chunk = b.chunk ? b.chunk : (n >= (16u << 20) ? 128u : 64u);
    t.refill_min = b.refill_min;
    t.w_adv = b.w_adv;
    t.w_node = b.w_node;
    t.w_leaf = b.w_leaf;
    // segments: at most max_seg, each at least 2048 chunks long (a drained segment costs every wave one probe),
    // never more than the launch has blocks
    uint32_t seg = n / (t.chunk * 2048u);
    if (seg > b.max_seg) seg = b.max_seg;
    if (seg > grid) seg = grid;
    if (seg < 1u) seg = 1u;
    t.n_seg = seg;
    const unsigned long long len = ((unsigned long long)n + seg - 1u) / seg;
    const unsigned long long rounded = (len + t.chunk - 1u) / t.chunk * t.chunk;
    t.seg_len = rounded > 0xFFFFFFFFull ? 0xFFFFFFFFu / t.chunk * t.chunk : (uint32_t)rounded;
    return t;
}

constexpr uint32_t RAY_NONE = 0xFFFFFFFFu;
constexpr uint32_t REF_ADVANCE = 0xFFFFFFFFu;  // "current object finished": never a valid leaf ref
constexpr uint32_t REF_DONE = 0xFFFFFFFEu;     // TRACE_PATH: ray finished, waiting for its continuation

// What a lane's ray is: the kind fixed per kernel, or (TRACE_PATH) chosen per ray by the IO policy,
// which then also supplies the NEXT ray of the same path when one finishes (k_tail).
enum TraceMode { TRACE_CLOSEST = 0, TRACE_ANY = 1, TRACE_PATH = 2 };

// 64-byte child pair / 48-byte triangle fetches with a 32-bit byte offset from a uniform base
// (global_load_dwordx4 v, v_off, s[base]); rt_upload_scene bounds both arrays to < 4 GiB.
RT_DEV float4 ld16(const void* base, uint32_t byte_off) {
    return *reinterpret_cast<const float4*>(static_cast<const char*>(base) + byte_off);
}

// IO policy (one object per lane):
//   load(i, &o, &d) -> tag                 world-space ray i of the launch; the tag (never 0xFFFFFFFF) names
//                                          the ray in every later call -- `i` below -- so that a policy whose
//                                          rays sit behind a queue looks its slot up once
//   hit(i, t, n, meta, prim)               closest-hit rays: a hit closer than any before it on ray i
//   escaped(i, o, d)                       closest-hit ray i (world-space o, d) finished without a hit; before emit_closest / next
//   emit_closest(i, got, vis)              closest-hit ray i is finished
//   emit_any(i, occluded, vis)             any-hit ray i is finished
//   TRACE_PATH only (CONTINUES = true: next() may continue the lane with another ray, so finished lanes are
//   parked for a continuation phase; false: next() is called at once and its return value ignored):
//   first_is_any(i)                        kind of the ray load() returned
//   next(i, was_any, got, occluded, vis, &o, &d, &any) -> bool
//                                          ray of item i finished; true = the lane continues with (o, d, any)
// FEAT: what the scene's objects can be (SceneFeatures).  The advance phase -- transforms, analytic
// sphere and triangle objects -- is what sets the register count of the whole loop: 94 VGPRs with
// everything, 72 with meshes and planes only, i.e. 5 against 7 waves per SIMD.  rt_upload_scene
// records what the scene uses and the hot kernels exist in a lean and a full instantiation.
enum SceneFeatures : uint32_t { FEAT_TRANSFORM = 1u, FEAT_SPHERE_TRIANGLE = 2u, FEAT_ALL = 3u };

template <class IO, bool PATH> struct ParksLanes { static constexpr bool value = false; };
template <class IO> struct ParksLanes<IO, true> { static constexpr bool value = IO::CONTINUES; };

// COUNT: 0 = no visit counters; COUNT_REFERENCE = the reference's visits (no root-box skip, no top-level cull:
// equal to the oracle's counters ray for ray); COUNT_FETCHES = the visits this loop performs when it is timed
// (root-box skip and cull active), i.e. the fetches the roofline figure prices.
enum CountMode { COUNT_NONE = 0, COUNT_REFERENCE = 1, COUNT_FETCHES = 2 };

template <int MODE, int COUNT, int STACK_LDS, int BLOCK, uint32_t FEAT = FEAT_ALL, bool OVERFLOW = true, class IO>
RT_DEV void trace_loop(const DScene& sc, IO& io, uint32_t n, uint32_t* work_counter, const TraceTune tune, float t_min,
                       float t_max_in, uint32_t* lds, uint32_t* overflow, uint32_t overflow_stride,
                       const uint32_t* seg_counts = nullptr, uint32_t counter_first = 0u, const uint32_t* live_segs = nullptr) {
    // The launch's items: segment s holds items [s * seg_len, s * seg_len + count(s)).  count(s) comes from device
    // memory (rt_render: the survivors of a bounce are compacted per segment, rt_render.hip) or, without it, the n
    // items are contiguous and every segment but the last is full.
    // live_segs (with seg_counts): [0] = number of segments that have items, [1..] = their indices.  A wave's home segment is
    // one of THOSE (block index mod their number).  With block index mod n_seg, every wave whose home segment is empty --
    // a frame whose geometry covers the middle bands of the image leaves the outer bands' segments without survivors --
    // moves to the first segment with items after it: all of them to the same one, which they drain on one counter (~15 ns
    // an atomic), then on together to the next.  Config 1's second bounce took 0.74 ms that way instead of 0.15.
    // work_counter: WORK_BUF_U32 zeroed u32 -- 2 * WORK_MAX_SEG counters, one 128-byte line each, then two 64-bit masks of
    // the counters whose last chunk has been handed out.  Segment s of this launch draws from counter counter_first + s
    // (rt_render runs a shadow launch on counters 64.. and the next closest-hit launch on counters 0.. between two
    // resets; a mixed launch uses all 128).
    constexpr bool PATH = MODE == TRACE_PATH;
    // A policy whose rays have no continuation (k_mixed: two kinds of ray, each finished for good) is served at
    // once like the single-kind kernels; only a policy that may hand the lane its NEXT ray parks it for phase D.
    constexpr bool PARK = ParksLanes<IO, PATH>::value;
    const uint32_t lane = threadIdx.x & 63u;
    Stack<STACK_LDS, BLOCK, OVERFLOW> st;
    st.set_lds(lds);
    st.overflow = overflow;
    st.overflow_stride = overflow_stride;
    st.sp = 0;

    const uint32_t refill_min = tune.refill_min, chunk = tune.chunk;
    uint32_t ray = RAY_NONE;
    bool more = true;
    uint32_t pool_next = 0, pool_end = 0;  // wave-uniform
    auto seg_count = [&](uint32_t s_) -> uint32_t {
        if (seg_counts) return seg_counts[s_];
        const unsigned long long first = (unsigned long long)s_ * tune.seg_len;
        return first >= n ? 0u : (uint32_t)min((unsigned long long)tune.seg_len, (unsigned long long)n - first);
    };
    uint32_t seg = blockIdx.x % tune.n_seg;  // wave-uniform: the segment the wave draws from (first: its home segment)
    if (live_segs) {
        const uint32_t n_live = live_segs[0];
        if (n_live != 0u && n_live <= tune.n_seg) seg = min(live_segs[1u + blockIdx.x % n_live], tune.n_seg - 1u);
    }
    // wave-uniform: segments this wave need not ask (again): empty from the start, or found drained
    unsigned long long dead_lo = __builtin_amdgcn_ballot_w64(lane >= tune.n_seg || seg_count(lane) == 0u);
    unsigned long long dead_hi = tune.n_seg > 64u ? __builtin_amdgcn_ballot_w64(lane + 64u >= tune.n_seg || seg_count(lane + 64u) == 0u) : ~0ull;
    uint32_t seg_cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)seg_count(seg));  // items of the current segment
    unsigned long long* const drained = reinterpret_cast<unsigned long long*>(work_counter + 2u * WORK_U32);
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
    unsigned long long cand = ~0ull;                  // object groups this ray can hit at all
    bool lane_any = MODE == TRACE_ANY;                // TRACE_PATH: kind of this lane's current ray
    bool done_occluded = false;                       // TRACE_PATH: outcome of a finished any-hit ray
    Visits vis = {0, 0};
#if RT_TRACE_STATS
    unsigned long long s_stat[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long s_uni[4] = {0, 0, 0, 0};  // node steps, steps with one node for all lanes, lanes on the first lane's node, lanes
    unsigned long long s_cyc[5] = {0, 0, 0, 0, 0};  // wave cycles in: scheduler+refill, advance, node, leaf, (total)
    unsigned long long t_mark = __builtin_readcyclecounter();
    const unsigned long long t_start = t_mark;
#define RT_CYC(i) { unsigned long long t_now = __builtin_readcyclecounter(); s_cyc[i] += t_now - t_mark; t_mark = t_now; }
#else
#define RT_CYC(i)
#endif

    // a ray is finished: hand it to the IO policy now, or (TRACE_PATH) park the lane for phase D
    auto finish = [&](bool occluded) {
        if constexpr (MODE != TRACE_ANY) {  // a closest-hit ray that met nothing: the policy gets the ray itself (rt_render marks
            if (!(PATH && lane_any) && !got) io.escaped(ray, wo, wd);  // the ones no light lies on, see k_classify)
        }
        if constexpr (PARK) {
            cur = REF_DONE;
            done_occluded = occluded;
        } else if constexpr (PATH) {
            V3 no, nd;
            bool nany;
            (void)io.next(ray, lane_any, got, occluded, vis, &no, &nd, &nany);
            ray = RAY_NONE;
        } else {
            if constexpr (MODE == TRACE_ANY) io.emit_any(ray, occluded, vis);
            else io.emit_closest(ray, got, vis);
            ray = RAY_NONE;
        }
    };
    auto begin_ray = [&]() {
        best_t = t_max_in;
        got = false;
        cur = REF_ADVANCE;
        obj_hit = false;
        fresh = true;
        st.sp = 0;
        if (COUNT) vis = {0, 0};
        // Top level: one slab test per bounded object against the WORLD ray, before any per-object transform.
        // Objects whose bounds the ray misses drop out of Engine::hit's loop for this ray; the others are
        // still met in vector order with the reference's shrinking t_max.  Not when counting reference visits.
        // The test runs over the whole half-line t > 0, not over (t_min, t_max): Object::hit hands the limits
        // UNSCALED to the object-space ray (object.rs:39-47), so under a scaling transform object k accepts the
        // world range (t_min * nr, t_max * nr) with nr = 1 / |M^-1 d| depending on the ray -- a mesh scaled x2 is
        // hit at world t = 1.5 with t_max = 1.  Every accepted hit has t_obj > t_min, hence world t > 0 when
        // t_min >= 0 (a negative t_min opens the line backwards as well).
        if (COUNT != COUNT_REFERENCE && sc.cull_test != 0ull) {  // uniform
            const V3 winv = v3(recip(wd.x), recip(wd.y), recip(wd.z));
            const float w_lo = t_min < 0.0f ? -__builtin_inff() : 0.0f;
            cand = ~sc.cull_test;
            for (unsigned long long m = sc.cull_test; m != 0ull; m &= m - 1ull) {
                const uint32_t k = (uint32_t)__builtin_ctzll(m);
                CFloat* b = (CFloat*)sc.wbox + 8u * k;
                float t_entry;
                if (aabb_hit(b, b + 3, wo, winv, w_lo, __builtin_inff(), &t_entry)) cand |= 1ull << k;
            }
        }
    };
    // the first object >= from that the ray can hit: `from` itself when its group is a candidate, else the first
    // object of the next candidate group (n_objects when there is none)
    auto next_cand = [&](uint32_t from) -> uint32_t {
        if (COUNT == COUNT_REFERENCE) return from;
        const uint32_t g = from >> sc.cull_shift;
        const unsigned long long rem = g < 64u ? cand & (~0ull << g) : 0ull;
        if (rem == 0ull) return sc.n_objects;
        const uint32_t g2 = (uint32_t)__builtin_ctzll(rem);
        return g2 == g ? from : g2 << sc.cull_shift;
    };

    for (;;) {
        // ---------------------------------------------------------------- refill idle lanes
        // The wave owns a pool [pool_next, pool_end) of consecutive ray indices; idle lanes draw
        // from it, and it is restocked `chunk` rays at a time from the launch's work counters.
        bool refilled = false;  // wave-uniform: lanes received new rays in this iteration
        if (more) {
            const bool idle = ray == RAY_NONE;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(idle);
            const uint32_t n_idle = (uint32_t)__popcll(m);
            if (n_idle >= refill_min || (n_idle > 0 && __builtin_amdgcn_ballot_w64(!idle) == 0ull)) {
                // A wave draws `chunk` consecutive rays at a time from the counter of its current
                // segment (home segment = block index mod n_seg; the next one when that is drained).
                // One counter for the whole launch was the bottleneck of the first version: device-
                // scope atomics on one address complete at the memory side, about one per 15 ns, and
                // a frame needs 4.4 M of them at chunk 64 (profiles/r01/tuning.md).
                while (pool_next == pool_end && more) {
                    const uint32_t gseg = counter_first + seg;
                    uint32_t b = seg_cnt;
                    if (seg_cnt != 0u) {
                        if (lane == 0) b = atomicAdd(work_counter + gseg * WORK_STRIDE, chunk);
                        b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
                    }
                    const bool last_chunk = b >= seg_cnt || b + chunk >= seg_cnt;
                    if (b < seg_cnt) {
                        pool_next = seg * tune.seg_len + b;
                        pool_end = pool_next + min(chunk, seg_cnt - b);
                        // whoever takes a segment's last chunk says so to every wave of the launch
                        if (last_chunk && lane == 0) atomicOr(&drained[gseg >> 6], 1ull << (gseg & 63u));
                    }
                    if (last_chunk) {
                        if (seg < 64u) dead_lo |= 1ull << seg;
                        else dead_hi |= 1ull << (seg - 64u);
                    }
                    if (b >= seg_cnt) {
                        // Drained.  One look at the launch's drained masks (a load that bypasses the L1; a stale word can
                        // only send the wave to a segment that is drained too, which it then marks and never asks again)
                        // instead of one atomic per drained segment and wave: at the end of a launch every wave comes here.
                        const unsigned long long g_lo = __hip_atomic_load(&drained[counter_first >> 6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned long long g_hi = tune.n_seg > 64u ? __hip_atomic_load(&drained[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                        const unsigned long long av_lo = ~(dead_lo | g_lo), av_hi = ~(dead_hi | g_hi);
                        if ((av_lo | av_hi) == 0ull) {
                            more = false;
                        } else {  // the first segment with work after this one, cyclically
                            const uint32_t nx = seg + 1u;
                            const unsigned long long a_lo = nx < 64u ? av_lo & (~0ull << nx) : 0ull;
                            const unsigned long long a_hi = nx < 64u ? av_hi : (nx < 128u ? av_hi & (~0ull << (nx - 64u)) : 0ull);
                            if (a_lo) seg = (uint32_t)__builtin_ctzll(a_lo);
                            else if (a_hi) seg = 64u + (uint32_t)__builtin_ctzll(a_hi);
                            else if (av_lo) seg = (uint32_t)__builtin_ctzll(av_lo);
                            else seg = 64u + (uint32_t)__builtin_ctzll(av_hi);
                            seg_cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)seg_count(seg));
                        }
                    }
                }
                if (more) {
                    const uint32_t take = min(n_idle, pool_end - pool_next);  // pool_end > pool_next here
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));  // idle lanes below this one
                    if (idle && rank < take) {
                        const uint32_t idx = pool_next + rank;
                        ray = io.load(idx, &wo, &wd);
                        if constexpr (PATH) lane_any = io.first_is_any(idx);
                        begin_ray();
                    }
                    pool_next += take;
                    refilled = take > 0u;
                }
            }
        }
        // ---------------------------------------------------------------- pick ONE phase
        // PMC of the run-every-phase version: 24 % VALU lane utilisation (a phase ran whenever a
        // single lane needed it).  Lanes now wait in their state until their phase is the best use
        // of the next issue slots; only that phase runs in this iteration.
        const bool any = PATH ? lane_any : MODE == TRACE_ANY;
        const bool adv = ray != RAY_NONE && cur == REF_ADVANCE;
        const bool isdone = PARK && ray != RAY_NONE && cur == REF_DONE;
        const bool isnode = ray != RAY_NONE && (cur & LEAF_BIT) == 0u;
        const bool isleaf = ray != RAY_NONE && (cur & LEAF_BIT) != 0u && cur < REF_DONE;
        const uint32_t n_adv = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(adv)), n_node = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(isnode)),
                       n_leaf = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(isleaf));
        const uint32_t n_done = PARK ? (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(isdone)) : 0u;
        if ((n_adv | n_node | n_leaf | n_done) == 0u) break;
        const uint32_t s_adv = n_adv * tune.w_adv, s_node = n_node * tune.w_node, s_leaf = n_leaf * tune.w_leaf;
        // TRACE_PATH: continuations go first once a fifth of the busy lanes wait for one (running every
        // phase in every iteration instead was measured slower: 6.3 ms against 4.0 ms for 219 000 paths)
        const bool run_done = PARK && n_done > 0u && 4u * n_done >= n_adv + n_node + n_leaf;
#ifndef RT_ADV_AT_REFILL
#define RT_ADV_AT_REFILL 1
#endif
        // new rays start with an advance (first object); run it at once for the lanes just refilled (>= refill_min
        // of them, plus whoever finished a tree) instead of letting them wait until advance lanes outweigh node lanes
        const bool adv_now = RT_ADV_AT_REFILL && refilled;
        const bool run_node = !run_done && !adv_now && n_node > 0u && s_node >= s_leaf && s_node >= s_adv;
        const bool run_leaf = !run_done && !adv_now && !run_node && n_leaf > 0u && s_leaf >= s_adv;
        const bool run_adv = !run_done && !run_node && !run_leaf;
#if RT_TRACE_STATS
        if (lane == 0) {
            const uint32_t run = run_node ? n_node : run_leaf ? n_leaf : n_adv;
            RT_STAT(0, 1);                         // iterations
            RT_STAT(1, run);                       // lanes that advance
            RT_STAT(2, n_adv + n_node + n_leaf - run);  // lanes waiting for another phase
            RT_STAT(more ? 3 : 4, 64u - n_adv - n_node - n_leaf);  // idle lanes (queue not empty / draining)
#if RT_STAT_LANES  // lanes served per phase kind instead of phase counts (divide by the counts of a normal stats run)
            RT_STAT(5, run_node ? n_node : 0);
            RT_STAT(6, run_leaf ? n_leaf : 0);
            RT_STAT(7, run_adv ? n_adv : 0);
#else
            RT_STAT(5, run_node ? 1 : 0);
            RT_STAT(6, run_leaf ? 1 : 0);
            RT_STAT(7, run_adv ? 1 : 0);
#endif
        }
#endif

        RT_CYC(0)
        // ---------------------------------------------------------------- D: continuation (TRACE_PATH)
        if constexpr (PARK) {
            if (run_done && isdone) {
                bool next_any = false;
                if (io.next(ray, lane_any, got, done_occluded, vis, &wo, &wd, &next_any)) {
                    lane_any = next_any;
                    begin_ray();
                } else {
                    ray = RAY_NONE;
                    cur = REF_ADVANCE;
                }
            }
        }

        // ---------------------------------------------------------------- A: next object
        if (run_adv) {
            if (adv) {
                if (obj_hit) {  // close the mesh traversal that just ended (Object::hit tail, object.rs:56-65)
                    if constexpr (IO::DEFER_NORMAL) {
                        // The consumer rebuilds the normal from the triangle (hit_mesh's `tri`), with the same
                        // arithmetic: no dependent triangle fetch stalls the wave here.  nr is 1 for an object
                        // without a transform, and t * 1 is t.
                        const float t = (FEAT & FEAT_TRANSFORM) ? tmax * nr : tmax;
                        best_t = t;
                        got = true;
                        io.hit_mesh(ray, t, obj | (hit_ff ? 0x80000000u : 0u), hit_tri);
                    } else {
                        const DObject& ob = sc.objects[obj];
                        float4 q0 = ld16(sc.tris, hit_tri * 48u), q1 = ld16(sc.tris, hit_tri * 48u + 16u),
                               q2 = ld16(sc.tris, hit_tri * 48u + 32u);
                        V3 nrm = normalise(cross(v3(q1.x, q1.y, q1.z), v3(q2.x, q2.y, q2.z)));  // primitives.rs:175
                        float t = tmax;
                        if ((FEAT & FEAT_TRANSFORM) && ob.has_transform) {
                            t = t * nr;
                            nrm = normalise(mat_transpose_mul(ob.inv_m, nrm));
                        }
                        best_t = t;
                        got = true;
                        io.hit(ray, t, nrm, obj | (hit_ff ? 0x80000000u : 0u), __float_as_uint(q0.w));
                    }
                    obj_hit = false;
                }
            }
            // Engine::hit's loop over objects (path_integrator.rs:90-95), walked by the WAVE: each step takes
            // the next object index k of its first waiting lane and serves every lane waiting for that
            // same k, so the object record is read through a uniform address (scalar registers) instead
            // of one copy per lane -- a lane still meets the objects in vector order.
            uint32_t next = next_cand(fresh ? 0u : obj + 1u);
            bool pending = adv;
            if (adv) fresh = false;
            for (;;) {
                const unsigned long long pm = __builtin_amdgcn_ballot_w64(pending);
                if (pm == 0ull) break;
                const uint32_t first = (uint32_t)__builtin_ctzll(pm);
                const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)next, (int)first);
                const bool mine = pending && next == k;
                if (k >= sc.n_objects) {  // uniform
                    if (mine) {
                        finish(false);
                        pending = false;
                    }
                    continue;
                }
                CObject& ob = ((CObject*)sc.objects)[k];  // k is uniform: scalar loads
                if (mine) {
                    obj = k;
                    next = next_cand(k + 1u);
                    o = wo;
                    d = wd;
                    nr = 1.0f;
                    if ((FEAT & FEAT_TRANSFORM) && ob.has_transform) {  // object.rs:39-45
                        o = mat_mul(ob.inv_m, wo) + v3(ob.inv_t[0], ob.inv_t[1], ob.inv_t[2]);
                        d = mat_mul(ob.inv_m, wd);
                        nr = norm_recip(d);
                        d = d * nr;
                    }
                    const float t_lim = any ? t_max_in : best_t;
                    if (ob.kind == RT_SURF_MESH) {
                        inv = v3(recip(d.x), recip(d.y), recip(d.z));
                        // Skip a mesh whose ROOT box the ray misses.  The reference never tests the
                        // root (bvh.rs:252) and would fetch its child pair and miss both children;
                        // that outcome is certain: a child box lies inside the root box bound for
                        // bound (both are min/max over subsets of the same vertices) and every step
                        // of AABB::hit is monotone in the bounds, so a ray that fails the root test
                        // fails both child tests with the same arithmetic.  No triangle is tested in
                        // either case: the result is identical, one node visit and two phase changes
                        // cheaper.  Not applied when counting REFERENCE visits (those stay the reference's) or
                        // when the root is itself a leaf.
                        bool enter = true;
                        if (COUNT != COUNT_REFERENCE && (ob.root_ref & LEAF_BIT) == 0u) {
                            float t_entry;
                            enter = aabb_hit(&ob.surf[0], &ob.surf[3], o, inv, t_min, t_lim, &t_entry);
                        }
                        if (enter) {
                            tmax = t_lim;
                            cur = ob.root_ref;
                            st.sp = 0;
                            pending = false;
                        }
                    } else {
                        float t = 0.0f;
                        V3 nrm = wo;
                        uint32_t ff = 0;
                        bool h = false;
                        if ((FEAT & FEAT_SPHERE_TRIANGLE) && ob.kind == RT_SURF_SPHERE) {
                            h = sphere_hit(v3(ob.surf[0], ob.surf[1], ob.surf[2]), ob.surf[3], o, d, t_min, t_lim, &t, &nrm, &ff);
                        } else if (ob.kind == RT_SURF_PLANE) {
                            nrm = v3(ob.surf[0], ob.surf[1], ob.surf[2]);
                            h = plane_hit(nrm, ob.surf[3], o, d, t_min, t_lim, &t, &ff);
                        } else if ((FEAT & FEAT_SPHERE_TRIANGLE) && ob.kind == RT_SURF_TRIANGLE) {
                            V3 e1 = v3(ob.surf[3], ob.surf[4], ob.surf[5]), e2 = v3(ob.surf[6], ob.surf[7], ob.surf[8]);
                            float det;
                            h = tri_hit(v3(ob.surf[0], ob.surf[1], ob.surf[2]), e1, e2, o, d, t_min, t_lim, &t, &det);
                            nrm = normalise(cross(e1, e2));
                            ff = det > 0.0f;
                        }
                        if (h) {
                            if (any) {
                                finish(true);
                                pending = false;
                            } else {
                                if ((FEAT & FEAT_TRANSFORM) && ob.has_transform) {
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
            }
        }

        RT_CYC(1)
        // ---------------------------------------------------------------- B: internal node
#ifndef RT_NODE_REPS
#define RT_NODE_REPS 3
#endif
        if (run_node) {
            // Several node steps per scheduling decision: the ballots, counts and branches around a step
            // cost about as many scalar instructions as the step itself (PMC: 90 SALU against 100 VALU per
            // iteration), and 4 steps in 5 are node steps.  Lanes that leave the node state sit out the
            // remaining repetitions.  1 / 2 / 3 / 4 / 6 repetitions: extend 34.4 / 32.5 / 32.1 / 32.0 / 32.5 ms.
            for (int rep = 0; rep < RT_NODE_REPS; ++rep)
            if (ray != RAY_NONE && (cur & LEAF_BIT) == 0u) {
#if RT_TRACE_STATS  // how often do all lanes of a node step visit the SAME node (candidates for one scalar fetch)?
                {
                    const uint32_t c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur);
                    const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
                    const unsigned long long same = __builtin_amdgcn_ballot_w64(cur == c0);
                    if (lane == (uint32_t)__builtin_ctzll(act)) {
                        s_uni[0] += 1;
                        s_uni[1] += same == act ? 1 : 0;
                        s_uni[2] += (unsigned long long)__popcll(same);
                        s_uni[3] += (unsigned long long)__popcll(act);
                    }
                }
#endif
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
                // bvh.rs:287-295: both hit -> far pushed, near next, near = child1 unless t1 > t2; one hit -> that
                // child; none -> pop.  Written as two predicated regions instead of a four-way branch.
                const bool swap = t1 > t2;
                cur = (h2 && (!h1 || swap)) ? ref2 : ref1;
                if (h1 && h2) st.push(swap ? ref1 : ref2);
                if (!(h1 || h2)) cur = st.sp == 0 ? REF_ADVANCE : st.pop();
            }
        }

        RT_CYC(2)
        // ---------------------------------------------------------------- C: leaf
        if (run_leaf) {
            if (isleaf) {
                uint32_t k = cur & ~LEAF_BIT;
                bool go_on = true, occluded = false;
                // primitives in `indices` order (bvh.rs:256-266).  One exit condition, evaluated at the end of
                // the body: an any-hit lane leaving through a `break` while closest-hit lanes of the same wave
                // go on made the visit counter of TRACE_PATH instantiations count one test too many.
                while (go_on) {
                    float4 q0 = ld16(sc.tris, k * 48u), q1 = ld16(sc.tris, k * 48u + 16u), q2 = ld16(sc.tris, k * 48u + 32u);
                    if (COUNT) vis.n_tri += 1;
                    float t, det;
                    const bool h = tri_hit(v3(q0.x, q0.y, q0.z), v3(q1.x, q1.y, q1.z), v3(q2.x, q2.y, q2.z), o, d, t_min, tmax, &t, &det);
                    if (h && !any) {
                        tmax = t;
                        hit_tri = k;
                        hit_ff = det > 0.0f;
                        obj_hit = true;
                    }
                    occluded = h && any;
                    go_on = !occluded && __float_as_uint(q1.w) == 0u;
                    ++k;
                }
                if (occluded) finish(true);
                else cur = st.sp == 0 ? REF_ADVANCE : st.pop();
            }
        }
        RT_CYC(3)
    }
#if RT_TRACE_STATS
    if (lane == 0) {
        for (int i = 0; i < 8; ++i) atomicAdd(&g_trace_stats[(MODE == TRACE_ANY ? 8 : 0) + i], s_stat[i]);
        s_cyc[4] = __builtin_readcyclecounter() - t_start;
        for (int i = 0; i < 5; ++i) atomicAdd(&g_trace_stats[(MODE == TRACE_ANY ? 24 : 16) + i], s_cyc[i]);
    }
    for (int i = 0; i < 4; ++i)
        if (s_uni[i]) atomicAdd(&g_trace_stats[32 + (MODE == TRACE_ANY ? 4 : 0) + i], s_uni[i]);
#endif
}

}  // namespace rtd
