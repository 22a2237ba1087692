// rt_ctx: per-GPU context (device scene, stream, scratch pools) shared by the device sources.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "device_trace.h"

namespace rth {
void set_error(const char* fmt, ...);
int on_exception(const char* who) noexcept;  // host/mesh.cpp: maps the exception in flight to a status code
}

#define RT_HIP(expr)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            rth::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return RT_ERR_HIP;                                                                  \
        }                                                                                       \
    } while (0)

// Occupancy target of the traversal kernels: the compiler keeps their VGPR use within what lets this
// many waves share a SIMD (512 VGPRs / waves, 8-register granules).
#ifndef RT_WAVES_EU
#define RT_WAVES_EU 5
#endif
#ifndef RT_WAVES_EU_ANY
#define RT_WAVES_EU_ANY 6  // the any-hit loop needs fewer registers: 6 waves fit without spills
#endif
#define RT_TRACE_ATTR __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_WAVES_EU)))
#define RT_TRACE_ATTR_ANY __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_WAVES_EU_ANY)))
#ifndef RT_WAVES_EU_LEAN
#define RT_WAVES_EU_LEAN 7  // scenes without analytic sphere/triangle objects (device_trace.h, SceneFeatures); 5 / 6 / 7 waves:
                            // 54.4 / 54.2 / 53.0 ms per 1080p 64-spp frame (profiles/r02/tuning.md); 7 blocks of 20 KiB LDS fit a CU
#endif
#ifndef RT_LEAN_FEAT
#define RT_LEAN_FEAT 1u     // what the lean kernels still support: rtd::FEAT_TRANSFORM
#endif
#ifndef RT_WAVES_EU_LEAN_ANY
#define RT_WAVES_EU_LEAN_ANY 7
#endif
#define RT_TRACE_ATTR_LEAN __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_WAVES_EU_LEAN)))
#define RT_TRACE_ATTR_LEAN_ANY __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_WAVES_EU_LEAN_ANY)))
constexpr int RT_BLOCK = 256;      // threads per workgroup in the traversal kernels (4 waves)
#ifndef RT_STACK_LDS_N
#define RT_STACK_LDS_N 24
#endif
#ifndef RT_STACK_ANY_N
#define RT_STACK_ANY_N RT_STACK_LDS_N
#endif
constexpr int RT_STACK_LDS = RT_STACK_LDS_N;  // traversal-stack entries per lane kept in LDS, closest-hit kernels (1 KiB/block each)
constexpr int RT_STACK_ANY = RT_STACK_ANY_N;  // same for the any-hit kernels
#ifndef RT_STACK_LEAN_N
#define RT_STACK_LEAN_N 20
#endif
#ifndef RT_STACK_LEAN_ANY_N
#define RT_STACK_LEAN_ANY_N 20
#endif
constexpr int RT_STACK_LEAN = RT_STACK_LEAN_N;          // lean kernels: 20 KiB per block, 7 blocks per CU (71 VGPRs, no spills)
constexpr int RT_STACK_LEAN_ANY = RT_STACK_LEAN_ANY_N;  // lean any-hit kernel: the same
constexpr int rt_min_i(int a, int b) { return a < b ? a : b; }
constexpr int RT_STACK_MIN = rt_min_i(rt_min_i(RT_STACK_LEAN, RT_STACK_LEAN_ANY), rt_min_i(RT_STACK_LDS, RT_STACK_ANY));
constexpr int RT_TRACE_GRID = 2048;  // workgroups of the grid-stride trace kernels (8 per CU)

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need);  // grow-only
    void release();
};

struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool has_scene = false;
    uint32_t features = 0;  // rtd::SceneFeatures the uploaded scene needs
    rtd::DScene scene{};  // host copy of the kernel argument
    DevBuf pairs, objects, materials, lights, media, wbox;  // pairs: child pairs + triangles (one allocation)
    DevBuf overflow;      // traversal-stack overflow strips
    uint32_t overflow_levels = 0;
    DevBuf staging[16];   // rt_trace_* host<->device staging
    // rt_render state
    DevBuf path_state, counters, sample_sum;
    uint64_t max_paths_in_flight = 0;
    hipEvent_t ev[2] = {nullptr, nullptr};
    float last_kernel_ms = 0.0f;
    // rt_render: counts of every bounce arrive in pinned host memory (k_publish writes them), one ring slot of
    // 4 u32 and one event per bounce in flight; event pairs of the stage timers, reused from call to call
    uint32_t* rb_host = nullptr;
    hipEvent_t rb_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> ev_pool;
    uint32_t* scratch = nullptr;  // this device's copy of the static scratch array (rt_device.hip), resolved by rt_create
    uint32_t count_mode = RT_COUNT_REFERENCE;  // what rt_trace_*'s n_int / n_tri count (rt_set_visit_count_mode)
};

constexpr size_t RT_READBACK_RING = 8;  // the host is at most two bounces ahead of the counts it has read
int rt_ctx_ensure_overflow(rt_ctx* ctx, size_t lanes);
int rt_ctx_ensure_readback(rt_ctx* ctx);
// rt_render.hip: the renderer behind rt_render (user_rays null) and rt_ray_colour (device pointers to `width` rays)
int rt_render_impl(rt_ctx* ctx, uint32_t width, uint32_t height, const rt_rays* user_rays, uint32_t spp_first,
                   uint32_t spp_count, uint32_t depth, uint64_t seed, uint32_t flags, float* rgb_sum, rt_stats* stats);

#ifndef RT_LEAN_OVERFLOW
#define RT_LEAN_OVERFLOW true  // false: no overflow path in the lean kernels (-1 ms), scenes deeper than the LDS stack take the full ones
#endif
// the lean kernels serve scenes without analytic sphere/triangle objects whose trees fit their LDS stack
inline bool rt_scene_is_lean(const rt_ctx* ctx) {
    return (ctx->features & ~RT_LEAN_FEAT) == 0 &&
           (RT_LEAN_OVERFLOW || ctx->scene.stack_need <= (uint32_t)(RT_STACK_LEAN < RT_STACK_LEAN_ANY ? RT_STACK_LEAN : RT_STACK_LEAN_ANY));
}

// Scheduler parameters of the persistent trace kernels for a launch of n rays on `grid` blocks.
rtd::TraceTune rt_trace_tune(size_t n, int grid);
rtd::TuneBase rt_tune_base();
int rt_trace_grid(size_t n);
