// Device context, scene upload and the batch intersector entry points (rt_trace_closest /
// rt_trace_any).  gfx950 only; compiled with -ffp-contract=off (see device_math.h).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "context.h"
#include "host/hostmath.h"

using namespace rtd;

// ---------------------------------------------------------------- buffers
int DevBuf::ensure(size_t need) {
    if (need <= bytes) return RT_OK;
    if (p) {
        RT_HIP(hipFree(p));
        p = nullptr;
        bytes = 0;
    }
    size_t cap = need + need / 8 + 256;
    RT_HIP(hipMalloc(&p, cap));
    bytes = cap;
    return RT_OK;
}
void DevBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
}

int rt_ctx_ensure_overflow(rt_ctx* ctx, size_t lanes) {
    if (ctx->overflow_levels == 0) return RT_OK;
    return ctx->overflow.ensure(lanes * ctx->overflow_levels * sizeof(uint32_t));
}

// Workgroups of a trace launch over n rays: one lane per ray up to the full persistent grid.
int rt_trace_grid(size_t n) {
    size_t blocks = (n + RT_BLOCK - 1) / RT_BLOCK;
    if (blocks > (size_t)RT_TRACE_GRID) blocks = RT_TRACE_GRID;
    return blocks == 0 ? 1 : (int)blocks;
}

int rt_ctx_ensure_readback(rt_ctx* ctx) {
    if (ctx->rb_host) return RT_OK;
    uint32_t* host = nullptr;
    RT_HIP(hipHostMalloc((void**)&host, RT_READBACK_RING * 4 * sizeof(uint32_t), hipHostMallocDefault));
    memset(host, 0, RT_READBACK_RING * 4 * sizeof(uint32_t));
    for (size_t i = 0; i < RT_READBACK_RING; ++i) {
        if (ctx->rb_ev[i]) continue;  // kept from an attempt that failed further on
        const hipError_t e = hipEventCreateWithFlags(&ctx->rb_ev[i], hipEventDisableTiming);
        if (e != hipSuccess) {  // all or nothing: rb_host set means every event exists
            ctx->rb_ev[i] = nullptr;
            (void)hipHostFree(host);
            RT_HIP(e);
        }
    }
    ctx->rb_host = host;
    return RT_OK;
}

TuneBase rt_tune_base() {
    // Defaults from sweeps on MI355X (profiles/r01/tuning.md);
    // RT_TUNE="chunk,refill,w_adv,w_node,w_leaf,max_seg" overrides them for experiments.
    static TuneBase base = [] {
        unsigned d[6] = {0, 24, 1, 3, 4, WORK_MAX_SEG};
        if (const char* e = getenv("RT_TUNE")) sscanf(e, "%u,%u,%u,%u,%u,%u", &d[0], &d[1], &d[2], &d[3], &d[4], &d[5]);
        if (d[0] > 4096) d[0] = 0;
        if (d[1] == 0 || d[1] > 64) d[1] = 24;
        if (!d[2] || !d[3] || !d[4]) d[2] = 1, d[3] = 3, d[4] = 4;
        if (d[5] == 0 || d[5] > WORK_MAX_SEG) d[5] = WORK_MAX_SEG;
        return TuneBase{d[0], d[1], d[2], d[3], d[4], d[5]};
    }();
    return base;
}

TraceTune rt_trace_tune(size_t n, int grid) { return make_tune(rt_tune_base(), (uint32_t)n, (uint32_t)grid); }

// ---------------------------------------------------------------- kernels
struct TraceIO {  // caller-provided SoA ray batch and hit arrays (rt_rays / rt_hits)
    rt_rays rays;
    rt_hits out;
    uint8_t* occluded;
    RT_DEV uint32_t load(uint32_t i, V3* o, V3* d) const {
        *o = v3(rays.ox[i], rays.oy[i], rays.oz[i]);
        *d = v3(rays.dx[i], rays.dy[i], rays.dz[i]);
        return i;
    }
    // a closer hit than any before it on this ray (Engine::hit's `closest = Some(..)`, :92-94)
    static constexpr bool DEFER_NORMAL = false;  // callers get the normal and the primitive id
    RT_DEV void hit(uint32_t i, float t, V3 n, uint32_t meta, uint32_t prim) const {
        out.t[i] = t;
        if (out.prim) out.prim[i] = prim;
        if (out.obj) out.obj[i] = meta & 0x7FFFFFFFu;
        if (out.nx) out.nx[i] = n.x;
        if (out.ny) out.ny[i] = n.y;
        if (out.nz) out.nz[i] = n.z;
        if (out.front_face) out.front_face[i] = (uint8_t)(meta >> 31);
    }
    RT_DEV void escaped(uint32_t, const V3&, const V3&) const {}
    RT_DEV void emit_closest(uint32_t i, bool got, const Visits& vis) const {
        if (!got) {
            out.t[i] = __builtin_inff();
            if (out.prim) out.prim[i] = RT_MISS;
            if (out.obj) out.obj[i] = RT_MISS;
            if (out.nx) out.nx[i] = 0.0f;
            if (out.ny) out.ny[i] = 0.0f;
            if (out.nz) out.nz[i] = 0.0f;
            if (out.front_face) out.front_face[i] = (uint8_t)0;
        }
        if (out.n_int) out.n_int[i] = vis.n_int;
        if (out.n_tri) out.n_tri[i] = vis.n_tri;
    }
    RT_DEV void emit_any(uint32_t i, bool occ, const Visits& vis) const {
        occluded[i] = occ ? 1 : 0;
        if (out.n_int) out.n_int[i] = vis.n_int;
        if (out.n_tri) out.n_tri[i] = vis.n_tri;
    }
};

template <bool ANY, int COUNT>
__global__ RT_TRACE_ATTR void k_trace(DScene sc, TraceIO io, uint32_t n, float t_min, float t_max,
                                                    uint32_t* work_counter, TraceTune tune, uint32_t* overflow,
                                                    uint32_t overflow_stride) {
    constexpr int STACK = ANY ? RT_STACK_ANY : RT_STACK_LDS;
    __shared__ uint32_t lds_stack[STACK * RT_BLOCK];
    const uint32_t gtid = blockIdx.x * RT_BLOCK + threadIdx.x;
    trace_loop<ANY ? TRACE_ANY : TRACE_CLOSEST, COUNT, STACK, RT_BLOCK>(sc, io, n, work_counter, tune, t_min, t_max,
                                                                        lds_stack + threadIdx.x, overflow + gtid,
                                                                        overflow_stride);
}

// lean form (scenes without analytic sphere / triangle objects, rtd::SceneFeatures): 6 waves per SIMD
template <bool ANY>
__global__ RT_TRACE_ATTR_LEAN void k_trace_lean(DScene sc, TraceIO io, uint32_t n, float t_min, float t_max,
                                                uint32_t* work_counter, TraceTune tune, uint32_t* overflow,
                                                uint32_t overflow_stride) {
    __shared__ uint32_t lds_stack[RT_STACK_LEAN * RT_BLOCK];
    const uint32_t gtid = blockIdx.x * RT_BLOCK + threadIdx.x;
    trace_loop<ANY ? TRACE_ANY : TRACE_CLOSEST, false, RT_STACK_LEAN, RT_BLOCK, RT_LEAN_FEAT, RT_LEAN_OVERFLOW>(
        sc, io, n, work_counter, tune, t_min, t_max, lds_stack + threadIdx.x, overflow + gtid, overflow_stride);
}

// ---------------------------------------------------------------- scratch of the allocation-free entry point
// rt_trace_closest_noalloc (include/rt_amd.h) may not call hipMalloc: the two scratch buffers of a trace launch --
// the work counters and the traversal-stack overflow strips -- live in this file-scope device array instead.  Every
// GPU the library runs on holds its own copy (24.008 MiB) until the process ends; the array's NAME is only a
// host-side stand-in, so rt_create resolves its address on the context's device once (hipGetSymbolAddress) and
// host code uses that pointer everywhere.  All calls on one GPU share the array: calls that overlap on a GPU race.
constexpr uint32_t NOALLOC_LEVELS = RT_NOALLOC_MAX_STACK_DEPTH - RT_STACK_MIN;       // overflow levels the strips hold
constexpr size_t NOALLOC_LANES = (size_t)RT_TRACE_GRID * RT_BLOCK;                   // lanes of the largest trace grid
constexpr size_t SCRATCH_U32 = WORK_BUF_U32 + NOALLOC_LANES * NOALLOC_LEVELS;
static_assert(RT_NOALLOC_MAX_STACK_DEPTH > RT_STACK_MIN, "the allocation-free entry point needs at least one overflow level");
static __device__ __align__(256) uint32_t scratch[SCRATCH_U32];

// ---------------------------------------------------------------- helpers
namespace {

bool is_device_ptr(const void* p) {
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // plain host memory: not an error for us
        return false;
    }
    return a.type == hipMemoryTypeDevice;
}

struct Stager {  // stages host arrays through device buffers of the context
    rt_ctx* ctx;
    int next = 0;
    struct Back {
        void* host;
        void* dev;
        size_t bytes;
    };
    std::vector<Back> backs;
    int in(const void* src, size_t bytes, const void** dev) {
        if (!src) {
            *dev = nullptr;
            return RT_OK;
        }
        if (is_device_ptr(src)) {
            *dev = src;
            return RT_OK;
        }
        DevBuf& b = ctx->staging[next++];
        int rc = b.ensure(bytes);
        if (rc != RT_OK) return rc;
        RT_HIP(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        *dev = b.p;
        return RT_OK;
    }
    int out(void* dst, size_t bytes, void** dev) {
        if (!dst) {
            *dev = nullptr;
            return RT_OK;
        }
        if (is_device_ptr(dst)) {
            *dev = dst;
            return RT_OK;
        }
        DevBuf& b = ctx->staging[next++];
        int rc = b.ensure(bytes);
        if (rc != RT_OK) return rc;
        *dev = b.p;
        backs.push_back({dst, b.p, bytes});
        return RT_OK;
    }
    int finish() {
        for (const Back& b : backs)
            RT_HIP(hipMemcpyAsync(b.host, b.dev, b.bytes, hipMemcpyDeviceToHost, ctx->stream));
        RT_HIP(hipStreamSynchronize(ctx->stream));
        return RT_OK;
    }
};

int check_ctx(rt_ctx* ctx, bool need_scene) {
    if (!ctx) {
        rth::set_error("null context");
        return RT_ERR_INVALID;
    }
    if (need_scene && !ctx->has_scene) {
        rth::set_error("no scene uploaded (call rt_upload_scene first)");
        return RT_ERR_STATE;
    }
    RT_HIP(hipSetDevice(ctx->device));
    return RT_OK;
}

int trace_grid(size_t n) { return rt_trace_grid(n); }

}  // namespace

// ---------------------------------------------------------------- C ABI
extern "C" {

int rt_create(int device, rt_ctx** out) try {
    if (!out) {
        rth::set_error("rt_create: null out");
        return RT_ERR_INVALID;
    }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        rth::set_error("rt_create: no HIP device available (%s)", hipGetErrorString(e));
        return RT_ERR_HIP;
    }
    if (device < 0 || device >= count) {
        rth::set_error("rt_create: device %d out of range (%d devices)", device, count);
        return RT_ERR_INVALID;
    }
    RT_HIP(hipSetDevice(device));
    rt_ctx* ctx = new rt_ctx;
    ctx->device = device;
    // address of THIS device's copy of `scratch` (the device is current); never the array's name in host code
    hipError_t es = hipGetSymbolAddress((void**)&ctx->scratch, HIP_SYMBOL(scratch));
    if (es != hipSuccess || !ctx->scratch) {
        rth::set_error("rt_create: hipGetSymbolAddress(scratch) failed: %s", hipGetErrorString(es));
        delete ctx;
        return RT_ERR_HIP;
    }
    hipError_t ec = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (ec == hipSuccess) ec = hipEventCreate(&ctx->ev[0]);
    if (ec == hipSuccess) ec = hipEventCreate(&ctx->ev[1]);
    if (ec != hipSuccess) {
        rth::set_error("rt_create: stream / event creation failed: %s", hipGetErrorString(ec));
        rt_destroy(ctx);
        return RT_ERR_HIP;
    }
    *out = ctx;
    return RT_OK;
} catch (...) {
    return rth::on_exception("rt_create");
}

void rt_destroy(rt_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    DevBuf* bufs[] = {&ctx->pairs, &ctx->objects, &ctx->materials, &ctx->lights, &ctx->media, &ctx->wbox,
                      &ctx->overflow, &ctx->path_state, &ctx->counters, &ctx->sample_sum};
    for (DevBuf* b : bufs) b->release();
    for (DevBuf& b : ctx->staging) b.release();
    if (ctx->rb_host) (void)hipHostFree(ctx->rb_host);
    for (hipEvent_t e : ctx->rb_ev)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->ev_pool)
        if (e) (void)hipEventDestroy(e);
    if (ctx->ev[0]) (void)hipEventDestroy(ctx->ev[0]);
    if (ctx->ev[1]) (void)hipEventDestroy(ctx->ev[1]);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

void* rt_stream(rt_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int rt_upload_scene(rt_ctx* ctx, const rt_scene_flat* s) try {
    int rc = check_ctx(ctx, false);
    if (rc != RT_OK) return rc;
    if (!s) {
        rth::set_error("rt_upload_scene: null scene");
        return RT_ERR_INVALID;
    }
    ctx->has_scene = false;

    // ---- BVHs: validate every mesh tree on the host (a malformed tree must never reach a kernel),
    //      measure its depth, and re-lay it out as aligned child pairs.
    std::vector<DPair> pairs;
    if ((uint64_t)s->n_tris * 48ull >= (1ull << 32)) {
        rth::set_error("rt_upload_scene: more than 4 GiB of triangles (89 478 485)");
        return RT_ERR_LIMIT;
    }
    std::vector<float4> tris((size_t)s->n_tris * 3);
    std::vector<uint32_t> root_ref(s->n_bvhs);
    uint32_t stack_need = 0;
    for (uint32_t t = 0; t < s->n_tris; ++t) {
        const float *a = s->tri_v0 + 3 * (size_t)t, *b = s->tri_v1 + 3 * (size_t)t, *c = s->tri_v2 + 3 * (size_t)t;
        float4 q0, q1, q2;
        q0.x = a[0]; q0.y = a[1]; q0.z = a[2];
        uint32_t pid = s->tri_prim_id[t];
        memcpy(&q0.w, &pid, 4);
        q1.x = b[0] - a[0]; q1.y = b[1] - a[1]; q1.z = b[2] - a[2]; q1.w = 0.0f;  // e1 (primitives.rs:147)
        q2.x = c[0] - a[0]; q2.y = c[1] - a[1]; q2.z = c[2] - a[2]; q2.w = 0.0f;  // e2 (primitives.rs:148)
        tris[3 * (size_t)t] = q0;
        tris[3 * (size_t)t + 1] = q1;
        tris[3 * (size_t)t + 2] = q2;
    }
    for (uint32_t bi = 0; bi < s->n_bvhs; ++bi) {
        const rt_bvh_flat& b = s->bvhs[bi];
        if (b.node_count == 0 || (uint64_t)b.node_offset + b.node_count > s->n_nodes ||
            (uint64_t)b.tri_offset + b.tri_count > s->n_tris || b.tri_count == 0) {
            rth::set_error("rt_upload_scene: bvh %u: ranges outside the node/triangle arrays", bi);
            return RT_ERR_INVALID;
        }
        const uint32_t* nidx = s->node_index + b.node_offset;
        const uint32_t* nnp = s->node_num_prim + b.node_offset;
        std::vector<uint32_t> pair_id(b.node_count, 0xFFFFFFFFu);
        uint32_t pair_base = (uint32_t)pairs.size(), n_internal = 0;
        for (uint32_t k = 0; k < b.node_count; ++k)
            if (nnp[k] == 0) pair_id[k] = pair_base + n_internal++;
        if (((uint64_t)pair_base + n_internal) * sizeof(DPair) >= (1ull << 32)) {
            rth::set_error("rt_upload_scene: more than 4 GiB of BVH nodes (2^26 internal nodes)");
            return RT_ERR_LIMIT;
        }
        pairs.resize((size_t)pair_base + n_internal);
        auto ref_of = [&](uint32_t k) -> uint32_t {
            return nnp[k] > 0 ? (LEAF_BIT | (b.tri_offset + nidx[k])) : pair_id[k];
        };
        // walk from the root: every node reached exactly once, children in range, leaves in range
        std::vector<uint8_t> seen(b.node_count, 0);
        std::vector<std::pair<uint32_t, uint32_t>> todo;  // node, depth
        todo.push_back({0, 0});
        uint32_t depth_max = 0;
        while (!todo.empty()) {
            auto [k, depth] = todo.back();
            todo.pop_back();
            if (seen[k]) {
                rth::set_error("rt_upload_scene: bvh %u: node %u reachable twice (not a tree)", bi, k);
                return RT_ERR_INVALID;
            }
            seen[k] = 1;
            if (depth > depth_max) depth_max = depth;
            if (nnp[k] > 0) {
                if ((uint64_t)nidx[k] + nnp[k] > b.tri_count) {
                    rth::set_error("rt_upload_scene: bvh %u: leaf %u outside the triangle range", bi, k);
                    return RT_ERR_INVALID;
                }
                uint32_t last = b.tri_offset + nidx[k] + nnp[k] - 1;
                uint32_t one = 1;
                memcpy(&tris[3 * (size_t)last + 1].w, &one, 4);  // last-in-leaf flag
                continue;
            }
            uint32_t c0 = nidx[k];
            if (c0 == 0 || (uint64_t)c0 + 1 >= b.node_count) {
                rth::set_error("rt_upload_scene: bvh %u: node %u has children out of range", bi, k);
                return RT_ERR_INVALID;
            }
            DPair& p = pairs[pair_id[k]];
            for (int c = 0; c < 2; ++c) {
                uint32_t ck = c0 + c;
                const float* up = s->node_upper + 3 * (size_t)(b.node_offset + ck);
                const float* lw = s->node_lower + 3 * (size_t)(b.node_offset + ck);
                for (int a = 0; a < 3; ++a) {
                    p.c[c].lo[a] = lw[a];
                    p.c[c].hi[a] = up[a];
                }
                p.c[c].ref = ref_of(ck);
                p.c[c].num_prim = nnp[ck];
                todo.push_back({ck, depth + 1});
            }
        }
        root_ref[bi] = ref_of(0);
        if (depth_max > stack_need) stack_need = depth_max;
    }

    // ---- objects
    std::vector<DObject> objs(s->n_objects);
    uint32_t features = 0;
    for (uint32_t i = 0; i < s->n_objects; ++i) {
        const rt_object_flat& f = s->objects[i];
        DObject& o = objs[i];
        memset(&o, 0, sizeof o);
        o.kind = f.surface_kind;
        if (f.surface_kind > RT_SURF_POINT) {
            rth::set_error("rt_upload_scene: object %u: unknown surface kind", i);
            return RT_ERR_INVALID;
        }
        if (f.surface_kind == RT_SURF_MESH) {
            if (f.bvh >= s->n_bvhs) {
                rth::set_error("rt_upload_scene: object %u: bvh index out of range", i);
                return RT_ERR_INVALID;
            }
            o.root_ref = root_ref[f.bvh];
        }
        memcpy(o.surf, f.surf, 9 * sizeof(float));
        if (f.surface_kind == RT_SURF_MESH) {
            // bounds of the mesh's root node (nodes[0] of bvh.rs:60-66): the reference never tests
            // them, the traversal uses them to skip a mesh the ray cannot enter (device_trace.h)
            const rt_bvh_flat& b = s->bvhs[f.bvh];
            for (int a = 0; a < 3; ++a) {
                o.surf[a] = s->node_lower[3 * (size_t)b.node_offset + a];
                o.surf[3 + a] = s->node_upper[3 * (size_t)b.node_offset + a];
            }
        }
        if (f.surface_kind == RT_SURF_TRIANGLE) {
            for (int a = 0; a < 3; ++a) {
                o.surf[3 + a] = f.surf[3 + a] - f.surf[a];
                o.surf[6 + a] = f.surf[6 + a] - f.surf[a];
            }
        }
        o.has_transform = f.has_transform;
        if (f.has_transform) features |= rtd::FEAT_TRANSFORM;
        if (f.surface_kind == RT_SURF_SPHERE || f.surface_kind == RT_SURF_TRIANGLE) features |= rtd::FEAT_SPHERE_TRIANGLE;
        memcpy(o.inv_m, f.inv_matrix, sizeof o.inv_m);
        memcpy(o.inv_t, f.inv_translation, sizeof o.inv_t);
        if (f.material >= s->n_materials || f.medium_inside >= (int32_t)s->n_media ||
            f.medium_outside >= (int32_t)s->n_media) {
            rth::set_error("rt_upload_scene: object %u: material/medium index out of range", i);
            return RT_ERR_INVALID;
        }
        o.material = f.material;
        o.has_medium_interface = f.has_medium_interface;
        o.medium_inside = f.medium_inside;
        o.medium_outside = f.medium_outside;
    }
    // ---- top level over objects (DScene::wbox): world-space bounds of up to 64 GROUPS of 2^cull_shift consecutive
    // objects (one object per group while the scene has at most 64).  A group has an entry when every object in it
    // is a mesh; its box is the union of its objects' boxes.  With one object per group a mesh without a transform
    // and with an internal root gets no entry: the object loop's exact root-box test (device_trace.h) already does
    // that at the same price.  An object's box is the hull of the 8 root-box corners taken to world space in
    // f64, widened by 1e-4 of the scene scale: a hit the reference finds lies within a few f32 roundings (1e-7
    // relative) of a triangle, all triangles lie inside the root box, so its ray passes the widened box with a
    // margin three orders above every rounding on the way.
    uint32_t cull_shift = 0;
    while (((uint64_t)64 << cull_shift) < s->n_objects) ++cull_shift;
    auto world_box = [&](const DObject& o, double* lo, double* hi) -> bool {
        double m[9], inv[9];  // forward matrix = inverse of the stored inverse; identity without a transform
        for (int a = 0; a < 9; ++a) m[a] = o.has_transform ? (double)o.inv_m[a] : (a % 4 == 0 ? 1.0 : 0.0);
        const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
                           m[2] * (m[3] * m[7] - m[4] * m[6]);
        if (!(std::fabs(det) > 1e-30)) return false;
        inv[0] = (m[4] * m[8] - m[5] * m[7]) / det, inv[1] = (m[2] * m[7] - m[1] * m[8]) / det, inv[2] = (m[1] * m[5] - m[2] * m[4]) / det;
        inv[3] = (m[5] * m[6] - m[3] * m[8]) / det, inv[4] = (m[0] * m[8] - m[2] * m[6]) / det, inv[5] = (m[2] * m[3] - m[0] * m[5]) / det;
        inv[6] = (m[3] * m[7] - m[4] * m[6]) / det, inv[7] = (m[1] * m[6] - m[0] * m[7]) / det, inv[8] = (m[0] * m[4] - m[1] * m[3]) / det;
        for (int a = 0; a < 3; ++a) lo[a] = 1e300, hi[a] = -1e300;
        for (int c = 0; c < 8; ++c) {
            double x[3];
            for (int a = 0; a < 3; ++a)
                x[a] = (double)((c >> a) & 1 ? o.surf[3 + a] : o.surf[a]) - (o.has_transform ? (double)o.inv_t[a] : 0.0);
            for (int a = 0; a < 3; ++a) {
                const double w = inv[3 * a] * x[0] + inv[3 * a + 1] * x[1] + inv[3 * a + 2] * x[2];
                lo[a] = std::min(lo[a], w), hi[a] = std::max(hi[a], w);
            }
        }
        double scale = 0.0;
        for (int a = 0; a < 3; ++a) scale = std::max(scale, std::max(std::max(std::fabs(lo[a]), std::fabs(hi[a])), hi[a] - lo[a]));
        const double pad = 1e-4 * scale + 1e-30;
        for (int a = 0; a < 3; ++a) lo[a] -= pad, hi[a] += pad;
        return std::isfinite(lo[0] + lo[1] + lo[2] + hi[0] + hi[1] + hi[2]);
    };
    std::vector<float> wbox(8 * 64, 0.0f);
    unsigned long long cull_test = 0;
    for (uint32_t g = 0; g < 64u && ((uint64_t)g << cull_shift) < s->n_objects; ++g) {
        const uint32_t first = g << cull_shift, last = (uint32_t)std::min<uint64_t>((uint64_t)first + (1u << cull_shift), s->n_objects);
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        bool ok = true;
        for (uint32_t i = first; i < last && ok; ++i) {
            const DObject& o = objs[i];
            double l[3], h[3];
            ok = o.kind == RT_SURF_MESH && !(cull_shift == 0 && !o.has_transform && (o.root_ref & rtd::LEAF_BIT) == 0u) &&
                 world_box(o, l, h);
            for (int a = 0; a < 3 && ok; ++a) lo[a] = std::min(lo[a], l[a]), hi[a] = std::max(hi[a], h[a]);
        }
        for (int a = 0; a < 3 && ok; ++a) {
            const float l = (float)lo[a], h = (float)hi[a];
            if (!std::isfinite(l) || !std::isfinite(h) || !(l <= h)) ok = false;
            wbox[8 * g + a] = (double)l <= lo[a] ? l : std::nextafterf(l, -INFINITY);
            wbox[8 * g + 3 + a] = (double)h >= hi[a] ? h : std::nextafterf(h, INFINITY);
        }
        if (ok) cull_test |= 1ull << g;
    }
    // ---- materials, lights, media: records the kernels switch on and loop over.  An unknown kind would fall into
    // an else branch silently; a checkered medium with a cell size <= 0 or NaN never advances its DDA
    // (device_shade.h: t_delta = 0), which is a hung GPU rather than an error code.
    auto finite3 = [](const float* v) { return std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]); };
    for (uint32_t i = 0; i < s->n_materials; ++i)
        if (s->materials[i].kind > RT_MAT_CONDUCTOR) {
            rth::set_error("rt_upload_scene: material %u: unknown kind %u", i, s->materials[i].kind);
            return RT_ERR_INVALID;
        }
    for (uint32_t i = 0; i < s->n_lights; ++i) {
        const rt_light& l = s->lights[i];
        if (l.kind > RT_LIGHT_AREA || !finite3(l.pos) || (l.kind == RT_LIGHT_AREA && !(l.scalar > 0.0f && std::isfinite(l.scalar)))) {
            rth::set_error("rt_upload_scene: light %u: unknown kind, non-finite position or non-positive radius", i);
            return RT_ERR_INVALID;
        }
    }
    for (uint32_t i = 0; i < s->n_media; ++i) {
        const rt_medium& m = s->media[i];
        bool ok = m.kind <= RT_MEDIUM_CHECKERED && finite3(m.sigma_a) && finite3(m.sigma_s) && std::isfinite(m.g);
        for (int a = 0; ok && a < 3; ++a) ok = m.sigma_a[a] >= 0.0f && m.sigma_s[a] >= 0.0f;
        if (ok && m.kind == RT_MEDIUM_CHECKERED)
            ok = std::isfinite(m.cell_size) && m.cell_size > 0.0f && std::isfinite(m.sigma_a_black) && m.sigma_a_black >= 0.0f &&
                 std::isfinite(m.sigma_s_black) && m.sigma_s_black >= 0.0f;
        if (!ok) {
            rth::set_error("rt_upload_scene: medium %u: unknown kind, negative or non-finite coefficient, or cell_size <= 0", i);
            return RT_ERR_INVALID;
        }
    }
    if (s->initial_medium >= (int32_t)s->n_media || s->initial_medium < RT_MEDIUM_AUTO) {
        rth::set_error("rt_upload_scene: initial_medium out of range");
        return RT_ERR_INVALID;
    }

    // ---- copy
    auto up = [&](DevBuf& b, const void* src, size_t bytes) -> int {
        int r = b.ensure(bytes > 256 ? bytes : 256);  // record 0 of every table is mapped (k_shade reads it on a miss)
        if (r != RT_OK) return r;
        if (bytes) RT_HIP(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return RT_OK;
    };
    // child pairs and triangles share ONE allocation so that the traversal step can fetch either with
    // the same uniform base + 32-bit lane offset; 64 B of slack lets a leaf lane issue the 4th 16-B
    // load of the common 64-B fetch shape.
    const size_t pair_bytes = pairs.size() * sizeof(DPair), tri_bytes = tris.size() * sizeof(float4);
    const size_t tri_off = (pair_bytes + 255) & ~(size_t)255;
    if (tri_off + tri_bytes + 64 >= (1ull << 32)) {
        rth::set_error("rt_upload_scene: BVH nodes + triangles exceed the 4 GiB geometry window");
        return RT_ERR_LIMIT;
    }
    if ((rc = ctx->pairs.ensure(tri_off + tri_bytes + 64)) != RT_OK) return rc;
    RT_HIP(hipMemsetAsync((char*)ctx->pairs.p + tri_off + tri_bytes, 0, 64, ctx->stream));
    if (pair_bytes) RT_HIP(hipMemcpyAsync(ctx->pairs.p, pairs.data(), pair_bytes, hipMemcpyHostToDevice, ctx->stream));
    if (tri_bytes)
        RT_HIP(hipMemcpyAsync((char*)ctx->pairs.p + tri_off, tris.data(), tri_bytes, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = up(ctx->objects, objs.data(), objs.size() * sizeof(DObject))) != RT_OK) return rc;
    if ((rc = up(ctx->wbox, wbox.data(), wbox.size() * sizeof(float))) != RT_OK) return rc;
    if ((rc = up(ctx->materials, s->materials, s->n_materials * sizeof(rt_material))) != RT_OK) return rc;
    if ((rc = up(ctx->lights, s->lights, s->n_lights * sizeof(rt_light))) != RT_OK) return rc;
    if ((rc = up(ctx->media, s->media, s->n_media * sizeof(rt_medium))) != RT_OK) return rc;
    RT_HIP(hipStreamSynchronize(ctx->stream));  // the host vectors go out of scope

    DScene& d = ctx->scene;
    d.pairs = (const DPair*)ctx->pairs.p;
    d.tris = (const float4*)((const char*)ctx->pairs.p + tri_off);
    d.tri_off = (uint32_t)tri_off;
    d.objects = (const DObject*)ctx->objects.p;
    d.materials = (const rt_material*)ctx->materials.p;
    d.lights = (const rt_light*)ctx->lights.p;
    d.media = (const rt_medium*)ctx->media.p;
    d.n_objects = s->n_objects;
    d.n_lights = s->n_lights;
    d.n_media = s->n_media;
    d.n_materials = s->n_materials;
    d.camera = s->camera;
    d.initial_medium = s->initial_medium == RT_MEDIUM_AUTO ? RT_MEDIUM_NONE : s->initial_medium;
    d.stack_need = stack_need;
    d.wbox = (const float*)ctx->wbox.p;
    d.cull_test = cull_test;
    d.cull_shift = cull_shift;
    ctx->overflow_levels = stack_need > (uint32_t)RT_STACK_MIN ? stack_need - RT_STACK_MIN : 0;
    ctx->features = features;
    ctx->has_scene = true;

    if (s->initial_medium == RT_MEDIUM_AUTO) {
        // Engine::set_initial_medium (path_integrator.rs:65-85) with the device intersector:
        // walk camera.ray(0.5, 0.5) from hit to hit, counting medium-interface entries and exits.
        const rt_camera_flat& c = s->camera;
        rth::V3 origin = rth::v3(c.origin[0], c.origin[1], c.origin[2]);
        rth::V3 llc = rth::v3(c.lower_left_corner[0], c.lower_left_corner[1], c.lower_left_corner[2]);
        rth::V3 hor = rth::v3(c.horizontal[0], c.horizontal[1], c.horizontal[2]);
        rth::V3 ver = rth::v3(c.vertical[0], c.vertical[1], c.vertical[2]);
        rth::V3 dir = rth::normalise(llc + 0.5f * hor + 0.5f * ver - origin, nullptr);  // camera.rs:60-63
        int32_t depth = 0;
        for (int guard = 0; guard < 100000; ++guard) {
            float ox = origin.x, oy = origin.y, oz = origin.z, dx = dir.x, dy = dir.y, dz = dir.z;
            rt_rays rays = {&ox, &oy, &oz, &dx, &dy, &dz};
            float t;
            uint32_t obj;
            uint8_t ff;
            rt_hits hits;
            memset(&hits, 0, sizeof hits);
            hits.t = &t;
            hits.obj = &obj;
            hits.front_face = &ff;
            rc = rt_trace_closest(ctx, &rays, 1, 1e-3f, INFINITY, &hits);
            if (rc != RT_OK) {
                ctx->has_scene = false;
                return rc;
            }
            if (obj == RT_MISS) break;
            const rt_object_flat& o = s->objects[obj];
            bool entering = ff != 0;
            if (o.has_medium_interface) {
                if (!entering && depth == 0) {
                    d.initial_medium = o.medium_outside;
                    break;
                }
                depth += entering ? 1 : -1;
            }
            origin = origin + dir * t;  // ray.origin = ray.at(intr.t)
        }
    }
    return RT_OK;
} catch (...) {
    return rth::on_exception("rt_upload_scene");
}

int rt_initial_medium(const rt_ctx* ctx, int32_t* out) {
    if (!ctx || !out || !ctx->has_scene) {
        rth::set_error("rt_initial_medium: no scene");
        return RT_ERR_STATE;
    }
    *out = ctx->scene.initial_medium;
    return RT_OK;
}

// One closest-hit / any-hit launch over device-resident arrays: the part rt_trace_* and rt_trace_*_noalloc share.
// `work` = WORK_BUF_U32 zeroed u32 (work counters + drained masks), `overflow` = traversal-stack overflow strips for `lanes` lanes.
static int launch_trace(rt_ctx* ctx, bool any, const TraceIO& io, size_t n, float t_min, float t_max, int grid,
                        uint32_t* work, uint32_t* overflow, uint32_t lanes) {
    const bool want_count = io.out.n_int || io.out.n_tri;
    const int count = !want_count ? COUNT_NONE : (ctx->count_mode == RT_COUNT_FETCHES ? COUNT_FETCHES : COUNT_REFERENCE);
    const TraceTune tune = rt_trace_tune(n, grid);
#define RT_LAUNCH_TRACE(...)                                                                                           \
    hipLaunchKernelGGL((__VA_ARGS__), dim3(grid), dim3(RT_BLOCK), 0, ctx->stream, ctx->scene, io, (uint32_t)n, t_min, t_max, \
                       work, tune, overflow, lanes)
    RT_HIP(hipEventRecord(ctx->ev[0], ctx->stream));
    if (any) {
        if (count == COUNT_REFERENCE) RT_LAUNCH_TRACE(k_trace<true, COUNT_REFERENCE>);
        else if (count == COUNT_FETCHES) RT_LAUNCH_TRACE(k_trace<true, COUNT_FETCHES>);
        else if (rt_scene_is_lean(ctx)) RT_LAUNCH_TRACE(k_trace_lean<true>);
        else RT_LAUNCH_TRACE(k_trace<true, COUNT_NONE>);
    } else {
        if (count == COUNT_REFERENCE) RT_LAUNCH_TRACE(k_trace<false, COUNT_REFERENCE>);
        else if (count == COUNT_FETCHES) RT_LAUNCH_TRACE(k_trace<false, COUNT_FETCHES>);
        else if (rt_scene_is_lean(ctx)) RT_LAUNCH_TRACE(k_trace_lean<false>);
        else RT_LAUNCH_TRACE(k_trace<false, COUNT_NONE>);
    }
#undef RT_LAUNCH_TRACE
    RT_HIP(hipGetLastError());
    RT_HIP(hipEventRecord(ctx->ev[1], ctx->stream));
    return RT_OK;
}

static bool rays_null(const rt_rays* r) { return !r || !r->ox || !r->oy || !r->oz || !r->dx || !r->dy || !r->dz; }

int rt_trace_closest(rt_ctx* ctx, const rt_rays* rays, size_t n, float t_min, float t_max, const rt_hits* hits) try {
    int rc = check_ctx(ctx, true);
    if (rc != RT_OK) return rc;
    if (rays_null(rays) || !hits || !hits->t) {
        rth::set_error("rt_trace_closest: null argument");
        return RT_ERR_INVALID;
    }
    if (n == 0) return RT_OK;
    if (n > 0xFFFFFFF0ull) {
        rth::set_error("rt_trace_closest: batch larger than 2^32 rays");
        return RT_ERR_LIMIT;
    }
    Stager sg{ctx};
    rt_rays dr;
    rt_hits dh;
    const size_t fb = n * sizeof(float);
    if ((rc = sg.in(rays->ox, fb, (const void**)&dr.ox)) || (rc = sg.in(rays->oy, fb, (const void**)&dr.oy)) ||
        (rc = sg.in(rays->oz, fb, (const void**)&dr.oz)) || (rc = sg.in(rays->dx, fb, (const void**)&dr.dx)) ||
        (rc = sg.in(rays->dy, fb, (const void**)&dr.dy)) || (rc = sg.in(rays->dz, fb, (const void**)&dr.dz)))
        return rc;
    if ((rc = sg.out(hits->t, fb, (void**)&dh.t)) || (rc = sg.out(hits->prim, fb, (void**)&dh.prim)) ||
        (rc = sg.out(hits->obj, fb, (void**)&dh.obj)) || (rc = sg.out(hits->nx, fb, (void**)&dh.nx)) ||
        (rc = sg.out(hits->ny, fb, (void**)&dh.ny)) || (rc = sg.out(hits->nz, fb, (void**)&dh.nz)) ||
        (rc = sg.out(hits->front_face, n, (void**)&dh.front_face)) ||
        (rc = sg.out(hits->n_int, fb, (void**)&dh.n_int)) || (rc = sg.out(hits->n_tri, fb, (void**)&dh.n_tri)))
        return rc;
    const int grid = trace_grid(n);
    const size_t lanes = (size_t)grid * RT_BLOCK;
    if ((rc = rt_ctx_ensure_overflow(ctx, lanes)) != RT_OK) return rc;
    if ((rc = ctx->counters.ensure(WORK_BUF_U32 * 4)) != RT_OK) return rc;
    RT_HIP(hipMemsetAsync(ctx->counters.p, 0, WORK_BUF_U32 * 4, ctx->stream));
    TraceIO io{dr, dh, nullptr};
    if ((rc = launch_trace(ctx, false, io, n, t_min, t_max, grid, (uint32_t*)ctx->counters.p, (uint32_t*)ctx->overflow.p,
                           (uint32_t)lanes)) != RT_OK)
        return rc;
    if ((rc = sg.finish()) != RT_OK) return rc;
    RT_HIP(hipEventElapsedTime(&ctx->last_kernel_ms, ctx->ev[0], ctx->ev[1]));
    return RT_OK;
} catch (...) {
    return rth::on_exception("rt_trace_closest");
}

int rt_ray_colour(rt_ctx* ctx, const rt_rays* rays, size_t n, uint32_t spp_first, uint32_t spp_count, uint32_t depth,
                  uint64_t seed, uint32_t flags, float* rgb_sum, rt_stats* stats) try {
    int rc = check_ctx(ctx, true);
    if (rc != RT_OK) return rc;
    if (rays_null(rays) || !rgb_sum) {
        rth::set_error("rt_ray_colour: null argument");
        return RT_ERR_INVALID;
    }
    if (n == 0) {  // nothing to write into rgb_sum; the statistics are those of an empty render
        if (stats) memset(stats, 0, sizeof *stats);
        return RT_OK;
    }
    if (n > 0x7FFFFFFFull) {
        rth::set_error("rt_ray_colour: batch larger than 2^31 - 1 rays");
        return RT_ERR_LIMIT;
    }
    Stager sg{ctx};
    rt_rays dr;
    const size_t fb = n * sizeof(float);
    if ((rc = sg.in(rays->ox, fb, (const void**)&dr.ox)) || (rc = sg.in(rays->oy, fb, (const void**)&dr.oy)) ||
        (rc = sg.in(rays->oz, fb, (const void**)&dr.oz)) || (rc = sg.in(rays->dx, fb, (const void**)&dr.dx)) ||
        (rc = sg.in(rays->dy, fb, (const void**)&dr.dy)) || (rc = sg.in(rays->dz, fb, (const void**)&dr.dz)))
        return rc;
    // the renderer with the caller's rays in the place of the camera's: "pixel" k is ray k, one row of n pixels
    return rt_render_impl(ctx, (uint32_t)n, 1u, &dr, spp_first, spp_count, depth, seed, flags, rgb_sum, stats);
} catch (...) {
    return rth::on_exception("rt_ray_colour");
}

int rt_trace_any(rt_ctx* ctx, const rt_rays* rays, size_t n, float t_min, float t_max, uint8_t* occluded,
                 uint32_t* n_int, uint32_t* n_tri) try {
    int rc = check_ctx(ctx, true);
    if (rc != RT_OK) return rc;
    if (rays_null(rays) || !occluded) {
        rth::set_error("rt_trace_any: null argument");
        return RT_ERR_INVALID;
    }
    if (n == 0) return RT_OK;
    if (n > 0xFFFFFFF0ull) {
        rth::set_error("rt_trace_any: batch larger than 2^32 rays");
        return RT_ERR_LIMIT;
    }
    Stager sg{ctx};
    rt_rays dr;
    uint8_t* d_occ;
    uint32_t *d_ni, *d_nt;
    const size_t fb = n * sizeof(float);
    if ((rc = sg.in(rays->ox, fb, (const void**)&dr.ox)) || (rc = sg.in(rays->oy, fb, (const void**)&dr.oy)) ||
        (rc = sg.in(rays->oz, fb, (const void**)&dr.oz)) || (rc = sg.in(rays->dx, fb, (const void**)&dr.dx)) ||
        (rc = sg.in(rays->dy, fb, (const void**)&dr.dy)) || (rc = sg.in(rays->dz, fb, (const void**)&dr.dz)))
        return rc;
    if ((rc = sg.out(occluded, n, (void**)&d_occ)) || (rc = sg.out(n_int, fb, (void**)&d_ni)) ||
        (rc = sg.out(n_tri, fb, (void**)&d_nt)))
        return rc;
    const int grid = trace_grid(n);
    const size_t lanes = (size_t)grid * RT_BLOCK;
    if ((rc = rt_ctx_ensure_overflow(ctx, lanes)) != RT_OK) return rc;
    if ((rc = ctx->counters.ensure(WORK_BUF_U32 * 4)) != RT_OK) return rc;
    RT_HIP(hipMemsetAsync(ctx->counters.p, 0, WORK_BUF_U32 * 4, ctx->stream));
    rt_hits dh;
    memset(&dh, 0, sizeof dh);
    dh.n_int = d_ni;
    dh.n_tri = d_nt;
    TraceIO io{dr, dh, d_occ};
    if ((rc = launch_trace(ctx, true, io, n, t_min, t_max, grid, (uint32_t*)ctx->counters.p, (uint32_t*)ctx->overflow.p,
                           (uint32_t)lanes)) != RT_OK)
        return rc;
    if ((rc = sg.finish()) != RT_OK) return rc;
    RT_HIP(hipEventElapsedTime(&ctx->last_kernel_ms, ctx->ev[0], ctx->ev[1]));
    return RT_OK;
} catch (...) {
    return rth::on_exception("rt_trace_any");
}

int rt_trace_closest_noalloc(rt_ctx* ctx, const rt_rays* rays, size_t n, float t_min, float t_max, const rt_hits* hits) try {
    int rc = check_ctx(ctx, true);
    if (rc != RT_OK) return rc;
    if (rays_null(rays) || !hits || !hits->t) {
        rth::set_error("rt_trace_closest_noalloc: null argument");
        return RT_ERR_INVALID;
    }
    if (n == 0) return RT_OK;
    if (n > RT_NOALLOC_MAX_RAYS) {
        rth::set_error("rt_trace_closest_noalloc: %zu rays exceed the limit of %llu per call", n,
                       (unsigned long long)RT_NOALLOC_MAX_RAYS);
        return RT_ERR_LIMIT;
    }
    if (ctx->scene.stack_need > (uint32_t)RT_NOALLOC_MAX_STACK_DEPTH) {
        rth::set_error("rt_trace_closest_noalloc: the scene's deepest BVH (%u levels) exceeds the %d levels the static "
                       "scratch holds; use rt_trace_closest", ctx->scene.stack_need, RT_NOALLOC_MAX_STACK_DEPTH);
        return RT_ERR_LIMIT;
    }
    // staging host arrays would need device buffers: every array must already be device memory
    const void* ptrs[] = {rays->ox, rays->oy, rays->oz, rays->dx, rays->dy, rays->dz, hits->t, hits->prim, hits->obj,
                          hits->nx, hits->ny, hits->nz, hits->front_face, hits->n_int, hits->n_tri};
    for (const void* p : ptrs)
        if (p && !is_device_ptr(p)) {
            rth::set_error("rt_trace_closest_noalloc: every ray / hit array must be device memory");
            return RT_ERR_INVALID;
        }
    const int grid = trace_grid(n);
    uint32_t* work = ctx->scratch;               // [0, WORK_BUF_U32): work counters and drained masks, zeroed per call
    uint32_t* overflow = ctx->scratch + WORK_BUF_U32;  // the strips need no initial value (written before they are read)
    RT_HIP(hipMemsetAsync(work, 0, WORK_BUF_U32 * sizeof(uint32_t), ctx->stream));
    TraceIO io{*rays, *hits, nullptr};
    if ((rc = launch_trace(ctx, false, io, n, t_min, t_max, grid, work, overflow, (uint32_t)NOALLOC_LANES)) != RT_OK) return rc;
    RT_HIP(hipStreamSynchronize(ctx->stream));
    RT_HIP(hipEventElapsedTime(&ctx->last_kernel_ms, ctx->ev[0], ctx->ev[1]));
    return RT_OK;
} catch (...) {
    return rth::on_exception("rt_trace_closest_noalloc");
}

int rt_set_visit_count_mode(rt_ctx* ctx, uint32_t mode) {
    if (!ctx || mode > RT_COUNT_FETCHES) {
        rth::set_error("rt_set_visit_count_mode: bad argument");
        return RT_ERR_INVALID;
    }
    ctx->count_mode = mode;
    return RT_OK;
}

int rt_set_max_paths_in_flight(rt_ctx* ctx, uint64_t max_paths) {
    if (!ctx) {
        rth::set_error("rt_set_max_paths_in_flight: null context");
        return RT_ERR_INVALID;
    }
    ctx->max_paths_in_flight = max_paths;
    return RT_OK;
}

int rt_last_kernel_ms(rt_ctx* ctx, float* ms) {
    if (!ctx || !ms) {
        rth::set_error("rt_last_kernel_ms: null argument");
        return RT_ERR_INVALID;
    }
    *ms = ctx->last_kernel_ms;
    return RT_OK;
}

#ifndef RT_BUILD_ID
#define RT_BUILD_ID "unknown"
#endif
const char* rt_version(void) { return "ray_tracing_amd 0.2 (gfx950) build " RT_BUILD_ID; }

}  // extern "C"
