// Multi-GPU: the per-GPU radiance accumulators are summed with ONE RCCL reduce over xGMI, enqueued on the
// context's stream right behind rt_render's last kernel (no host-side ordering needed).  One process per GPU;
// the host distributes the 128-byte unique id of rank 0 by whatever channel it has (file, MPI, a torch store).
// The reference has no counterpart (one thread, path_integrator.rs:286-308): rendering shards by global sample
// index, and this is the only exchange step of the path.
//
// RCCL is bound at run time (dlopen of librccl.so.1, the SONAME both /opt/rocm/lib and PyTorch's bundled copy carry),
// so the library loads on hosts without RCCL and shares the copy a host such as PyTorch has already loaded.
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "context.h"

namespace {

// The slice of the NCCL ABI this file uses (rccl.h: ncclUniqueId is 128 opaque bytes passed BY VALUE).
struct NcclUniqueId {
    char internal[128];
};
static_assert(sizeof(NcclUniqueId) == RT_COMM_ID_BYTES, "rt_amd.h and the NCCL ABI disagree on the id size");
typedef void* NcclComm;
enum { NCCL_SUCCESS = 0, NCCL_FLOAT32 = 7, NCCL_SUM = 0 };

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(NcclUniqueId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*Reduce)(const void*, void*, size_t, int, int, int, NcclComm, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char* n : names)
            if ((r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!r.handle) {
            const char* why = dlerror();  // one call: dlerror() clears the message it returns
            r.error = std::string("librccl.so.1 not found (") + (why ? why : "?") + ")";
            return;
        }
        auto sym = [&](const char* name) -> void* {
            void* p = dlsym(r.handle, name);
            if (!p && r.error.empty()) r.error = std::string("librccl: missing symbol ") + name;
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.Reduce = (decltype(r.Reduce))sym("ncclReduce");
        r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return &r;
}

int need_rccl(const char* who, Rccl** out) {
    Rccl* r = rccl();
    if (!r->error.empty()) {
        rth::set_error("%s: %s", who, r->error.c_str());
        return RT_ERR_STATE;
    }
    *out = r;
    return RT_OK;
}

#define RT_NCCL(r, expr)                                                                                   \
    do {                                                                                                   \
        int e_ = (expr);                                                                                   \
        if (e_ != NCCL_SUCCESS) {                                                                          \
            rth::set_error("%s failed: %s (%s:%d)", #expr, (r)->GetErrorString(e_), __FILE__, __LINE__);  \
            return RT_ERR_HIP;                                                                             \
        }                                                                                                  \
    } while (0)

}  // namespace

struct rt_comm {
    NcclComm comm = nullptr;
    rt_ctx* ctx = nullptr;
    int world = 1, rank = 0;
};

extern "C" {

int rt_comm_unique_id(uint8_t id[RT_COMM_ID_BYTES]) try {
    if (!id) {
        rth::set_error("rt_comm_unique_id: null argument");
        return RT_ERR_INVALID;
    }
    Rccl* r;
    int rc = need_rccl("rt_comm_unique_id", &r);
    if (rc != RT_OK) return rc;
    NcclUniqueId u;
    RT_NCCL(r, r->GetUniqueId(&u));
    memcpy(id, u.internal, sizeof u.internal);
    return RT_OK;
} catch (...) {
    return rth::on_exception("rt_comm_unique_id");
}

int rt_comm_create(rt_ctx* ctx, const uint8_t id[RT_COMM_ID_BYTES], int world_size, int rank, rt_comm** out) try {
    if (!ctx || !id || !out || world_size < 1 || rank < 0 || rank >= world_size) {
        rth::set_error("rt_comm_create: bad argument (world_size %d, rank %d)", world_size, rank);
        return RT_ERR_INVALID;
    }
    Rccl* r;
    int rc = need_rccl("rt_comm_create", &r);
    if (rc != RT_OK) return rc;
    RT_HIP(hipSetDevice(ctx->device));  // the communicator binds to the current device
    NcclUniqueId u;
    memcpy(u.internal, id, sizeof u.internal);
    NcclComm comm = nullptr;
    RT_NCCL(r, r->CommInitRank(&comm, world_size, u, rank));
    rt_comm* c = new rt_comm;
    c->comm = comm;
    c->ctx = ctx;
    c->world = world_size;
    c->rank = rank;
    *out = c;
    return RT_OK;
} catch (...) {
    return rth::on_exception("rt_comm_create");
}

void rt_comm_destroy(rt_comm* comm) {
    if (!comm) return;
    Rccl* r = rccl();
    if (comm->comm && r->CommDestroy) {
        (void)hipSetDevice(comm->ctx->device);
        (void)hipStreamSynchronize(comm->ctx->stream);
        (void)r->CommDestroy(comm->comm);
    }
    delete comm;
}

int rt_reduce_accum(rt_comm* comm, float* rgb_sum, uint32_t width, uint32_t height, int root) try {
    if (!comm || !rgb_sum || root >= comm->world) {
        rth::set_error("rt_reduce_accum: bad argument");
        return RT_ERR_INVALID;
    }
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, rgb_sum) != hipSuccess || a.type != hipMemoryTypeDevice) {
        (void)hipGetLastError();
        rth::set_error("rt_reduce_accum: rgb_sum must be device memory (render into a device accumulator)");
        return RT_ERR_INVALID;
    }
    Rccl* r;
    int rc = need_rccl("rt_reduce_accum", &r);
    if (rc != RT_OK) return rc;
    rt_ctx* ctx = comm->ctx;
    RT_HIP(hipSetDevice(ctx->device));
    const size_t count = (size_t)width * height * 3;
    // in place, on the stream rt_render used: ordered behind its resolve kernel without a host sync
    if (root < 0) RT_NCCL(r, r->AllReduce(rgb_sum, rgb_sum, count, NCCL_FLOAT32, NCCL_SUM, comm->comm, ctx->stream));
    else RT_NCCL(r, r->Reduce(rgb_sum, rgb_sum, count, NCCL_FLOAT32, NCCL_SUM, root, comm->comm, ctx->stream));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    return RT_OK;
} catch (...) {
    return rth::on_exception("rt_reduce_accum");
}

}  // extern "C"
