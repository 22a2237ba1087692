// Frames in flight: rt_render calls of DIFFERENT frames overlapped on one GPU.
//
// A frame's late bounces are latency-bound: under 1 % of its rays, a handful of waves per launch, each launch as long
// as its longest walk -- the GPU idles through a tenth of a 64-spp frame and through 40 % of an
// 8-spp share.  Splitting ONE frame does not help (every part has its own late bounces), a second frame does: its
// large launches fill the CUs the first frame's late launches leave empty.  A pipeline holds `frames_in_flight`
// contexts of one device (each its own stream, scene copy and path state) and one host thread per context, because
// rt_render drives its launches from the host (it reads the bounce's counts back before it sizes the next launch).
// Frames are dealt to the contexts round-robin and each is rendered by the unchanged rt_render: same bits as a
// lone call.  Host code only; the reference has no counterpart (Engine::render is one synchronous loop,
// path_integrator.rs:278-318).
//
// Pacing: frames that start together stay together (equal work, and the reduce order below re-aligns them frame after
// frame), and two frames in the SAME phase gain nothing from each other -- their large launches and their late launches
// coincide (bench.py, one rank with reduces: 52.4 ms per frame, as without a pipeline, against 49.5).  A worker therefore
// starts a frame no earlier than RT_PIPELINE_PACE (default 0.8) x latency / frames_in_flight after the previous start of
// any worker, latency being the running mean of the frames' own render times in the pipeline.
//
// Multi-GPU: one communicator per context, and the frames' reduces are issued in SUBMISSION order, one at a time
// (a worker whose frame is rendered waits for the reduce of every earlier frame) -- every rank submits the same
// frames in the same order, so at any moment at most one collective is outstanding per rank and it is the same one on
// every rank, whatever the relative speed of the contexts.
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <map>
#include <mutex>
#include <set>
#include <thread>

#include "context.h"

namespace rth {
const char* get_error();  // host/mesh.cpp: the calling thread's rt_last_error() text
}

namespace {

struct Job {
    uint64_t ticket = 0;
    uint32_t width = 0, height = 0, spp_first = 0, spp_count = 0, depth = 0, flags = 0;
    uint64_t seed = 0;
    float* rgb_sum = nullptr;
    int reduce_root = RT_PIPELINE_NO_REDUCE;
    uint64_t reduce_seq = 0;  // position among the frames that reduce
};

struct Done {
    int rc = RT_OK;
    std::string error;
    rt_stats stats{};
};

struct Slot {
    rt_ctx* ctx = nullptr;
    rt_comm* comm = nullptr;
    std::deque<Job> jobs;
    std::thread worker;
};

}  // namespace

struct rt_pipeline {
    int device = 0;
    std::vector<Slot> slots;
    std::mutex m;
    std::condition_variable cv;  // jobs queued, frames done, reduce turns: all waiters re-check their own condition
    std::map<uint64_t, Done> done;
    std::set<uint64_t> pending;  // submitted and not yet waited for
    uint64_t next_ticket = 0, next_reduce_seq = 0, reduce_turn = 0;
    bool stop = false;
    bool hooked = false;  // rt_pipeline_create_hooked: the caller's render / reduce steps instead of rt_render / rt_reduce_accum
    rt_pipeline_hooks hooks{};
    uint32_t separate = 0;  // slots whose stream was seen to run beside those of all slots before it (slot 0 counts)
    // pacing (see the top of the file)
    double pace = 0.8;
    double latency_s = 0.0;  // running mean of rt_render's duration inside the pipeline; 0 = none measured yet
    std::chrono::steady_clock::time_point last_start{};
};

namespace {

void worker_main(rt_pipeline* p, size_t s) {
    Slot& slot = p->slots[s];
    for (;;) {
        Job j;
        {
            std::unique_lock<std::mutex> lk(p->m);
            p->cv.wait(lk, [&] { return p->stop || !slot.jobs.empty(); });
            if (slot.jobs.empty()) return;  // stopped and drained
            j = slot.jobs.front();
            slot.jobs.pop_front();
            while (!p->stop && p->latency_s > 0.0 && p->pace > 0.0) {  // re-evaluated after every wake-up: another worker may have started
                const auto earliest = p->last_start + std::chrono::duration_cast<std::chrono::steady_clock::duration>(
                                                          std::chrono::duration<double>(p->pace * p->latency_s / (double)p->slots.size()));
                if (std::chrono::steady_clock::now() >= earliest) break;
                p->cv.wait_until(lk, earliest);
            }
            p->last_start = std::chrono::steady_clock::now();
        }
        Done d;
        const auto t_start = std::chrono::steady_clock::now();
        if (p->hooked) d.rc = p->hooks.render(p->hooks.user, slot.ctx, (uint32_t)s, j.ticket);
        else d.rc = rt_render(slot.ctx, j.width, j.height, j.spp_first, j.spp_count, j.depth, j.seed, j.flags, j.rgb_sum, &d.stats);
        if (d.rc != RT_OK) d.error = rth::get_error();
        const double took = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        std::unique_lock<std::mutex> lk(p->m);
        p->latency_s = p->latency_s > 0.0 ? 0.75 * p->latency_s + 0.25 * took : took;
        if (j.reduce_root != RT_PIPELINE_NO_REDUCE) {
            p->cv.wait(lk, [&] { return p->reduce_turn == j.reduce_seq; });
            lk.unlock();
            if (d.rc == RT_OK) {  // a frame that failed here leaves the other ranks waiting in theirs: as with rt_render + rt_reduce_accum by hand
                if (p->hooked) d.rc = p->hooks.reduce(p->hooks.user, slot.ctx, (uint32_t)s, j.ticket);
                else d.rc = rt_reduce_accum(slot.comm, j.rgb_sum, j.width, j.height, j.reduce_root);
                if (d.rc != RT_OK) d.error = rth::get_error();
            }
            lk.lock();
            ++p->reduce_turn;
        }
        p->done.emplace(j.ticket, std::move(d));
        p->cv.notify_all();
    }
}

// ---- streams that really run side by side
// HIP multiplexes a process's streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default), a new stream joining
// the least-shared queue of its priority -- which can be the queue of the stream created just before it, and work of two
// streams on one hardware queue runs strictly in order: bench.py's multi-GPU path (PyTorch and RCCL hold six streams by
// then) had both contexts on one queue and no overlap at all.  Priorities would separate them (one queue pool per
// priority) but cost more than a millisecond per frame.  So the slots' streams are TESTED: a 2 ms spin on one, an empty
// kernel on the other -- finished while the spin still runs, or the two share a queue and the stream is replaced (the
// rejects stay alive until the end, so that the next stream goes elsewhere).
__global__ void k_spin(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
    }
}
__global__ void k_nop() {}

int runs_beside(hipStream_t busy, hipStream_t other, hipEvent_t e_busy, hipEvent_t e_other, unsigned long long ticks, bool* out) {
    k_spin<<<1, 64, 0, busy>>>(ticks);
    RT_HIP(hipGetLastError());
    RT_HIP(hipEventRecord(e_busy, busy));
    k_nop<<<1, 64, 0, other>>>();
    RT_HIP(hipGetLastError());
    RT_HIP(hipEventRecord(e_other, other));
    RT_HIP(hipEventSynchronize(e_other));
    const hipError_t q = hipEventQuery(e_busy);  // still spinning: `other` did not have to wait for it
    if (q != hipSuccess && q != hipErrorNotReady) RT_HIP(q);
    (void)hipGetLastError();
    *out = q == hipErrorNotReady;
    RT_HIP(hipEventSynchronize(e_busy));
    return RT_OK;
}

int separate_queues(rt_pipeline* p) {
    p->separate = 1;
    if (p->slots.size() < 2) return RT_OK;
    RT_HIP(hipSetDevice(p->device));
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, p->device) != hipSuccess || khz <= 0) {
        (void)hipGetLastError();
        khz = 100000;  // the constant 100 MHz counter of gfx9
    }
    const unsigned long long ticks = (unsigned long long)khz * 2ull;  // 2 ms
    hipEvent_t ea = nullptr, eb = nullptr;
    RT_HIP(hipEventCreateWithFlags(&ea, hipEventDisableTiming));
    if (hipError_t e = hipEventCreateWithFlags(&eb, hipEventDisableTiming); e != hipSuccess) {
        (void)hipEventDestroy(ea);
        RT_HIP(e);
    }
    std::vector<hipStream_t> rejects;
    int rc = RT_OK;
    for (size_t k = 1; k < p->slots.size() && rc == RT_OK; ++k) {
        bool ok = false;
        for (int attempt = 0; attempt < 12 && rc == RT_OK; ++attempt) {
            ok = true;
            for (size_t j = 0; j < k && ok && rc == RT_OK; ++j)
                rc = runs_beside(p->slots[j].ctx->stream, p->slots[k].ctx->stream, ea, eb, ticks, &ok);
            if (ok || rc != RT_OK) break;
            hipStream_t st = nullptr;
            if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
                (void)hipGetLastError();
                break;
            }
            rejects.push_back(p->slots[k].ctx->stream);
            p->slots[k].ctx->stream = st;
        }
        if (ok && rc == RT_OK) ++p->separate;  // else: more slots than free hardware queues -- the slot shares one
    }
    for (hipStream_t st : rejects) (void)hipStreamDestroy(st);
    (void)hipEventDestroy(ea);
    (void)hipEventDestroy(eb);
    return rc;
}

void shut_down(rt_pipeline* p) {
    {
        std::lock_guard<std::mutex> lk(p->m);
        p->stop = true;
    }
    p->cv.notify_all();
    for (Slot& s : p->slots)
        if (s.worker.joinable()) s.worker.join();
    for (Slot& s : p->slots) {
        if (s.comm) rt_comm_destroy(s.comm);
        s.comm = nullptr;
    }
    for (Slot& s : p->slots) {
        if (s.ctx) rt_destroy(s.ctx);
        s.ctx = nullptr;
    }
}

bool idle(rt_pipeline* p) {  // caller holds p->m
    for (const Slot& s : p->slots)
        if (!s.jobs.empty()) return false;
    return p->done.size() == p->pending.size();  // every submitted frame is rendered
}

}  // namespace

extern "C" {

static int create_pipeline(const char* who, int device, uint32_t frames_in_flight, const rt_pipeline_hooks* hooks, rt_pipeline** out) {
    if (!out || frames_in_flight < 1 || frames_in_flight > RT_PIPELINE_MAX_FRAMES) {
        rth::set_error("%s: frames_in_flight must be 1..%d", who, RT_PIPELINE_MAX_FRAMES);
        return RT_ERR_INVALID;
    }
    rt_pipeline* p = new rt_pipeline;
    p->device = device;
    if (hooks) p->hooked = true, p->hooks = *hooks;
    if (const char* e = getenv("RT_PIPELINE_PACE")) p->pace = atof(e);
    p->slots.resize(frames_in_flight);
    int rc = RT_OK;
    if (device >= 0) {  // (a hooked pipeline may run without a device: its frames are then the caller's own host work)
        for (Slot& s : p->slots)
            if ((rc = rt_create(device, &s.ctx)) != RT_OK) break;
        if (rc == RT_OK) rc = separate_queues(p);
    }
    if (rc != RT_OK) {
        shut_down(p);
        delete p;
        return rc;
    }
    for (size_t s = 0; s < p->slots.size(); ++s) p->slots[s].worker = std::thread(worker_main, p, s);
    *out = p;
    return RT_OK;
}

int rt_pipeline_create(int device, uint32_t frames_in_flight, rt_pipeline** out) try {
    if (device < 0) {
        rth::set_error("rt_pipeline_create: device %d", device);
        return RT_ERR_INVALID;
    }
    return create_pipeline("rt_pipeline_create", device, frames_in_flight, nullptr, out);
} catch (...) {
    return rth::on_exception("rt_pipeline_create");
}

int rt_pipeline_create_hooked(int device, uint32_t frames_in_flight, const rt_pipeline_hooks* hooks, rt_pipeline** out) try {
    if (!hooks || !hooks->render) {
        rth::set_error("rt_pipeline_create_hooked: a render hook is needed");
        return RT_ERR_INVALID;
    }
    return create_pipeline("rt_pipeline_create_hooked", device, frames_in_flight, hooks, out);
} catch (...) {
    return rth::on_exception("rt_pipeline_create_hooked");
}

int rt_pipeline_submit_hooked(rt_pipeline* p, int reduce, uint64_t* ticket) try {
    if (!p || !ticket || !p->hooked) {
        rth::set_error("rt_pipeline_submit_hooked: not a pipeline of rt_pipeline_create_hooked");
        return RT_ERR_INVALID;
    }
    if (reduce && !p->hooks.reduce) {
        rth::set_error("rt_pipeline_submit_hooked: a reduce was asked for, but the pipeline has no reduce hook");
        return RT_ERR_STATE;
    }
    Job j;
    j.reduce_root = reduce ? 0 : RT_PIPELINE_NO_REDUCE;
    {
        std::lock_guard<std::mutex> lk(p->m);
        j.ticket = p->next_ticket;
        if (reduce) j.reduce_seq = p->next_reduce_seq;
        p->slots[j.ticket % p->slots.size()].jobs.push_back(j);
        p->pending.insert(j.ticket);
        ++p->next_ticket;
        if (reduce) ++p->next_reduce_seq;
    }
    p->cv.notify_all();
    *ticket = j.ticket;
    return RT_OK;
} catch (...) {
    return rth::on_exception("rt_pipeline_submit_hooked");
}

void rt_pipeline_destroy(rt_pipeline* p) {
    if (!p) return;
    shut_down(p);  // renders what is still queued first
    delete p;
}

uint32_t rt_pipeline_frames(const rt_pipeline* p) { return p ? (uint32_t)p->slots.size() : 0u; }

uint32_t rt_pipeline_separate_queues(const rt_pipeline* p) { return p ? p->separate : 0u; }

rt_ctx* rt_pipeline_context(rt_pipeline* p, uint32_t slot) {
    if (!p || slot >= p->slots.size()) {
        rth::set_error("rt_pipeline_context: no such slot");
        return nullptr;
    }
    return p->slots[slot].ctx;
}

int rt_pipeline_upload_scene(rt_pipeline* p, const rt_scene_flat* scene) try {
    if (!p || !scene) {
        rth::set_error("rt_pipeline_upload_scene: null argument");
        return RT_ERR_INVALID;
    }
    if (p->device < 0) {
        rth::set_error("rt_pipeline_upload_scene: the pipeline has no device");
        return RT_ERR_STATE;
    }
    {
        std::lock_guard<std::mutex> lk(p->m);
        if (!idle(p)) {
            rth::set_error("rt_pipeline_upload_scene: frames are in flight (wait for them first)");
            return RT_ERR_STATE;
        }
    }
    size_t reusable = 0;
    for (Slot& s : p->slots) {
        int rc = rt_upload_scene(s.ctx, scene);
        if (rc != RT_OK) return rc;
        reusable += s.ctx->path_state.bytes;
    }
    // rt_render sizes a context's path state from the free HBM it finds (40 %): the first context of several would
    // take most of it.  Share 80 % of what is free now among the contexts instead (196 B per path slot, as there).
    RT_HIP(hipSetDevice(p->device));
    size_t free_b = 0, total_b = 0;
    RT_HIP(hipMemGetInfo(&free_b, &total_b));
    const double share = (double)(free_b + reusable) * 0.8 / (double)p->slots.size();
    uint64_t max_paths = (uint64_t)(share / 196.0);
    if (max_paths < ((uint64_t)1 << 20)) max_paths = (uint64_t)1 << 20;
    for (Slot& s : p->slots) {
        int rc = rt_set_max_paths_in_flight(s.ctx, max_paths);
        if (rc != RT_OK) return rc;
    }
    return RT_OK;
} catch (...) {
    return rth::on_exception("rt_pipeline_upload_scene");
}

int rt_pipeline_comm_create(rt_pipeline* p, const uint8_t* ids, int world_size, int rank) try {
    if (!p || !ids) {
        rth::set_error("rt_pipeline_comm_create: null argument");
        return RT_ERR_INVALID;
    }
    if (p->device < 0) {
        rth::set_error("rt_pipeline_comm_create: the pipeline has no device");
        return RT_ERR_STATE;
    }
    {
        std::lock_guard<std::mutex> lk(p->m);
        if (!idle(p)) {
            rth::set_error("rt_pipeline_comm_create: frames are in flight (wait for them first)");
            return RT_ERR_STATE;
        }
    }
    for (Slot& s : p->slots)
        if (s.comm) {
            rth::set_error("rt_pipeline_comm_create: the pipeline has communicators already");
            return RT_ERR_STATE;
        }
    // slot by slot, in the same order on every rank (each creation is collective)
    for (size_t s = 0; s < p->slots.size(); ++s) {
        int rc = rt_comm_create(p->slots[s].ctx, ids + s * RT_COMM_ID_BYTES, world_size, rank, &p->slots[s].comm);
        if (rc != RT_OK) return rc;
    }
    return RT_OK;
} catch (...) {
    return rth::on_exception("rt_pipeline_comm_create");
}

int rt_pipeline_submit(rt_pipeline* p, uint32_t width, uint32_t height, uint32_t spp_first, uint32_t spp_count, uint32_t depth,
                       uint64_t seed, uint32_t flags, float* rgb_sum, int reduce_root, uint64_t* ticket) try {
    if (!p || !rgb_sum || !ticket) {
        rth::set_error("rt_pipeline_submit: null argument");
        return RT_ERR_INVALID;
    }
    if (p->hooked) {
        rth::set_error("rt_pipeline_submit: the pipeline renders through hooks (rt_pipeline_submit_hooked)");
        return RT_ERR_STATE;
    }
    if (reduce_root != RT_PIPELINE_NO_REDUCE && !p->slots[0].comm) {
        rth::set_error("rt_pipeline_submit: a reduce was asked for, but rt_pipeline_comm_create has not been called");
        return RT_ERR_STATE;
    }
    Job j;
    j.width = width, j.height = height, j.spp_first = spp_first, j.spp_count = spp_count, j.depth = depth, j.flags = flags;
    j.seed = seed, j.rgb_sum = rgb_sum, j.reduce_root = reduce_root;
    {
        std::lock_guard<std::mutex> lk(p->m);
        j.ticket = p->next_ticket;
        Slot& slot = p->slots[j.ticket % p->slots.size()];
        if (reduce_root != RT_PIPELINE_NO_REDUCE) j.reduce_seq = p->next_reduce_seq;
        slot.jobs.push_back(j);
        p->pending.insert(j.ticket);  // nothing below throws: the counters move only once the job is queued
        ++p->next_ticket;
        if (reduce_root != RT_PIPELINE_NO_REDUCE) ++p->next_reduce_seq;
    }
    p->cv.notify_all();
    *ticket = j.ticket;
    return RT_OK;
} catch (...) {
    return rth::on_exception("rt_pipeline_submit");
}

int rt_pipeline_wait(rt_pipeline* p, uint64_t ticket, rt_stats* stats) try {
    if (!p) {
        rth::set_error("rt_pipeline_wait: null argument");
        return RT_ERR_INVALID;
    }
    Done d;
    {
        std::unique_lock<std::mutex> lk(p->m);
        if (!p->pending.count(ticket)) {
            rth::set_error("rt_pipeline_wait: ticket %llu was never handed out or has been waited for", (unsigned long long)ticket);
            return RT_ERR_INVALID;
        }
        p->cv.wait(lk, [&] { return p->done.count(ticket) != 0; });
        auto it = p->done.find(ticket);
        d = std::move(it->second);
        p->done.erase(it);
        p->pending.erase(ticket);
    }
    if (stats) *stats = d.stats;
    if (d.rc != RT_OK) rth::set_error("%s", d.error.c_str());  // the worker's message, on the caller's thread
    return d.rc;
} catch (...) {
    return rth::on_exception("rt_pipeline_wait");
}

}  // extern "C"
