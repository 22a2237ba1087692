// Binned-SAH BVH builder: the crate's BoundingVolumeHierarchy::build restated for flat arrays.
//
// Follows ray_tracer/src/render/bvh.rs:60-241 decision for decision so that the node array
// (numbering, boxes, leaf ranges) and the `indices` permutation are the ones the reference builds:
//   * depth-first numbering: both children are appended when their parent is split, then the left
//     subtree is completed before the right one (bvh.rs:130-143);
//   * split only if the best binned cost beats num_prim * half_area (bvh.rs:112-115);
//   * 100 bins per axis, axes whose centroid extent is < f32::EPSILON skipped (bvh.rs:149,:167);
//   * the bin index comes from centroid_axis() = (a+b+c)/3.0 (triangle_mesh.rs:51-57) while the
//     centroid box comes from centroid() = (A+B+C) * recip(3.0) (triangle_mesh.rs:60-63);
//   * empty side => cost 0*inf = NaN => never selected (bvh.rs:208-210).
// What differs is the data layout: per-triangle boxes and centroids are computed once up front
// (the same float operations the reference repeats at every level) and the recursion is an
// explicit stack.
//
// Parallel build, node for node identical to the serial one:
//   * nodes with more than TASK_PRIMS primitives are split one after another, with the centroid
//     bounds and the 3 x 100 bins gathered by all threads (min/max and integer counts: the merged
//     result does not depend on how the range was cut);
//   * every subtree at or below TASK_PRIMS is an independent task on its own slice of `indices`,
//     built by the serial code into task-local arrays;
//   * the reference's depth-first node numbering is then reproduced by walking the upper tree in
//     that order and splicing each task's nodes in where the serial build would have created them.
// The split decisions, the partition (same two-pointer swaps) and every box are the serial code's,
// so tests/test_host_scene.py compares the output with the oracle's serial builder bit for bit.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <exception>
#include <mutex>
#include <thread>

#include "host_internal.h"

namespace rth {

namespace {

constexpr int BINS = 100;                         // bvh.rs:149
constexpr float F32_EPSILON = 1.1920928955078125e-7f;  // f32::EPSILON = 2^-23

struct Prims {
    std::vector<Box> box;         // TriangleRef::bounding_box (triangle_mesh.rs:42-48)
    std::vector<V3> centroid;     // TriangleRef::centroid     (triangle_mesh.rs:60-63)
    std::vector<V3> centroid_ax;  // TriangleRef::centroid_axis per axis (triangle_mesh.rs:51-57)
};

inline uint32_t float_to_usize_saturating(float f) {  // Rust `as usize`: NaN -> 0, negative -> 0
    if (!(f > 0.0f)) return 0;
    if (f >= 4294967040.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}

struct Nodes {  // node arrays of one (sub)tree; `index` of an internal node is local to these arrays
    std::vector<Box> node_box;
    std::vector<uint32_t> node_index, node_num_prim;
    uint32_t max_depth = 0;
};

struct Builder {
    const Prims& p;
    Nodes& b;
    std::vector<uint32_t>& indices;  // the mesh-wide permutation; a builder touches only its node's slice

    void new_node(uint32_t first_prim, uint32_t num_prim) {  // bvh.rs:89-100
        Box bb = box_empty();
        for (uint32_t i = first_prim; i < first_prim + num_prim; ++i)
            bb = box_combine(bb, p.box[indices[i]]);
        b.node_box.push_back(bb);
        b.node_index.push_back(first_prim);
        b.node_num_prim.push_back(num_prim);
    }

    struct Bins {
        Box box[3][BINS];
        int32_t count[3][BINS];
        void clear() {
            for (int a = 0; a < 3; ++a)
                for (int k = 0; k < BINS; ++k) {
                    box[a][k] = box_empty();
                    count[a][k] = 0;
                }
        }
    };

    // bins of the slice [lo, hi) of the node's primitives on every usable axis (bvh.rs:167-183)
    void gather_bins(uint32_t lo, uint32_t hi, const Box& centroid_aabb, Bins* bins) const {
        for (int plane_axis = 0; plane_axis < 3; ++plane_axis) {
            float max = centroid_aabb.upper[plane_axis];
            float min = centroid_aabb.lower[plane_axis];
            if (std::fabs(max - min) < F32_EPSILON) continue;
            float scale = (float)BINS / (max - min);
            for (uint32_t i = lo; i < hi; ++i) {
                uint32_t prim = indices[i];
                uint32_t bin = float_to_usize_saturating((p.centroid_ax[prim][plane_axis] - min) * scale);
                if (bin > BINS - 1) bin = BINS - 1;
                bins->box[plane_axis][bin] = box_combine(bins->box[plane_axis][bin], p.box[prim]);
                bins->count[plane_axis][bin] += 1;
            }
        }
    }

    // the cheapest of the 3 x 99 planes (bvh.rs:185-216)
    static void best_plane(const Box& centroid_aabb, const Bins& bins, int* axis_out, float* split_out, float* cost_out) {
        int axis = 0;
        float split_point = 0.0f;
        float cost = INFINITY;
        float left_area[BINS - 1], right_area[BINS - 1];
        int32_t left_count[BINS - 1], right_count[BINS - 1];
        for (int plane_axis = 0; plane_axis < 3; ++plane_axis) {
            float max = centroid_aabb.upper[plane_axis];
            float min = centroid_aabb.lower[plane_axis];
            if (std::fabs(max - min) < F32_EPSILON) continue;
            const Box* bin_box = bins.box[plane_axis];
            const int32_t* bin_count = bins.count[plane_axis];
            Box left_box = box_empty(), right_box = box_empty();
            int32_t left_sum = 0, right_sum = 0;
            for (int i = 0; i < BINS - 1; ++i) {
                left_sum += bin_count[i];
                left_count[i] = left_sum;
                left_box = box_combine(left_box, bin_box[i]);
                left_area[i] = box_half_area(left_box);

                right_sum += bin_count[BINS - 1 - i];
                right_count[BINS - 2 - i] = right_sum;
                right_box = box_combine(right_box, bin_box[BINS - 1 - i]);
                right_area[BINS - 2 - i] = box_half_area(right_box);
            }
            float scale2 = (max - min) / (float)BINS;
            for (int i = 0; i < BINS - 1; ++i) {
                float plane_cost =
                    (float)left_count[i] * left_area[i] + (float)right_count[i] * right_area[i];
                if (plane_cost < cost) {
                    axis = plane_axis;
                    split_point = min + scale2 * (float)(i + 1);
                    cost = plane_cost;
                }
            }
        }
        *axis_out = axis;
        *split_out = split_point;
        *cost_out = cost;
    }

    // bvh.rs:147-219
    void binned_sah_partition_plane(uint32_t node, int* axis_out, float* split_out, float* cost_out) {
        uint32_t first_prim = b.node_index[node];
        uint32_t num_prim = b.node_num_prim[node];
        Box centroid_aabb = box_empty();
        for (uint32_t i = first_prim; i < first_prim + num_prim; ++i)
            centroid_aabb = box_grow(centroid_aabb, p.centroid[indices[i]]);
        Bins bins;
        bins.clear();
        gather_bins(first_prim, first_prim + num_prim, centroid_aabb, &bins);
        best_plane(centroid_aabb, bins, axis_out, split_out, cost_out);
    }

    // bvh.rs:221-241.  Signed indices: where the reference's `j -= 1` would wrap below zero
    // (every primitive on the right with first_prim == 0) this returns an empty left side, which
    // the caller treats like any failed partition.
    uint32_t partition_sort(uint32_t first_prim, uint32_t num_prim, int axis, float split_point) {
        int64_t i = first_prim;
        int64_t j = (int64_t)first_prim + num_prim - 1;
        while (i <= j) {
            if (p.centroid_ax[indices[(size_t)i]][axis] < split_point) {
                i += 1;
            } else {
                std::swap(indices[(size_t)i], indices[(size_t)j]);
                j -= 1;
            }
        }
        return (uint32_t)i;
    }

    void build() {
        struct Item {
            uint32_t node, depth;
        };
        std::vector<Item> stack;
        stack.push_back({0, 0});
        while (!stack.empty()) {  // subdivide_recursive, bvh.rs:102-144
            Item it = stack.back();
            stack.pop_back();
            if (it.depth > b.max_depth) b.max_depth = it.depth;
            int axis;
            float split_point, cost;
            binned_sah_partition_plane(it.node, &axis, &split_point, &cost);
            uint32_t first_prim = b.node_index[it.node];
            uint32_t num_prim = b.node_num_prim[it.node];
            float current_cost = (float)num_prim * box_half_area(b.node_box[it.node]);
            if (cost >= current_cost) continue;
            uint32_t partition_point = partition_sort(first_prim, num_prim, axis, split_point);
            uint32_t left_count = partition_point - first_prim;
            uint32_t right_count = num_prim - left_count;
            if (left_count == 0 || right_count == 0) continue;
            uint32_t left_child = (uint32_t)b.node_index.size();
            b.node_index[it.node] = left_child;
            b.node_num_prim[it.node] = 0;
            new_node(first_prim, left_count);
            new_node(partition_point, right_count);
            stack.push_back({left_child + 1, it.depth + 1});  // right: handled after the left subtree
            stack.push_back({left_child, it.depth + 1});
        }
    }
};


constexpr uint32_t TASK_PRIMS = 16384;  // subtrees at or below this size are independent tasks

unsigned build_threads() {  // RT_BUILD_THREADS, else OMP_NUM_THREADS, else the hardware's count; at most 16
    unsigned t = 0;
    if (const char* e = getenv("RT_BUILD_THREADS")) t = (unsigned)atoi(e);
    if (t == 0)
        if (const char* e = getenv("OMP_NUM_THREADS")) t = (unsigned)atoi(e);
    if (t == 0) t = std::thread::hardware_concurrency();
    if (t == 0) t = 1;
    return t > 16 ? 16 : t;
}

template <class F>
void parallel_for(unsigned n_threads, F&& f) {  // f(thread index) on n_threads threads
    // An exception in a worker (std::bad_alloc of a large build) must not reach std::terminate: every thread is
    // joined, then the first exception is rethrown on the calling thread, where the C ABI maps it to a status.
    std::vector<std::thread> pool;
    std::exception_ptr first;
    std::mutex m;
    auto run = [&](unsigned t) {
        try {
            f(t);
        } catch (...) {
            std::lock_guard<std::mutex> g(m);
            if (!first) first = std::current_exception();
        }
    };
    try {
        for (unsigned t = 1; t < n_threads; ++t) pool.emplace_back([&run, t] { run(t); });
    } catch (...) {  // thread creation failed: the part not started runs here
        std::lock_guard<std::mutex> g(m);
        if (!first) first = std::current_exception();
    }
    run(0);
    for (auto& th : pool) th.join();
    if (first) std::rethrow_exception(first);
}

struct Upper {  // a node of the upper tree
    Box box;
    uint32_t first, num;
    int32_t left = -1, right = -1;  // upper-tree children, or
    int32_t task = -1;              // the task that builds this node's subtree
    uint32_t depth = 0;
};

void build_parallel(const Prims& p, uint32_t n, unsigned n_threads, BuiltBvh* out) {
    std::vector<uint32_t>& indices = out->indices;
    Nodes scratch;
    Builder top{p, scratch, indices};
    std::vector<Upper> upper;
    std::vector<int32_t> tasks;  // upper index of each task root

    auto make_upper = [&](uint32_t first, uint32_t num, uint32_t depth) -> int32_t {
        Upper u;
        u.first = first;
        u.num = num;
        u.depth = depth;
        // the node box (new_node, bvh.rs:89-100): min/max over the slice, gathered by all threads
        std::vector<Box> part(n_threads, box_empty());
        unsigned nt = num > 4 * TASK_PRIMS ? n_threads : 1;
        parallel_for(nt, [&](unsigned t) {
            uint32_t lo = first + (uint32_t)((uint64_t)num * t / nt), hi = first + (uint32_t)((uint64_t)num * (t + 1) / nt);
            Box bb = box_empty();
            for (uint32_t i = lo; i < hi; ++i) bb = box_combine(bb, p.box[indices[i]]);
            part[t] = bb;
        });
        u.box = box_empty();
        for (unsigned t = 0; t < nt; ++t) u.box = box_combine(u.box, part[t]);
        upper.push_back(u);
        return (int32_t)upper.size() - 1;
    };

    // ---- upper tree: split every node above TASK_PRIMS (any order: slices are disjoint)
    std::vector<int32_t> todo;
    todo.push_back(make_upper(0, n, 0));
    std::vector<Builder::Bins> tbins(n_threads);
    while (!todo.empty()) {
        int32_t ui = todo.back();
        todo.pop_back();
        const uint32_t first = upper[ui].first, num = upper[ui].num, depth = upper[ui].depth;
        if (num <= TASK_PRIMS) {
            upper[ui].task = (int32_t)tasks.size();
            tasks.push_back(ui);
            continue;
        }
        auto slice = [&](unsigned t, uint32_t* lo, uint32_t* hi) {
            *lo = first + (uint32_t)((uint64_t)num * t / n_threads);
            *hi = first + (uint32_t)((uint64_t)num * (t + 1) / n_threads);
        };
        std::vector<Box> cpart(n_threads, box_empty());
        parallel_for(n_threads, [&](unsigned t) {
            uint32_t lo, hi;
            slice(t, &lo, &hi);
            Box cb = box_empty();
            for (uint32_t i = lo; i < hi; ++i) cb = box_grow(cb, p.centroid[indices[i]]);
            cpart[t] = cb;
        });
        Box centroid_aabb = box_empty();
        for (unsigned t = 0; t < n_threads; ++t) centroid_aabb = box_combine(centroid_aabb, cpart[t]);
        parallel_for(n_threads, [&](unsigned t) {
            uint32_t lo, hi;
            slice(t, &lo, &hi);
            tbins[t].clear();
            top.gather_bins(lo, hi, centroid_aabb, &tbins[t]);
        });
        Builder::Bins& bins = tbins[0];
        for (unsigned t = 1; t < n_threads; ++t)
            for (int a = 0; a < 3; ++a)
                for (int k = 0; k < BINS; ++k) {
                    bins.box[a][k] = box_combine(bins.box[a][k], tbins[t].box[a][k]);
                    bins.count[a][k] += tbins[t].count[a][k];
                }
        int axis;
        float split_point, cost;
        Builder::best_plane(centroid_aabb, bins, &axis, &split_point, &cost);
        float current_cost = (float)num * box_half_area(upper[ui].box);
        if (cost >= current_cost) continue;  // stays a leaf (bvh.rs:112-115)
        uint32_t partition_point = top.partition_sort(first, num, axis, split_point);
        uint32_t left_count = partition_point - first, right_count = num - left_count;
        if (left_count == 0 || right_count == 0) continue;
        int32_t l = make_upper(first, left_count, depth + 1);
        int32_t r = make_upper(partition_point, right_count, depth + 1);
        upper[ui].left = l;
        upper[ui].right = r;
        todo.push_back(r);
        todo.push_back(l);
    }

    // ---- tasks: serial builds of the small subtrees, largest first
    std::vector<Nodes> built(tasks.size());
    std::vector<uint32_t> order(tasks.size());
    for (uint32_t k = 0; k < order.size(); ++k) order[k] = k;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return upper[tasks[a]].num > upper[tasks[b]].num; });
    std::atomic<uint32_t> next(0);
    parallel_for(n_threads, [&](unsigned) {
        for (;;) {
            uint32_t k = next.fetch_add(1);
            if (k >= order.size()) break;
            const Upper& u = upper[tasks[order[k]]];
            Nodes& nd = built[order[k]];
            Builder bld{p, nd, indices};
            nd.node_box.push_back(u.box);  // local node 0 = the task root (already boxed)
            nd.node_index.push_back(u.first);
            nd.node_num_prim.push_back(u.num);
            bld.build();
        }
    });

    // ---- number the nodes as the serial depth-first build does (bvh.rs:130-143)
    out->node_box.reserve((size_t)2 * n - 1);
    out->node_index.reserve((size_t)2 * n - 1);
    out->node_num_prim.reserve((size_t)2 * n - 1);
    auto append = [&](const Box& bx, uint32_t index, uint32_t num_prim) {
        out->node_box.push_back(bx);
        out->node_index.push_back(index);
        out->node_num_prim.push_back(num_prim);
    };
    append(upper[0].box, upper[0].first, upper[0].num);
    struct Visit {
        int32_t ui;
        uint32_t gid;
    };
    std::vector<Visit> walk;
    walk.push_back({0, 0});
    while (!walk.empty()) {
        Visit v = walk.back();
        walk.pop_back();
        const Upper& u = upper[v.ui];
        if (u.depth > out->max_depth) out->max_depth = u.depth;
        if (u.task >= 0) {
            const Nodes& nd = built[u.task];
            if (u.depth + nd.max_depth > out->max_depth) out->max_depth = u.depth + nd.max_depth;
            if (nd.node_num_prim[0] != 0) continue;  // the task root stayed a leaf
            const uint32_t base = (uint32_t)out->node_index.size();  // local node k >= 1 becomes base + k - 1
            out->node_index[v.gid] = base + nd.node_index[0] - 1;
            out->node_num_prim[v.gid] = 0;
            for (size_t k = 1; k < nd.node_index.size(); ++k) {
                const bool internal = nd.node_num_prim[k] == 0;
                append(nd.node_box[k], internal ? base + nd.node_index[k] - 1 : nd.node_index[k], nd.node_num_prim[k]);
            }
        } else if (u.left >= 0) {
            const uint32_t l = (uint32_t)out->node_index.size();
            out->node_index[v.gid] = l;
            out->node_num_prim[v.gid] = 0;
            append(upper[u.left].box, upper[u.left].first, upper[u.left].num);
            append(upper[u.right].box, upper[u.right].first, upper[u.right].num);
            walk.push_back({u.right, l + 1});  // right: numbered after the whole left subtree
            walk.push_back({u.left, l});
        }
    }
}

}  // namespace

bool build_bvh(const rt_mesh& mesh, BuiltBvh* out) {
    uint32_t n = mesh.n_triangles();
    if (n == 0) {  // the reference underflows `2 * n - 1` (bvh.rs:66)
        set_error("build_bvh: mesh has no triangles");
        return false;
    }
    if ((uint64_t)2 * n - 1 >= 0xFFFFFFFFull) {  // assert!(2 * n - 1 < u32::MAX), bvh.rs:66
        set_error("build_bvh: too many triangles (%u)", n);
        return false;
    }
    Prims p;
    p.box.resize(n);
    p.centroid.resize(n);
    p.centroid_ax.resize(n);
    for (uint32_t t = 0; t < n; ++t) {
        V3 a = mesh.vertices[mesh.triangles[3 * t]];
        V3 b = mesh.vertices[mesh.triangles[3 * t + 1]];
        V3 c = mesh.vertices[mesh.triangles[3 * t + 2]];
        p.box[t] = box_grow(box_grow(box_grow(box_empty(), a), b), c);
        p.centroid[t] = (a + b + c) / 3.0f;
        p.centroid_ax[t] = v3((a.x + b.x + c.x) / 3.0f, (a.y + b.y + c.y) / 3.0f, (a.z + b.z + c.z) / 3.0f);
    }
    BuiltBvh& b = *out;
    b = BuiltBvh();
    b.indices.resize(n);
    for (uint32_t i = 0; i < n; ++i) b.indices[i] = i;

    unsigned n_threads = build_threads();
    if (n_threads <= 1 || n <= 2 * TASK_PRIMS) {  // serial build
        Nodes nodes;
        nodes.node_box.reserve((size_t)2 * n - 1);
        nodes.node_index.reserve((size_t)2 * n - 1);
        nodes.node_num_prim.reserve((size_t)2 * n - 1);
        Builder bld{p, nodes, b.indices};
        bld.new_node(0, n);
        bld.build();
        b.node_box.swap(nodes.node_box);
        b.node_index.swap(nodes.node_index);
        b.node_num_prim.swap(nodes.node_num_prim);
        b.max_depth = nodes.max_depth;
        return true;
    }
    build_parallel(p, n, n_threads, &b);
    return true;
}

}  // namespace rth
