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
#include <algorithm>

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

struct Builder {
    const Prims& p;
    BuiltBvh& b;

    void new_node(uint32_t first_prim, uint32_t num_prim) {  // bvh.rs:89-100
        Box bb = box_empty();
        for (uint32_t i = first_prim; i < first_prim + num_prim; ++i)
            bb = box_combine(bb, p.box[b.indices[i]]);
        b.node_box.push_back(bb);
        b.node_index.push_back(first_prim);
        b.node_num_prim.push_back(num_prim);
    }

    // bvh.rs:147-219
    void binned_sah_partition_plane(uint32_t node, int* axis_out, float* split_out, float* cost_out) {
        uint32_t first_prim = b.node_index[node];
        uint32_t num_prim = b.node_num_prim[node];
        Box centroid_aabb = box_empty();
        for (uint32_t i = first_prim; i < first_prim + num_prim; ++i)
            centroid_aabb = box_grow(centroid_aabb, p.centroid[b.indices[i]]);

        int axis = 0;
        float split_point = 0.0f;
        float cost = INFINITY;
        Box bin_box[BINS];
        int32_t bin_count[BINS];
        float left_area[BINS - 1], right_area[BINS - 1];
        int32_t left_count[BINS - 1], right_count[BINS - 1];
        for (int plane_axis = 0; plane_axis < 3; ++plane_axis) {
            float max = centroid_aabb.upper[plane_axis];
            float min = centroid_aabb.lower[plane_axis];
            if (std::fabs(max - min) < F32_EPSILON) continue;
            float scale = (float)BINS / (max - min);
            for (int k = 0; k < BINS; ++k) {
                bin_box[k] = box_empty();
                bin_count[k] = 0;
            }
            for (uint32_t i = first_prim; i < first_prim + num_prim; ++i) {
                uint32_t prim = b.indices[i];
                uint32_t bin = float_to_usize_saturating((p.centroid_ax[prim][plane_axis] - min) * scale);
                if (bin > BINS - 1) bin = BINS - 1;
                bin_box[bin] = box_combine(bin_box[bin], p.box[prim]);
                bin_count[bin] += 1;
            }
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

    // bvh.rs:221-241.  Signed indices: where the reference's `j -= 1` would wrap below zero
    // (every primitive on the right with first_prim == 0) this returns an empty left side, which
    // the caller treats like any failed partition.
    uint32_t partition_sort(uint32_t first_prim, uint32_t num_prim, int axis, float split_point) {
        int64_t i = first_prim;
        int64_t j = (int64_t)first_prim + num_prim - 1;
        while (i <= j) {
            if (p.centroid_ax[b.indices[(size_t)i]][axis] < split_point) {
                i += 1;
            } else {
                std::swap(b.indices[(size_t)i], b.indices[(size_t)j]);
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
    b.node_box.reserve((size_t)2 * n - 1);
    b.node_index.reserve((size_t)2 * n - 1);
    b.node_num_prim.reserve((size_t)2 * n - 1);
    Builder bld{p, b};
    bld.new_node(0, n);
    bld.build();
    return true;
}

}  // namespace rth
