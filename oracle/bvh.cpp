// ORACLE -- test infrastructure only (see vec3d.h).
// ray_tracer/src/render/bvh.rs restated: build (binned SAH, :60-241) and traverse_bvh (:243-300).
#include "geometry.h"

namespace orc {

thread_local VisitCounters g_visits;
thread_local bool g_normalise_failed = false;

static inline size_t f32_as_usize(float f) {  // Rust `as usize`: saturating, NaN -> 0
    if (!(f > 0.0f)) return 0;
    if (f >= 18446744073709551615.0f) return SIZE_MAX;
    return (size_t)f;
}

std::shared_ptr<BoundingVolumeHierarchy> BoundingVolumeHierarchy::build(std::vector<TriangleRef> prims) {
    auto bvh = std::make_shared<BoundingVolumeHierarchy>();
    size_t n = prims.size();
    bvh->primitives = std::move(prims);
    bvh->indices.resize(n);
    for (size_t i = 0; i < n; ++i) bvh->indices[i] = i;
    bvh->nodes.reserve(2 * n - 1);
    bvh->nodes.push_back(bvh->new_node(0, (uint32_t)n));
    bvh->subdivide_recursive(0);
    bvh->nodes.shrink_to_fit();
    return bvh;
}

Node BoundingVolumeHierarchy::new_node(uint32_t first_prim, uint32_t num_prim) const {
    AABB bb = AABB::empty();
    for (uint32_t i = first_prim; i < first_prim + num_prim; ++i)
        bb = AABB::combine(bb, primitives[indices[i]].bounding_box());
    return Node{bb, first_prim, num_prim};
}

void BoundingVolumeHierarchy::subdivide_recursive(size_t node_index) {
    int axis;
    float split_point, cost;
    binned_sah_partition_plane(node_index, &axis, &split_point, &cost);

    uint32_t first_prim = nodes[node_index].index;
    uint32_t num_prim = nodes[node_index].num_prim;

    float current_cost = (float)num_prim * nodes[node_index].bounding_box.half_area();
    if (cost >= current_cost) return;

    uint32_t partition_point = partition_sort(first_prim, num_prim, axis, split_point);

    uint32_t left_count = partition_point - first_prim;
    uint32_t right_count = num_prim - left_count;
    if (left_count == 0 || right_count == 0) return;

    size_t left_child_index = nodes.size();
    nodes[node_index].index = (uint32_t)left_child_index;
    nodes[node_index].num_prim = 0;

    Node left_node = new_node(first_prim, left_count);
    Node right_node = new_node(partition_point, right_count);
    nodes.push_back(left_node);
    nodes.push_back(right_node);

    subdivide_recursive(left_child_index);
    subdivide_recursive(left_child_index + 1);
}

void BoundingVolumeHierarchy::binned_sah_partition_plane(size_t node_index, int* axis_out, float* split_out,
                                                         float* cost_out) const {
    const size_t BINS = 100;
    const Node& node = nodes[node_index];
    uint32_t first_prim = node.index;
    uint32_t num_prim = node.num_prim;

    AABB centroid_aabb = AABB::empty();
    for (uint32_t i = first_prim; i < first_prim + num_prim; ++i)
        centroid_aabb = centroid_aabb.grow(primitives[indices[i]].centroid());

    int axis = 0;
    float split_point = 0.0f;
    float cost = INFINITY;
    for (int plane_axis = 0; plane_axis < 3; ++plane_axis) {
        float max = centroid_aabb.upper[plane_axis];
        float min = centroid_aabb.lower[plane_axis];
        if (std::fabs(max - min) < F32_EPSILON) continue;
        float scale = (float)BINS / (max - min);

        std::vector<std::pair<AABB, int32_t>> bins(BINS, {AABB::empty(), 0});
        for (uint32_t i = first_prim; i < first_prim + num_prim; ++i) {
            const TriangleRef& primitive = primitives[indices[i]];
            size_t bin_index = f32_as_usize((primitive.centroid_axis(plane_axis) - min) * scale);
            if (bin_index > BINS - 1) bin_index = BINS - 1;
            bins[bin_index].first = AABB::combine(bins[bin_index].first, primitive.bounding_box());
            bins[bin_index].second += 1;
        }

        std::vector<float> left_area(BINS - 1, 0.0f), right_area(BINS - 1, 0.0f);
        std::vector<int32_t> left_count(BINS - 1, 0), right_count(BINS - 1, 0);

        AABB left_box = AABB::empty(), right_box = AABB::empty();
        int32_t left_sum = 0, right_sum = 0;
        for (size_t i = 0; i < BINS - 1; ++i) {
            left_sum += bins[i].second;
            left_count[i] = left_sum;
            left_box = AABB::combine(left_box, bins[i].first);
            left_area[i] = left_box.half_area();

            right_sum += bins[BINS - 1 - i].second;
            right_count[BINS - 2 - i] = right_sum;
            right_box = AABB::combine(right_box, bins[BINS - 1 - i].first);
            right_area[BINS - 2 - i] = right_box.half_area();
        }

        float scale2 = (max - min) / (float)BINS;
        for (size_t i = 0; i < BINS - 1; ++i) {
            float plane_cost = (float)left_count[i] * left_area[i] + (float)right_count[i] * right_area[i];
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

uint32_t BoundingVolumeHierarchy::partition_sort(uint32_t first_prim, uint32_t num_prim, int axis,
                                                 float split_point) {
    // The reference uses u32 i, j and `j -= 1`; with first_prim == 0 and every primitive on the
    // right that would wrap (panic in debug).  Signed arithmetic here; the caller sees an empty
    // left side and keeps the node a leaf.
    int64_t i = first_prim;
    int64_t j = (int64_t)first_prim + num_prim - 1;
    while (i <= j) {
        if (primitives[indices[(size_t)i]].centroid_axis(axis) < split_point) {
            i += 1;
        } else {
            std::swap(indices[(size_t)i], indices[(size_t)j]);
            j -= 1;
        }
    }
    return (uint32_t)i;
}

bool BoundingVolumeHierarchy::traverse_bvh(const Ray& ray, float t_min, float t_max, TraverseMode mode,
                                           SurfaceIntersection* out) const {
    bool have_hit = false;
    std::vector<const Node*> queue;
    queue.push_back(&nodes[0]);
    while (!queue.empty()) {
        const Node* node = queue.back();
        queue.pop_back();
        if (node->is_leaf()) {
            for (uint32_t i = node->index; i < node->index + node->num_prim; ++i) {
                const TriangleRef& primitive = primitives[indices[i]];
                g_visits.n_tri += 1;
                SurfaceIntersection new_hit;
                if (primitive.hit(ray, t_min, t_max, &new_hit)) {
                    if (mode == TraverseMode::Any) {
                        *out = new_hit;
                        return true;
                    }
                    t_max = new_hit.t;
                    *out = new_hit;
                    have_hit = true;
                }
            }
            continue;
        }

        g_visits.n_int += 1;
        const Node* child1 = &nodes[node->index];
        const Node* child2 = &nodes[(size_t)node->index + 1];
        float t1, t2;
        bool hit1 = child1->bounding_box.hit(ray, t_min, t_max, &t1);
        bool hit2 = child2->bounding_box.hit(ray, t_min, t_max, &t2);

        if (!hit1 && !hit2) continue;
        if (hit1 && !hit2) {
            queue.push_back(child1);
            continue;
        }
        if (!hit1 && hit2) {
            queue.push_back(child2);
            continue;
        }
        if (t1 > t2) std::swap(child1, child2);
        queue.push_back(child2);
        queue.push_back(child1);
    }
    return have_hit;
}

}  // namespace orc
