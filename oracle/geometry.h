// ORACLE -- test infrastructure only (see vec3d.h).
//
// CPU restatement of the reference's geometry and BVH:
//   ray_tracer/src/core/ray.rs, ray_tracer/src/geometry/traits.rs,
//   ray_tracer/src/geometry/primitives.rs, ray_tracer/src/geometry/triangle_mesh.rs:16-81,
//   ray_tracer/src/render/bvh.rs.
// Kept in the reference's own shape (trait objects, indexed mesh behind TriangleRef, the `indices`
// indirection, a heap stack per traversal) so that it can be read side by side with the Rust.
// Two additions the reference lacks, both needed to check the device path: the hit record carries
// the primitive index, and traversal counts the internal nodes and triangles it visits.
#pragma once
#include <memory>
#include <vector>

#include "vec3d.h"

namespace orc {

struct Ray {  // ray.rs:4-18
    Vec3D origin, direction;
    Vec3D at(float t) const { return origin + direction * t; }
};

struct SurfaceIntersection {  // traits.rs:13-17
    float t;
    Vec3D normal;
    bool front_face;
    uint32_t prim;  // addition: index of the triangle inside its mesh (0 for analytic surfaces)
};

struct VisitCounters {  // addition: per-thread traversal counters
    uint64_t n_int = 0;  // internal nodes whose two children were box-tested (bvh.rs:271-274)
    uint64_t n_tri = 0;  // primitive.hit calls inside leaves (bvh.rs:256-257)
};
extern thread_local VisitCounters g_visits;

struct AxisAlignedBoundingBox {  // primitives.rs:250-253
    Vec3D upper, lower;
    static AxisAlignedBoundingBox make(Vec3D a, Vec3D b) { return {vmax(a, b), vmin(a, b)}; }  // :257-261
    static AxisAlignedBoundingBox empty() { return {fill(-INFINITY), fill(INFINITY)}; }        // :264-269
    float half_area() const {  // :282-285
        Vec3D w = upper - lower;
        return w.x * w.y + w.y * w.z + w.z * w.x;
    }
    static AxisAlignedBoundingBox combine(const AxisAlignedBoundingBox& a, const AxisAlignedBoundingBox& b) {
        return {vmax(a.upper, b.upper), vmin(a.lower, b.lower)};  // :288-292
    }
    AxisAlignedBoundingBox grow(Vec3D p) const { return {vmax(upper, p), vmin(lower, p)}; }  // :295-300
    // :304-321.  Returns false for None, otherwise the clipped entry distance.
    bool hit(const Ray& ray, float t_min, float t_max, float* t_out) const {
        for (int axis = 0; axis < 3; ++axis) {
            float d_inv = recip(ray.direction[axis]);
            float t0 = (lower[axis] - ray.origin[axis]) * d_inv;
            float t1 = (upper[axis] - ray.origin[axis]) * d_inv;
            if (d_inv < 0.0f) {
                float tmp = t0;
                t0 = t1;
                t1 = tmp;
            }
            t_min = fmax_(t_min, t0);
            t_max = fmin_(t_max, t1);
            if (t_max <= t_min) return false;
        }
        *t_out = t_min;
        return true;
    }
};
using AABB = AxisAlignedBoundingBox;

struct Surface {  // traits.rs:5-11
    virtual ~Surface() {}
    virtual bool hit(const Ray& ray, float t_min, float t_max, SurfaceIntersection* out) const = 0;
    virtual bool hit_bool(const Ray& ray, float t_min, float t_max) const {
        SurfaceIntersection tmp;
        return hit(ray, t_min, t_max, &tmp);
    }
};

struct Point : Surface {  // primitives.rs:19-23
    bool hit(const Ray&, float, float, SurfaceIntersection*) const override { return false; }
};

struct Sphere : Surface {  // primitives.rs:26-95
    Vec3D center;
    float radius;
    Sphere(Vec3D c, float r) : center(c), radius(r) {}
    float area() const { return 4.0f * PI * radius * radius; }            // :38-40
    Vec3D normal(Vec3D point) const { return (point - center) / radius; }  // :43-45 (mul by recip)
    bool make_intersection(const Ray& ray, float t, float t_min, float t_max, SurfaceIntersection* out) const {
        if (t > t_min && t < t_max) {  // :56-64
            Vec3D n = normal(ray.at(t));
            out->t = t;
            out->normal = n;
            out->front_face = dot(ray.direction, n) < 0.0f;
            out->prim = 0;
            return true;
        }
        return false;
    }
    bool hit(const Ray& ray, float t_min, float t_max, SurfaceIntersection* out) const override {  // :70-94
        Vec3D r_oc = ray.origin - center;
        float a = norm_squared(ray.direction);
        float half_b = dot(ray.direction, r_oc);
        float c = norm_squared(r_oc) - radius * radius;
        float d = half_b * half_b - a * c;
        if (d < 0.0f) return false;
        float sqrt_d = std::sqrt(d);
        float t1 = (-(half_b + sqrt_d)) / a;
        if (make_intersection(ray, t1, t_min, t_max, out)) return true;
        float t2 = (-half_b + sqrt_d) / a;
        if (make_intersection(ray, t2, t_min, t_max, out)) return true;
        return false;
    }
};

// primitives.rs:138-186
inline bool triangle_intersection(Vec3D v0, Vec3D v1, Vec3D v2, const Ray& ray, float t_min, float t_max,
                                  SurfaceIntersection* out) {
    Vec3D e1 = v1 - v0;
    Vec3D e2 = v2 - v0;
    Vec3D ray_cross_e2 = cross(ray.direction, e2);
    float det = dot(e1, ray_cross_e2);
    if (std::fabs(det) < F32_EPSILON) return false;
    float inv_det = 1.0f / det;
    Vec3D s = ray.origin - v0;
    float u = inv_det * dot(s, ray_cross_e2);
    if (u < 0.0f || u > 1.0f) return false;
    Vec3D s_cross_e1 = cross(s, e1);
    float v = inv_det * dot(ray.direction, s_cross_e1);
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = inv_det * dot(e2, s_cross_e1);
    if (t > t_min && t < t_max) {
        out->t = t;
        out->normal = normalise(cross(e1, e2));
        out->front_face = det > 0.0f;
        out->prim = 0;
        return true;
    }
    return false;
}

struct Triangle : Surface {  // primitives.rs:111-192
    Vec3D v0, v1, v2;
    Triangle(Vec3D a, Vec3D b, Vec3D c) : v0(a), v1(b), v2(c) {}
    bool hit(const Ray& ray, float t_min, float t_max, SurfaceIntersection* out) const override {
        return triangle_intersection(v0, v1, v2, ray, t_min, t_max, out);
    }
};

struct Plane : Surface {  // primitives.rs:215-247
    float d;
    Vec3D normal;
    Plane(Vec3D p, Vec3D n) {  // :222-226
        normal = normalise(n);
        d = -dot(p, normal);
    }
    bool hit(const Ray& ray, float t_min, float t_max, SurfaceIntersection* out) const override {
        float denom = dot(normal, ray.direction);
        if (std::fabs(denom) <= F32_EPSILON) return false;
        float t = -(dot(normal, ray.origin) + d) / denom;
        if (t < t_max && t > t_min) {
            out->t = t;
            out->normal = normal;
            out->front_face = denom > 0.0f;
            out->prim = 0;
            return true;
        }
        return false;
    }
};

struct TriangleMesh {  // triangle_mesh.rs:16-19
    std::vector<Vec3D> vertices;
    std::vector<uint32_t> triangles;  // [usize; 3] per triangle
};

struct TriangleRef {  // triangle_mesh.rs:24-64
    const TriangleMesh* mesh;
    uint32_t index;
    bool hit(const Ray& ray, float t_min, float t_max, SurfaceIntersection* out) const {  // :30-37
        const uint32_t* tri = &mesh->triangles[3 * (size_t)index];
        Vec3D v0 = mesh->vertices[tri[0]], v1 = mesh->vertices[tri[1]], v2 = mesh->vertices[tri[2]];
        bool h = triangle_intersection(v0, v1, v2, ray, t_min, t_max, out);
        if (h) out->prim = index;
        return h;
    }
    AABB bounding_box() const {  // :42-48
        const uint32_t* tri = &mesh->triangles[3 * (size_t)index];
        return AABB::empty().grow(mesh->vertices[tri[0]]).grow(mesh->vertices[tri[1]]).grow(mesh->vertices[tri[2]]);
    }
    float centroid_axis(int axis) const {  // :51-57 -- true division by 3.0
        const uint32_t* tri = &mesh->triangles[3 * (size_t)index];
        return (mesh->vertices[tri[0]][axis] + mesh->vertices[tri[1]][axis] + mesh->vertices[tri[2]][axis]) / 3.0f;
    }
    Vec3D centroid() const {  // :60-63 -- Vec3D / f32 = multiply by recip(3.0)
        const uint32_t* tri = &mesh->triangles[3 * (size_t)index];
        return (mesh->vertices[tri[0]] + mesh->vertices[tri[1]] + mesh->vertices[tri[2]]) / 3.0f;
    }
};

struct Node {  // bvh.rs:13-24
    AABB bounding_box;
    uint32_t index;
    uint32_t num_prim;
    bool is_leaf() const { return num_prim > 0; }
};

enum class TraverseMode { Nearest, Any };  // bvh.rs:54-57

struct BoundingVolumeHierarchy : Surface {  // bvh.rs:31-35, T = TriangleRef
    std::vector<TriangleRef> primitives;
    std::vector<size_t> indices;
    std::vector<Node> nodes;

    bool hit(const Ray& ray, float t_min, float t_max, SurfaceIntersection* out) const override {  // :44-46
        return traverse_bvh(ray, t_min, t_max, TraverseMode::Nearest, out);
    }
    bool hit_bool(const Ray& ray, float t_min, float t_max) const override {  // :48-51
        SurfaceIntersection tmp;
        return traverse_bvh(ray, t_min, t_max, TraverseMode::Any, &tmp);
    }

    static std::shared_ptr<BoundingVolumeHierarchy> build(std::vector<TriangleRef> prims);  // :60-81
    Node new_node(uint32_t first_prim, uint32_t num_prim) const;                             // :89-100
    void subdivide_recursive(size_t node_index);                                             // :102-144
    void binned_sah_partition_plane(size_t node_index, int* axis, float* split_point, float* cost) const;  // :147-219
    uint32_t partition_sort(uint32_t first_prim, uint32_t num_prim, int axis, float split_point);          // :221-241
    bool traverse_bvh(const Ray& ray, float t_min, float t_max, TraverseMode mode,
                      SurfaceIntersection* out) const;  // :243-300
};

}  // namespace orc
