// Device-resident scene layout (what rt_upload_scene builds in HBM) and the intersection routines.
//
// Layout, chosen for gfx950 memory transactions rather than for the reference's pointer graph:
//   * BVH nodes are stored as CHILD PAIRS.  The reference keeps the two children of a node adjacent
//     (bvh.rs:5-12) but the array starts with the singleton root, so every pair straddles a 64-byte
//     line.  Here the root's fields live in the object record and pair p (reference nodes 2p+1,
//     2p+2 of the mesh) sits at byte 64*p: one internal-node visit = one aligned 64-byte fetch.
//   * A child reference is one u32: bit 31 = leaf, low bits = pair index (internal) or first
//     triangle (leaf).  That is all a traversal-stack entry has to hold (4 bytes in LDS).
//   * Triangles are in leaf order, 48 bytes each: (v0, prim_id) (e1, last-in-leaf flag) (e2, 0),
//     e1 = v1-v0 and e2 = v2-v0 precomputed with the same f32 subtraction the reference performs
//     per test (primitives.rs:147-148).  A leaf is walked until the flag, so no count is needed.
#pragma once
#include "../../include/rt_amd.h"
#include "device_math.h"

namespace rtd {

constexpr uint32_t LEAF_BIT = 0x80000000u;

struct __attribute__((aligned(32))) DNode {  // one child: 32 bytes
    float lo[3];
    float hi[3];
    uint32_t ref;       // LEAF_BIT | first triangle (global, leaf order)  or  pair index (global)
    uint32_t num_prim;  // Node.num_prim (bvh.rs:17); informational on the device
};
struct __attribute__((aligned(64))) DPair {
    DNode c[2];
};

struct DObject {
    uint32_t kind;           // enum rt_surface_kind
    uint32_t root_ref;       // RT_SURF_MESH: child reference of the root node
    uint32_t has_transform;
    uint32_t material;
    float surf[12];          // SPHERE c,r / PLANE n,d / TRIANGLE v0,e1,e2 / unused
    float inv_m[9];          // stored inverse transform (object.rs:25)
    float inv_t[3];
    int32_t medium_inside, medium_outside;
    uint32_t has_medium_interface;
    uint32_t pad;
};

struct DScene {
    const DPair* pairs;
    const float4* tris;  // 3 float4 per triangle; lives tri_off bytes after `pairs` in the same allocation
    const DObject* objects;
    const rt_material* materials;
    const rt_light* lights;
    const rt_medium* media;
    uint32_t n_objects, n_lights, n_media, n_materials;
    rt_camera_flat camera;
    int32_t initial_medium;
    uint32_t stack_need;  // deepest traversal stack any mesh of the scene can need
    uint32_t tri_off;     // byte offset of tris from pairs (one allocation, < 4 GiB, 64 B of slack after it)
    // Top level over OBJECTS (flat): world-space bounds of up to 64 groups of 2^cull_shift consecutive objects (one
    // object per group in scenes of at most 64), 8 floats each (lower xyz, upper xyz, 2 unused); bit g of cull_test =
    // group g has bounds worth testing.  A ray that misses group g's bounds cannot hit any object of the group
    // (rt_upload_scene says why), so the traversal leaves them out of its object loop.
    const float* wbox;
    unsigned long long cull_test;
    uint32_t cull_shift;
};

// The same records seen through the CONSTANT address space: a load through these with a wave-uniform index
// is a scalar load (s_load into SGPRs, scalar cache) instead of a vector load that every lane repeats and the
// wave waits a full vector-memory round trip for.  Only for data no kernel writes (objects, wbox).
typedef const __attribute__((address_space(4))) DObject CObject;
typedef const __attribute__((address_space(4))) float CFloat;

struct Visits {
    uint32_t n_int, n_tri;
};

// Per-lane traversal stack: the first STACK_LDS entries in LDS ([level][thread]: a lane's level l
// is at bank (l*BLOCK + tid) % 32 = tid % 32, so divergent depths never conflict), deeper entries
// in a per-lane global overflow strip (only trees deeper than STACK_LDS ever touch it).
// OVERFLOW = false: the caller guarantees the trees of the scene are no deeper than STACK_LDS (rt_upload_scene
// measures them), and the two overflow branches -- scalar work in every push and pop -- are not compiled.
template <int STACK_LDS, int BLOCK, bool OVERFLOW = true>
struct Stack {
    // The LDS half is addressed through an explicit address-space-3 pointer: with two generic
    // pointers the compiler merges pop() into ONE flat_load of a selected address, which sends every
    // LDS pop through the texture-address path and makes it wait on both memory counters.
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    lds_u32* lds;         // &shared[0][threadIdx.x]
    uint32_t* overflow;   // &overflow[global lane], stride = overflow_stride
    uint32_t overflow_stride;
    int sp;
    RT_DEV void set_lds(uint32_t* p) { lds = (lds_u32*)p; }
    RT_DEV void push(uint32_t v) {
        if (!OVERFLOW || sp < STACK_LDS) lds[sp * BLOCK] = v;
        else overflow[(size_t)(sp - STACK_LDS) * overflow_stride] = v;
        ++sp;
    }
    RT_DEV uint32_t pop() {
        --sp;
        uint32_t v;
        if (!OVERFLOW || sp < STACK_LDS) v = lds[sp * BLOCK];
        else v = overflow[(size_t)(sp - STACK_LDS) * overflow_stride];
        return v;
    }
};

// AxisAlignedBoundingBox::hit (primitives.rs:304-321).  t_min only grows and t_max only shrinks
// over the three axes, so the per-axis early return is equivalent to one test at the end.
template <class P>  // P: pointer to 3 floats in any address space
RT_DEV bool aabb_hit(P lo, P hi, V3 o, V3 inv, float t_min, float t_max, float* t_entry) {
    {
        float t0 = (lo[0] - o.x) * inv.x, t1 = (hi[0] - o.x) * inv.x;
        bool neg = inv.x < 0.0f;
        t_min = fmaxf(t_min, neg ? t1 : t0);
        t_max = fminf(t_max, neg ? t0 : t1);
    }
    {
        float t0 = (lo[1] - o.y) * inv.y, t1 = (hi[1] - o.y) * inv.y;
        bool neg = inv.y < 0.0f;
        t_min = fmaxf(t_min, neg ? t1 : t0);
        t_max = fminf(t_max, neg ? t0 : t1);
    }
    {
        float t0 = (lo[2] - o.z) * inv.z, t1 = (hi[2] - o.z) * inv.z;
        bool neg = inv.z < 0.0f;
        t_min = fmaxf(t_min, neg ? t1 : t0);
        t_max = fminf(t_max, neg ? t0 : t1);
    }
    *t_entry = t_min;
    return !(t_max <= t_min);
}

// triangle_intersection (primitives.rs:138-186) with e1, e2 given.  The rejections are evaluated
// as one predicate; each comparison is the reference's, so NaNs fall the same way.
RT_DEV bool tri_hit(V3 v0, V3 e1, V3 e2, V3 o, V3 d, float t_min, float t_max, float* t_out, float* det_out) {
    V3 p = cross(d, e2);
    float det = dot(e1, p);
    float inv_det = 1.0f / det;
    V3 s = o - v0;
    float u = inv_det * dot(s, p);
    V3 q = cross(s, e1);
    float v = inv_det * dot(d, q);
    float t = inv_det * dot(e2, q);
    bool ok = !(fabsf(det) < F32_EPSILON) && !(u < 0.0f || u > 1.0f) && !(v < 0.0f || u + v > 1.0f) &&
              (t > t_min && t < t_max);
    *t_out = t;
    *det_out = det;
    return ok;
}

// Sphere::hit (primitives.rs:70-94 with make_intersection :49-66)
RT_DEV bool sphere_hit(V3 center, float radius, V3 o, V3 d, float t_min, float t_max, float* t_out, V3* n_out,
                       uint32_t* ff_out) {
    V3 r_oc = o - center;
    float a = norm_squared(d);
    float half_b = dot(d, r_oc);
    float c = norm_squared(r_oc) - radius * radius;
    float disc = half_b * half_b - a * c;
    if (disc < 0.0f) return false;
    float sqrt_d = sqrtf(disc);
    float t = (-(half_b + sqrt_d)) / a;
    if (!(t > t_min && t < t_max)) {
        t = (-half_b + sqrt_d) / a;
        if (!(t > t_min && t < t_max)) return false;
    }
    V3 n = ((o + d * t) - center) / radius;
    *t_out = t;
    *n_out = n;
    *ff_out = dot(d, n) < 0.0f;
    return true;
}

// Plane::hit (primitives.rs:229-247); front_face = denom > 0 (the reference's sense, kept)
RT_DEV bool plane_hit(V3 normal, float dd, V3 o, V3 d, float t_min, float t_max, float* t_out, uint32_t* ff_out) {
    float denom = dot(normal, d);
    if (fabsf(denom) <= F32_EPSILON) return false;
    float t = -(dot(normal, o) + dd) / denom;
    if (t < t_max && t > t_min) {
        *t_out = t;
        *ff_out = denom > 0.0f;
        return true;
    }
    return false;
}

}  // namespace rtd
