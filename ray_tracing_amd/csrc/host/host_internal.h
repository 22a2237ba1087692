// Internal declarations shared by the host-side scene-preparation sources.
#pragma once
#include <cstdarg>
#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/rt_amd.h"
#include "hostmath.h"

struct rt_mesh {  // TriangleMesh (triangle_mesh.rs:16-19)
    std::vector<rth::V3> vertices;
    std::vector<uint32_t> triangles;  // 3 per triangle
    uint32_t n_triangles() const { return (uint32_t)(triangles.size() / 3); }
};

namespace rth {

// rt_last_error() storage (thread local).
void set_error(const char* fmt, ...);
const char* get_error();
// Called from the catch (...) clause of an extern "C" entry point: nothing may unwind across the C ABI.
// std::bad_alloc -> RT_ERR_NOMEM, anything else -> RT_ERR_INVALID; the text goes to rt_last_error().
int on_exception(const char* who) noexcept;

struct BuiltBvh {  // BoundingVolumeHierarchy<TriangleRef> after build (bvh.rs:31-35)
    std::vector<Box> node_box;
    std::vector<uint32_t> node_index;
    std::vector<uint32_t> node_num_prim;
    std::vector<uint32_t> indices;  // leaf order -> original triangle
    uint32_t max_depth = 0;
};

// BoundingVolumeHierarchy::build over TriangleRef primitives (bvh.rs:60-81).
bool build_bvh(const rt_mesh& mesh, BuiltBvh* out);

}  // namespace rth
