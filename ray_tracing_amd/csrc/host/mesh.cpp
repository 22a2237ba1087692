// TriangleMesh, binary-STL ingest and Loop subdivision: the host-side mesh sources that feed the
// BVH builder.  Restated from the reference's behaviour (file:line cited per function); the float
// operation order is kept so that generated meshes are reproducible vertex for vertex.
#include <array>
#include <cstdio>
#include <cstring>
#include <unordered_map>

#include <exception>
#include <new>

#include "host_internal.h"

using namespace rth;

namespace rth {
static thread_local std::string g_error;
void set_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
}
const char* get_error() { return g_error.c_str(); }
int on_exception(const char* who) noexcept {
    try {
        throw;
    } catch (const std::bad_alloc&) {
        set_error("%s: out of host memory", who);
        return RT_ERR_NOMEM;
    } catch (const std::exception& e) {
        set_error("%s: %s", who, e.what());
        return RT_ERR_INVALID;
    } catch (...) {
        set_error("%s: unknown exception", who);
        return RT_ERR_INVALID;
    }
}
}  // namespace rth

namespace {

const float PI_F = 3.14159265358979323846f;

// ---------------------------------------------------------------- Loop subdivision
// geometry/subdivision_surface.rs.  None is -1.
struct SDVertex {
    V3 point;
    uint32_t reference_face;
};
struct SDFace {
    uint32_t vertices[3];
    int64_t neighbours[3];
};
inline int nxt(int i) { return (i + 1) % 3; }
inline int prv(int i) { return (i + 2) % 3; }
inline uint64_t edge_key(uint32_t a, uint32_t b) {  // SDEdge::new sorts the pair (:51-59)
    uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
    return ((uint64_t)lo << 32) | hi;
}

struct SubdivisionSurface {
    std::vector<SDVertex> vertices;
    std::vector<SDFace> faces;
    bool bad = false;  // the reference panics ("Logic error!" / "Bad topology!")

    int vertex_index(const SDFace& f, uint32_t v) {  // :77-84
        for (int i = 0; i < 3; ++i)
            if (f.vertices[i] == v) return i;
        bad = true;
        return 0;
    }
    uint32_t next_vertex(uint32_t v, uint32_t face) {  // :330-333
        const SDFace& f = faces[face];
        return f.vertices[nxt(vertex_index(f, v))];
    }
    uint32_t prev_vertex(uint32_t v, uint32_t face) {  // :336-339
        const SDFace& f = faces[face];
        return f.vertices[prv(vertex_index(f, v))];
    }
    int64_t next_face(uint32_t v, uint32_t face) {  // :344-347
        const SDFace& f = faces[face];
        return f.neighbours[vertex_index(f, v)];
    }
    int64_t prev_face(uint32_t v, uint32_t face) {  // :350-353
        const SDFace& f = faces[face];
        return f.neighbours[prv(vertex_index(f, v))];
    }

    // :357-391
    bool vertex_neighbours(uint32_t vertex, std::vector<uint32_t>& nb) {
        nb.clear();
        uint32_t start_face = vertices[vertex].reference_face;
        uint32_t face = start_face;
        nb.push_back(next_vertex(vertex, face));
        for (;;) {
            int64_t nf = next_face(vertex, face);
            if (nf < 0) break;
            if ((uint32_t)nf == start_face) return false;  // closed fan: interior vertex
            uint32_t n = next_vertex(vertex, (uint32_t)nf);
            for (uint32_t o : nb)
                if (o == n) {
                    bad = true;
                    return false;
                }
            nb.push_back(n);
            face = (uint32_t)nf;
            if (bad) return false;
        }
        // boundary: reverse, then walk the other way
        for (size_t a = 0, b = nb.size(); a + 1 < b; ++a, --b) std::swap(nb[a], nb[b - 1]);
        face = start_face;
        nb.push_back(prev_vertex(vertex, face));
        for (;;) {
            int64_t pf = prev_face(vertex, face);
            if (pf < 0) break;
            face = (uint32_t)pf;
            nb.push_back(prev_vertex(vertex, face));
            if (bad || nb.size() > faces.size() * 3 + 3) {
                bad = true;
                break;
            }
        }
        return true;
    }

    // :94-149
    static SubdivisionSurface from_mesh(const rt_mesh& mesh) {
        SubdivisionSurface s;
        s.vertices.reserve(mesh.vertices.size());
        for (const V3& p : mesh.vertices) s.vertices.push_back({p, 0});
        uint32_t nf = mesh.n_triangles();
        s.faces.reserve(nf);
        for (uint32_t f = 0; f < nf; ++f) {
            SDFace face;
            for (int k = 0; k < 3; ++k) {
                face.vertices[k] = mesh.triangles[3 * f + k];
                face.neighbours[k] = -1;
            }
            s.faces.push_back(face);
        }
        for (uint32_t f = 0; f < nf; ++f)
            for (int k = 0; k < 3; ++k) s.vertices[s.faces[f].vertices[k]].reference_face = f;
        std::unordered_map<uint64_t, std::pair<uint32_t, int>> edges;
        edges.reserve((size_t)nf * 2);
        for (uint32_t f = 0; f < nf; ++f) {
            for (int e = 0; e < 3; ++e) {
                uint64_t key = edge_key(s.faces[f].vertices[e], s.faces[f].vertices[nxt(e)]);
                auto it = edges.find(key);
                if (it != edges.end()) {
                    s.faces[f].neighbours[e] = it->second.first;
                    s.faces[it->second.first].neighbours[it->second.second] = f;
                    edges.erase(it);
                } else {
                    edges.emplace(key, std::make_pair(f, e));
                }
            }
        }
        return s;
    }

    static float loop_a(float n) {  // :165 / :300
        float c = 3.0f + 2.0f * cosf(2.0f * PI_F / n);
        return 5.0f / 8.0f - (c * c) / 64.0f;
    }

    // :151-289
    SubdivisionSurface subdivide() {
        SubdivisionSurface out;
        out.vertices.reserve(2 * faces.size());
        std::vector<uint32_t> nb;
        for (uint32_t v = 0; v < vertices.size(); ++v) {
            const SDVertex& vert = vertices[v];
            bool on_boundary = vertex_neighbours(v, nb);
            size_t valence = nb.size();
            V3 new_point;
            if (!on_boundary) {
                float n = (float)valence;
                float a = loop_a(n);
                float alpha = n * (1.0f - a) / a;
                V3 sum = v3(0, 0, 0);
                for (uint32_t i : nb) sum = sum + vertices[i].point;
                new_point = (alpha * vert.point + sum) / (alpha + n);
            } else {
                V3 sum = vertices[nb[0]].point + vertices[nb[valence - 1]].point;
                new_point = (6.0f * vert.point + sum) / 8.0f;
            }
            int i = vertex_index(faces[vert.reference_face], v);
            out.vertices.push_back({new_point, 4 * vert.reference_face + (uint32_t)i});
            if (bad) {
                out.bad = true;
                return out;
            }
        }

        std::unordered_map<uint64_t, uint32_t> edges;
        edges.reserve(faces.size() * 2);
        std::vector<uint32_t> nb2;
        for (uint32_t f = 0; f < faces.size(); ++f) {
            const SDFace& face = faces[f];
            for (int i = 0; i < 3; ++i) {
                uint32_t ea = face.vertices[i], eb = face.vertices[nxt(i)];
                uint64_t key = edge_key(ea, eb);
                if (edges.count(key)) continue;
                edges.emplace(key, (uint32_t)out.vertices.size());
                uint32_t v1 = ea < eb ? ea : eb, v2 = ea < eb ? eb : ea;  // edge.v1 <= edge.v2
                uint32_t reference_face = 4 * f + 3;
                if (face.neighbours[i] >= 0) {
                    const SDFace& face2 = faces[(size_t)face.neighbours[i]];
                    int i2 = vertex_index(face2, face.vertices[i]);
                    V3 point = (3.0f * vertices[v1].point + 3.0f * vertices[v2].point +
                                vertices[face.vertices[prv(i)]].point +
                                vertices[face2.vertices[nxt(i2)]].point) /
                               8.0f;
                    out.vertices.push_back({point, reference_face});
                } else {
                    vertex_neighbours(v1, nb);
                    vertex_neighbours(v2, nb2);
                    float w1 = nb.size() == 4 ? 5.0f : 3.0f;
                    float w2 = nb2.size() == 4 ? 5.0f : 3.0f;
                    V3 point = (w1 * vertices[v1].point + w2 * vertices[v2].point) / (w1 + w2);
                    out.vertices.push_back({point, reference_face});
                }
            }
        }

        out.faces.resize(4 * faces.size());
        for (SDFace& nf : out.faces)
            for (int k = 0; k < 3; ++k) {
                nf.vertices[k] = 0;
                nf.neighbours[k] = -1;
            }
        for (uint32_t f = 0; f < faces.size(); ++f) {
            const SDFace& face = faces[f];
            auto child = [f](int i) { return (size_t)4 * f + i; };
            for (int i = 0; i < 3; ++i) {
                out.faces[child(i)].vertices[i] = face.vertices[i];
                uint32_t vertex = edges[edge_key(face.vertices[i], face.vertices[nxt(i)])];
                out.faces[child(i)].vertices[nxt(i)] = vertex;
                out.faces[child(nxt(i))].vertices[i] = vertex;
                out.faces[child(3)].vertices[i] = vertex;

                out.faces[child(i)].neighbours[nxt(i)] = (int64_t)child(3);
                out.faces[child(3)].neighbours[prv(i)] = (int64_t)child(i);

                if (face.neighbours[i] >= 0) {
                    uint32_t f2 = (uint32_t)face.neighbours[i];
                    int f2_i = vertex_index(faces[f2], face.vertices[i]);
                    out.faces[child(i)].neighbours[i] = (int64_t)4 * f2 + f2_i;
                }
                if (face.neighbours[prv(i)] >= 0) {
                    uint32_t f2 = (uint32_t)face.neighbours[prv(i)];
                    int f2_i = vertex_index(faces[f2], face.vertices[i]);
                    out.faces[child(i)].neighbours[prv(i)] = (int64_t)4 * f2 + f2_i;
                }
            }
        }
        out.bad = bad;
        return out;
    }

    // :291-318
    void push_to_limit_surface() {
        std::vector<V3> limit(vertices.size());
        std::vector<uint32_t> nb;
        for (uint32_t v = 0; v < vertices.size(); ++v) {
            bool on_boundary = vertex_neighbours(v, nb);
            if (bad) return;
            size_t valence = nb.size();
            V3 point = vertices[v].point;
            if (!on_boundary) {
                float n = (float)valence;
                float a = loop_a(n);
                float omega = 3.0f * n / (8.0f * a);
                V3 sum = v3(0, 0, 0);
                for (uint32_t i : nb) sum = sum + vertices[i].point;
                limit[v] = (omega * point + sum) / (omega + n);
            } else {
                float weight = valence == 4 ? 4.0f : 3.0f;
                V3 sum = vertices[nb[0]].point + vertices[nb[valence - 1]].point;
                limit[v] = (weight * point + sum) / (weight + 2.0f);
            }
        }
        for (uint32_t v = 0; v < vertices.size(); ++v) vertices[v].point = limit[v];
    }
};

// ---------------------------------------------------------------- binary STL (render/stl.rs)
struct KeyHash {
    size_t operator()(const std::array<uint32_t, 3>& k) const {
        uint64_t h = 0x9E3779B97F4A7C15ull;
        for (uint32_t w : k) {
            h ^= w + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
        }
        return (size_t)h;
    }
};

}  // namespace

extern "C" {

int rt_mesh_create(const float* vertices, uint32_t n_vertices, const uint32_t* triangles,
                   uint32_t n_triangles, rt_mesh** out) try {
    if (!out || (!vertices && n_vertices) || (!triangles && n_triangles)) {
        set_error("rt_mesh_create: null argument");
        return RT_ERR_INVALID;
    }
    for (size_t i = 0; i < (size_t)n_triangles * 3; ++i)
        if (triangles[i] >= n_vertices) {
            set_error("rt_mesh_create: triangle %zu references vertex %u >= %u", i / 3, triangles[i],
                      n_vertices);
            return RT_ERR_INVALID;  // the reference would panic on the first lookup
        }
    rt_mesh* m = new rt_mesh;
    m->vertices.resize(n_vertices);
    for (uint32_t i = 0; i < n_vertices; ++i)
        m->vertices[i] = v3(vertices[3 * i], vertices[3 * i + 1], vertices[3 * i + 2]);
    m->triangles.assign(triangles, triangles + (size_t)n_triangles * 3);
    *out = m;
    return RT_OK;
} catch (...) {
    return rth::on_exception("rt_mesh_create");
}

// stl.rs:8-143.  80-byte header, u32 count, per triangle: normal (ignored), 3 vertices, u16
// attribute; trailing bytes are an error (:89-99); vertices unified on their bit patterns in
// first-seen order (:114-143).
int rt_mesh_read_stl(const char* path, rt_mesh** out) try {
    if (!path || !out) {
        set_error("rt_mesh_read_stl: null argument");
        return RT_ERR_INVALID;
    }
    FILE* f = fopen(path, "rb");
    if (!f) {
        set_error("rt_mesh_read_stl: cannot open %s", path);
        return RT_ERR_IO;
    }
    unsigned char header[84];
    if (fread(header, 1, 84, f) != 84) {
        fclose(f);
        set_error("rt_mesh_read_stl: %s: short header", path);
        return RT_ERR_IO;
    }
    uint32_t n;
    memcpy(&n, header + 80, 4);  // little-endian host
    // the count comes from an untrusted header: check it against the file's size before reserving anything
    long here = ftell(f), size = -1;
    if (here >= 0 && fseek(f, 0, SEEK_END) == 0) size = ftell(f);
    if (here < 0 || size < 0 || fseek(f, here, SEEK_SET) != 0) {
        fclose(f);
        set_error("rt_mesh_read_stl: %s: cannot determine the file size", path);
        return RT_ERR_IO;
    }
    if ((uint64_t)(size - 84) / 50 < n) {
        fclose(f);
        set_error("rt_mesh_read_stl: %s: truncated (header announces %u triangles, the file holds %llu)", path, n,
                  (unsigned long long)((size - 84) / 50));
        return RT_ERR_IO;
    }
    rt_mesh* m = new rt_mesh;
    m->triangles.reserve((size_t)n * 3);
    m->vertices.reserve(n / 2);
    std::unordered_map<std::array<uint32_t, 3>, uint32_t, KeyHash> vertex_map;
    vertex_map.reserve(n / 2);
    for (uint32_t t = 0; t < n; ++t) {
        unsigned char rec[50];
        if (fread(rec, 1, 50, f) != 50) {
            fclose(f);
            delete m;
            set_error("rt_mesh_read_stl: %s: truncated at triangle %u of %u", path, t, n);
            return RT_ERR_IO;
        }
        for (int k = 0; k < 3; ++k) {
            std::array<uint32_t, 3> bits;
            memcpy(bits.data(), rec + 12 + 12 * k, 12);
            auto it = vertex_map.find(bits);
            if (it != vertex_map.end()) {
                m->triangles.push_back(it->second);
            } else {
                uint32_t idx = (uint32_t)m->vertices.size();
                vertex_map.emplace(bits, idx);
                V3 p;
                memcpy(&p, bits.data(), 12);
                m->vertices.push_back(p);
                m->triangles.push_back(idx);
            }
        }
    }
    unsigned char extra;
    if (fread(&extra, 1, 1, f) != 0) {
        fclose(f);
        delete m;
        set_error("rt_mesh_read_stl: %s: unexpected data after last triangle", path);
        return RT_ERR_IO;
    }
    fclose(f);
    *out = m;
    return RT_OK;
} catch (...) {
    return rth::on_exception("rt_mesh_read_stl");
}

int rt_mesh_loop_subdivide(const rt_mesh* mesh, uint32_t divisions, rt_mesh** out) try {
    if (!mesh || !out) {
        set_error("rt_mesh_loop_subdivide: null argument");
        return RT_ERR_INVALID;
    }
    SubdivisionSurface s = SubdivisionSurface::from_mesh(*mesh);
    for (uint32_t d = 0; d < divisions && !s.bad; ++d) s = s.subdivide();
    if (!s.bad) s.push_to_limit_surface();
    if (s.bad) {
        set_error("rt_mesh_loop_subdivide: bad topology (the reference panics here)");
        return RT_ERR_INVALID;
    }
    rt_mesh* m = new rt_mesh;  // to_triangle_mesh, :320-327
    m->vertices.reserve(s.vertices.size());
    for (const SDVertex& v : s.vertices) m->vertices.push_back(v.point);
    m->triangles.reserve(s.faces.size() * 3);
    for (const SDFace& f : s.faces)
        for (int k = 0; k < 3; ++k) m->triangles.push_back(f.vertices[k]);
    *out = m;
    return RT_OK;
} catch (...) {
    return rth::on_exception("rt_mesh_loop_subdivide");
}

int rt_mesh_apply_transform(rt_mesh* mesh, const float matrix[9], const float translation[3]) {
    if (!mesh || !matrix || !translation) {
        set_error("rt_mesh_apply_transform: null argument");
        return RT_ERR_INVALID;
    }
    M3 m;
    memcpy(m.d, matrix, sizeof m.d);
    V3 t = v3(translation[0], translation[1], translation[2]);
    for (V3& v : mesh->vertices) v = mat_mul(m, v) + t;  // affine_transform.rs:39-41
    return RT_OK;
}

int rt_mesh_volume_centroid(const rt_mesh* mesh, float out[3]) {  // triangle_mesh.rs:103-118
    if (!mesh || !out) {
        set_error("rt_mesh_volume_centroid: null argument");
        return RT_ERR_INVALID;
    }
    float total_volume = 0.0f;
    V3 vc = v3(0, 0, 0);
    for (uint32_t t = 0; t < mesh->n_triangles(); ++t) {
        V3 a = mesh->vertices[mesh->triangles[3 * t]];
        V3 b = mesh->vertices[mesh->triangles[3 * t + 1]];
        V3 c = mesh->vertices[mesh->triangles[3 * t + 2]];
        V3 centroid = (((v3(0, 0, 0) + a) + b) + c) / 4.0f;
        float signed_volume = dot(a, cross(b, c)) / 6.0f;
        total_volume += signed_volume;
        vc = vc + signed_volume * centroid;
    }
    V3 r = vc / total_volume;
    out[0] = r.x;
    out[1] = r.y;
    out[2] = r.z;
    return RT_OK;
}

int rt_mesh_counts(const rt_mesh* mesh, uint32_t* n_vertices, uint32_t* n_triangles) {
    if (!mesh) {
        set_error("rt_mesh_counts: null mesh");
        return RT_ERR_INVALID;
    }
    if (n_vertices) *n_vertices = (uint32_t)mesh->vertices.size();
    if (n_triangles) *n_triangles = mesh->n_triangles();
    return RT_OK;
}

int rt_mesh_copy_out(const rt_mesh* mesh, float* vertices, uint32_t* triangles) try {
    if (!mesh) {
        set_error("rt_mesh_copy_out: null mesh");
        return RT_ERR_INVALID;
    }
    if (vertices) memcpy(vertices, mesh->vertices.data(), mesh->vertices.size() * sizeof(V3));
    if (triangles)
        memcpy(triangles, mesh->triangles.data(), mesh->triangles.size() * sizeof(uint32_t));
    return RT_OK;
} catch (...) {
    return rth::on_exception("rt_mesh_copy_out");
}

void rt_mesh_destroy(rt_mesh* mesh) { delete mesh; }

const char* rt_last_error(void) { return get_error(); }

}  // extern "C"
