// ORACLE -- test infrastructure only (see vec3d.h).  C interface of the CPU restatement, loaded by
// tests/ (ctypes), __graft_entry__.smoke() and the cpu_baseline leg of bench.py.
//
// The oracle builds its OWN scene from a description (its own BVH with the faithful builder of
// bvh.cpp), so the device path and the oracle share nothing but the input arrays.
#include <atomic>
#include <chrono>
#include <cstring>
#include <string>
#include <thread>

#include "../include/rt_amd.h"
#include "engine.h"

using namespace orc;

struct orc_mesh {
    TriangleMesh mesh;
};

struct orc_scene {
    Engine engine;
    std::vector<std::shared_ptr<BoundingVolumeHierarchy>> bvhs;  // per object (null for analytic)
    std::string error;
};

static thread_local std::string g_err;

static Vec3D v3(const float* p) { return vec(p[0], p[1], p[2]); }

extern "C" {

const char* orc_last_error(void) { return g_err.c_str(); }

orc_mesh* orc_mesh_create(const float* vertices, uint32_t nv, const uint32_t* triangles, uint32_t nt) {
    orc_mesh* m = new orc_mesh;
    m->mesh.vertices.resize(nv);
    for (uint32_t i = 0; i < nv; ++i) m->mesh.vertices[i] = v3(vertices + 3 * i);
    m->mesh.triangles.assign(triangles, triangles + (size_t)nt * 3);
    return m;
}
void orc_mesh_destroy(orc_mesh* m) { delete m; }

static std::shared_ptr<Medium> make_medium(const rt_medium& d) {
    if (d.kind == RT_MEDIUM_HOMOGENEOUS)
        return std::make_shared<HomogeneousMedium>(v3(d.sigma_a), v3(d.sigma_s), d.g, v3(d.emission));
    auto c = std::make_shared<CheckeredMedium>();
    c->cell_size = d.cell_size;
    c->sigma_a_white = d.sigma_a[0];
    c->sigma_s_white = d.sigma_s[0];
    c->emission_white = v3(d.emission);
    c->sigma_a_black = d.sigma_a_black;
    c->sigma_s_black = d.sigma_s_black;
    c->emission_black = v3(d.emission_black);
    c->g = d.g;
    return c;
}

// Engine::new over a description.  object_meshes[i] is the oracle mesh of object i (NULL unless
// the object is a mesh surface); desc->objects[i].mesh (the product's opaque handle) is ignored.
orc_scene* orc_scene_create(const rt_scene_desc* desc, const orc_mesh* const* object_meshes) {
    orc_scene* s = new orc_scene;
    Engine& e = s->engine;
    std::vector<std::shared_ptr<BSDF>> materials;
    for (uint32_t i = 0; i < desc->n_materials; ++i) {
        const rt_material& m = desc->materials[i];
        if (m.kind == RT_MAT_DIFFUSE) materials.push_back(std::make_shared<Diffuse>(v3(m.a)));
        else if (m.kind == RT_MAT_DIELECTRIC) materials.push_back(std::make_shared<Dielectric>(m.a[0]));
        else materials.push_back(std::make_shared<Conductor>(v3(m.a), v3(m.b)));
    }
    for (uint32_t i = 0; i < desc->n_media; ++i) e.media.push_back(make_medium(desc->media[i]));
    for (uint32_t i = 0; i < desc->n_lights; ++i) {
        const rt_light& l = desc->lights[i];
        if (l.kind == RT_LIGHT_POINT) e.lights.push_back(std::make_shared<PointLight>(v3(l.pos), v3(l.intensity)));
        else if (l.kind == RT_LIGHT_SPOT)
            e.lights.push_back(std::make_shared<Spotlight>(v3(l.pos), v3(l.dir), l.scalar, v3(l.intensity)));
        else e.lights.push_back(std::make_shared<AreaLight>(v3(l.pos), l.scalar, v3(l.intensity)));
    }
    std::vector<std::pair<const orc_mesh*, std::shared_ptr<BoundingVolumeHierarchy>>> built;
    for (uint32_t i = 0; i < desc->n_objects; ++i) {
        const rt_object_desc& d = desc->objects[i];
        Object o;
        std::shared_ptr<BoundingVolumeHierarchy> bvh;
        switch (d.surface_kind) {
            case RT_SURF_MESH: {
                const orc_mesh* om = object_meshes ? object_meshes[i] : nullptr;
                if (!om) {
                    g_err = "orc_scene_create: mesh object without oracle mesh";
                    delete s;
                    return nullptr;
                }
                for (auto& b : built)
                    if (b.first == om) bvh = b.second;
                if (!bvh) {  // TriangleMesh::build_bvh, triangle_mesh.rs:73-81
                    std::vector<TriangleRef> prims;
                    uint32_t nt = (uint32_t)(om->mesh.triangles.size() / 3);
                    for (uint32_t t = 0; t < nt; ++t) prims.push_back(TriangleRef{&om->mesh, t});
                    bvh = BoundingVolumeHierarchy::build(std::move(prims));
                    built.push_back({om, bvh});
                }
                o.surface = bvh;
                break;
            }
            case RT_SURF_SPHERE: o.surface = std::make_shared<Sphere>(v3(d.surf), d.surf[3]); break;
            case RT_SURF_PLANE: o.surface = std::make_shared<Plane>(v3(d.surf), v3(d.surf + 3)); break;
            case RT_SURF_TRIANGLE:
                o.surface = std::make_shared<Triangle>(v3(d.surf), v3(d.surf + 3), v3(d.surf + 6));
                break;
            default: o.surface = std::make_shared<Point>(); break;
        }
        s->bvhs.push_back(bvh);
        o.material = materials.at(d.material);
        if (d.has_transform) {  // Object::new, object.rs:17-29
            AffineTransform at;
            memcpy(at.matrix.data, d.matrix, sizeof at.matrix.data);
            at.translation = v3(d.translation);
            if (!at.inverse(&o.affine_transform)) {
                g_err = "orc_scene_create: singular transform";
                delete s;
                return nullptr;
            }
            o.has_transform = true;
        }
        if (d.has_medium_interface) {
            o.has_mediuminterface = true;
            o.medium_inside_id = d.medium_inside;
            o.medium_outside_id = d.medium_outside;
            if (d.medium_inside >= 0) o.mediuminterface.inside = e.media.at(d.medium_inside);
            if (d.medium_outside >= 0) o.mediuminterface.outside = e.media.at(d.medium_outside);
        }
        e.objects.push_back(std::move(o));
    }
    e.camera = Camera(v3(desc->camera.origin), v3(desc->camera.direction), desc->camera.aspect_ratio);
    if (desc->initial_medium == RT_MEDIUM_AUTO) {
        e.set_initial_medium();
    } else if (desc->initial_medium >= 0) {
        e.initial_medium = e.media.at(desc->initial_medium);
        e.initial_medium_id = desc->initial_medium;
    }
    return s;
}
void orc_scene_destroy(orc_scene* s) { delete s; }
int32_t orc_initial_medium(const orc_scene* s) { return s->engine.initial_medium_id; }

// The oracle's own BVH of object `obj` (node arrays in the reference's order + indices).
int orc_bvh_sizes(const orc_scene* s, uint32_t obj, uint32_t* n_nodes, uint32_t* n_prims) {
    if (obj >= s->bvhs.size() || !s->bvhs[obj]) return -1;
    *n_nodes = (uint32_t)s->bvhs[obj]->nodes.size();
    *n_prims = (uint32_t)s->bvhs[obj]->indices.size();
    return 0;
}
int orc_bvh_copy(const orc_scene* s, uint32_t obj, float* upper, float* lower, uint32_t* index, uint32_t* num_prim,
                 uint32_t* indices) {
    if (obj >= s->bvhs.size() || !s->bvhs[obj]) return -1;
    const BoundingVolumeHierarchy& b = *s->bvhs[obj];
    for (size_t k = 0; k < b.nodes.size(); ++k) {
        for (int a = 0; a < 3; ++a) {
            upper[3 * k + a] = b.nodes[k].bounding_box.upper[a];
            lower[3 * k + a] = b.nodes[k].bounding_box.lower[a];
        }
        index[k] = b.nodes[k].index;
        num_prim[k] = b.nodes[k].num_prim;
    }
    for (size_t i = 0; i < b.indices.size(); ++i) indices[i] = (uint32_t)b.indices[i];
    return 0;
}

// Engine::hit on a batch (path_integrator.rs:88-97).  All outputs optional except t.
int orc_trace_closest(const orc_scene* s, const float* ox, const float* oy, const float* oz, const float* dx,
                      const float* dy, const float* dz, size_t n, float t_min, float t_max, float* t, uint32_t* prim,
                      uint32_t* obj, float* nx, float* ny, float* nz, uint8_t* front_face, uint32_t* n_int,
                      uint32_t* n_tri) {
    const Engine& e = s->engine;
    for (size_t i = 0; i < n; ++i) {
        Ray ray{vec(ox[i], oy[i], oz[i]), vec(dx[i], dy[i], dz[i])};
        g_visits = VisitCounters();
        SurfaceIntersection intr;
        const Object* o = nullptr;
        bool h = e.hit(ray, t_min, t_max, &intr, &o);
        t[i] = h ? intr.t : INFINITY;
        if (prim) prim[i] = h ? intr.prim : RT_MISS;
        if (obj) obj[i] = h ? (uint32_t)(o - e.objects.data()) : RT_MISS;
        if (nx) nx[i] = h ? intr.normal.x : 0.0f;
        if (ny) ny[i] = h ? intr.normal.y : 0.0f;
        if (nz) nz[i] = h ? intr.normal.z : 0.0f;
        if (front_face) front_face[i] = h ? (intr.front_face ? 1 : 0) : 0;
        if (n_int) n_int[i] = (uint32_t)g_visits.n_int;
        if (n_tri) n_tri[i] = (uint32_t)g_visits.n_tri;
    }
    return 0;
}

// Engine::hit_bool on a batch (path_integrator.rs:100-104).
int orc_trace_any(const orc_scene* s, const float* ox, const float* oy, const float* oz, const float* dx,
                  const float* dy, const float* dz, size_t n, float t_min, float t_max, uint8_t* occluded,
                  uint32_t* n_int, uint32_t* n_tri) {
    const Engine& e = s->engine;
    for (size_t i = 0; i < n; ++i) {
        Ray ray{vec(ox[i], oy[i], oz[i]), vec(dx[i], dy[i], dz[i])};
        g_visits = VisitCounters();
        occluded[i] = e.hit_bool(ray, t_min, t_max) ? 1 : 0;
        if (n_int) n_int[i] = (uint32_t)g_visits.n_int;
        if (n_tri) n_tri[i] = (uint32_t)g_visits.n_tri;
    }
    return 0;
}

// Camera::ray for pixel-grid positions (camera.rs:60-63): ray k <- (u[k], v[k]).
int orc_camera_rays(const orc_scene* s, const float* u, const float* v, size_t n, float* ox, float* oy, float* oz,
                    float* dx, float* dy, float* dz) {
    for (size_t k = 0; k < n; ++k) {
        Ray r = s->engine.camera.ray(u[k], v[k]);
        ox[k] = r.origin.x; oy[k] = r.origin.y; oz[k] = r.origin.z;
        dx[k] = r.direction.x; dy[k] = r.direction.y; dz[k] = r.direction.z;
    }
    return 0;
}

struct orc_stats {
    uint64_t rays_closest, rays_shadow, iterations, nodes_visited, tris_tested;
    double seconds;
    uint32_t threads;
};

// Engine::render (path_integrator.rs:278-318) without the PPM text: rgb_sum[(j*w+i)*3..] = sum
// over samples [spp_first, spp_first+spp_count) of ray_colour.  Rows [row_begin,row_end) only
// (for bounded baseline samples); rows are dealt to `threads` workers, each pixel is independent.
int orc_render(const orc_scene* s, uint32_t width, uint32_t height, uint32_t spp_first, uint32_t spp_count,
               uint32_t depth, uint64_t seed, uint32_t row_begin, uint32_t row_end, uint32_t threads,
               float* rgb_sum, orc_stats* stats) {
    if (threads == 0) threads = 1;
    if (row_end > height) row_end = height;
    std::atomic<uint32_t> next_row(row_begin);
    std::vector<RayCounters> rcs(threads);
    std::vector<VisitCounters> vcs(threads);
    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&](uint32_t tid) {
        g_visits = VisitCounters();
        RayCounters rc;
        for (;;) {
            uint32_t j = next_row.fetch_add(1);
            if (j >= row_end) break;
            for (uint32_t i = 0; i < width; ++i) {
                Vec3D c = s->engine.render_pixel(i, j, width, height, spp_first, spp_count, depth, seed, &rc);
                float* o = rgb_sum + ((size_t)j * width + i) * 3;
                o[0] = c.x; o[1] = c.y; o[2] = c.z;
            }
        }
        rcs[tid] = rc;
        vcs[tid] = g_visits;
    };
    if (threads == 1) {
        worker(0);
    } else {
        std::vector<std::thread> pool;
        for (uint32_t t = 0; t < threads; ++t) pool.emplace_back(worker, t);
        for (auto& th : pool) th.join();
    }
    auto t1 = std::chrono::steady_clock::now();
    if (stats) {
        memset(stats, 0, sizeof *stats);
        for (uint32_t t = 0; t < threads; ++t) {
            stats->rays_closest += rcs[t].closest;
            stats->rays_shadow += rcs[t].shadow;
            stats->iterations += rcs[t].iterations;
            stats->nodes_visited += vcs[t].n_int;
            stats->tris_tested += vcs[t].n_tri;
        }
        stats->seconds = std::chrono::duration<double>(t1 - t0).count();
        stats->threads = threads;
    }
    return 0;
}

// The same loops for an arbitrary list of pixels (pixel = j*width + i): out[k*3..] = sum over the
// samples of that pixel.  Used to check full-size GPU frames on a random subset of pixels.
int orc_render_pixels(const orc_scene* s, uint32_t width, uint32_t height, uint32_t spp_first, uint32_t spp_count,
                      uint32_t depth, uint64_t seed, const uint32_t* pixels, size_t n_pixels, uint32_t threads,
                      float* out, orc_stats* stats) {
    if (threads == 0) threads = 1;
    std::atomic<size_t> next(0);
    std::vector<RayCounters> rcs(threads);
    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&](uint32_t tid) {
        RayCounters rc;
        for (;;) {
            size_t k = next.fetch_add(1);
            if (k >= n_pixels) break;
            uint32_t i = pixels[k] % width, j = pixels[k] / width;
            Vec3D c = s->engine.render_pixel(i, j, width, height, spp_first, spp_count, depth, seed, &rc);
            out[3 * k] = c.x; out[3 * k + 1] = c.y; out[3 * k + 2] = c.z;
        }
        rcs[tid] = rc;
    };
    if (threads == 1) worker(0);
    else {
        std::vector<std::thread> pool;
        for (uint32_t t = 0; t < threads; ++t) pool.emplace_back(worker, t);
        for (auto& th : pool) th.join();
    }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        for (auto& rc : rcs) {
            stats->rays_closest += rc.closest;
            stats->rays_shadow += rc.shadow;
            stats->iterations += rc.iterations;
        }
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        stats->threads = threads;
    }
    return 0;
}

// Engine::ray_colour (path_integrator.rs:138-276) for a list of rays: out[k*3..] = sum over the samples of
// ray_colour(ray k, depth) with the random numbers of (pixel index k, sample s); the ray is taken as given.
int orc_ray_colour(const orc_scene* s, const float* ox, const float* oy, const float* oz, const float* dx,
                   const float* dy, const float* dz, size_t n, uint32_t spp_first, uint32_t spp_count, uint32_t depth,
                   uint64_t seed, float* out, orc_stats* stats) {
    RayCounters rc;
    g_visits = VisitCounters();
    for (size_t k = 0; k < n; ++k) {
        Vec3D colour = ZERO;
        for (uint32_t sm = spp_first; sm < spp_first + spp_count; ++sm) {
            Rng rng(seed, (uint32_t)k, sm);
            colour += s->engine.ray_colour(Ray{Vec3D{ox[k], oy[k], oz[k]}, Vec3D{dx[k], dy[k], dz[k]}}, depth, rng, &rc);
        }
        out[3 * k] = colour.x; out[3 * k + 1] = colour.y; out[3 * k + 2] = colour.z;
    }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->rays_closest = rc.closest;
        stats->rays_shadow = rc.shadow;
        stats->iterations = rc.iterations;
        stats->nodes_visited = g_visits.n_int;
        stats->tris_tested = g_visits.n_tri;
        stats->threads = 1;
    }
    return 0;
}

// ---------------------------------------------------------------- unit probes for the golden tests
// AxisAlignedBoundingBox::new + hit (primitives.rs:257-261, :304-321). Returns 1 and *t for Some(t).
int orc_aabb_hit(const float a[3], const float b[3], const float o[3], const float d[3], float t_min, float t_max,
                 float* t) {
    AABB box = AABB::make(v3(a), v3(b));
    return box.hit(Ray{v3(o), v3(d)}, t_min, t_max, t) ? 1 : 0;
}
void orc_aabb_combine(const float a1[3], const float b1[3], const float a2[3], const float b2[3], float upper[3],
                      float lower[3]) {
    AABB r = AABB::combine(AABB::make(v3(a1), v3(b1)), AABB::make(v3(a2), v3(b2)));
    for (int k = 0; k < 3; ++k) {
        upper[k] = r.upper[k];
        lower[k] = r.lower[k];
    }
}
int orc_sphere_hit(const float c[3], float r, const float o[3], const float d[3], float t_min, float t_max, float* t,
                   float normal[3], int* front_face) {
    Sphere sp(v3(c), r);
    SurfaceIntersection intr;
    if (!sp.hit(Ray{v3(o), v3(d)}, t_min, t_max, &intr)) return 0;
    *t = intr.t;
    normal[0] = intr.normal.x; normal[1] = intr.normal.y; normal[2] = intr.normal.z;
    *front_face = intr.front_face;
    return 1;
}
int orc_triangle_hit(const float v0[3], const float v1[3], const float v2[3], const float o[3], const float d[3],
                     float t_min, float t_max, float* t, float normal[3], int* front_face) {
    SurfaceIntersection intr;
    if (!triangle_intersection(v3(v0), v3(v1), v3(v2), Ray{v3(o), v3(d)}, t_min, t_max, &intr)) return 0;
    *t = intr.t;
    normal[0] = intr.normal.x; normal[1] = intr.normal.y; normal[2] = intr.normal.z;
    *front_face = intr.front_face;
    return 1;
}
int orc_plane_hit(const float p[3], const float n[3], const float o[3], const float d[3], float t_min, float t_max,
                  float* t, float normal[3], int* front_face) {
    Plane pl(v3(p), v3(n));
    SurfaceIntersection intr;
    if (!pl.hit(Ray{v3(o), v3(d)}, t_min, t_max, &intr)) return 0;
    *t = intr.t;
    normal[0] = intr.normal.x; normal[1] = intr.normal.y; normal[2] = intr.normal.z;
    *front_face = intr.front_face;
    return 1;
}
// Object{Sphere} with a STORED transform given directly (the reference test sets the field, object.rs:129-148)
int orc_object_sphere_hit(const float c[3], float r, int has_transform, const float matrix[9], const float tr[3],
                          const float o[3], const float d[3], float t_min, float t_max, float* t) {
    Object obj;
    obj.surface = std::make_shared<Sphere>(v3(c), r);
    obj.has_transform = has_transform != 0;
    if (has_transform) {
        memcpy(obj.affine_transform.matrix.data, matrix, 36);
        obj.affine_transform.translation = v3(tr);
    }
    SurfaceIntersection intr;
    if (!obj.hit(Ray{v3(o), v3(d)}, t_min, t_max, &intr)) return 0;
    *t = intr.t;
    return 1;
}
void orc_matrix_rotation(const float axis[3], float angle, float out[9]) {
    Matrix m = Matrix::rotation(v3(axis), angle);
    memcpy(out, m.data, 36);
}
int orc_matrix_inverse(const float m[9], float out[9]) {
    Matrix a, r;
    memcpy(a.data, m, 36);
    if (!a.inverse(&r)) return 0;
    memcpy(out, r.data, 36);
    return 1;
}
void orc_matrix_mul(const float a[9], const float b[9], float out[9]) {
    Matrix x, y;
    memcpy(x.data, a, 36);
    memcpy(y.data, b, 36);
    Matrix r = Matrix::matrix_mul(x, y);
    memcpy(out, r.data, 36);
}
void orc_tangent_space(const float v[3], float t1[3], float t2[3]) {
    Vec3D a, b;
    tangent_space(v3(v), &a, &b);
    t1[0] = a.x; t1[1] = a.y; t1[2] = a.z;
    t2[0] = b.x; t2[1] = b.y; t2[2] = b.z;
}
// Vec3D algebra probe: op 0 normalise(a), 1 cross(a,b), 2 a+b, 3 a-b, 4 -a, 5 a*s, 6 s*a, 7 a/s (Div), 8 a/=s
void orc_vec_op(int op, const float a[3], const float b[3], float s, float out[3]) {
    Vec3D x = v3(a), y = v3(b), r = ZERO;
    switch (op) {
        case 0: r = normalise(x); break;
        case 1: r = cross(x, y); break;
        case 2: r = x + y; break;
        case 3: r = x - y; break;
        case 4: r = -x; break;
        case 5: r = x * s; break;
        case 6: r = s * x; break;
        case 7: r = x / s; break;
        case 8: r = x; r /= s; break;
    }
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float orc_vec_dot(const float a[3], const float b[3]) { return dot(v3(a), v3(b)); }

void orc_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox4x32_10(ctr, key, out); }
float orc_rng_f32(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t dim) {
    Rng r(seed, pixel, sample);
    r.set_bounce(bounce);
    r.dim = dim;
    return r.f32();
}
float orc_pm(int fn, float x) {
    switch (fn) {
        case 0: return pm::sinf_(x);
        case 1: return pm::cosf_(x);
        case 2: return pm::expf_(x);
        default: return pm::logf_(x);
    }
}
// BSDF probe: sample material `m` of the scene with local wo and the rng at (seed,pixel,sample,bounce)
void orc_bsdf_sample(const rt_material* m, const float wo[3], uint64_t seed, uint32_t pixel, uint32_t sample,
                     uint32_t bounce, float spectrum[3], float* pdf, float dir_in[3], int* flags) {
    std::shared_ptr<BSDF> b;
    if (m->kind == RT_MAT_DIFFUSE) b = std::make_shared<Diffuse>(v3(m->a));
    else if (m->kind == RT_MAT_DIELECTRIC) b = std::make_shared<Dielectric>(m->a[0]);
    else b = std::make_shared<Conductor>(v3(m->a), v3(m->b));
    Rng r(seed, pixel, sample);
    r.set_bounce(bounce);
    BSDFSample sm = b->sample(v3(wo), r);
    spectrum[0] = sm.spectrum.x; spectrum[1] = sm.spectrum.y; spectrum[2] = sm.spectrum.z;
    *pdf = sm.pdf;
    dir_in[0] = sm.dir_in.x; dir_in[1] = sm.dir_in.y; dir_in[2] = sm.dir_in.z;
    *flags = (sm.is_specular ? 1 : 0) | (sm.is_transmission ? 2 : 0);
}
int orc_normalise_failed(void) {
    int f = g_normalise_failed;
    g_normalise_failed = false;
    return f;
}

}  // extern "C"
