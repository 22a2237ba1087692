// ray_tracer.hpp -- C++ host-side mirror of the `ray_tracer` crate's public interface for the
// render path, implemented on top of the C ABI (include/rt_amd.h).
//
// The reference's host language is Rust; this image has no Rust toolchain, so the host side above
// the C ABI is C++ with the crate's names, argument meaning and error behaviour (a Rust panic
// becomes a std::runtime_error carrying rt_last_error()).  A user of the crate should recognise
// src/main.rs line for line -- see examples/main.cpp.
//
//   ray_tracer::Vec3D, Ray                         math/vec3d, core/ray.rs
//   AffineTransform::{translate, scale, ...}       math/affine_transform.rs:43-81
//   TriangleMesh, read_stl, loop_subdivide         geometry/triangle_mesh.rs, render/stl.rs:8, subdivision_surface.rs:18
//   TriangleMesh::build_bvh -> Surface             geometry/triangle_mesh.rs:73
//   Sphere, Plane, Triangle                        geometry/primitives.rs
//   Diffuse, Dielectric, Conductor                 materials/bsdf.rs
//   PointLight, Spotlight, AreaLight               materials/lights.rs
//   HomogeneousMedium, CheckeredMedium, MediumInterface   materials/medium.rs, testmedium.rs
//   Camera::{new, look_at}, Object::new            core/camera.rs, core/object.rs
//   Engine::{new, render}                          render/path_integrator.rs:50, :278
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/rt_amd.h"

namespace ray_tracer {

inline void check(int rc) {
    if (rc != RT_OK) throw std::runtime_error(std::string("ray_tracer: ") + rt_last_error());
}

struct Vec3D {
    float x = 0, y = 0, z = 0;
    static Vec3D X() { return {1, 0, 0}; }
    static Vec3D Y() { return {0, 1, 0}; }
    static Vec3D Z() { return {0, 0, 1}; }
    static Vec3D ONES() { return {1, 1, 1}; }
    static Vec3D ZERO() { return {0, 0, 0}; }
    static Vec3D fill(float v) { return {v, v, v}; }
    static Vec3D new_polar(float r, float theta, float phi) {  // vec3d.rs:53-59
        return {r * std::sin(theta) * std::cos(phi), r * std::sin(theta) * std::sin(phi), r * std::cos(theta)};
    }
};
inline Vec3D operator+(Vec3D a, Vec3D b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3D operator-(Vec3D a, Vec3D b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3D operator-(Vec3D a) { return {-a.x, -a.y, -a.z}; }
inline Vec3D operator*(Vec3D a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3D operator*(float s, Vec3D a) { return {s * a.x, s * a.y, s * a.z}; }
inline Vec3D cross(Vec3D a, Vec3D b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

struct AffineTransform {  // affine_transform.rs:5-8 (row-major 3x3 + translation)
    float matrix[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    Vec3D translation;
    static AffineTransform translate(Vec3D t) { AffineTransform a; a.translation = t; return a; }
    static AffineTransform scale(float s) { return scale_translate(s, Vec3D::ZERO()); }
    static AffineTransform scale_translate(float s, Vec3D t) {
        AffineTransform a;
        a.matrix[0] = a.matrix[4] = a.matrix[8] = s;
        a.translation = t;
        return a;
    }
    static AffineTransform rotate_translate(Vec3D axis, float angle, Vec3D t) {  // matrix.rs:111-140
        float n = 1.0f / std::sqrt(axis.x * axis.x + axis.y * axis.y + axis.z * axis.z);
        float x = axis.x * n, y = axis.y * n, z = axis.z * n, c = std::cos(angle), s = std::sin(angle), o = 1.0f - c;
        AffineTransform a;
        float m[9] = {c + x * x * o, x * y * o - z * s, x * z * o + y * s, y * x * o + z * s, c + y * y * o,
                      y * z * o - x * s, z * x * o - y * s, z * y * o + x * s, c + z * z * o};
        for (int i = 0; i < 9; ++i) a.matrix[i] = m[i];
        a.translation = t;
        return a;
    }
    static AffineTransform rotation(Vec3D axis, float angle) { return rotate_translate(axis, angle, Vec3D::ZERO()); }
};

class TriangleMesh {  // triangle_mesh.rs:16-19, held by the host library
   public:
    TriangleMesh(const std::vector<Vec3D>& vertices, const std::vector<std::array<uint32_t, 3>>& triangles) {
        rt_mesh* m = nullptr;
        check(rt_mesh_create(&vertices[0].x, (uint32_t)vertices.size(), &triangles[0][0], (uint32_t)triangles.size(), &m));
        h_.reset(m, rt_mesh_destroy);
    }
    explicit TriangleMesh(rt_mesh* m) : h_(m, rt_mesh_destroy) {}
    Vec3D volume_centroid() const {  // triangle_mesh.rs:103
        float c[3];
        check(rt_mesh_volume_centroid(h_.get(), c));
        return {c[0], c[1], c[2]};
    }
    void apply_transform(const AffineTransform& t) {  // triangle_mesh.rs:83
        check(rt_mesh_apply_transform(h_.get(), t.matrix, &t.translation.x));
    }
    std::vector<Vec3D> vertices() const {
        uint32_t nv = 0, nt = 0;
        check(rt_mesh_counts(h_.get(), &nv, &nt));
        std::vector<Vec3D> v(nv);
        check(rt_mesh_copy_out(h_.get(), nv ? &v[0].x : nullptr, nullptr));
        return v;
    }
    uint32_t num_triangles() const {
        uint32_t nv = 0, nt = 0;
        check(rt_mesh_counts(h_.get(), &nv, &nt));
        return nt;
    }
    rt_mesh* handle() const { return h_.get(); }

   private:
    std::shared_ptr<rt_mesh> h_;
};

namespace stl {
inline TriangleMesh read_stl(const std::string& path) {  // stl.rs:8
    rt_mesh* m = nullptr;
    check(rt_mesh_read_stl(path.c_str(), &m));
    return TriangleMesh(m);
}
}  // namespace stl
inline TriangleMesh loop_subdivide(const TriangleMesh& mesh, uint32_t divisions) {  // subdivision_surface.rs:18
    rt_mesh* m = nullptr;
    check(rt_mesh_loop_subdivide(mesh.handle(), divisions, &m));
    return TriangleMesh(m);
}

// ---- surfaces (trait Surface): value types recorded into the scene description
struct Surface {
    uint32_t kind = RT_SURF_POINT;
    float surf[9] = {0};
    std::shared_ptr<TriangleMesh> mesh;  // RT_SURF_MESH keeps the mesh alive
};
inline Surface Sphere(Vec3D c, float r) { Surface s; s.kind = RT_SURF_SPHERE; s.surf[0] = c.x; s.surf[1] = c.y; s.surf[2] = c.z; s.surf[3] = r; return s; }
inline Surface Plane(Vec3D p, Vec3D n) { Surface s; s.kind = RT_SURF_PLANE; float v[6] = {p.x, p.y, p.z, n.x, n.y, n.z}; for (int i = 0; i < 6; ++i) s.surf[i] = v[i]; return s; }
inline Surface Triangle(Vec3D a, Vec3D b, Vec3D c) { Surface s; s.kind = RT_SURF_TRIANGLE; float v[9] = {a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z}; for (int i = 0; i < 9; ++i) s.surf[i] = v[i]; return s; }
inline Surface build_bvh(const std::shared_ptr<TriangleMesh>& mesh) {  // TriangleMesh::build_bvh (built in Engine::new)
    Surface s;
    s.kind = RT_SURF_MESH;
    s.mesh = mesh;
    return s;
}

// ---- materials / lights / media: the C structs, with the crate's constructors
inline rt_material Diffuse(Vec3D reflectance) { rt_material m{}; m.kind = RT_MAT_DIFFUSE; m.a[0] = reflectance.x; m.a[1] = reflectance.y; m.a[2] = reflectance.z; return m; }
inline rt_material Dielectric(float refractive_index) { rt_material m{}; m.kind = RT_MAT_DIELECTRIC; m.a[0] = refractive_index; return m; }
inline rt_material Conductor(Vec3D n, Vec3D k) { rt_material m{}; m.kind = RT_MAT_CONDUCTOR; m.a[0] = n.x; m.a[1] = n.y; m.a[2] = n.z; m.b[0] = k.x; m.b[1] = k.y; m.b[2] = k.z; return m; }
inline rt_light PointLight(Vec3D pos, Vec3D intensity) { rt_light l{}; l.kind = RT_LIGHT_POINT; l.pos[0] = pos.x; l.pos[1] = pos.y; l.pos[2] = pos.z; l.intensity[0] = intensity.x; l.intensity[1] = intensity.y; l.intensity[2] = intensity.z; return l; }
inline rt_light Spotlight(Vec3D pos, Vec3D dir, float cos_angle, Vec3D intensity) { rt_light l = PointLight(pos, intensity); l.kind = RT_LIGHT_SPOT; l.dir[0] = dir.x; l.dir[1] = dir.y; l.dir[2] = dir.z; l.scalar = cos_angle; return l; }
inline rt_light AreaLight(Vec3D center, float radius, Vec3D emmited_radiance) { rt_light l = PointLight(center, emmited_radiance); l.kind = RT_LIGHT_AREA; l.scalar = radius; return l; }
inline rt_medium HomogeneousMedium(Vec3D sa, Vec3D ss, std::optional<float> g, Vec3D emission) {
    rt_medium m{};
    m.kind = RT_MEDIUM_HOMOGENEOUS;
    float a[3] = {sa.x, sa.y, sa.z}, s[3] = {ss.x, ss.y, ss.z}, e[3] = {emission.x, emission.y, emission.z};
    for (int i = 0; i < 3; ++i) { m.sigma_a[i] = a[i]; m.sigma_s[i] = s[i]; m.emission[i] = e[i]; }
    m.g = g.value_or(0.0f);
    return m;
}
inline rt_medium CheckeredMedium(float cell, float sa_w, float ss_w, Vec3D em_w, float sa_b, float ss_b, Vec3D em_b, float g) {
    rt_medium m{};
    m.kind = RT_MEDIUM_CHECKERED;
    m.cell_size = cell; m.sigma_a[0] = sa_w; m.sigma_s[0] = ss_w; m.sigma_a_black = sa_b; m.sigma_s_black = ss_b; m.g = g;
    m.emission[0] = em_w.x; m.emission[1] = em_w.y; m.emission[2] = em_w.z;
    m.emission_black[0] = em_b.x; m.emission_black[1] = em_b.y; m.emission_black[2] = em_b.z;
    return m;
}
struct MediumInterface {  // MediumInterface::new(inside, outside), medium.rs:16
    std::optional<rt_medium> inside, outside;
};

struct Camera {  // Camera::new / look_at (camera.rs:14, :55)
    Vec3D origin, direction;
    float aspect_ratio;
    Camera(Vec3D o, Vec3D d, float aspect) : origin(o), direction(d), aspect_ratio(aspect) {}
    void look_at(Vec3D point) { direction = point - origin; }
};

struct Object {  // Object::new(surface, material, affine_transform, mediuminterface), object.rs:17
    Surface surface;
    rt_material material;
    std::optional<AffineTransform> affine_transform;
    std::optional<MediumInterface> mediuminterface;
};

class Engine {  // path_integrator.rs:41-46
   public:
    Engine(std::vector<Object> objects, std::vector<rt_light> lights, Camera camera, int device = 0)
        : objects_(std::move(objects)) {
        std::vector<rt_object_desc> od(objects_.size());
        std::vector<rt_material> mats;
        std::vector<rt_medium> media;
        for (size_t i = 0; i < objects_.size(); ++i) {
            const Object& o = objects_[i];
            rt_object_desc& d = od[i];
            d = rt_object_desc{};
            d.surface_kind = o.surface.kind;
            d.mesh = o.surface.mesh ? o.surface.mesh->handle() : nullptr;
            for (int k = 0; k < 9; ++k) d.surf[k] = o.surface.surf[k];
            if (o.affine_transform) {
                d.has_transform = 1;
                for (int k = 0; k < 9; ++k) d.matrix[k] = o.affine_transform->matrix[k];
                d.translation[0] = o.affine_transform->translation.x;
                d.translation[1] = o.affine_transform->translation.y;
                d.translation[2] = o.affine_transform->translation.z;
            }
            d.material = (uint32_t)mats.size();
            mats.push_back(o.material);
            d.medium_inside = d.medium_outside = RT_MEDIUM_NONE;
            if (o.mediuminterface) {
                d.has_medium_interface = 1;
                if (o.mediuminterface->inside) { d.medium_inside = (int32_t)media.size(); media.push_back(*o.mediuminterface->inside); }
                if (o.mediuminterface->outside) { d.medium_outside = (int32_t)media.size(); media.push_back(*o.mediuminterface->outside); }
            }
        }
        rt_scene_desc sd{};
        sd.n_objects = (uint32_t)od.size(); sd.objects = od.data();
        sd.n_materials = (uint32_t)mats.size(); sd.materials = mats.data();
        sd.n_lights = (uint32_t)lights.size(); sd.lights = lights.data();
        sd.n_media = (uint32_t)media.size(); sd.media = media.data();
        sd.camera = {{camera.origin.x, camera.origin.y, camera.origin.z}, {camera.direction.x, camera.direction.y, camera.direction.z}, camera.aspect_ratio};
        sd.initial_medium = RT_MEDIUM_AUTO;  // Engine::new -> set_initial_medium (:61)
        rt_scene* sc = nullptr;
        check(rt_scene_build(&sd, &sc));
        scene_.reset(sc, rt_scene_destroy);
        rt_ctx* c = nullptr;
        check(rt_create(device, &c));
        ctx_.reset(c, rt_destroy);
        rt_scene_flat flat;
        check(rt_scene_view(sc, &flat));
        check(rt_upload_scene(c, &flat));
    }

    // Engine::render(aspect_ratio, image_width, samples_per_pixel, ray_depth) (:278-318): P3 PPM to
    // `path` ("-" = stdout, as the reference streams it).  `seed` replaces Rng::new()'s clock seed.
    void render(float aspect_ratio, size_t image_width, uint32_t samples_per_pixel, uint32_t ray_depth,
                const char* path = "-", uint64_t seed = 0, rt_stats* stats = nullptr) const {
        uint32_t w = (uint32_t)image_width, h = (uint32_t)((float)image_width / aspect_ratio);  // :285
        std::vector<float> sum((size_t)w * h * 3);
        check(rt_render(ctx_.get(), w, h, 0, samples_per_pixel, ray_depth, seed, 0, sum.data(), stats));
        std::vector<uint8_t> rgb((size_t)w * h * 3);
        check(rt_tonemap_u8(sum.data(), w, h, samples_per_pixel, rgb.data()));
        check(rt_write_ppm(path, rgb.data(), w, h));
        std::fprintf(stderr, "Done.\n");  // :315
    }

   private:
    std::vector<Object> objects_;
    std::shared_ptr<rt_scene> scene_;
    std::shared_ptr<rt_ctx> ctx_;
};

}  // namespace ray_tracer
