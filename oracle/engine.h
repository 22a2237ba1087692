// ORACLE -- test infrastructure only (see vec3d.h).
//
// CPU restatement of ray_tracer/src/core/{camera,object}.rs and
// ray_tracer/src/render/path_integrator.rs (Engine: hit, hit_bool, hit_light, ray_colour, render).
#pragma once
#include <memory>
#include <vector>

#include "materials.h"

namespace orc {

const float T_MIN = 1e-3f;      // path_integrator.rs:15
const float T_MAX = INFINITY;   // path_integrator.rs:16

struct Camera {  // camera.rs:4-64
    Vec3D origin, direction;
    float aspect_ratio;
    Vec3D horizontal, vertical, lower_left_corner;
    Camera() {}
    Camera(Vec3D o, Vec3D d, float aspect) : origin(o), direction(d), aspect_ratio(aspect) { set_direction(d); }
    void set_direction(Vec3D dir) {  // :41-53
        float window_height = 2.0f;
        float window_width = window_height * aspect_ratio;
        float focal_length = 2.0f;
        dir = normalise(dir);
        direction = dir;
        horizontal = normalise(cross(dir, Z)) * window_width;
        vertical = normalise(cross(horizontal, dir)) * window_height;
        lower_left_corner = origin + dir * focal_length - horizontal / 2.0f - vertical / 2.0f;
    }
    Ray ray(float u, float v) const {  // :60-63
        Vec3D d = lower_left_corner + u * horizontal + v * vertical - origin;
        return Ray{origin, normalise(d)};
    }
};

struct Object {  // object.rs:8-87
    std::shared_ptr<Surface> surface;
    std::shared_ptr<BSDF> material;
    bool has_transform = false;
    AffineTransform affine_transform;  // stored INVERSE (object.rs:25)
    bool has_mediuminterface = false;
    MediumInterface mediuminterface;
    int32_t medium_inside_id = -1, medium_outside_id = -1;  // addition: ids for reporting

    bool hit(const Ray& ray, float t_min, float t_max, SurfaceIntersection* out) const {  // :32-68
        if (!has_transform) return surface->hit(ray, t_min, t_max, out);
        const AffineTransform& transform = affine_transform;
        Vec3D origin = transform.matrix.mul(ray.origin) + transform.translation;
        Vec3D direction = transform.matrix.mul(ray.direction);
        float nr = norm_recip(direction);
        Ray obj_ray{origin, direction * nr};
        SurfaceIntersection intersection;
        if (!surface->hit(obj_ray, t_min, t_max, &intersection)) return false;
        out->t = intersection.t * nr;
        out->normal = normalise(transform.matrix.transpose_mul(intersection.normal));
        out->front_face = intersection.front_face;
        out->prim = intersection.prim;
        return true;
    }
    bool hit_bool(const Ray& ray, float t_min, float t_max) const {  // :71-87
        if (!has_transform) return surface->hit_bool(ray, t_min, t_max);
        const AffineTransform& transform = affine_transform;
        Vec3D origin = transform.matrix.mul(ray.origin) + transform.translation;
        Vec3D direction = transform.matrix.mul(ray.direction);
        float nr = norm_recip(direction);
        Ray obj_ray{origin, direction * nr};
        return surface->hit_bool(obj_ray, t_min, t_max);
    }
};

inline float mis_weight(float pa, float pb) {  // path_integrator.rs:30-39 with power = 2
    float a = pa * pa;
    float b = pb * pb;
    if (a + b == 0.0f) return 0.0f;
    return a / (a + b);
}

struct RayCounters {  // addition: the metric's ray counts
    uint64_t closest = 0;  // Engine::hit calls
    uint64_t shadow = 0;   // Engine::hit_bool calls
    uint64_t iterations = 0;
};

struct Engine {  // path_integrator.rs:41-46
    std::vector<Object> objects;
    std::vector<std::shared_ptr<Light>> lights;
    Camera camera;
    std::shared_ptr<Medium> initial_medium;
    int32_t initial_medium_id = -1;
    std::vector<std::shared_ptr<Medium>> media;  // addition: id -> medium, for reporting

    void set_initial_medium() {  // :65-85
        Ray ray = camera.ray(0.5f, 0.5f);
        int32_t depth = 0;
        SurfaceIntersection intr;
        const Object* obj;
        while (hit(ray, T_MIN, T_MAX, &intr, &obj)) {
            bool entering = intr.front_face;
            if (obj->has_mediuminterface) {
                if (!entering && depth == 0) {
                    initial_medium = obj->mediuminterface.outside;
                    initial_medium_id = obj->medium_outside_id;
                    break;
                }
                if (entering) depth += 1; else depth -= 1;
            }
            ray.origin = ray.at(intr.t);
        }
    }

    bool hit(const Ray& ray, float t_min, float t_max, SurfaceIntersection* out, const Object** obj_out) const {  // :88-97
        bool result = false;
        for (const Object& object : objects) {
            SurfaceIntersection intr;
            if (object.hit(ray, t_min, t_max, &intr)) {
                t_max = intr.t;
                *out = intr;
                *obj_out = &object;
                result = true;
            }
        }
        return result;
    }
    bool hit_bool(const Ray& ray, float t_min, float t_max) const {  // :100-104
        for (const Object& obj : objects)
            if (obj.hit_bool(ray, t_min, t_max)) return true;
        return false;
    }
    bool hit_light(const Ray& ray, float t_min, float t_max, Vec3D* radiance, const Light** light_out) const {  // :107-121
        bool result = false;
        for (const auto& light : lights) {
            float t;
            Vec3D rad;
            if (light->hit(ray, t_min, t_max, &t, &rad)) {
                t_max = t;
                *radiance = rad;
                *light_out = light.get();
                result = true;
            }
        }
        return result;
    }

    // :138-276
    Vec3D ray_colour(Ray ray, uint32_t depth, Rng& rng, RayCounters* rc) const {
        const std::shared_ptr<Medium>* current_medium = &initial_medium;
        static const std::shared_ptr<Medium> none;
        Vec3D radiance = ZERO;
        Vec3D beta = ONES;
        float prev_pdf = 1.0f;
        bool prev_specular = true;
        uint32_t bounce = 0;

        while (depth > 0) {
            depth -= 1;
            rng.set_bounce(++bounce);
            rc->iterations += 1;

            SurfaceIntersection hit_rec;
            const Object* object = nullptr;
            rc->closest += 1;
            bool have_hit = hit(ray, T_MIN, T_MAX, &hit_rec, &object);
            float t_hit = have_hit ? hit_rec.t : T_MAX;

            Vec3D emitted;
            const Light* light;
            if (hit_light(ray, T_MIN, t_hit, &emitted, &light)) {
                if (prev_specular) {
                    radiance += mul_elemwise(beta, emitted);
                } else {
                    float p_light = (1.0f / (float)lights.size()) * light->pdf(ray);  // lights_pdf, :134-136
                    float w = mis_weight(prev_pdf, p_light);
                    radiance += w * mul_elemwise(beta, emitted);
                }
                break;
            }
            if (!have_hit) break;

            if (*current_medium) {
                float transmittance;
                MediumInteraction mi = (*current_medium)->sample_interaction(ray, t_hit, rng, &transmittance);
                if (mi.kind == MediumInteraction::Null) {
                    beta *= transmittance;
                } else if (mi.kind == MediumInteraction::Absorption) {
                    radiance += mul_elemwise(beta, mi.emission);
                    break;
                } else {
                    depth -= 1;  // u32: wraps at 0 like release-mode Rust (:193)
                    beta *= transmittance * mi.f / mi.pdf;
                    ray = mi.ray;
                    continue;
                }
            }

            Vec3D interaction_point = ray.at(hit_rec.t);
            Vec3D wo_local = to_shading_coords(-ray.direction, hit_rec.normal);

            BSDFSample bsdf_sample = object->material->sample(wo_local, rng);
            prev_specular = bsdf_sample.is_specular;

            if (!prev_specular) {
                size_t n = lights.size();  // sample_lights_uniform, :124-131
                uint32_t light_index = rng.usize((uint32_t)n);
                const Light& l = *lights[light_index];
                float select_light_pdf = 1.0f / (float)n;
                LightSample light_sample = l.sample(interaction_point, rng);

                Ray shadow_ray{interaction_point, light_sample.direction};
                rc->shadow += 1;
                if (!hit_bool(shadow_ray, T_MIN, T_MAX)) {
                    Vec3D wi_local = to_shading_coords(shadow_ray.direction, hit_rec.normal);
                    float p_light = light_sample.pdf * select_light_pdf;
                    Vec3D bsdf_val = object->material->eval(wi_local, wo_local);
                    float cosv = std::fabs(wi_local.z);
                    Vec3D contribution = mul_elemwise(light_sample.spectrum, bsdf_val * cosv);
                    float p_bsdf = object->material->pdf(wi_local, wo_local);
                    float w = mis_weight(p_light, p_bsdf);
                    if (p_light > 0.0f) radiance += w * mul_elemwise(beta, contribution) / p_light;
                }
            }

            Vec3D bsdf_val_cos = bsdf_sample.spectrum * std::fabs(bsdf_sample.dir_in.z);
            beta = mul_elemwise(beta, bsdf_val_cos) / bsdf_sample.pdf;
            prev_pdf = bsdf_sample.pdf;
            prev_specular = bsdf_sample.is_specular;

            ray = Ray{interaction_point, to_world_coords(bsdf_sample.dir_in, hit_rec.normal)};

            if (bsdf_sample.is_transmission) {
                if (object->has_mediuminterface) {
                    if (hit_rec.front_face) current_medium = &object->mediuminterface.inside;
                    else current_medium = &object->mediuminterface.outside;
                }
            }

            float q = 1.0f - max_elem(beta);
            if (rng.f32() < q) break;
            beta /= 1.0f - q;
        }
        (void)none;
        return radiance;
    }

    // The pixel/sample loops of render (:295-308) for one pixel; returns the SUM over samples.
    Vec3D render_pixel(uint32_t i, uint32_t j, uint32_t width, uint32_t height, uint32_t spp_first,
                       uint32_t spp_count, uint32_t ray_depth, uint64_t seed, RayCounters* rc) const {
        Vec3D colour = ZERO;
        for (uint32_t s = spp_first; s < spp_first + spp_count; ++s) {
            Rng rng(seed, j * width + i, s);
            float u = ((float)i + rng.f32()) / (float)(width - 1);
            float v = ((float)j + rng.f32()) / (float)(height - 1);
            Ray ray = camera.ray(u, v);
            colour += ray_colour(ray, ray_depth, rng, rc);
        }
        return colour;
    }
};

}  // namespace orc
