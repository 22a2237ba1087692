// ORACLE -- test infrastructure only (see vec3d.h).
//
// CPU restatement of the reference's materials module:
//   ray_tracer/src/math/complex/mod.rs, ray_tracer/src/materials/bsdf.rs,
//   ray_tracer/src/materials/lights.rs, ray_tracer/src/materials/medium.rs,
//   ray_tracer/src/materials/testmedium.rs, ray_tracer/src/math/utils.rs:30-42.
// cos/sin/ln/exp go through pmath.h (see there); the RNG is the counter-based one of rng.h.
#pragma once
#include <memory>

#include "geometry.h"
#include "pmath.h"
#include "rng.h"

namespace orc {

// ---------------------------------------------------------------- math/complex/mod.rs
struct Complex {
    float real, imag;
    Complex conj() const { return {real, -imag}; }                     // :11-16
    float abs_square() const { return real * real + imag * imag; }     // :22-24
    float norm() const { return std::sqrt(abs_square()); }             // :26-28
};
inline Complex operator*(Complex a, Complex b) {  // :61-70
    return {a.real * b.real - a.imag * b.imag, a.real * b.imag + a.imag * b.real};
}
inline Complex operator*(Complex a, float s) { return {a.real * s, a.imag * s}; }  // :72-81
inline Complex operator*(float s, Complex a) { return {s * a.real, s * a.imag}; }  // :83-92
inline Complex operator/(Complex a, float s) { return a * recip(s); }              // :102-108
inline Complex crecip(Complex a) { return a.conj() / a.abs_square(); }             // :18-20
inline Complex operator/(Complex a, Complex b) { return a * crecip(b); }           // :94-100
inline Complex operator/(float s, Complex a) { return s * crecip(a); }             // :110-116
inline Complex operator+(Complex a, Complex b) { return {a.real + b.real, a.imag + b.imag}; }
inline Complex operator+(float s, Complex a) { return {a.real + s, a.imag}; }      // :140-149
inline Complex operator-(Complex a, Complex b) { return {a.real - b.real, a.imag - b.imag}; }
// :173-182 -- `f32 - Complex` computes `Complex - f32` (operand swap); reproduced, not fixed.
inline Complex operator-(float s, Complex a) { return {a.real - s, a.imag}; }
inline Complex csqrt(Complex z) {  // :30-35
    float r = z.norm();
    float a = safe_sqrt((r + z.real) / 2.0f);
    float b = safe_sqrt((r - z.real) / 2.0f) * signum(z.imag);
    return {a, b};
}

// ---------------------------------------------------------------- math/utils.rs
inline Vec3D sample_unit_sphere(Rng& rng) {  // :30-35
    float z = 1.0f - 2.0f * rng.f32();
    float rho = std::sqrt(1.0f - z * z);
    float phi = 2.0f * PI * rng.f32();
    return vec(rho * pm::cosf_(phi), rho * pm::sinf_(phi), z);
}
inline float sample_exponential(float a, Rng& rng) {  // :39-42
    float u = rng.f32();
    return -pm::logf_(1.0f - u) / a;
}

// ---------------------------------------------------------------- materials/bsdf.rs
const float PI_RECIP = 1.0f / PI;  // :8

struct BSDFSample {  // :13-19
    Vec3D spectrum;
    float pdf;
    Vec3D dir_in;
    bool is_specular;
    bool is_transmission;
};

inline Vec3D to_shading_coords(Vec3D dir, Vec3D normal) {  // :21-28
    Vec3D v1, v2;
    tangent_space(normal, &v1, &v2);
    return vec(dot(dir, v1), dot(dir, v2), dot(dir, normal));
}
inline Vec3D to_world_coords(Vec3D dir, Vec3D normal) {  // :30-33
    Vec3D v1, v2;
    tangent_space(normal, &v1, &v2);
    return dir.x * v1 + dir.y * v2 + dir.z * normal;
}

struct BSDF {  // :35-41
    virtual ~BSDF() {}
    virtual Vec3D eval(Vec3D dir_in, Vec3D dir_out) const = 0;
    virtual float pdf(Vec3D dir_in, Vec3D dir_out) const = 0;
    virtual BSDFSample sample(Vec3D dir_out, Rng& rng) const = 0;
};

struct Diffuse : BSDF {  // :43-86
    Vec3D reflectance;
    explicit Diffuse(Vec3D r) : reflectance(r) {}
    Vec3D eval(Vec3D dir_in, Vec3D dir_out) const override {
        if (dir_in.z * dir_out.z < 0.0f) return ZERO;
        return reflectance * PI_RECIP;
    }
    float pdf(Vec3D dir_in, Vec3D dir_out) const override {
        if (dir_in.z * dir_out.z < 0.0f) return 0.0f;
        return std::fabs(dir_in.z) * PI_RECIP;
    }
    BSDFSample sample(Vec3D dir_out, Rng& rng) const override {
        Vec3D dir_in;
        if (!try_normalise(Z + sample_unit_sphere(rng), &dir_in)) dir_in = Z;
        if (dir_out.z < 0.0f) dir_in.z = -dir_in.z;
        return {reflectance * PI_RECIP, std::fabs(dir_in.z) * PI_RECIP, dir_in, false, false};
    }
};

struct SnellResult {  // :183-188
    float cos_i, cos_t, eta_rel_it;
    bool total_internal_reflection;
};
inline SnellResult compute_snell(Vec3D dir_in, float eta_inside, float eta_outside) {  // :193-225
    bool incident_is_inside = dir_in.z < 0.0f;
    float eta_i = incident_is_inside ? eta_inside : eta_outside;
    float eta_t = incident_is_inside ? eta_outside : eta_inside;
    float eta_rel_it = eta_i / eta_t;
    float cos_i = std::fabs(dir_in.z);
    float sin_i_2 = fmax_(1.0f - cos_i * cos_i, 0.0f);
    float sin_t_2 = sin_i_2 * (eta_rel_it * eta_rel_it);
    if (sin_t_2 > 1.0f) return {cos_i, 0.0f, eta_rel_it, true};
    return {cos_i, safe_sqrt(1.0f - sin_t_2), eta_rel_it, false};
}
inline Vec3D refract(Vec3D dir_in, const SnellResult& s) {  // :228-234
    Vec3D dir_out = -dir_in * s.eta_rel_it;
    float sign = dir_in.z < 0.0f ? -1.0f : 1.0f;
    dir_out.z += sign * (s.cos_i * s.eta_rel_it - s.cos_t);
    return dir_out;
}
inline Vec3D reflect(Vec3D dir_in) { return vec(-dir_in.x, -dir_in.y, dir_in.z); }  // :237-239
inline float fresnell_reflectance(const SnellResult& s) {                          // :242-258
    if (s.total_internal_reflection) return 1.0f;
    float ci = s.cos_i, ct = s.cos_t, eta = s.eta_rel_it;
    float r_perpendicular = (eta * ci - ct) / (eta * ci + ct);
    float r_parallel = (ci - eta * ct) / (ci + eta * ct);
    return (r_parallel * r_parallel + r_perpendicular * r_perpendicular) / 2.0f;
}

struct Dielectric : BSDF {  // :88-140 (roughness None)
    float refractive_index;
    explicit Dielectric(float n) : refractive_index(n) {}
    Vec3D eval(Vec3D, Vec3D) const override { return ZERO; }
    float pdf(Vec3D, Vec3D) const override { return 0.0f; }
    BSDFSample sample(Vec3D dir_out, Rng& rng) const override {
        SnellResult snell = compute_snell(dir_out, refractive_index, 1.0f);
        float reflectance = fresnell_reflectance(snell);
        if (rng.f32() < reflectance) {
            Vec3D dir_in = reflect(dir_out);
            float cos_in = std::fabs(dir_in.z);
            return {fill(reflectance / cos_in), reflectance, dir_in, true, false};
        }
        Vec3D dir_in = refract(dir_out, snell);
        float cos_in = std::fabs(dir_in.z);
        float transmission = 1.0f - reflectance;
        Vec3D spectrum = fill(transmission / cos_in);
        spectrum = spectrum * snell.eta_rel_it * snell.eta_rel_it;
        return {spectrum, transmission, dir_in, true, true};
    }
};

inline float complex_fresnell_reflectance(float cos_in, Complex eta) {  // :276-297
    cos_in = cos_in < 0.0f ? 0.0f : (cos_in > 1.0f ? 1.0f : cos_in);  // f32::clamp
    float sin_in_2 = fmax_(1.0f - cos_in * cos_in, 0.0f);
    Complex sin_t_2 = sin_in_2 / (eta * eta);
    Complex cos_t = csqrt(1.0f - sin_t_2);
    Complex r_parallel = (eta * cos_in - cos_t) / (eta * cos_in + cos_t);
    Complex r_perpendicular = (cos_in - eta * cos_t) / (cos_in + eta * cos_t);
    return (r_parallel.abs_square() + r_perpendicular.abs_square()) / 2.0f;
}

struct Conductor : BSDF {  // :142-179
    Vec3D refractive_index, extinction_coefficient;
    Conductor(Vec3D n, Vec3D k) : refractive_index(n), extinction_coefficient(k) {}
    Vec3D eval(Vec3D, Vec3D) const override { return ZERO; }
    float pdf(Vec3D, Vec3D) const override { return 0.0f; }
    BSDFSample sample(Vec3D dir_out, Rng&) const override {
        Vec3D dir_in = reflect(dir_out);
        float cos_in = std::fabs(dir_in.z);
        Vec3D r = vec(complex_fresnell_reflectance(cos_in, {refractive_index.x, extinction_coefficient.x}),
                      complex_fresnell_reflectance(cos_in, {refractive_index.y, extinction_coefficient.y}),
                      complex_fresnell_reflectance(cos_in, {refractive_index.z, extinction_coefficient.z}));
        return {r / cos_in, 1.0f, dir_in, true, false};
    }
};

// ---------------------------------------------------------------- materials/lights.rs
struct LightSample {  // :22-27
    Vec3D spectrum, point, direction;
    float pdf;
};

struct Light {  // :11-19
    virtual ~Light() {}
    virtual LightSample sample(Vec3D ref_point, Rng& rng) const = 0;
    virtual float pdf(const Ray&) const { return 0.0f; }
    virtual bool hit(const Ray&, float, float, float*, Vec3D*) const { return false; }
};

struct PointLight : Light {  // :30-47
    Vec3D pos, intensity;
    PointLight(Vec3D p, Vec3D i) : pos(p), intensity(i) {}
    LightSample sample(Vec3D ref_point, Rng&) const override {
        Vec3D direction = pos - ref_point;
        float d2 = norm_squared(direction);
        return {intensity / d2, pos, direction / std::sqrt(d2), 1.0f};
    }
};

struct Spotlight : Light {  // :50-75
    Vec3D pos, dir;
    float cos_angle;
    Vec3D intensity;
    Spotlight(Vec3D p, Vec3D d, float c, Vec3D i) : pos(p), dir(d), cos_angle(c), intensity(i) {}
    LightSample sample(Vec3D ref_point, Rng&) const override {
        Vec3D direction = normalise(pos - ref_point);
        float d2 = norm_squared(direction);  // of the NORMALISED direction: ~1 (quirk, :59-61)
        direction = direction / std::sqrt(d2);
        LightSample s = {ZERO, pos, direction, 1.0f};
        if (-dot(direction, dir) > cos_angle) s.spectrum = intensity / d2;
        return s;
    }
};

struct AreaLight : Light {  // :77-178
    Sphere surface;
    Vec3D emmited_radiance;
    AreaLight(Vec3D c, float r, Vec3D e) : surface(c, r), emmited_radiance(e) {}
    bool hit(const Ray& ray, float t_min, float t_max, float* t, Vec3D* radiance) const override {  // :83-87
        SurfaceIntersection intr;
        if (!surface.hit(ray, t_min, t_max, &intr)) return false;
        *t = intr.t;
        *radiance = emmited_radiance;
        return true;
    }
    LightSample sample(Vec3D ref_point, Rng& rng) const override {  // :89-156
        Vec3D z = surface.center - ref_point;
        float d2 = norm_squared(z);
        float sin_theta_max_2 = surface.radius * surface.radius / d2;
        if (sin_theta_max_2 > 1.0f) {
            Vec3D normal = sample_unit_sphere(rng);
            Vec3D point = surface.center + surface.radius * normal;
            Vec3D direction = point - ref_point;
            float norm_2 = norm_squared(direction);
            direction = direction / std::sqrt(norm_2);
            float pdf = norm_2 / (surface.area() * std::fabs(dot(direction, normal)));
            return {emmited_radiance, point, direction, pdf};
        }
        float sin_theta_max = std::sqrt(sin_theta_max_2);
        float cos_theta_max = safe_sqrt(1.0f - sin_theta_max_2);
        float one_minus_cos_theta_max = 1.0f - cos_theta_max;
        const float POINT_LIGHT_EPS = 1e-9f;
        if (one_minus_cos_theta_max <= POINT_LIGHT_EPS) {
            Vec3D intensity = emmited_radiance * surface.area() * PI;
            Vec3D direction = z / std::sqrt(d2);
            return {intensity / d2, surface.center - direction * surface.radius, direction, 1.0f};
        }
        float cos_theta = (cos_theta_max - 1.0f) * rng.f32() + 1.0f;
        float sin_theta_2 = 1.0f - cos_theta * cos_theta;
        float cos_alpha = sin_theta_2 / sin_theta_max + cos_theta * safe_sqrt(1.0f - sin_theta_2 / sin_theta_max_2);
        float sin_alpha = safe_sqrt(1.0f - cos_alpha * cos_alpha);
        float phi = 2.0f * PI * rng.f32();
        z = z / std::sqrt(d2);
        Vec3D x, y;
        tangent_space(z, &x, &y);
        Vec3D point = surface.center +
                      surface.radius * (cos_alpha * pm::cosf_(phi) * x + cos_alpha * pm::sinf_(phi) * y + sin_alpha * z);
        Vec3D direction = normalise(point - ref_point);
        float pdf = 1.0f / (2.0f * PI * one_minus_cos_theta_max);
        return {emmited_radiance, point, direction, pdf};
    }
    float pdf(const Ray& ray) const override {  // :158-177
        SurfaceIntersection intr;
        if (surface.hit(ray, 0.0f, INFINITY, &intr)) {
            float t = intr.t;
            Vec3D normal = intr.normal;
            float d2 = norm_squared(ray.origin - surface.center);
            float sin_theta_max_2 = surface.radius * surface.radius / d2;
            if (sin_theta_max_2 > 1.0f) return t * t / (surface.area() * std::fabs(dot(ray.direction, normal)));
            float cos_theta_max = std::sqrt(1.0f - sin_theta_max_2);
            return 1.0f / (2.0f * PI * (1.0f - cos_theta_max));
        }
        return 0.0f;
    }
};

// ---------------------------------------------------------------- materials/medium.rs
struct ConstantMajorantSegment {  // :21-25
    float t_start, t_end, sigma_majorant;
};
struct MediumProperties {  // :27-31
    float absorbtion_coefficient, scattering_coefficient;
    Vec3D emission;
};
struct MediumInteraction {  // :33-37
    enum Kind { Null, Absorption, Scatter } kind;
    float transmittance;  // Null
    Vec3D emission;       // Absorption
    Ray ray;              // Scatter
    float pdf, f;
};

inline float henyey_greenstein_phase_function(float cos_theta, float g) {  // :194-197
    float denom = 1.0f + g * g - 2.0f * g * cos_theta;
    return (1.0f - g * g) / (4.0f * PI * denom * safe_sqrt(denom));
}
inline Vec3D sample_henyey_greenstein_direction(Vec3D wo, float g, Rng& rng) {  // :199-218
    float u = rng.f32();
    float v = rng.f32();
    float cos_theta;
    if (std::fabs(g) < 1e-3f) {
        cos_theta = 1.0f - 2.0f * u;
    } else {
        float term = (1.0f - g * g) / (1.0f + g - 2.0f * g * u);
        cos_theta = -(1.0f + g * g - term * term) / (2.0f * g);
    }
    float sin_theta = safe_sqrt(1.0f - cos_theta * cos_theta);
    float phi = 2.0f * PI * v;
    Vec3D wi = vec(sin_theta * pm::cosf_(phi), sin_theta * pm::sinf_(phi), cos_theta);
    return to_world_coords(wi, wo);
}
inline void sample_henyey_greenstein(float g, Vec3D wo, Rng& rng, Vec3D* wi, float* pdf, float* f) {  // :220-226
    *wi = sample_henyey_greenstein_direction(wo, g, rng);
    float cos_theta = dot(*wi, wo);
    *f = henyey_greenstein_phase_function(cos_theta, g);
    *pdf = *f;
}

struct Medium {  // :39-137
    virtual ~Medium() {}
    virtual MediumProperties evaluate(Vec3D point, size_t wavelength) const = 0;
    // majorant_iterator as a pull interface: begin() then next() until it returns false
    struct MajorantIter {
        virtual ~MajorantIter() {}
        virtual bool next(ConstantMajorantSegment* seg) = 0;
    };
    virtual std::unique_ptr<MajorantIter> majorant_iterator(const Ray& ray, float t_max) const = 0;
    virtual void sample_scattering_direction(Vec3D point, Vec3D wo, Rng& rng, Vec3D* wi, float* pdf, float* f) const = 0;

    MediumInteraction sample_segment(const Ray& ray, const ConstantMajorantSegment& segment, Rng& rng) const {  // :58-114
        MediumInteraction mi{};
        float majorant = segment.sigma_majorant;
        if (majorant == 0.0f) {
            mi.kind = MediumInteraction::Null;
            mi.transmittance = 1.0f;
            return mi;
        }
        float t = segment.t_start;
        for (;;) {
            float dt = sample_exponential(majorant, rng);
            t += dt;
            if (t >= segment.t_end) {
                float dt2 = segment.t_end - segment.t_start;
                mi.kind = MediumInteraction::Null;
                mi.transmittance = pm::expf_(-majorant * dt2);
                return mi;
            }
            Vec3D p = ray.at(t);
            MediumProperties mp = evaluate(p, 0);
            float u = rng.f32();
            float p_absorption = mp.absorbtion_coefficient / majorant;
            float p_scattering = mp.scattering_coefficient / majorant;
            if (u < p_absorption) {
                mi.kind = MediumInteraction::Absorption;
                mi.emission = mp.emission;
                return mi;
            } else if (u < p_absorption + p_scattering) {
                Vec3D wi;
                float pdf, f;
                sample_scattering_direction(p, ray.direction, rng, &wi, &pdf, &f);
                mi.kind = MediumInteraction::Scatter;
                mi.ray = Ray{p, wi};
                mi.pdf = pdf;
                mi.f = f;
                return mi;
            }
        }
    }

    MediumInteraction sample_interaction(const Ray& ray, float t_max, Rng& rng, float* transmittance_out) const {  // :116-136
        float transmittance = 1.0f;
        auto it = majorant_iterator(ray, t_max);
        ConstantMajorantSegment segment;
        while (it->next(&segment)) {
            MediumInteraction interaction = sample_segment(ray, segment, rng);
            if (interaction.kind == MediumInteraction::Null) {
                transmittance *= interaction.transmittance;
                continue;
            }
            *transmittance_out = transmittance;
            return interaction;
        }
        MediumInteraction mi{};
        mi.kind = MediumInteraction::Null;
        mi.transmittance = transmittance;
        *transmittance_out = transmittance;
        return mi;
    }
};

struct MediumInterface {  // :10-19
    std::shared_ptr<Medium> outside, inside;
};

struct HomogeneousMedium : Medium {  // :139-192
    Vec3D absorbtion_coefficient, scattering_coefficient, emission;
    float g;
    HomogeneousMedium(Vec3D a, Vec3D s, float g_, Vec3D e)
        : absorbtion_coefficient(a), scattering_coefficient(s), emission(e), g(g_) {}
    MediumProperties evaluate(Vec3D, size_t wavelength) const override {
        return {absorbtion_coefficient[(int)wavelength], scattering_coefficient[(int)wavelength], emission};
    }
    struct Once : MajorantIter {
        ConstantMajorantSegment seg;
        bool done = false;
        bool next(ConstantMajorantSegment* out) override {
            if (done) return false;
            done = true;
            *out = seg;
            return true;
        }
    };
    std::unique_ptr<MajorantIter> majorant_iterator(const Ray&, float t_max) const override {
        Vec3D sigma_t = absorbtion_coefficient + scattering_coefficient;
        auto it = std::make_unique<Once>();
        it->seg = {0.0f, t_max, max_elem(sigma_t)};
        return it;
    }
    void sample_scattering_direction(Vec3D, Vec3D wo, Rng& rng, Vec3D* wi, float* pdf, float* f) const override {
        sample_henyey_greenstein(g, wo, rng, wi, pdf, f);
    }
};

// ---------------------------------------------------------------- materials/testmedium.rs
struct CheckeredMedium : Medium {  // :12-27
    float cell_size;
    float sigma_a_white, sigma_s_white;
    Vec3D emission_white;
    float sigma_a_black, sigma_s_black;
    Vec3D emission_black;
    float g;

    static int32_t f32_as_i32(float f) {  // Rust `as i32`: saturating, NaN -> 0
        if (std::isnan(f)) return 0;
        if (f <= -2147483648.0f) return INT32_MIN;
        if (f >= 2147483648.0f) return INT32_MAX;
        return (int32_t)f;
    }
    void cell_index(Vec3D p, int32_t idx[3]) const {  // :53-60
        float s = cell_size;
        idx[0] = f32_as_i32(std::floor(p.x / s));
        idx[1] = f32_as_i32(std::floor(p.y / s));
        idx[2] = f32_as_i32(std::floor(p.z / s));
    }
    static bool is_white(const int32_t idx[3]) {  // :63-65 (y ignored: `idx[1] * 0`)
        return (((uint32_t)idx[0] + (uint32_t)idx[1] * 0u + (uint32_t)idx[2]) & 1u) == 0;
    }
    void props_at_cell(const int32_t idx[3], float* sa, float* ss, Vec3D* em) const {  // :68-74
        if (is_white(idx)) {
            *sa = sigma_a_white; *ss = sigma_s_white; *em = emission_white;
        } else {
            *sa = sigma_a_black; *ss = sigma_s_black; *em = emission_black;
        }
    }
    MediumProperties evaluate(Vec3D p, size_t) const override {  // :78-86
        int32_t idx[3];
        cell_index(p, idx);
        MediumProperties mp;
        props_at_cell(idx, &mp.absorbtion_coefficient, &mp.scattering_coefficient, &mp.emission);
        return mp;
    }
    struct Iter : MajorantIter {  // :96-211
        const CheckeredMedium* m;
        float t, t_max;
        int32_t idx[3], step[3];
        float t_next[3], t_delta[3];
        bool finished = false;
        bool next(ConstantMajorantSegment* out) override {
            if (finished) return false;
            if (t >= t_max) {
                finished = true;
                return false;
            }
            int axis = 0;
            if (t_next[1] < t_next[axis]) axis = 1;
            if (t_next[2] < t_next[axis]) axis = 2;
            float t_end = fmin_(t_next[axis], t_max);
            float sa, ss;
            Vec3D em;
            m->props_at_cell(idx, &sa, &ss, &em);
            float sigma_maj = sa + ss;
            *out = {t, t_end, sigma_maj};
            if (std::isfinite(t_next[axis])) {
                idx[axis] = (int32_t)((uint32_t)idx[axis] + (uint32_t)step[axis]);  // wrapping, as release Rust
                t_next[axis] += t_delta[axis];
            } else {
                t_next[axis] = t_max;
            }
            t = t_end;
            return true;
        }
    };
    std::unique_ptr<MajorantIter> majorant_iterator(const Ray& ray, float t_max) const override {  // :113-162
        auto it = std::make_unique<Iter>();
        it->m = this;
        it->t = 0.0f;
        it->t_max = t_max;
        float s = cell_size;
        cell_index(ray.origin, it->idx);
        for (int a = 0; a < 3; ++a) {
            float o = ray.origin[a], d = ray.direction[a];
            it->step[a] = d >= 0.0f ? 1 : -1;
            it->t_next[a] = INFINITY;
            it->t_delta[a] = INFINITY;
            if (std::fabs(d) > 0.0f) {
                float boundary = (float)(int32_t)((uint32_t)it->idx[a] + (it->step[a] > 0 ? 1u : 0u)) * s;
                float t_to_boundary = (boundary - o) / d;
                it->t_next[a] = t_to_boundary >= 0.0f ? t_to_boundary : 0.0f;
                it->t_delta[a] = fmax_(s / std::fabs(d), 0.0f);
            }
        }
        return it;
    }
    void sample_scattering_direction(Vec3D, Vec3D wo, Rng& rng, Vec3D* wi, float* pdf, float* f) const override {
        sample_henyey_greenstein(g, wo, rng, wi, pdf, f);
    }
};

}  // namespace orc
