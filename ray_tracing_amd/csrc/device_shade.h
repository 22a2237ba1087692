// Device-side materials: BSDF sample/eval/pdf, light sample/pdf/hit, medium majorant step.
// One lane = one path.  Every expression keeps the reference's operation order (cited per
// function, paths relative to ray_tracer/src) so that a lane reproduces the CPU result bit for bit.
#pragma once
#include "device_scene.h"

namespace rtd {

RT_DEV V3 ld3(const float* p) { return v3(p[0], p[1], p[2]); }

// ---------------------------------------------------------------- math/complex/mod.rs
struct Cx {
    float re, im;
};
RT_DEV Cx cx(float re, float im) { return Cx{re, im}; }
RT_DEV Cx cmul(Cx a, Cx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }  // :61-70
RT_DEV Cx cmulf(Cx a, float s) { return {a.re * s, a.im * s}; }                                // :72-81
RT_DEV Cx fmulc(float s, Cx a) { return {s * a.re, s * a.im}; }                                // :83-92
RT_DEV float cabs2(Cx a) { return a.re * a.re + a.im * a.im; }                                 // :22-24
RT_DEV Cx crecip(Cx a) { return cmulf(cx(a.re, -a.im), recip(cabs2(a))); }                     // :18-20, :102-108
RT_DEV Cx cdiv(Cx a, Cx b) { return cmul(a, crecip(b)); }                                      // :94-100
RT_DEV Cx fdivc(float s, Cx a) { return fmulc(s, crecip(a)); }                                 // :110-116
RT_DEV Cx cadd(Cx a, Cx b) { return {a.re + b.re, a.im + b.im}; }
RT_DEV Cx faddc(float s, Cx a) { return {a.re + s, a.im}; }                                    // :140-149
RT_DEV Cx csub(Cx a, Cx b) { return {a.re - b.re, a.im - b.im}; }
RT_DEV Cx fsubc(float s, Cx a) { return {a.re - s, a.im}; }  // :173-182: `f32 - Complex` computes Complex - f32 (kept)
RT_DEV Cx csqrt_(Cx z) {                                     // :30-35
    float r = sqrtf(cabs2(z));
    float a = safe_sqrt((r + z.re) / 2.0f);
    float b = safe_sqrt((r - z.re) / 2.0f) * signum(z.im);
    return {a, b};
}

// ---------------------------------------------------------------- math/utils.rs
RT_DEV V3 sample_unit_sphere(Rng& rng) {  // :30-35
    float z = 1.0f - 2.0f * rng.f32();
    float rho = sqrtf(1.0f - z * z);
    float phi = 2.0f * PI_F * rng.f32();
    float s, c;
    sincosf_(phi, &s, &c);
    return v3(rho * c, rho * s, z);
}
RT_DEV float sample_exponential(float a, Rng& rng) {  // :39-42
    float u = rng.f32();
    return -logf_(1.0f - u) / a;
}

// ---------------------------------------------------------------- materials/bsdf.rs
struct BsdfSample {  // :13-19
    V3 spectrum;
    float pdf;
    V3 dir_in;
    bool is_specular, is_transmission;
};

struct Snell {  // :183-188
    float cos_i, cos_t, eta_rel_it;
    bool tir;
};
RT_DEV Snell compute_snell(V3 dir_in, float eta_inside, float eta_outside) {  // :193-225
    bool incident_is_inside = dir_in.z < 0.0f;
    float eta_i = incident_is_inside ? eta_inside : eta_outside;
    float eta_t = incident_is_inside ? eta_outside : eta_inside;
    float eta_rel_it = eta_i / eta_t;
    float cos_i = fabsf(dir_in.z);
    float sin_i_2 = fmaxf(1.0f - cos_i * cos_i, 0.0f);
    float sin_t_2 = sin_i_2 * (eta_rel_it * eta_rel_it);
    if (sin_t_2 > 1.0f) return {cos_i, 0.0f, eta_rel_it, true};
    return {cos_i, safe_sqrt(1.0f - sin_t_2), eta_rel_it, false};
}
RT_DEV float fresnell_reflectance(const Snell& s) {  // :242-258
    if (s.tir) return 1.0f;
    float ci = s.cos_i, ct = s.cos_t, eta = s.eta_rel_it;
    float r_perpendicular = (eta * ci - ct) / (eta * ci + ct);
    float r_parallel = (ci - eta * ct) / (ci + eta * ct);
    return (r_parallel * r_parallel + r_perpendicular * r_perpendicular) / 2.0f;
}
RT_DEV float complex_fresnell_reflectance(float cos_in, Cx eta) {  // :276-297
    cos_in = cos_in < 0.0f ? 0.0f : (cos_in > 1.0f ? 1.0f : cos_in);
    float sin_in_2 = fmaxf(1.0f - cos_in * cos_in, 0.0f);
    Cx sin_t_2 = fdivc(sin_in_2, cmul(eta, eta));
    Cx cos_t = csqrt_(fsubc(1.0f, sin_t_2));
    Cx r_parallel = cdiv(csub(cmulf(eta, cos_in), cos_t), cadd(cmulf(eta, cos_in), cos_t));
    Cx r_perpendicular = cdiv(fsubc(cos_in, cmul(eta, cos_t)), faddc(cos_in, cmul(eta, cos_t)));
    return (cabs2(r_parallel) + cabs2(r_perpendicular)) / 2.0f;
}

RT_DEV BsdfSample bsdf_sample(const rt_material& m, V3 dir_out, Rng& rng) {
    BsdfSample r;
    if (m.kind == RT_MAT_DIFFUSE) {  // :65-85
        V3 v = v3(0.0f, 0.0f, 1.0f) + sample_unit_sphere(rng);
        float nr = norm_recip(v);
        V3 dir_in = isfinite(nr) ? v * nr : v3(0.0f, 0.0f, 1.0f);  // try_normalise().unwrap_or(Z)
        if (dir_out.z < 0.0f) dir_in.z = -dir_in.z;
        r.spectrum = ld3(m.a) * PI_RECIP;
        r.pdf = fabsf(dir_in.z) * PI_RECIP;
        r.dir_in = dir_in;
        r.is_specular = false;
        r.is_transmission = false;
    } else if (m.kind == RT_MAT_DIELECTRIC) {  // :99-134
        Snell snell = compute_snell(dir_out, m.a[0], 1.0f);
        float reflectance = fresnell_reflectance(snell);
        if (rng.f32() < reflectance) {
            V3 dir_in = v3(-dir_out.x, -dir_out.y, dir_out.z);  // reflect, :237-239
            float cos_in = fabsf(dir_in.z);
            r.spectrum = fill(reflectance / cos_in);
            r.pdf = reflectance;
            r.dir_in = dir_in;
            r.is_specular = true;
            r.is_transmission = false;
        } else {
            V3 dir_in = -dir_out * snell.eta_rel_it;  // refract, :228-234
            float sign = dir_out.z < 0.0f ? -1.0f : 1.0f;
            dir_in.z += sign * (snell.cos_i * snell.eta_rel_it - snell.cos_t);
            float cos_in = fabsf(dir_in.z);
            float transmission = 1.0f - reflectance;
            V3 spectrum = fill(transmission / cos_in);
            r.spectrum = spectrum * snell.eta_rel_it * snell.eta_rel_it;
            r.pdf = transmission;
            r.dir_in = dir_in;
            r.is_specular = true;
            r.is_transmission = true;
        }
    } else {  // Conductor, :159-178
        V3 dir_in = v3(-dir_out.x, -dir_out.y, dir_out.z);
        float cos_in = fabsf(dir_in.z);
        V3 f = v3(complex_fresnell_reflectance(cos_in, cx(m.a[0], m.b[0])),
                  complex_fresnell_reflectance(cos_in, cx(m.a[1], m.b[1])),
                  complex_fresnell_reflectance(cos_in, cx(m.a[2], m.b[2])));
        r.spectrum = f / cos_in;
        r.pdf = 1.0f;
        r.dir_in = dir_in;
        r.is_specular = true;
        r.is_transmission = false;
    }
    return r;
}
RT_DEV V3 bsdf_eval(const rt_material& m, V3 dir_in, V3 dir_out) {  // :48-55 / :94-97 / :149-152
    if (m.kind != RT_MAT_DIFFUSE) return fill(0.0f);
    if (dir_in.z * dir_out.z < 0.0f) return fill(0.0f);
    return ld3(m.a) * PI_RECIP;
}
RT_DEV float bsdf_pdf(const rt_material& m, V3 dir_in, V3 dir_out) {  // :57-63 / :136-139 / :154-157
    if (m.kind != RT_MAT_DIFFUSE) return 0.0f;
    if (dir_in.z * dir_out.z < 0.0f) return 0.0f;
    return fabsf(dir_in.z) * PI_RECIP;
}

// ---------------------------------------------------------------- materials/lights.rs
struct LightSample {  // :22-27 (the `point` field is never read by the integrator)
    V3 spectrum, direction;
    float pdf;
};
RT_DEV float sphere_area(float radius) { return 4.0f * PI_F * radius * radius; }  // primitives.rs:38-40

RT_DEV LightSample light_sample(const rt_light& l, V3 ref_point, Rng& rng) {
    LightSample s;
    if (l.kind == RT_LIGHT_POINT) {  // :36-46
        V3 direction = ld3(l.pos) - ref_point;
        float d2 = norm_squared(direction);
        s.spectrum = ld3(l.intensity) / d2;
        s.direction = direction / sqrtf(d2);
        s.pdf = 1.0f;
        return s;
    }
    if (l.kind == RT_LIGHT_SPOT) {  // :58-74 (d2 of the already normalised direction, as written)
        V3 direction = normalise(ld3(l.pos) - ref_point);
        float d2 = norm_squared(direction);
        direction = direction / sqrtf(d2);
        s.spectrum = fill(0.0f);
        s.direction = direction;
        s.pdf = 1.0f;
        if (-dot(direction, ld3(l.dir)) > l.scalar) s.spectrum = ld3(l.intensity) / d2;
        return s;
    }
    // AreaLight, :89-156
    V3 center = ld3(l.pos);
    float radius = l.scalar;
    V3 radiance = ld3(l.intensity);
    V3 z = center - ref_point;
    float d2 = norm_squared(z);
    float sin_theta_max_2 = radius * radius / d2;
    if (sin_theta_max_2 > 1.0f) {
        V3 normal = sample_unit_sphere(rng);
        V3 point = center + radius * normal;
        V3 direction = point - ref_point;
        float norm_2 = norm_squared(direction);
        direction = direction / sqrtf(norm_2);
        s.spectrum = radiance;
        s.direction = direction;
        s.pdf = norm_2 / (sphere_area(radius) * fabsf(dot(direction, normal)));
        return s;
    }
    float sin_theta_max = sqrtf(sin_theta_max_2);
    float cos_theta_max = safe_sqrt(1.0f - sin_theta_max_2);
    float one_minus_cos_theta_max = 1.0f - cos_theta_max;
    if (one_minus_cos_theta_max <= 1e-9f) {  // POINT_LIGHT_EPS, :118
        V3 intensity = radiance * sphere_area(radius) * PI_F;
        V3 direction = z / sqrtf(d2);
        s.spectrum = intensity / d2;
        s.direction = direction;
        s.pdf = 1.0f;
        return s;
    }
    float cos_theta = (cos_theta_max - 1.0f) * rng.f32() + 1.0f;
    float sin_theta_2 = 1.0f - cos_theta * cos_theta;
    float cos_alpha = sin_theta_2 / sin_theta_max + cos_theta * safe_sqrt(1.0f - sin_theta_2 / sin_theta_max_2);
    float sin_alpha = safe_sqrt(1.0f - cos_alpha * cos_alpha);
    float phi = 2.0f * PI_F * rng.f32();
    float sp, cp;
    sincosf_(phi, &sp, &cp);
    z = z / sqrtf(d2);
    V3 x, y;
    tangent_space(z, &x, &y);
    V3 point = center + radius * (cos_alpha * cp * x + cos_alpha * sp * y + sin_alpha * z);
    s.spectrum = radiance;
    s.direction = normalise(point - ref_point);
    s.pdf = 1.0f / (2.0f * PI_F * one_minus_cos_theta_max);
    return s;
}

// Light::hit (:16-18, :83-87): only the sphere emitter can be hit.
RT_DEV bool light_hit(const rt_light& l, V3 o, V3 d, float t_min, float t_max, float* t) {
    if (l.kind != RT_LIGHT_AREA) return false;
    V3 n;
    uint32_t ff;
    return sphere_hit(ld3(l.pos), l.scalar, o, d, t_min, t_max, t, &n, &ff);
}
// Light::pdf (:13-15, :158-177)
RT_DEV float light_pdf(const rt_light& l, V3 o, V3 d) {
    if (l.kind != RT_LIGHT_AREA) return 0.0f;
    float t;
    V3 normal;
    uint32_t ff;
    V3 center = ld3(l.pos);
    float radius = l.scalar;
    if (!sphere_hit(center, radius, o, d, 0.0f, __builtin_inff(), &t, &normal, &ff)) return 0.0f;
    float d2 = norm_squared(o - center);
    float sin_theta_max_2 = radius * radius / d2;
    if (sin_theta_max_2 > 1.0f) return t * t / (sphere_area(radius) * fabsf(dot(d, normal)));
    float cos_theta_max = sqrtf(1.0f - sin_theta_max_2);
    return 1.0f / (2.0f * PI_F * (1.0f - cos_theta_max));
}

// ---------------------------------------------------------------- materials/medium.rs, testmedium.rs
enum { MI_NULL = 0, MI_ABSORPTION = 1, MI_SCATTER = 2 };
struct MediumInteraction {  // medium.rs:33-37
    int kind;
    V3 emission;       // Absorption
    V3 origin, dir;    // Scatter ray
    float pdf, f;
};

RT_DEV float hg_phase(float cos_theta, float g) {  // medium.rs:194-197
    float denom = 1.0f + g * g - 2.0f * g * cos_theta;
    return (1.0f - g * g) / (4.0f * PI_F * denom * safe_sqrt(denom));
}
RT_DEV void sample_hg(float g, V3 wo, Rng& rng, V3* wi_out, float* pdf, float* f) {  // medium.rs:199-226
    float u = rng.f32();
    float v = rng.f32();
    float cos_theta;
    if (fabsf(g) < 1e-3f) {
        cos_theta = 1.0f - 2.0f * u;
    } else {
        float term = (1.0f - g * g) / (1.0f + g - 2.0f * g * u);
        cos_theta = -(1.0f + g * g - term * term) / (2.0f * g);
    }
    float sin_theta = safe_sqrt(1.0f - cos_theta * cos_theta);
    float phi = 2.0f * PI_F * v;
    float sp, cp;
    sincosf_(phi, &sp, &cp);
    V3 wi = to_world_coords(v3(sin_theta * cp, sin_theta * sp, cos_theta), wo);
    float ct = dot(wi, wo);
    *f = hg_phase(ct, g);
    *pdf = *f;
    *wi_out = wi;
}

RT_DEV int f32_as_i32(float f) {  // Rust `as i32`: saturating, NaN -> 0
    if (f != f) return 0;
    if (f <= -2147483648.0f) return (int)0x80000000;
    if (f >= 2147483648.0f) return 0x7FFFFFFF;
    return (int)f;
}
RT_DEV void checker_cell(const rt_medium& m, V3 p, int idx[3]) {  // testmedium.rs:53-60
    float s = m.cell_size;
    idx[0] = f32_as_i32(floorf(p.x / s));
    idx[1] = f32_as_i32(floorf(p.y / s));
    idx[2] = f32_as_i32(floorf(p.z / s));
}
RT_DEV bool checker_white(const int idx[3]) {  // testmedium.rs:63-65 (y multiplied by 0 in the reference)
    return (((uint32_t)idx[0] + (uint32_t)idx[2]) & 1u) == 0;
}

// Medium::evaluate at wavelength 0 (medium.rs:164-170, testmedium.rs:78-86)
RT_DEV void medium_evaluate(const rt_medium& m, V3 p, float* sa, float* ss, V3* em) {
    if (m.kind == RT_MEDIUM_HOMOGENEOUS) {
        *sa = m.sigma_a[0];
        *ss = m.sigma_s[0];
        *em = ld3(m.emission);
        return;
    }
    int idx[3];
    checker_cell(m, p, idx);
    if (checker_white(idx)) {
        *sa = m.sigma_a[0]; *ss = m.sigma_s[0]; *em = ld3(m.emission);
    } else {
        *sa = m.sigma_a_black; *ss = m.sigma_s_black; *em = ld3(m.emission_black);
    }
}

// Medium::sample_segment (medium.rs:58-114): delta tracking inside one constant-majorant segment.
// Returns MI_NULL with *transmittance set, or fills `mi`.
RT_DEV int sample_segment(const rt_medium& m, V3 o, V3 d, float t_start, float t_end, float majorant, Rng& rng,
                          float* transmittance, MediumInteraction* mi) {
    if (majorant == 0.0f) {
        *transmittance = 1.0f;
        return MI_NULL;
    }
    float t = t_start;
    for (;;) {
        float dt = sample_exponential(majorant, rng);
        t += dt;
        if (t >= t_end) {
            float dt2 = t_end - t_start;
            *transmittance = expf_(-majorant * dt2);
            return MI_NULL;
        }
        V3 p = o + d * t;
        float sa, ss;
        V3 em;
        medium_evaluate(m, p, &sa, &ss, &em);
        float u = rng.f32();
        float p_absorption = sa / majorant;
        float p_scattering = ss / majorant;
        if (u < p_absorption) {
            mi->kind = MI_ABSORPTION;
            mi->emission = em;
            return MI_ABSORPTION;
        } else if (u < p_absorption + p_scattering) {
            mi->kind = MI_SCATTER;
            mi->origin = p;
            sample_hg(m.g, d, rng, &mi->dir, &mi->pdf, &mi->f);
            return MI_SCATTER;
        }
    }
}

// Medium::sample_interaction (medium.rs:116-136) over the medium's majorant iterator:
// HomogeneousMedium yields one segment [0, t_max] (medium.rs:172-187); CheckeredMedium walks the
// cells with the 3-D DDA of testmedium.rs:96-211.
RT_DEV int medium_sample_interaction(const rt_medium& m, V3 o, V3 d, float t_max, Rng& rng, float* transmittance_out,
                                     MediumInteraction* mi) {
    float transmittance = 1.0f;
    if (m.kind == RT_MEDIUM_HOMOGENEOUS) {
        V3 sigma_t = ld3(m.sigma_a) + ld3(m.sigma_s);
        float seg_t;
        int k = sample_segment(m, o, d, 0.0f, t_max, max_elem(sigma_t), rng, &seg_t, mi);
        if (k == MI_NULL) transmittance *= seg_t;
        *transmittance_out = transmittance;
        return k;
    }
    // CheckeredMedium iterator state (testmedium.rs:113-162)
    float s = m.cell_size;
    int idx[3], step[3];
    float t_next[3], t_delta[3];
    checker_cell(m, o, idx);
    float oa[3] = {o.x, o.y, o.z}, da[3] = {d.x, d.y, d.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        step[a] = da[a] >= 0.0f ? 1 : -1;
        t_next[a] = __builtin_inff();
        t_delta[a] = __builtin_inff();
        if (fabsf(da[a]) > 0.0f) {
            float boundary = (float)(int)((uint32_t)idx[a] + (step[a] > 0 ? 1u : 0u)) * s;
            float t_to_boundary = (boundary - oa[a]) / da[a];
            t_next[a] = t_to_boundary >= 0.0f ? t_to_boundary : 0.0f;
            t_delta[a] = fmaxf(s / fabsf(da[a]), 0.0f);
        }
    }
    float t = 0.0f;
    for (;;) {  // Iterator::next, testmedium.rs:167-210
        if (t >= t_max) break;
        int axis = 0;
        if (t_next[1] < t_next[axis]) axis = 1;
        if (t_next[2] < t_next[axis]) axis = 2;
        float tn = axis == 0 ? t_next[0] : (axis == 1 ? t_next[1] : t_next[2]);
        float t_end = fminf(tn, t_max);
        bool white = checker_white(idx);
        float sigma_maj = white ? (m.sigma_a[0] + m.sigma_s[0]) : (m.sigma_a_black + m.sigma_s_black);
        float t_start = t;
        if (isfinite(tn)) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
                if (a == axis) {
                    idx[a] = (int)((uint32_t)idx[a] + (uint32_t)step[a]);
                    t_next[a] += t_delta[a];
                }
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a)
                if (a == axis) t_next[a] = t_max;
        }
        t = t_end;
        float seg_t;
        int k = sample_segment(m, o, d, t_start, t_end, sigma_maj, rng, &seg_t, mi);
        if (k != MI_NULL) {
            *transmittance_out = transmittance;
            return k;
        }
        transmittance *= seg_t;
    }
    *transmittance_out = transmittance;
    return MI_NULL;
}

}  // namespace rtd
