// rt_scene_build: description -> flattened SoA scene (the layout uploaded once), plus the image
// output helpers that sit after the render loop.
#include <cstdio>
#include <cstring>
#include <map>

#include "host_internal.h"

using namespace rth;

struct rt_scene {
    std::vector<float> node_upper, node_lower;
    std::vector<uint32_t> node_index, node_num_prim;
    std::vector<float> tri_v0, tri_v1, tri_v2;
    std::vector<uint32_t> tri_prim_id;
    std::vector<rt_bvh_flat> bvhs;
    std::vector<rt_object_flat> objects;
    std::vector<rt_material> materials;
    std::vector<rt_light> lights;
    std::vector<rt_medium> media;
    rt_camera_flat camera;
    int32_t initial_medium;
};

namespace {

inline void put3(float* dst, V3 v) {
    dst[0] = v.x;
    dst[1] = v.y;
    dst[2] = v.z;
}
inline V3 get3(const float* s) { return v3(s[0], s[1], s[2]); }

// Camera::new -> set_direction (camera.rs:14-53)
bool make_camera(const rt_camera_desc& d, rt_camera_flat* out) {
    const float window_height = 2.0f;
    const float window_width = window_height * d.aspect_ratio;
    const float focal_length = 2.0f;
    bool ok = true, ok1;
    V3 origin = get3(d.origin);
    V3 direction = normalise(get3(d.direction), &ok1);
    ok &= ok1;
    V3 horizontal = normalise(cross(direction, v3(0, 0, 1)), &ok1) * window_width;
    ok &= ok1;
    V3 vertical = normalise(cross(horizontal, direction), &ok1) * window_height;
    ok &= ok1;
    V3 llc = origin + direction * focal_length - horizontal / 2.0f - vertical / 2.0f;
    put3(out->origin, origin);
    put3(out->lower_left_corner, llc);
    put3(out->horizontal, horizontal);
    put3(out->vertical, vertical);
    return ok;
}

int append_bvh(rt_scene* s, const rt_mesh& mesh) {
    BuiltBvh b;
    if (!build_bvh(mesh, &b)) return RT_ERR_INVALID;
    rt_bvh_flat f;
    f.node_offset = (uint32_t)s->node_index.size();
    f.node_count = (uint32_t)b.node_index.size();
    f.tri_offset = (uint32_t)s->tri_prim_id.size();
    f.tri_count = mesh.n_triangles();
    f.max_depth = b.max_depth;
    if ((uint64_t)f.node_offset + f.node_count > 0x7FFFFFFFull ||
        (uint64_t)f.tri_offset + f.tri_count > 0x7FFFFFFFull) {
        set_error("rt_scene_build: scene exceeds 2^31 nodes or triangles");
        return RT_ERR_LIMIT;
    }
    size_t n0 = s->node_index.size();
    s->node_upper.resize((n0 + f.node_count) * 3);
    s->node_lower.resize((n0 + f.node_count) * 3);
    for (uint32_t k = 0; k < f.node_count; ++k) {
        put3(&s->node_upper[(n0 + k) * 3], b.node_box[k].upper);
        put3(&s->node_lower[(n0 + k) * 3], b.node_box[k].lower);
    }
    s->node_index.insert(s->node_index.end(), b.node_index.begin(), b.node_index.end());
    s->node_num_prim.insert(s->node_num_prim.end(), b.node_num_prim.begin(), b.node_num_prim.end());
    size_t t0 = s->tri_prim_id.size();
    s->tri_v0.resize((t0 + f.tri_count) * 3);
    s->tri_v1.resize((t0 + f.tri_count) * 3);
    s->tri_v2.resize((t0 + f.tri_count) * 3);
    s->tri_prim_id.resize(t0 + f.tri_count);
    for (uint32_t i = 0; i < f.tri_count; ++i) {  // leaf order: primitives[indices[i]] (bvh.rs:84-86)
        uint32_t t = b.indices[i];
        put3(&s->tri_v0[(t0 + i) * 3], mesh.vertices[mesh.triangles[3 * t]]);
        put3(&s->tri_v1[(t0 + i) * 3], mesh.vertices[mesh.triangles[3 * t + 1]]);
        put3(&s->tri_v2[(t0 + i) * 3], mesh.vertices[mesh.triangles[3 * t + 2]]);
        s->tri_prim_id[t0 + i] = t;
    }
    s->bvhs.push_back(f);
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_scene_build(const rt_scene_desc* desc, rt_scene** out) try {
    if (!desc || !out) {
        set_error("rt_scene_build: null argument");
        return RT_ERR_INVALID;
    }
    if ((desc->n_objects && !desc->objects) || (desc->n_materials && !desc->materials) ||
        (desc->n_lights && !desc->lights) || (desc->n_media && !desc->media)) {
        set_error("rt_scene_build: count without array");
        return RT_ERR_INVALID;
    }
    rt_scene* s = new rt_scene;
    std::map<const rt_mesh*, uint32_t> mesh_to_bvh;
    for (uint32_t o = 0; o < desc->n_objects; ++o) {
        const rt_object_desc& d = desc->objects[o];
        rt_object_flat f;
        memset(&f, 0, sizeof f);
        f.surface_kind = d.surface_kind;
        switch (d.surface_kind) {
            case RT_SURF_MESH: {
                if (!d.mesh) {
                    set_error("rt_scene_build: object %u: mesh surface without mesh", o);
                    delete s;
                    return RT_ERR_INVALID;
                }
                auto it = mesh_to_bvh.find(d.mesh);
                if (it == mesh_to_bvh.end()) {
                    int rc = append_bvh(s, *d.mesh);
                    if (rc != RT_OK) {
                        delete s;
                        return rc;
                    }
                    it = mesh_to_bvh.emplace(d.mesh, (uint32_t)s->bvhs.size() - 1).first;
                }
                f.bvh = it->second;
                break;
            }
            case RT_SURF_SPHERE:
                memcpy(f.surf, d.surf, 4 * sizeof(float));
                break;
            case RT_SURF_PLANE: {  // Plane::new (primitives.rs:222-226)
                bool ok;
                V3 normal = normalise(get3(d.surf + 3), &ok);
                if (!ok) {
                    set_error("rt_scene_build: object %u: plane normal cannot be normalised", o);
                    delete s;
                    return RT_ERR_INVALID;
                }
                put3(f.surf, normal);
                f.surf[3] = -dot(get3(d.surf), normal);
                break;
            }
            case RT_SURF_TRIANGLE:
                memcpy(f.surf, d.surf, 9 * sizeof(float));
                break;
            case RT_SURF_POINT:
                memcpy(f.surf, d.surf, 3 * sizeof(float));
                break;
            default:
                set_error("rt_scene_build: object %u: unknown surface kind %u", o, d.surface_kind);
                delete s;
                return RT_ERR_INVALID;
        }
        if (d.has_transform) {  // Object::new stores the inverse (object.rs:25; affine_transform.rs:12-20)
            M3 m, inv;
            memcpy(m.d, d.matrix, sizeof m.d);
            if (!mat_inverse(m, &inv)) {
                set_error("rt_scene_build: object %u: singular transform (Matrix::inverse -> None)", o);
                delete s;
                return RT_ERR_INVALID;
            }
            f.has_transform = 1;
            memcpy(f.inv_matrix, inv.d, sizeof inv.d);
            put3(f.inv_translation, -mat_mul(inv, get3(d.translation)));
        }
        if (d.material >= desc->n_materials) {
            set_error("rt_scene_build: object %u: material %u out of range", o, d.material);
            delete s;
            return RT_ERR_INVALID;
        }
        f.material = d.material;
        f.has_medium_interface = d.has_medium_interface ? 1 : 0;
        f.medium_inside = d.has_medium_interface ? d.medium_inside : RT_MEDIUM_NONE;
        f.medium_outside = d.has_medium_interface ? d.medium_outside : RT_MEDIUM_NONE;
        if (f.medium_inside >= (int32_t)desc->n_media || f.medium_outside >= (int32_t)desc->n_media ||
            f.medium_inside < RT_MEDIUM_NONE || f.medium_outside < RT_MEDIUM_NONE) {
            set_error("rt_scene_build: object %u: medium index out of range", o);
            delete s;
            return RT_ERR_INVALID;
        }
        s->objects.push_back(f);
    }
    s->materials.assign(desc->materials, desc->materials + desc->n_materials);
    s->lights.assign(desc->lights, desc->lights + desc->n_lights);
    s->media.assign(desc->media, desc->media + desc->n_media);
    if (!make_camera(desc->camera, &s->camera)) {
        set_error("rt_scene_build: camera direction cannot be normalised");
        delete s;
        return RT_ERR_INVALID;
    }
    if (desc->initial_medium < RT_MEDIUM_AUTO || desc->initial_medium >= (int32_t)desc->n_media) {
        set_error("rt_scene_build: initial_medium out of range");
        delete s;
        return RT_ERR_INVALID;
    }
    s->initial_medium = desc->initial_medium;
    *out = s;
    return RT_OK;
} catch (...) {
    return rth::on_exception("rt_scene_build");
}

int rt_scene_view(const rt_scene* s, rt_scene_flat* out) try {
    if (!s || !out) {
        set_error("rt_scene_view: null argument");
        return RT_ERR_INVALID;
    }
    out->n_nodes = (uint32_t)s->node_index.size();
    out->node_upper = s->node_upper.data();
    out->node_lower = s->node_lower.data();
    out->node_index = s->node_index.data();
    out->node_num_prim = s->node_num_prim.data();
    out->n_tris = (uint32_t)s->tri_prim_id.size();
    out->tri_v0 = s->tri_v0.data();
    out->tri_v1 = s->tri_v1.data();
    out->tri_v2 = s->tri_v2.data();
    out->tri_prim_id = s->tri_prim_id.data();
    out->n_bvhs = (uint32_t)s->bvhs.size();
    out->bvhs = s->bvhs.data();
    out->n_objects = (uint32_t)s->objects.size();
    out->objects = s->objects.data();
    out->n_materials = (uint32_t)s->materials.size();
    out->materials = s->materials.data();
    out->n_lights = (uint32_t)s->lights.size();
    out->lights = s->lights.data();
    out->n_media = (uint32_t)s->media.size();
    out->media = s->media.data();
    out->camera = s->camera;
    out->initial_medium = s->initial_medium;
    return RT_OK;
} catch (...) {
    return rth::on_exception("rt_scene_view");
}

void rt_scene_destroy(rt_scene* scene) { delete scene; }

// path_integrator.rs:19-27 and :309-312
int rt_tonemap_u8(const float* rgb_sum, uint32_t width, uint32_t height, uint32_t spp, uint8_t* out) try {
    if (!rgb_sum || !out || spp == 0) {
        set_error("rt_tonemap_u8: bad argument");
        return RT_ERR_INVALID;
    }
    float inv = recip((float)spp);
    for (uint32_t row = 0; row < height; ++row) {
        uint32_t j = height - 1 - row;  // `for j in (0..image_height).rev()`
        for (uint32_t i = 0; i < width; ++i) {
            const float* c = rgb_sum + ((size_t)j * width + i) * 3;
            V3 colour = get3(c);
            if (std::isnan(colour.x) || std::isnan(colour.y) || std::isnan(colour.z))
                colour = v3(255.0f, 255.0f, 0.0f);
            colour = colour * inv;  // Vec3D / f32
            uint8_t* o = out + ((size_t)row * width + i) * 3;
            for (int k = 0; k < 3; ++k) {
                float s = std::sqrt(colour[k]);
                if (s < 0.0f) s = 0.0f;  // f32::clamp keeps NaN
                else if (s > 0.999f) s = 0.999f;
                float v = 256.0f * s;
                o[k] = !(v > 0.0f) ? 0 : (v >= 255.0f ? 255 : (uint8_t)v);  // `as u8` saturates
            }
        }
    }
    return RT_OK;
} catch (...) {
    return rth::on_exception("rt_tonemap_u8");
}

int rt_write_ppm(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height) try {
    if (!path || !rgb8) {
        set_error("rt_write_ppm: null argument");
        return RT_ERR_INVALID;
    }
    FILE* f = strcmp(path, "-") == 0 ? stdout : fopen(path, "w");
    if (!f) {
        set_error("rt_write_ppm: cannot open %s", path);
        return RT_ERR_IO;
    }
    fprintf(f, "P3\n%u %u\n255\n", width, height);
    for (size_t p = 0; p < (size_t)width * height; ++p)
        fprintf(f, "%u %u %u\n", rgb8[3 * p], rgb8[3 * p + 1], rgb8[3 * p + 2]);
    if (f != stdout) fclose(f);
    return RT_OK;
} catch (...) {
    return rth::on_exception("rt_write_ppm");
}

}  // extern "C"
