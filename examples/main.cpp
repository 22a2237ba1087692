// Host program equivalent to the reference binary (src/main.rs) on the MI355X backend:
//   ./rt_main [scene] [width] [spp] [depth] [seed] > image.ppm
// scene = volumetric_sphere (default, src/main.rs:25-83) | tetrahedra (:86-162) | primitives (:302-397)
//       | teapot <utah_teapot.stl> (:165-211) | baby_yoda <baby_yoda.stl> (:214-299)
// The scene PARAMETERS restate src/main.rs; everything else goes through ray_tracer.hpp -> rt_amd.h.
#include <cstdlib>
#include <cstring>
#include <tuple>

#include "../ray_tracing_amd/csrc/hostapi/ray_tracer.hpp"

using namespace ray_tracer;
using Scene = std::tuple<std::vector<Object>, std::vector<rt_light>, Camera>;

static const float PI = 3.14159265358979323846f;
static const float ASPECT_RATIO = 16.0f / 9.0f;  // src/main.rs:12

static Scene scene_volumetric_sphere() {
    MediumInterface mi{CheckeredMedium(0.5f, 0.2f, 0.1f, Vec3D::X() * 5.0f, 0.1f, 0.5f, Vec3D::ZERO(), 0.0f), std::nullopt};
    Object sphere{Sphere(Vec3D::ZERO(), 1.6f), Dielectric(1.0f), std::nullopt, mi};
    return {{sphere}, {AreaLight({0.0f, 100.0f, 0.0f}, 80.0f, Vec3D::ONES() * 1.0f)},
            Camera({0.1f, -5.0f, 0.1f}, {0.0f, 1.0f, 0.0f}, ASPECT_RATIO)};
}

static Scene scene_tetrahedra() {
    TriangleMesh tetra({{1, 1, 1}, {1, -1, -1}, {-1, 1, -1}, {-1, -1, 1}}, {{0, 1, 2}, {1, 3, 2}, {0, 3, 1}, {0, 2, 3}});
    rt_material mat = Diffuse({1.0f, 0.5f, 0.5f});
    const uint32_t subdivisions[4] = {0, 1, 2, 6};
    Vec3D separation = 0.65f * Vec3D::Y();
    std::vector<Object> scene;
    scene.push_back({Plane(-0.6f * Vec3D::Z(), Vec3D::Z()), Diffuse(Vec3D::ONES() * 0.8f), std::nullopt, std::nullopt});
    Vec3D loc = Vec3D::new_polar(1.8f, 0.3f * PI, 0.0f * PI);
    Vec3D centre = separation * 0.5f * 3.0f;
    for (int i = 0; i < 4; ++i) {
        auto mesh = std::make_shared<TriangleMesh>(loop_subdivide(tetra, subdivisions[i]));
        scene.push_back({build_bvh(mesh), mat, AffineTransform::translate((float)i * separation), std::nullopt});
    }
    return {scene, {PointLight(10.0f * Vec3D{10.0f, 2.0f, 10.0f}, Vec3D::ONES() * 60000.0f)}, Camera(loc + centre, -loc, ASPECT_RATIO)};
}

static Scene scene_mesh_file(const char* path, bool yoda) {
    auto mesh = std::make_shared<TriangleMesh>(yoda ? stl::read_stl(path) : loop_subdivide(stl::read_stl(path), 1));
    float ground = INFINITY;
    for (const Vec3D& v : mesh->vertices()) ground = std::fmin(ground, v.z);
    Vec3D centre = mesh->volume_centroid();
    std::vector<Object> scene;
    if (!yoda) {
        scene.push_back({build_bvh(mesh), Diffuse({0.9f, 0.3f, 0.3f}), std::nullopt, std::nullopt});
        scene.push_back({Plane(Vec3D::Z() * ground, Vec3D::Z()), Diffuse(Vec3D::ONES() * 0.8f), std::nullopt, std::nullopt});
        Camera cam(centre + Vec3D::new_polar(18.0f, 0.3f * PI, 0.5f * PI), Vec3D::ONES(), ASPECT_RATIO);
        cam.look_at(centre);
        return {scene, {PointLight(100.0f * Vec3D{1.0f, 0.0f, 1.0f}, Vec3D::ONES() * 60000.0f)}, cam};
    }
    MediumInterface mi{HomogeneousMedium({0.03f, 0.03f, 0.01f}, Vec3D::ZERO(), std::nullopt, Vec3D::ZERO()), std::nullopt};
    Camera cam(centre + 90.0f * Vec3D::new_polar(1.0f, PI / 2.0f, 0.4f * PI) + 40.0f * Vec3D::Z(), Vec3D::ONES(), ASPECT_RATIO);
    cam.look_at(centre);
    Vec3D away = cam.direction;  // get_direction() is normalised in the reference; the light is far enough not to care
    float n = 1.0f / std::sqrt(away.x * away.x + away.y * away.y + away.z * away.z);
    away = away * n;
    Vec3D left = cross(Vec3D::Z(), away);
    away.z = 0.0f;
    scene.push_back({build_bvh(mesh), Dielectric(1.5f), std::nullopt, mi});
    scene.push_back({Plane(Vec3D::Z() * ground, Vec3D::Z()), Diffuse(Vec3D::ONES() * 0.8f), std::nullopt, std::nullopt});
    scene.push_back({Sphere(Vec3D::ZERO(), 1.0f), Diffuse(Vec3D{0.8f, 0.2f, 0.2f} * 2.0f),
                     AffineTransform::scale_translate(5.0f, {-4.0f, 6.0f, 59.0f}), std::nullopt});
    return {scene, {AreaLight(centre + 2000.0f * Vec3D::Z() + 2000.0f * left - 1000.0f * away, 1000.0f, Vec3D::ONES() * 5.0f)}, cam};
}

static Scene scene_primitives() {
    std::vector<Object> s;
    auto none = std::nullopt;
    s.push_back({Sphere({-2.0f, 0.0f, 0.5f}, 0.5f), Diffuse({1.0f, 0.01f, 0.01f}), none, none});
    s.push_back({Sphere({-0.5f, 0.0f, 0.8f}, 0.75f), Conductor(Vec3D::ONES(), Vec3D::ONES()), none, none});
    s.push_back({Sphere({1.5f, 0.0f, 1.0f}, 1.0f), Diffuse({0.01f, 0.01f, 1.0f}), none, none});
    s.push_back({Triangle({-4.5f, 2.0f, 0.2f}, {-2.0f, 3.0f, 0.2f}, {-3.0f, 2.5f, 3.2f}), Conductor(Vec3D::ONES(), Vec3D::ONES()), none, none});
    s.push_back({Sphere({0.0f, -1.6f, 0.8f}, 0.75f), Dielectric(1.5f), none, none});
    s.push_back({Sphere({0.0f, 0.0f, -100.0f}, 100.0f), Diffuse(Vec3D::ONES() * 0.5f), none, none});
    return {s, {AreaLight({0.0f, 110.0f, 0.0f}, 100.0f, Vec3D::ONES() * 2.0f), AreaLight({2.5f, -2.5f, 0.5f}, 0.5f, {0.1f, 3.0f, 0.1f})},
            Camera({-2.5f, -4.0f, 3.0f}, {0.5f, 1.0f, -0.5f}, ASPECT_RATIO)};
}

int main(int argc, char** argv) {
    try {
        int a = 1;
        std::string name = argc > a ? argv[a++] : "volumetric_sphere";
        const char* file = (name == "teapot" || name == "baby_yoda") ? (argc > a ? argv[a++] : nullptr) : nullptr;
        size_t width = argc > a ? std::strtoul(argv[a++], nullptr, 10) : 400;         // src/main.rs:13
        uint32_t spp = argc > a ? (uint32_t)std::strtoul(argv[a++], nullptr, 10) : 256;  // :14
        uint32_t depth = argc > a ? (uint32_t)std::strtoul(argv[a++], nullptr, 10) : 16;  // :15
        uint64_t seed = argc > a ? std::strtoull(argv[a++], nullptr, 10) : 0;
        if ((name == "teapot" || name == "baby_yoda") && !file) {
            std::fprintf(stderr, "usage: rt_main %s <file.stl> [width spp depth seed]\n", name.c_str());
            return 2;
        }
        Scene sc = name == "tetrahedra" ? scene_tetrahedra()
                 : name == "primitives" ? scene_primitives()
                 : name == "teapot"     ? scene_mesh_file(file, false)
                 : name == "baby_yoda"  ? scene_mesh_file(file, true)
                                        : scene_volumetric_sphere();
        Engine eng(std::get<0>(sc), std::get<1>(sc), std::get<2>(sc));
        rt_stats st;
        eng.render(ASPECT_RATIO, width, spp, depth, "-", seed, &st);
        std::fprintf(stderr, "%llu rays in %.1f ms (%.1f Mrays/s)\n", (unsigned long long)(st.rays_closest + st.rays_shadow),
                     st.ms_total, (st.rays_closest + st.rays_shadow) / (st.ms_total * 1e3));
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
}
