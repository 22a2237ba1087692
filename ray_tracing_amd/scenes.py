"""Scene descriptions for the BASELINE.json configurations.

These restate the *parameters* of the scene functions in the reference's src/main.rs (cited per
function) plus the synthetic benchmark scenes SURVEY.md section 8d defines.  They only build
rt_scene_desc structs; geometry processing happens in the host library.
"""
import math

import numpy as np

from .api import Mesh, SceneDescription

ASPECT_RATIO = np.float32(16.0) / np.float32(9.0)  # src/main.rs:12
F = np.float32
PI = F(math.pi)

IDENTITY = np.eye(3, dtype=np.float32)

# src/main.rs:88-96
TETRA_VERTICES = np.array([[1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]], np.float32)
TETRA_TRIANGLES = np.array([[0, 1, 2], [1, 3, 2], [0, 3, 1], [0, 2, 3]], np.uint32)


def new_polar(r, theta, phi):  # Vec3D::new_polar (vec3d.rs:53-59), f32 arithmetic
    r, theta, phi = F(r), F(theta), F(phi)
    st, ct = F(math.sin(theta)), F(math.cos(theta))
    sp, cp = F(math.sin(phi)), F(math.cos(phi))
    return np.array([r * st * cp, r * st * sp, r * ct], np.float32)


_mesh_cache = {}


def tetrahedron_mesh(levels):
    """Tetrahedron of src/main.rs:88-96 Loop-subdivided `levels` times and pushed to the limit
    surface (4**(levels+1) triangles; levels=9 is the 1 048 576-triangle mesh of config 2)."""
    if levels not in _mesh_cache:
        base = Mesh.from_arrays(TETRA_VERTICES, TETRA_TRIANGLES)
        _mesh_cache[levels] = base.loop_subdivide(levels)
    return _mesh_cache[levels]


def volumetric_sphere():
    """Config 1: scene_volumetric_sphere, src/main.rs:25-83."""
    s = SceneDescription()
    mat = s.dielectric(1.0)
    med = s.checkered_medium(0.5, 0.2, 0.1, (5.0, 0.0, 0.0), 0.1, 0.5, (0.0, 0.0, 0.0), 0.0)
    s.add_sphere((0, 0, 0), 1.6, mat, medium=(med, None))
    s.area_light((0.0, 100.0, 0.0), 80.0, (1.0, 1.0, 1.0))
    s.set_camera((0.1, -5.0, 0.1), (0.0, 1.0, 0.0), ASPECT_RATIO)
    return s


def tetrahedron_scene(levels=9, camera_distance=0.55):
    """Config 2: one subdivided tetrahedron (Diffuse (1,0.5,0.5), src/main.rs:97-99), the ground
    plane (src/main.rs:104-111) and the point light (src/main.rs:114-118); camera on the polar
    direction of src/main.rs:121 looking at the origin, moved in so the mesh fills the frame."""
    s = SceneDescription()
    red = s.diffuse((1.0, 0.5, 0.5))
    grey = s.diffuse((0.8, 0.8, 0.8))
    s.add_mesh(tetrahedron_mesh(levels), red)
    s.add_plane((0.0, 0.0, -0.6), (0.0, 0.0, 1.0), grey)
    s.point_light(F(10.0) * np.array([10.0, 2.0, 10.0], np.float32), (60000.0, 60000.0, 60000.0))
    loc = new_polar(camera_distance, F(0.3) * PI, F(0.0))
    s.set_camera(loc, -loc, ASPECT_RATIO)
    return s


def path_scene(levels=9, glass_levels=6, medium=False, sigma_s=0.02):
    """Config 3 (and 4 with medium=True): the config-2 mesh as Diffuse, a second subdivided
    tetrahedron as Dielectric{1.5} (the role baby_yoda.stl plays in src/main.rs:223-226, replaced by
    a synthetic mesh because the reference's assets are not shipped), ground plane Diffuse 0.8,
    sphere AreaLight far away (pattern of src/main.rs:271-277), optional HomogeneousMedium inside
    the dielectric mesh (src/main.rs:227-235 with a non-zero sigma_s to exercise Scatter)."""
    s = SceneDescription()
    red = s.diffuse((1.0, 0.5, 0.5))
    grey = s.diffuse((0.8, 0.8, 0.8))
    glass = s.dielectric(1.5)
    s.add_mesh(tetrahedron_mesh(levels), red)
    med = None
    if medium:
        m = s.homogeneous_medium((0.03, 0.03, 0.01), (sigma_s, sigma_s, sigma_s), 0.0, (0.0, 0.0, 0.0))
        med = (m, None)
    s.add_mesh(tetrahedron_mesh(glass_levels), glass,
               transform=(IDENTITY * F(0.8), (0.1, -0.55, -0.07)), medium=med)
    s.add_plane((0.0, 0.0, -0.32), (0.0, 0.0, 1.0), grey)
    s.area_light((2000.0, 1000.0, 2000.0), 1000.0, (5.0, 5.0, 5.0))
    loc = new_polar(1.05, F(0.33) * PI, F(-0.15) * PI)
    s.look_at(loc, (0.05, -0.25, -0.1), ASPECT_RATIO)
    return s


def _rotation_xyz(ax, ay, az):
    cx, sx, cy, sy, cz, sz = math.cos(ax), math.sin(ax), math.cos(ay), math.sin(ay), math.cos(az), math.sin(az)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return (rz @ ry @ rx).astype(np.float32)


def distinct_tetrahedron_mesh(levels, i):
    """Copy i of the subdivided tetrahedron as a mesh of its OWN: vertices rotated and stretched per copy, so that
    TriangleMesh::build_bvh gives every copy a different tree (src/main.rs:149-159 builds one mesh and one BVH per
    object).  Copy 0 is the plain mesh."""
    key = ("distinct", levels, i)
    if key not in _mesh_cache:
        v, t = tetrahedron_mesh(levels).arrays()
        m = Mesh.from_arrays(v, t)
        if i:
            rot = _rotation_xyz(0.37 * i, 0.61 * i, 0.23 * i)
            stretch = np.diag([1.0 + 0.02 * (i % 3), 1.0 - 0.015 * (i % 4), 1.0 + 0.01 * (i % 5)]).astype(np.float32)
            m.apply_transform((rot @ stretch).astype(np.float32), (0.0, 0.0, 0.0))
        _mesh_cache[key] = m
    return _mesh_cache[key]


def multi_mesh_scene(levels=9, copies=10, distinct=False):
    """Config 5: `copies` subdivided tetrahedra, each its own Object with a translation (pattern of
    src/main.rs:149-159), over a ground plane under a sphere area light.  distinct=False: instances of ONE mesh
    and BVH (device geometry stays one mesh's size); distinct=True: every object has its own mesh and its own
    BVH, as the reference builds them -- 10 x 1 048 576 triangles = 1.1 GB of child pairs and triangles, beyond
    the 256 MiB Infinity Cache."""
    s = SceneDescription()
    grey = s.diffuse((0.8, 0.8, 0.8))
    glass = s.dielectric(1.5)
    s.add_plane((0.0, 0.0, -0.32), (0.0, 0.0, 1.0), grey)
    for i in range(copies):
        mesh = distinct_tetrahedron_mesh(levels, i) if distinct else tetrahedron_mesh(levels)
        col = s.diffuse((0.3 + 0.07 * (i % 10), 0.5, 1.0 - 0.07 * (i % 10)))
        s.add_mesh(mesh, glass if i % 5 == 4 else col,
                   transform=(IDENTITY, (0.0, 0.65 * (i - (copies - 1) * 0.5), 0.0)))
    s.area_light((2000.0, 1000.0, 2000.0), 1000.0, (5.0, 5.0, 5.0))
    loc = new_polar(0.9 + 0.22 * copies, F(0.3) * PI, F(0.0))
    s.set_camera(loc, -loc, ASPECT_RATIO)
    return s


def primitives_scene():
    """scene_primitives, src/main.rs:302-397 (spheres, a triangle, Conductor, Dielectric, 2 area lights)."""
    s = SceneDescription()
    s.add_sphere((-2.0, 0.0, 0.5), 0.5, s.diffuse((1.0, 0.01, 0.01)))
    s.add_sphere((-0.5, 0.0, 0.8), 0.75, s.conductor((1, 1, 1), (1, 1, 1)))
    s.add_sphere((1.5, 0.0, 1.0), 1.0, s.diffuse((0.01, 0.01, 1.0)))
    s.add_triangle((-4.5, 2.0, 0.2), (-2.0, 3.0, 0.2), (-3.0, 2.5, 3.2), s.conductor((1, 1, 1), (1, 1, 1)))
    s.add_sphere((0.0, -1.6, 0.8), 0.75, s.dielectric(1.5))
    s.add_sphere((0.0, 0.0, -100.0), 100.0, s.diffuse((0.5, 0.5, 0.5)))
    s.area_light((0.0, 110.0, 0.0), 100.0, (2.0, 2.0, 2.0))
    s.area_light((2.5, -2.5, 0.5), 0.5, (0.1, 3.0, 0.1))
    s.set_camera((-2.5, -4.0, 3.0), (0.5, 1.0, -0.5), ASPECT_RATIO)
    return s


def baby_yoda_scene(stl_path, sigma_s=0.0, extra_mesh_levels=None):
    """scene_baby_yoda, src/main.rs:214-299: the STL mesh (read_stl, stl.rs:8-143; 4 120 triangles) as
    Dielectric{1.5} with a HomogeneousMedium inside (sigma_a (0.03, 0.03, 0.01); sigma_s 0 in the reference),
    ground plane at the mesh's lowest vertex, a floating diffuse ball (unit sphere under scale_translate(5,
    (-4, 6, 59))), one far sphere AreaLight.  `extra_mesh_levels`: SURVEY.md's config 3 puts the subdivided
    tetrahedron (Diffuse (1, 0.5, 0.5)) beside it -- added scaled x48 next to the figure when given."""
    s = SceneDescription()
    mesh = Mesh.read_stl(stl_path)
    verts = mesh.arrays()[0]
    ground = F(np.min(verts[:, 2]))
    centre = mesh.volume_centroid()
    glass = s.dielectric(1.5)
    med = s.homogeneous_medium((0.03, 0.03, 0.01), (sigma_s, sigma_s, sigma_s), 0.0, (0.0, 0.0, 0.0))
    s.add_mesh(mesh, glass, medium=(med, None))
    s.add_plane((0.0, 0.0, float(ground)), (0.0, 0.0, 1.0), s.diffuse((0.8, 0.8, 0.8)))
    ball = s.diffuse(tuple(np.array([0.8, 0.2, 0.2], np.float32) * F(2.0)))
    s.add_sphere((0.0, 0.0, 0.0), 1.0, ball, transform=(IDENTITY * F(5.0), (-4.0, 6.0, 59.0)))
    # camera: centre + 90 * new_polar(1, pi/2, 0.4 pi) + 40 Z, looking at the centroid (:252-260)
    d = new_polar(1.0, PI / F(2.0), F(0.4) * PI)
    origin = ((centre + F(90.0) * d).astype(np.float32) + F(40.0) * np.array([0, 0, 1], np.float32)).astype(np.float32)
    s.look_at(origin, centre, ASPECT_RATIO)
    away = (centre - origin).astype(np.float32)  # Camera::get_direction: normalised (camera.rs:44-46)
    n2 = (away[0] * away[0] + away[1] * away[1]) + away[2] * away[2]
    away = (away * (F(1.0) / np.sqrt(F(n2)))).astype(np.float32)
    z = np.array([0, 0, 1], np.float32)
    left = np.cross(z, away).astype(np.float32)
    away[2] = F(0.0)
    lc = (((centre + F(2000.0) * z).astype(np.float32) + F(2000.0) * left).astype(np.float32) - F(1000.0) * away).astype(np.float32)
    s.area_light(tuple(lc), 1000.0, (5.0, 5.0, 5.0))
    if extra_mesh_levels is not None:
        pos = (centre - F(60.0) * left).astype(np.float32)
        s.add_mesh(tetrahedron_mesh(extra_mesh_levels), s.diffuse((1.0, 0.5, 0.5)),
                   transform=(IDENTITY * F(48.0), (float(pos[0]), float(pos[1]), float(ground) + 17.0)))
    return s


def teapot_scene(stl_path, divisions=1):
    """scene_teapot, src/main.rs:165-211: utah_teapot.stl through read_stl (stl.rs:8-143; 9 438 triangles, an OPEN mesh:
    the Loop rules for boundary vertices and edges apply, subdivision_surface.rs:169-173, :218-232), loop_subdivide(.., 1)
    (37 752 triangles), Diffuse (0.9, 0.3, 0.3); ground plane at the subdivided mesh's lowest vertex; PointLight at
    100 * (1, 0, 1); camera at centroid + new_polar(18, 0.3 pi, 0.5 pi) looking at the volume centroid."""
    s = SceneDescription()
    mesh = Mesh.read_stl(stl_path).loop_subdivide(divisions)
    verts = mesh.arrays()[0]
    ground = F(np.min(verts[:, 2]))
    s.add_mesh(mesh, s.diffuse((0.9, 0.3, 0.3)))
    s.add_plane((0.0, 0.0, float(ground)), (0.0, 0.0, 1.0), s.diffuse((0.8, 0.8, 0.8)))
    centre = mesh.volume_centroid()
    d = new_polar(18.0, F(0.3) * PI, F(0.5) * PI)
    s.look_at((centre + d).astype(np.float32), centre, ASPECT_RATIO)
    s.point_light(F(100.0) * np.array([1.0, 0.0, 1.0], np.float32), (60000.0, 60000.0, 60000.0))
    return s


def spotlight_scene(levels=3):
    """Spotlight (lights.rs:49-75) next to a PointLight: a subdivided tetrahedron and a glass ball over a ground
    plane.  The cone (cos_angle 0.9) covers the mesh and part of the plane, so shading points fall on both sides
    of the `-dot(direction, dir) > cos_angle` test; the spotlight's intensity is NOT divided by the squared
    distance (the reference normalises before it takes d2, lights.rs:59-61), hence the modest value."""
    s = SceneDescription()
    red = s.diffuse((1.0, 0.5, 0.5))
    grey = s.diffuse((0.8, 0.8, 0.8))
    s.add_mesh(tetrahedron_mesh(levels), red)
    s.add_plane((0.0, 0.0, -0.6), (0.0, 0.0, 1.0), grey)
    s.add_sphere((0.9, -0.9, -0.25), 0.35, s.dielectric(1.5))
    pos = np.array([2.0, -1.5, 4.0], np.float32)
    aim = (np.array([0.2, -0.2, -0.3], np.float32) - pos).astype(np.float32)
    aim = (aim / np.sqrt((aim * aim).sum(dtype=np.float32))).astype(np.float32)
    s.spotlight(pos, aim, 0.9, (2.5, 2.4, 2.0))
    s.point_light((-30.0, -20.0, 40.0), (900.0, 900.0, 1200.0))
    loc = new_polar(4.0, F(0.3) * PI, F(-0.4) * PI)
    s.look_at(loc, (0.2, -0.3, -0.3), ASPECT_RATIO)
    return s


def image_height(width, aspect=ASPECT_RATIO):
    """(image_width as f32 / aspect_ratio) as usize, path_integrator.rs:285"""
    return int(F(width) / F(aspect))


def pixel_center_rays(camera_flat, width, height):
    """Camera::ray (camera.rs:60-63) at u=(i+0.5)/(w-1), v=(j+0.5)/(h-1) in f32, for the
    primary-ray parity batches (pixel order j*w+i)."""
    o = np.array(camera_flat.origin[:], np.float32)
    llc = np.array(camera_flat.lower_left_corner[:], np.float32)
    hor = np.array(camera_flat.horizontal[:], np.float32)
    ver = np.array(camera_flat.vertical[:], np.float32)
    i = np.arange(width, dtype=np.float32)
    j = np.arange(height, dtype=np.float32)
    u = ((i + F(0.5)) / F(width - 1)).astype(np.float32)
    v = ((j + F(0.5)) / F(height - 1)).astype(np.float32)
    uu, vv = np.meshgrid(u, v)  # [h, w]
    uu = uu.reshape(-1, 1).astype(np.float32)
    vv = vv.reshape(-1, 1).astype(np.float32)
    d = ((llc + uu * hor).astype(np.float32) + vv * ver).astype(np.float32) - o
    d = d.astype(np.float32)
    n2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32) + d[:, 2] * d[:, 2]
    r = (F(1.0) / np.sqrt(n2.astype(np.float32))).astype(np.float32)
    d = (d * r[:, None]).astype(np.float32)
    origins = np.broadcast_to(o, d.shape).copy()
    return origins, d
