"""ctypes binding of include/rt_amd.h (ray_tracing_amd/_lib/librt_amd.so).

Plumbing only: struct mirrors, array marshalling and error mapping.  There is no CPU fallback: if
the shared library is missing, importing this module raises, and every device entry point fails
with RtError when no GPU is present.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "_lib", "librt_amd.so")  # env: tuning builds only

RT_OK = 0
RT_MISS = 0xFFFFFFFF
SURF_MESH, SURF_SPHERE, SURF_PLANE, SURF_TRIANGLE, SURF_POINT = range(5)
MAT_DIFFUSE, MAT_DIELECTRIC, MAT_CONDUCTOR = range(3)
LIGHT_POINT, LIGHT_SPOT, LIGHT_AREA = range(3)
MEDIUM_HOMOGENEOUS, MEDIUM_CHECKERED = range(2)
MEDIUM_NONE, MEDIUM_AUTO = -1, -2
RENDER_COUNT_VISITS, RENDER_TIME_STAGES, RENDER_NO_TAIL, RENDER_COUNT_FETCHES = 1, 2, 4, 8
COUNT_REFERENCE, COUNT_FETCHES = 0, 1

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rt error {code}: {msg}")
        self.code = code


class ObjectDesc(C.Structure):
    _fields_ = [("surface_kind", C.c_uint32), ("mesh", C.c_void_p), ("surf", C.c_float * 9),
                ("has_transform", C.c_uint32), ("matrix", C.c_float * 9), ("translation", C.c_float * 3),
                ("material", C.c_uint32), ("has_medium_interface", C.c_uint32),
                ("medium_inside", C.c_int32), ("medium_outside", C.c_int32)]


class Material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("a", C.c_float * 3), ("b", C.c_float * 3)]


class Light(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("pos", C.c_float * 3), ("dir", C.c_float * 3), ("scalar", C.c_float),
                ("intensity", C.c_float * 3)]


class Medium(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("sigma_a", C.c_float * 3), ("sigma_s", C.c_float * 3),
                ("emission", C.c_float * 3), ("g", C.c_float), ("cell_size", C.c_float),
                ("sigma_a_black", C.c_float), ("sigma_s_black", C.c_float), ("emission_black", C.c_float * 3)]


class CameraDesc(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("direction", C.c_float * 3), ("aspect_ratio", C.c_float)]


class SceneDesc(C.Structure):
    _fields_ = [("n_objects", C.c_uint32), ("objects", C.POINTER(ObjectDesc)),
                ("n_materials", C.c_uint32), ("materials", C.POINTER(Material)),
                ("n_lights", C.c_uint32), ("lights", C.POINTER(Light)),
                ("n_media", C.c_uint32), ("media", C.POINTER(Medium)),
                ("camera", CameraDesc), ("initial_medium", C.c_int32)]


class BvhFlat(C.Structure):
    _fields_ = [("node_offset", C.c_uint32), ("node_count", C.c_uint32), ("tri_offset", C.c_uint32),
                ("tri_count", C.c_uint32), ("max_depth", C.c_uint32)]


class ObjectFlat(C.Structure):
    _fields_ = [("surface_kind", C.c_uint32), ("bvh", C.c_uint32), ("surf", C.c_float * 9),
                ("has_transform", C.c_uint32), ("inv_matrix", C.c_float * 9), ("inv_translation", C.c_float * 3),
                ("material", C.c_uint32), ("has_medium_interface", C.c_uint32),
                ("medium_inside", C.c_int32), ("medium_outside", C.c_int32)]


class CameraFlat(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("lower_left_corner", C.c_float * 3), ("horizontal", C.c_float * 3),
                ("vertical", C.c_float * 3)]


class SceneFlat(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("node_upper", f32p), ("node_lower", f32p), ("node_index", u32p),
                ("node_num_prim", u32p), ("n_tris", C.c_uint32), ("tri_v0", f32p), ("tri_v1", f32p),
                ("tri_v2", f32p), ("tri_prim_id", u32p), ("n_bvhs", C.c_uint32), ("bvhs", C.POINTER(BvhFlat)),
                ("n_objects", C.c_uint32), ("objects", C.POINTER(ObjectFlat)),
                ("n_materials", C.c_uint32), ("materials", C.POINTER(Material)),
                ("n_lights", C.c_uint32), ("lights", C.POINTER(Light)),
                ("n_media", C.c_uint32), ("media", C.POINTER(Medium)),
                ("camera", CameraFlat), ("initial_medium", C.c_int32)]


class Rays(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ox", "oy", "oz", "dx", "dy", "dz")]


class Hits(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("t", "prim", "obj", "nx", "ny", "nz", "front_face", "n_int", "n_tri")]


class Stats(C.Structure):
    _fields_ = [("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64), ("paths", C.c_uint64),
                ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64), ("nodes_visited_shadow", C.c_uint64),
                ("tris_tested_shadow", C.c_uint64), ("iterations", C.c_uint32),
                ("batches", C.c_uint32), ("ms_total", C.c_float), ("ms_generate", C.c_float),
                ("ms_extend", C.c_float), ("ms_shade", C.c_float), ("ms_shadow", C.c_float),
                ("ms_compact", C.c_float), ("launches_extend", C.c_uint64), ("launches_shadow", C.c_uint64),
                ("paths_tail", C.c_uint64), ("rays_closest_tail", C.c_uint64), ("rays_shadow_tail", C.c_uint64),
                ("nodes_visited_tail", C.c_uint64), ("tris_tested_tail", C.c_uint64), ("launches_tail", C.c_uint64),
                ("ms_tail", C.c_float), ("lean_kernels", C.c_uint32),
                ("nodes_visited_shadow_tail", C.c_uint64), ("tris_tested_shadow_tail", C.c_uint64),
                ("rays_shadow_skipped", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# every symbol include/rt_amd.h declares (checked by tests/test_abi.py)
EXPORTS = ["rt_mesh_create", "rt_mesh_read_stl", "rt_mesh_loop_subdivide", "rt_mesh_apply_transform",
           "rt_mesh_volume_centroid", "rt_mesh_counts", "rt_mesh_copy_out", "rt_mesh_destroy", "rt_scene_build",
           "rt_scene_view", "rt_scene_destroy", "rt_tonemap_u8", "rt_write_ppm", "rt_create", "rt_destroy",
           "rt_upload_scene", "rt_initial_medium", "rt_trace_closest", "rt_trace_any", "rt_trace_closest_noalloc", "rt_render", "rt_ray_colour",
           "rt_comm_unique_id", "rt_comm_create", "rt_comm_destroy", "rt_reduce_accum",
           "rt_pipeline_create", "rt_pipeline_destroy", "rt_pipeline_frames", "rt_pipeline_separate_queues", "rt_pipeline_context",
           "rt_pipeline_upload_scene", "rt_pipeline_comm_create", "rt_pipeline_submit", "rt_pipeline_wait",
           "rt_pipeline_create_hooked", "rt_pipeline_submit_hooked",
           "rt_set_max_paths_in_flight", "rt_set_visit_count_mode", "rt_last_kernel_ms", "rt_stream", "rt_last_error", "rt_version"]

if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} is missing: build it with `make -C ray_tracing_amd/csrc` "
                      "(or __graft_entry__.build()); there is no fallback path")


def _preload_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64.  Two HIP runtimes in one
    process cannot both own the GPU (whichever comes second sees no device), so when torch is
    installed its runtime is loaded first and librt_amd.so binds to it by SONAME."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        tl = os.path.join(os.path.dirname(spec.origin), "lib")
        for name in ("libhsa-runtime64.so", "libamdhip64.so"):
            path = os.path.join(tl, name)
            if os.path.exists(path):
                C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError:
        pass


_preload_hip_runtime()
lib = C.CDLL(LIB_PATH)
lib.rt_last_error.restype = C.c_char_p
lib.rt_version.restype = C.c_char_p
lib.rt_stream.restype = C.c_void_p
lib.rt_stream.argtypes = [C.c_void_p]
lib.rt_destroy.argtypes = [C.c_void_p]
lib.rt_destroy.restype = None
lib.rt_mesh_destroy.argtypes = [C.c_void_p]
lib.rt_mesh_destroy.restype = None
lib.rt_scene_destroy.argtypes = [C.c_void_p]
lib.rt_scene_destroy.restype = None
lib.rt_trace_closest.argtypes = [C.c_void_p, C.POINTER(Rays), C.c_size_t, C.c_float, C.c_float, C.POINTER(Hits)]
lib.rt_trace_closest_noalloc.argtypes = lib.rt_trace_closest.argtypes
lib.rt_trace_any.argtypes = [C.c_void_p, C.POINTER(Rays), C.c_size_t, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                             C.c_void_p]
lib.rt_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                          C.c_uint32, C.c_void_p, C.POINTER(Stats)]
lib.rt_ray_colour.argtypes = [C.c_void_p, C.POINTER(Rays), C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                              C.c_uint32, C.c_void_p, C.POINTER(Stats)]
lib.rt_comm_unique_id.argtypes = [C.c_void_p]
lib.rt_comm_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
lib.rt_comm_destroy.argtypes = [C.c_void_p]
lib.rt_comm_destroy.restype = None
lib.rt_reduce_accum.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]
lib.rt_upload_scene.argtypes = [C.c_void_p, C.POINTER(SceneFlat)]
lib.rt_pipeline_create.argtypes = [C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]
lib.rt_pipeline_destroy.argtypes = [C.c_void_p]
lib.rt_pipeline_destroy.restype = None
lib.rt_pipeline_frames.argtypes = [C.c_void_p]
lib.rt_pipeline_frames.restype = C.c_uint32
lib.rt_pipeline_separate_queues.argtypes = [C.c_void_p]
lib.rt_pipeline_separate_queues.restype = C.c_uint32
lib.rt_pipeline_context.argtypes = [C.c_void_p, C.c_uint32]
lib.rt_pipeline_context.restype = C.c_void_p
lib.rt_pipeline_upload_scene.argtypes = [C.c_void_p, C.POINTER(SceneFlat)]
lib.rt_pipeline_comm_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
lib.rt_pipeline_submit.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                   C.c_uint32, C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
lib.rt_pipeline_wait.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(Stats)]
PIPELINE_HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64)  # (user, ctx, slot, ticket) -> status


class PipelineHooks(C.Structure):
    _fields_ = [("render", PIPELINE_HOOK), ("reduce", PIPELINE_HOOK), ("user", C.c_void_p)]


lib.rt_pipeline_create_hooked.argtypes = [C.c_int, C.c_uint32, C.POINTER(PipelineHooks), C.POINTER(C.c_void_p)]
lib.rt_pipeline_submit_hooked.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
lib.rt_set_max_paths_in_flight.argtypes = [C.c_void_p, C.c_uint64]
lib.rt_set_visit_count_mode.argtypes = [C.c_void_p, C.c_uint32]


def _check(code):
    if code != RT_OK:
        raise RtError(code, lib.rt_last_error().decode())


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(x):
    """numpy array, torch tensor (host or device), int address or None -> address"""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    return x.data_ptr()


class Mesh:
    """TriangleMesh (triangle_mesh.rs:16-19) held by the host library."""

    def __init__(self, handle):
        self.handle = C.c_void_p(handle)

    @classmethod
    def from_arrays(cls, vertices, triangles):
        v = _f32(vertices).reshape(-1, 3)
        t = np.ascontiguousarray(triangles, dtype=np.uint32).reshape(-1, 3)
        h = C.c_void_p()
        _check(lib.rt_mesh_create(v.ctypes.data_as(f32p), C.c_uint32(len(v)), t.ctypes.data_as(u32p),
                                  C.c_uint32(len(t)), C.byref(h)))
        return cls(h.value)

    @classmethod
    def read_stl(cls, path):
        h = C.c_void_p()
        _check(lib.rt_mesh_read_stl(path.encode(), C.byref(h)))
        return cls(h.value)

    def loop_subdivide(self, divisions):
        h = C.c_void_p()
        _check(lib.rt_mesh_loop_subdivide(self.handle, C.c_uint32(divisions), C.byref(h)))
        return Mesh(h.value)

    def apply_transform(self, matrix, translation):
        m = (C.c_float * 9)(*np.asarray(matrix, dtype=np.float32).ravel())
        t = (C.c_float * 3)(*np.asarray(translation, dtype=np.float32).ravel())
        _check(lib.rt_mesh_apply_transform(self.handle, m, t))

    def volume_centroid(self):
        out = (C.c_float * 3)()
        _check(lib.rt_mesh_volume_centroid(self.handle, out))
        return np.array(out[:], dtype=np.float32)

    def counts(self):
        nv, nt = C.c_uint32(), C.c_uint32()
        _check(lib.rt_mesh_counts(self.handle, C.byref(nv), C.byref(nt)))
        return nv.value, nt.value

    def arrays(self):
        nv, nt = self.counts()
        v = np.empty((nv, 3), np.float32)
        t = np.empty((nt, 3), np.uint32)
        _check(lib.rt_mesh_copy_out(self.handle, v.ctypes.data_as(f32p), t.ctypes.data_as(u32p)))
        return v, t

    def __del__(self):
        if getattr(self, "handle", None) and lib is not None:  # lib is None once the interpreter shuts down
            lib.rt_mesh_destroy(self.handle)
            self.handle = None


def _arr3(x):
    return (C.c_float * 3)(*[float(np.float32(v)) for v in x])


class SceneDescription:
    """Python-side builder for rt_scene_desc: what src/main.rs writes by hand."""

    def __init__(self):
        self.objects, self.materials, self.lights, self.media = [], [], [], []
        self.meshes = []  # per object: Mesh or None (also consumed by the oracle binding in tests/)
        self.camera = CameraDesc()
        self.initial_medium = MEDIUM_AUTO

    # materials
    def diffuse(self, reflectance):
        self.materials.append(Material(MAT_DIFFUSE, _arr3(reflectance), _arr3((0, 0, 0))))
        return len(self.materials) - 1

    def dielectric(self, refractive_index):
        self.materials.append(Material(MAT_DIELECTRIC, _arr3((refractive_index, 0, 0)), _arr3((0, 0, 0))))
        return len(self.materials) - 1

    def conductor(self, n, k):
        self.materials.append(Material(MAT_CONDUCTOR, _arr3(n), _arr3(k)))
        return len(self.materials) - 1

    # media
    def homogeneous_medium(self, sigma_a, sigma_s, g=0.0, emission=(0, 0, 0)):
        self.media.append(Medium(MEDIUM_HOMOGENEOUS, _arr3(sigma_a), _arr3(sigma_s), _arr3(emission), g, 0.0, 0.0,
                                 0.0, _arr3((0, 0, 0))))
        return len(self.media) - 1

    def checkered_medium(self, cell_size, sa_w, ss_w, em_w, sa_b, ss_b, em_b, g):
        self.media.append(Medium(MEDIUM_CHECKERED, _arr3((sa_w, 0, 0)), _arr3((ss_w, 0, 0)), _arr3(em_w), g,
                                 cell_size, sa_b, ss_b, _arr3(em_b)))
        return len(self.media) - 1

    # lights
    def point_light(self, pos, intensity):
        self.lights.append(Light(LIGHT_POINT, _arr3(pos), _arr3((0, 0, 0)), 0.0, _arr3(intensity)))

    def spotlight(self, pos, direction, cos_angle, intensity):
        self.lights.append(Light(LIGHT_SPOT, _arr3(pos), _arr3(direction), cos_angle, _arr3(intensity)))

    def area_light(self, center, radius, radiance):
        self.lights.append(Light(LIGHT_AREA, _arr3(center), _arr3((0, 0, 0)), radius, _arr3(radiance)))

    # objects (Object::new, object.rs:17)
    def _add(self, kind, surf, material, transform, medium, mesh=None):
        o = ObjectDesc()
        o.surface_kind = kind
        o.mesh = mesh.handle if mesh is not None else None
        s = list(surf) + [0.0] * (9 - len(surf))
        o.surf = (C.c_float * 9)(*[float(np.float32(v)) for v in s])
        if transform is not None:
            m, t = transform
            o.has_transform = 1
            o.matrix = (C.c_float * 9)(*np.asarray(m, dtype=np.float32).ravel())
            o.translation = _arr3(t)
        o.material = material
        if medium is not None:
            o.has_medium_interface = 1
            o.medium_inside, o.medium_outside = [MEDIUM_NONE if m is None else m for m in medium]
        else:
            o.medium_inside = o.medium_outside = MEDIUM_NONE
        self.objects.append(o)
        self.meshes.append(mesh)
        return len(self.objects) - 1

    def add_mesh(self, mesh, material, transform=None, medium=None):
        return self._add(SURF_MESH, [], material, transform, medium, mesh)

    def add_sphere(self, center, radius, material, transform=None, medium=None):
        return self._add(SURF_SPHERE, list(center) + [radius], material, transform, medium)

    def add_plane(self, point, normal, material, transform=None, medium=None):
        return self._add(SURF_PLANE, list(point) + list(normal), material, transform, medium)

    def add_triangle(self, v0, v1, v2, material, transform=None, medium=None):
        return self._add(SURF_TRIANGLE, list(v0) + list(v1) + list(v2), material, transform, medium)

    def set_camera(self, origin, direction, aspect_ratio):
        self.camera = CameraDesc(_arr3(origin), _arr3(direction), aspect_ratio)

    def look_at(self, origin, point, aspect_ratio):  # Camera::look_at (camera.rs:55-57)
        o = np.asarray(origin, np.float32)
        self.set_camera(o, np.asarray(point, np.float32) - o, aspect_ratio)

    def c_struct(self):
        d = SceneDesc()
        self._keep = [(ObjectDesc * max(1, len(self.objects)))(*self.objects),
                      (Material * max(1, len(self.materials)))(*self.materials),
                      (Light * max(1, len(self.lights)))(*self.lights),
                      (Medium * max(1, len(self.media)))(*self.media)]
        d.n_objects, d.objects = len(self.objects), self._keep[0]
        d.n_materials, d.materials = len(self.materials), self._keep[1]
        d.n_lights, d.lights = len(self.lights), self._keep[2]
        d.n_media, d.media = len(self.media), self._keep[3]
        d.camera = self.camera
        d.initial_medium = self.initial_medium
        return d


class Scene:
    """Flattened scene owned by the host library (rt_scene_build)."""

    def __init__(self, description):
        self.description = description
        d = description.c_struct()
        h = C.c_void_p()
        _check(lib.rt_scene_build(C.byref(d), C.byref(h)))
        self.handle = h
        self.flat = SceneFlat()
        _check(lib.rt_scene_view(self.handle, C.byref(self.flat)))

    def nodes(self):
        n = self.flat.n_nodes
        g = lambda p, shape, dt: np.ctypeslib.as_array(p, shape=shape).astype(dt, copy=True) if n else np.zeros(shape, dt)
        return dict(upper=g(self.flat.node_upper, (n, 3), np.float32), lower=g(self.flat.node_lower, (n, 3), np.float32),
                    index=g(self.flat.node_index, (n,), np.uint32), num_prim=g(self.flat.node_num_prim, (n,), np.uint32))

    def prim_ids(self):
        n = self.flat.n_tris
        return np.ctypeslib.as_array(self.flat.tri_prim_id, shape=(n,)).copy() if n else np.zeros(0, np.uint32)

    def bvhs(self):
        return [BvhFlat.from_buffer_copy(self.flat.bvhs[i]) for i in range(self.flat.n_bvhs)]

    def __del__(self):
        if getattr(self, "handle", None) and lib is not None:
            lib.rt_scene_destroy(self.handle)
            self.handle = None


class Context:
    """One GPU context (Engine): rt_create + rt_upload_scene."""

    def __init__(self, device=0, _borrowed=None):
        self._owner = None
        if _borrowed is not None:  # a FramePipeline's context: the pipeline destroys it
            self.handle, self._owner = C.c_void_p(_borrowed[0]), _borrowed[1]
            return
        h = C.c_void_p()
        _check(lib.rt_create(C.c_int(device), C.byref(h)))
        self.handle = h

    def upload(self, scene):
        flat = scene.flat if isinstance(scene, Scene) else scene
        _check(lib.rt_upload_scene(self.handle, C.byref(flat)))

    def initial_medium(self):
        v = C.c_int32()
        _check(lib.rt_initial_medium(self.handle, C.byref(v)))
        return v.value

    def stream(self):
        return lib.rt_stream(self.handle)

    def last_kernel_ms(self):
        v = C.c_float()
        _check(lib.rt_last_kernel_ms(self.handle, C.byref(v)))
        return v.value

    def set_max_paths_in_flight(self, n):
        _check(lib.rt_set_max_paths_in_flight(self.handle, n))

    def set_visit_count_mode(self, mode):
        _check(lib.rt_set_visit_count_mode(self.handle, mode))

    def trace_closest(self, origins, directions, t_min=1e-3, t_max=np.inf, counters=False):
        o = _f32(origins).reshape(-1, 3)
        d = _f32(directions).reshape(-1, 3)
        n = len(o)
        cols = [np.ascontiguousarray(o[:, k]) for k in range(3)] + [np.ascontiguousarray(d[:, k]) for k in range(3)]
        rays = Rays(*[c.ctypes.data for c in cols])
        out = dict(t=np.empty(n, np.float32), prim=np.empty(n, np.uint32), obj=np.empty(n, np.uint32),
                   nx=np.empty(n, np.float32), ny=np.empty(n, np.float32), nz=np.empty(n, np.float32),
                   front_face=np.empty(n, np.uint8))
        if counters:
            out["n_int"] = np.empty(n, np.uint32)
            out["n_tri"] = np.empty(n, np.uint32)
        hits = Hits(*[out[k].ctypes.data if k in out else None for k, _ in Hits._fields_])
        _check(lib.rt_trace_closest(self.handle, C.byref(rays), n, t_min, t_max, C.byref(hits)))
        return out

    def trace_any(self, origins, directions, t_min=1e-3, t_max=np.inf, counters=False):
        o = _f32(origins).reshape(-1, 3)
        d = _f32(directions).reshape(-1, 3)
        n = len(o)
        cols = [np.ascontiguousarray(o[:, k]) for k in range(3)] + [np.ascontiguousarray(d[:, k]) for k in range(3)]
        rays = Rays(*[c.ctypes.data for c in cols])
        occ = np.empty(n, np.uint8)
        ni = np.empty(n, np.uint32) if counters else None
        nt = np.empty(n, np.uint32) if counters else None
        _check(lib.rt_trace_any(self.handle, C.byref(rays), n, t_min, t_max, occ.ctypes.data, _ptr(ni), _ptr(nt)))
        return dict(occluded=occ, n_int=ni, n_tri=nt)

    def trace_closest_device(self, rays6, n, hits_ptrs, t_min=1e-3, t_max=np.inf):
        """Raw form: rays6 = six device/host addresses, hits_ptrs = dict name -> address."""
        rays = Rays(*rays6)
        hits = Hits(*[hits_ptrs.get(k) for k, _ in Hits._fields_])
        _check(lib.rt_trace_closest(self.handle, C.byref(rays), n, t_min, t_max, C.byref(hits)))

    def trace_closest_noalloc(self, rays6, n, hits_ptrs, t_min=1e-3, t_max=np.inf):
        """rt_trace_closest_noalloc: device addresses only, no allocation inside the library."""
        rays = Rays(*rays6)
        hits = Hits(*[hits_ptrs.get(k) for k, _ in Hits._fields_])
        _check(lib.rt_trace_closest_noalloc(self.handle, C.byref(rays), n, t_min, t_max, C.byref(hits)))

    def trace_any_device(self, rays6, n, occluded, t_min=1e-3, t_max=np.inf, n_int=None, n_tri=None):
        rays = Rays(*rays6)
        _check(lib.rt_trace_any(self.handle, C.byref(rays), n, t_min, t_max, occluded, n_int, n_tri))

    def render(self, width, height, spp, depth, seed=0, spp_first=0, flags=0, out=None):
        """Engine::render without the PPM text. out: optional device/host buffer address or array."""
        stats = Stats()
        if out is None:
            out = np.empty((height, width, 3), np.float32)
        _check(lib.rt_render(self.handle, width, height, spp_first, spp, depth, seed, flags, _ptr(out),
                             C.byref(stats)))
        return out, stats

    def ray_colour(self, origins, directions, spp, depth, seed=0, spp_first=0, flags=0):
        """Engine::ray_colour for a batch of rays: (n, 3) sums over the samples, and the statistics of the run."""
        o = _f32(origins).reshape(-1, 3)
        d = _f32(directions).reshape(-1, 3)
        n = len(o)
        cols = [np.ascontiguousarray(o[:, k]) for k in range(3)] + [np.ascontiguousarray(d[:, k]) for k in range(3)]
        rays = Rays(*[c.ctypes.data for c in cols])
        out = np.zeros((n, 3), np.float32)
        stats = Stats()
        _check(lib.rt_ray_colour(self.handle, C.byref(rays), n, spp_first, spp, depth, seed, flags, _ptr(out),
                                 C.byref(stats)))
        return out, stats

    def close(self):
        if getattr(self, "handle", None) and lib is not None:
            if getattr(self, "_owner", None) is None:
                lib.rt_destroy(self.handle)
            self.handle = None

    def __del__(self):
        self.close()


COMM_ID_BYTES = 128
PIPELINE_NO_REDUCE = -2


def comm_unique_id():
    """rt_comm_unique_id: 128 bytes created on ONE rank; hand them to every rank, then Comm(ctx, id, world, rank)."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    _check(lib.rt_comm_unique_id(buf))
    return bytes(buf)


class Comm:
    """RCCL communicator of one rank (one process per GPU), bound to a Context's device and stream."""

    def __init__(self, ctx, unique_id, world_size, rank):
        assert len(unique_id) == COMM_ID_BYTES
        self._ctx = ctx  # the communicator uses the context's stream: keep it alive
        h = C.c_void_p()
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        _check(lib.rt_comm_create(ctx.handle, buf, world_size, rank, C.byref(h)))
        self.handle = h

    def reduce_accum(self, rgb_sum, width, height, root=0):
        """Sum the ranks' device accumulators in place onto `root` (root < 0: onto every rank)."""
        _check(lib.rt_reduce_accum(self.handle, _ptr(rgb_sum), width, height, root))

    def close(self):
        if getattr(self, "handle", None) and lib is not None:
            lib.rt_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        self.close()


class FramePipeline:
    """rt_pipeline: up to `frames_in_flight` frames rendered at once on one GPU, each by the unchanged rt_render."""

    def __init__(self, device=0, frames_in_flight=2, render=None, reduce=None):
        """render / reduce: hooks (ctx_address, slot, ticket) -> status for rt_pipeline_create_hooked; device < 0 with
        hooks runs them without a GPU context (ctx_address is None)."""
        h = C.c_void_p()
        if render is None:
            _check(lib.rt_pipeline_create(C.c_int(device), frames_in_flight, C.byref(h)))
        else:
            def wrap(fn):
                def hook(_user, ctx, slot, ticket):
                    try:
                        return int(fn(ctx, slot, ticket) or 0)
                    except Exception:  # nothing may unwind into the worker thread
                        import traceback
                        traceback.print_exc()
                        return -1
                return PIPELINE_HOOK(hook)
            self._hooks = PipelineHooks(wrap(render), wrap(reduce) if reduce is not None else PIPELINE_HOOK(0), None)
            _check(lib.rt_pipeline_create_hooked(C.c_int(device), frames_in_flight, C.byref(self._hooks), C.byref(h)))
        self.handle = h
        self.frames_in_flight = lib.rt_pipeline_frames(h)
        self.separate_queues = lib.rt_pipeline_separate_queues(h)  # slots whose streams were seen to run side by side

    def upload(self, scene):
        flat = scene.flat if isinstance(scene, Scene) else scene
        _check(lib.rt_pipeline_upload_scene(self.handle, C.byref(flat)))

    def context(self, slot=0):
        """The slot's context, borrowed (settings; plain renders while no frame is in flight)."""
        h = lib.rt_pipeline_context(self.handle, slot)
        if not h:
            raise RtError(-1, lib.rt_last_error().decode())
        return Context(_borrowed=(h, self))

    def comm_create(self, unique_ids, world_size, rank):
        """unique_ids: one comm_unique_id() per slot, the same on every rank.  Collective."""
        assert len(unique_ids) == self.frames_in_flight and all(len(u) == COMM_ID_BYTES for u in unique_ids)
        buf = (C.c_uint8 * (COMM_ID_BYTES * len(unique_ids))).from_buffer_copy(b"".join(unique_ids))
        _check(lib.rt_pipeline_comm_create(self.handle, buf, world_size, rank))

    def submit(self, width, height, spp, depth, seed=0, spp_first=0, flags=0, out=None, reduce_root=PIPELINE_NO_REDUCE):
        """Queues one frame; returns its ticket.  `out` (array or address) must stay alive until wait(ticket)."""
        t = C.c_uint64()
        _check(lib.rt_pipeline_submit(self.handle, width, height, spp_first, spp, depth, seed, flags, _ptr(out), reduce_root,
                                      C.byref(t)))
        return t.value

    def submit_hooked(self, reduce=False):
        t = C.c_uint64()
        _check(lib.rt_pipeline_submit_hooked(self.handle, int(bool(reduce)), C.byref(t)))
        return t.value

    def wait(self, ticket):
        stats = Stats()
        _check(lib.rt_pipeline_wait(self.handle, ticket, C.byref(stats)))
        return stats

    def close(self):
        if getattr(self, "handle", None) and lib is not None:
            lib.rt_pipeline_destroy(self.handle)
            self.handle = None

    def __del__(self):
        self.close()


def version():
    return lib.rt_version().decode()


def build_id():
    """Hash of the sources the loaded library was built from (last word of rt_version())."""
    return version().split()[-1]


def tonemap_u8(rgb_sum, spp):
    a = _f32(rgb_sum)
    h, w = a.shape[0], a.shape[1]
    out = np.empty((h, w, 3), np.uint8)
    _check(lib.rt_tonemap_u8(a.ctypes.data_as(f32p), C.c_uint32(w), C.c_uint32(h), C.c_uint32(spp),
                             out.ctypes.data_as(u8p)))
    return out


def write_ppm(path, rgb8):
    a = np.ascontiguousarray(rgb8, dtype=np.uint8)
    _check(lib.rt_write_ppm(path.encode(), a.ctypes.data_as(u8p), C.c_uint32(a.shape[1]), C.c_uint32(a.shape[0])))
