"""ctypes binding of oracle/liboracle.so -- the CPU restatement used as the CHECKER.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product
package (ray_tracing_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")


def build():
    subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)


if not os.path.exists(LIB):
    build()
lib = C.CDLL(LIB)
lib.orc_mesh_create.restype = C.c_void_p
lib.orc_mesh_create.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
lib.orc_mesh_destroy.argtypes = [C.c_void_p]
lib.orc_scene_create.restype = C.c_void_p
lib.orc_scene_create.argtypes = [C.c_void_p, C.c_void_p]
lib.orc_scene_destroy.argtypes = [C.c_void_p]
lib.orc_last_error.restype = C.c_char_p
lib.orc_initial_medium.argtypes = [C.c_void_p]
lib.orc_initial_medium.restype = C.c_int32
lib.orc_rng_f32.restype = C.c_float
lib.orc_rng_f32.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
lib.orc_pm.restype = C.c_float
lib.orc_pm.argtypes = [C.c_int, C.c_float]
lib.orc_vec_dot.restype = C.c_float
lib.orc_trace_closest.argtypes = [C.c_void_p] + [C.c_void_p] * 6 + [C.c_size_t, C.c_float, C.c_float] + [C.c_void_p] * 9
lib.orc_trace_any.argtypes = [C.c_void_p] + [C.c_void_p] * 6 + [C.c_size_t, C.c_float, C.c_float] + [C.c_void_p] * 3
lib.orc_bvh_sizes.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
lib.orc_bvh_copy.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 5


class OrcStats(C.Structure):
    _fields_ = [("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64), ("iterations", C.c_uint64),
                ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64), ("seconds", C.c_double),
                ("threads", C.c_uint32)]


lib.orc_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                           C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(OrcStats)]


lib.orc_ray_colour.argtypes = [C.c_void_p] + [C.c_void_p] * 6 + [C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                               C.c_void_p, C.c_void_p]
lib.orc_render_pixels.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                  C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.POINTER(OrcStats)]


def f3(v):
    return (C.c_float * 3)(*[float(np.float32(x)) for x in v])


def f9(m):
    return (C.c_float * 9)(*[float(np.float32(x)) for x in np.asarray(m).ravel()])


def rng_f32(seed, pixel, sample, bounce, dim):
    """Draw `dim` of (pixel, sample, bounce) as an f32 in [0, 1) (oracle/rng.h)."""
    return lib.orc_rng_f32(seed, pixel, sample, bounce, dim)


class OracleScene:
    """Engine::new on the CPU from a ray_tracing_amd.api.SceneDescription (own BVH build)."""

    def __init__(self, description):
        self._meshes = {}
        handles = []
        for m in description.meshes:
            if m is None:
                handles.append(None)
                continue
            key = id(m)
            if key not in self._meshes:
                v, t = m.arrays()
                h = lib.orc_mesh_create(v.ctypes.data, len(v), t.ctypes.data, len(t))
                self._meshes[key] = (h, v, t)
            handles.append(self._meshes[key][0])
        arr = (C.c_void_p * max(1, len(handles)))(*handles)
        d = description.c_struct()
        self._desc = (d, description)
        self.handle = lib.orc_scene_create(C.byref(d), arr)
        if not self.handle:
            raise RuntimeError(lib.orc_last_error().decode())

    def initial_medium(self):
        return lib.orc_initial_medium(self.handle)

    def bvh(self, obj):
        nn, npr = C.c_uint32(), C.c_uint32()
        assert lib.orc_bvh_sizes(self.handle, obj, C.byref(nn), C.byref(npr)) == 0
        out = dict(upper=np.empty((nn.value, 3), np.float32), lower=np.empty((nn.value, 3), np.float32),
                   index=np.empty(nn.value, np.uint32), num_prim=np.empty(nn.value, np.uint32),
                   indices=np.empty(npr.value, np.uint32))
        lib.orc_bvh_copy(self.handle, obj, *[out[k].ctypes.data for k in ("upper", "lower", "index", "num_prim", "indices")])
        return out

    @staticmethod
    def _cols(origins, directions):
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, np.float32).reshape(-1, 3)
        return [np.ascontiguousarray(o[:, k]) for k in range(3)] + [np.ascontiguousarray(d[:, k]) for k in range(3)]

    def trace_closest(self, origins, directions, t_min=1e-3, t_max=np.inf):
        cols = self._cols(origins, directions)
        n = len(cols[0])
        out = dict(t=np.empty(n, np.float32), prim=np.empty(n, np.uint32), obj=np.empty(n, np.uint32),
                   nx=np.empty(n, np.float32), ny=np.empty(n, np.float32), nz=np.empty(n, np.float32),
                   front_face=np.empty(n, np.uint8), n_int=np.empty(n, np.uint32), n_tri=np.empty(n, np.uint32))
        lib.orc_trace_closest(self.handle, *[c.ctypes.data for c in cols], n, t_min, t_max,
                              *[out[k].ctypes.data for k in ("t", "prim", "obj", "nx", "ny", "nz", "front_face", "n_int", "n_tri")])
        return out

    def trace_any(self, origins, directions, t_min=1e-3, t_max=np.inf):
        cols = self._cols(origins, directions)
        n = len(cols[0])
        out = dict(occluded=np.empty(n, np.uint8), n_int=np.empty(n, np.uint32), n_tri=np.empty(n, np.uint32))
        lib.orc_trace_any(self.handle, *[c.ctypes.data for c in cols], n, t_min, t_max,
                          out["occluded"].ctypes.data, out["n_int"].ctypes.data, out["n_tri"].ctypes.data)
        return out

    def render(self, width, height, spp, depth, seed=0, spp_first=0, rows=None, threads=1):
        out = np.zeros((height, width, 3), np.float32)
        st = OrcStats()
        r0, r1 = rows if rows is not None else (0, height)
        lib.orc_render(self.handle, width, height, spp_first, spp, depth, seed, r0, r1, threads, out.ctypes.data,
                       C.byref(st))
        return out, st

    def render_pixels(self, width, height, pixels, spp, depth, seed=0, spp_first=0, threads=1):
        px = np.ascontiguousarray(pixels, np.uint32)
        out = np.zeros((len(px), 3), np.float32)
        st = OrcStats()
        lib.orc_render_pixels(self.handle, width, height, spp_first, spp, depth, seed, px.ctypes.data, len(px), threads,
                              out.ctypes.data, C.byref(st))
        return out, st

    def camera_rays(self, u, v):
        """Camera::ray(u, v) (camera.rs:60-63) for arrays of image-plane positions."""
        u = np.ascontiguousarray(u, np.float32)
        v = np.ascontiguousarray(v, np.float32)
        cols = [np.empty(len(u), np.float32) for _ in range(6)]
        lib.orc_camera_rays(C.c_void_p(self.handle), C.c_void_p(u.ctypes.data), C.c_void_p(v.ctypes.data),
                            C.c_size_t(len(u)), *[C.c_void_p(c.ctypes.data) for c in cols])
        return np.stack(cols[:3], 1), np.stack(cols[3:], 1)

    def ray_colour(self, origins, directions, spp, depth, seed=0, spp_first=0):
        cols = self._cols(origins, directions)
        n = len(cols[0])
        out = np.zeros((n, 3), np.float32)
        st = OrcStats()
        lib.orc_ray_colour(self.handle, *[c.ctypes.data for c in cols], n, spp_first, spp, depth, seed, out.ctypes.data,
                           C.byref(st))
        return out, st

    def __del__(self):
        if getattr(self, "handle", None):
            lib.orc_scene_destroy(self.handle)
            self.handle = None
            for h, _, _ in self._meshes.values():
                lib.orc_mesh_destroy(h)
