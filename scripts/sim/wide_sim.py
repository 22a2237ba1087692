"""CPU comparison of the reference's binary BVH walk with the 4-wide exact walk (oracle/wide_model.cpp): same
triangle-test sequence?  how many node fetches / stack words?   python scripts/sim/wide_sim.py [levels] [n_rays]"""
import ctypes as C, os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
so = os.path.join(ROOT, "oracle", "libwide_model.so")
subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-o", so,
                os.path.join(ROOT, "oracle", "wide_model.cpp")], check=True)
lib = C.CDLL(so)
levels = int(sys.argv[1]) if len(sys.argv) > 1 else 7
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
scene = rt.Scene(scenes.tetrahedron_scene(levels=levels))
fl = scene.flat
b = scene.bvhs()[0]
assert b.node_offset == 0 and b.tri_offset == 0
nt = b.tri_count
v0 = np.ctypeslib.as_array(fl.tri_v0, shape=(fl.n_tris, 3))[:nt]
v1 = np.ctypeslib.as_array(fl.tri_v1, shape=(fl.n_tris, 3))[:nt]
v2 = np.ctypeslib.as_array(fl.tri_v2, shape=(fl.n_tris, 3))[:nt]
fp = lambda a: a.ctypes.data_as(C.c_void_p)

def run(name, org, dr, defer_internal=0):
    org = np.ascontiguousarray(org, np.float32); dr = np.ascontiguousarray(dr, np.float32)
    out = np.zeros((len(org), 9), np.uint32)
    bad = lib.wide_model_compare(fl.node_upper, fl.node_lower, fl.node_index, fl.node_num_prim, fl.tri_v0, fl.tri_v1, fl.tri_v2,
                                 fp(org), fp(dr), C.c_size_t(len(org)), C.c_float(1e-3), C.c_float(np.inf), defer_internal, fp(out))
    o = out.astype(np.int64)
    print(f"{name:28s} rays {len(org)} mismatches {bad}  binary fetches/ray {o[:,1].mean():.2f}  wide {o[:,2].mean():.2f} "
          f"(ratio {o[:,2].sum()/max(1,o[:,1].sum()):.3f})  loads bin {4*o[:,1].mean():.1f} wide {7*o[:,2].mean():.1f}  tri/ray {o[:,3].mean():.2f}  "
          f"max fetches bin {o[:,1].max()} wide {o[:,2].max()}  max steps bin {(o[:,1]+o[:,3]).max()} wide {(o[:,2]+o[:,4]).max()}  "
          f"stack bin max {o[:,7].max()} wide max {o[:,8].max()} mean {o[:,8].mean():.2f}  hits {(o[:,5]!=0xFFFFFFFF).mean():.2f}", flush=True)
    return out

rng = np.random.default_rng(0)
w = 640; h = scenes.image_height(w)
org, dr = scenes.pixel_center_rays(fl.camera, w, h)
idx = rng.choice(len(org), min(n_rays, len(org)), replace=False)
run("primary", org[idx], dr[idx])
run("primary defer-internal", org[idx], dr[idx], 1)
# secondary-like: start just off a random triangle's centroid, random direction
t = rng.integers(0, nt, n_rays)
c = ((v0[t] + v1[t] + v2[t]) / 3).astype(np.float32)
d = rng.normal(size=(n_rays, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
run("surface, random dir", c + 2e-3 * d, d)
run("surface, defer-internal", c + 2e-3 * d, d, 1)
# through the mesh: from outside aiming at random surface points
o2 = (rng.normal(size=(n_rays, 3)).astype(np.float32)); o2 = 3 * o2 / np.linalg.norm(o2, axis=1, keepdims=True)
d2 = c - o2; d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
run("outside -> surface point", o2, d2.astype(np.float32))
# degenerate: axis-aligned rays starting on vertex coordinates
ax = np.eye(3, dtype=np.float32)[rng.integers(0, 3, n_rays)] * rng.choice([-1, 1], (n_rays, 1)).astype(np.float32)
run("axis-aligned from vertices", v0[t] - 2 * ax, ax)
run("axis-aligned, o on vertex", v0[t], ax)
out = run("primary again", org[idx], dr[idx])
bad = np.nonzero(out[:, 0] == 0)[0]
print("bad rows", out[bad[:10]])
print("same hit among bad:", (out[bad,5]==out[bad,6]).mean(), "same ntri:", (out[bad,3]==out[bad,4]).mean())
