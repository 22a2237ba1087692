"""Upper bound of what re-ordering secondary rays can buy: bounce rays (cosine-ish random directions from the
primary hit points of the bench scene) traced by rt_trace_closest in pixel order, binned by direction octant,
and sorted by origin cell + octant.  Kernel time only (HIP events around the launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ray_tracing_amd as rt
from ray_tracing_amd import scenes
desc = scenes.path_scene(levels=9)
scene = rt.Scene(desc); ctx = rt.Context(0); ctx.upload(scene)
w, h = 1920, 1080
org, dr = scenes.pixel_center_rays(scene.flat.camera, w, h)
rep = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hits = ctx.trace_closest(org, dr)
ok = hits["obj"] != rt.api.RT_MISS
p = (org[ok] + dr[ok] * hits["t"][ok, None]).astype(np.float32)
nrm = np.stack([hits["nx"][ok], hits["ny"][ok], hits["nz"][ok]], 1)
p = np.repeat(p, rep, 0); nrm = np.repeat(nrm, rep, 0)
rng = np.random.default_rng(0)
d = rng.normal(size=p.shape).astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True)
flip = (d * nrm).sum(1) < 0
nrm_ff = np.where(((dr[ok].repeat(rep, 0)) * nrm).sum(1)[:, None] > 0, -nrm, nrm)
d = np.where(((d * nrm_ff).sum(1) < 0)[:, None], -d, d).astype(np.float32)
n = len(p)
print("rays", n)

def run(order, label):
    o, dd = p[order], d[order]
    cols = [torch.from_numpy(np.ascontiguousarray(a[:, k])).cuda() for a in (o, dd) for k in range(3)]
    t = torch.empty(n, dtype=torch.float32, device="cuda"); pr = torch.empty(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ms = []
    for _ in range(4):
        ctx.trace_closest_device([c.data_ptr() for c in cols], n, dict(t=t.data_ptr(), prim=pr.data_ptr()))
        ms.append(ctx.last_kernel_ms())
    print(f"{label:34s} {min(ms):7.3f} ms  {n / min(ms) / 1e6:6.2f} Grays/s", flush=True)

ident = np.arange(n)
run(ident, "pixel order, random directions")
octant = (d[:, 0] < 0).astype(np.int64) | ((d[:, 1] < 0).astype(np.int64) << 1) | ((d[:, 2] < 0).astype(np.int64) << 2)
run(np.argsort(octant, kind="stable"), "stable partition by octant")
lo, hi = p.min(0), p.max(0)
def morton(bits):
    q = np.clip(((p - lo) / (hi - lo + 1e-9) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    m = np.zeros(n, np.int64)
    for b in range(bits):
        for a in range(3):
            m |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return m
for bits in (4, 6):
    m = morton(bits)
    run(np.argsort(m * 8 + octant, kind="stable"), f"morton {bits}b/axis + octant")
    run(np.argsort(octant * (1 << 3 * bits) + m, kind="stable"), f"octant major, morton {bits}b")
run(rng.permutation(n), "random permutation")
