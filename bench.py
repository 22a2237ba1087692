#!/usr/bin/env python3
"""bench.py -- Mrays/s of the path-tracing hot path on MI355X (one process per GPU).

    python bench.py [--gpus N --steps K --warmup W] [--workload render|trace]

workload "render" (default): BASELINE.json configs[2] -- full path tracing (diffuse/dielectric BSDFs,
    area-light NEE/MIS, Russian roulette) of the 1M-triangle scene at 1920x1080, 64 spp, depth 16.
    One step = one frame.  With N GPUs the 64 samples of the frame are sharded by global sample index
    (rank r renders sample_shard(64, N, r) of every pixel) and the radiance accumulators are summed to
    rank 0 with one RCCL reduce over xGMI through the C ABI (rt_reduce_accum), inside the timed region:
    STRONG scaling, the BASELINE metric "ms/frame @1080p 64 spp at 1/2/4/8 GPUs".  --scaling weak gives
    every rank its own 64 samples instead (the frame gains samples).  The K timed frames go through a frame pipeline
    (rt_pipeline_*, --frames-in-flight, default 2 at >= 32 spp per rank and 4 below): whole frames overlap on the GPU,
    each rendered -- and reduced -- by the same calls as a lone frame, so `value` and `ms_per_step` are THROUGHPUT
    figures; the latency of a lone frame (`ms_frame_alone`) and the kernel time the roofline uses are measured on frames
    rendered one at a time directly after the timed region (--frames-in-flight 1 times those instead).
workload "config5": BASELINE.json configs[4]'s scene -- a plane and 10 subdivided tetrahedra, each its OWN
    mesh and BVH (10.5 M triangles, 1.1 GB of geometry: beyond the 256 MiB Infinity Cache); --instanced
    shares one BVH instead.  --width 3840 for 4K.  Used for the cache-resident / out-of-cache comparison.
workload "trace": BASELINE.json configs[1] -- primary + shadow rays only, same mesh, 1080p 1 spp.
    One step = one closest-hit launch over 2 073 600 primary rays + one any-hit launch over the
    shadow rays of the pixels that hit.

Prints ONE JSON line (rank 0).  `value` = rays traced by all ranks / max-over-ranks wall time, rays
= Engine::hit + Engine::hit_bool calls.  `roofline` is about the dominant kernel (closest-hit
traversal, k_extend): `frac` = MEMORY-SIDE bytes per launch (rocprofv3 counters of this build, committed under
profiles/) / the kernel's duration (HIP events on the library's own stream, measured in this run) / 8 TB/s.  The
SURVEY.md 8(d) figure -- algorithmic bytes 48 + 64*N_int + 40*N_tri per ray / kernel time -- is printed beside it as
`requested_GBs`: those are bytes the lanes REQUEST, most of them served by the vector L1 and the L2 (it exceeds the HBM
peak), compared under `levels` with the measured rate of per-lane 64-byte gathers.  `cpu_baseline` times the oracle
(oracle/, the C++ restatement of the reference: "port") on a bounded sample of the same workload on
this box's host cores.  The oracle is used here only as that baseline/cross-check.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# MEASURED ceiling of the traversal's access shape when every line hits the vector L1: 64 lanes each reading their own
# 64-byte record with four global_load_dwordx4 (scripts/microbench/gather_lanes.hip, rows L64/1 of
# profiles/r03/gather_lanes*.txt: 202-259 G records/s on tables <= 0.5 MiB, median 214 = 13.7 TB/s; 58 G records/s =
# 3.7 TB/s from a 64 MiB table).  The requested-bytes figure is compared with this, not with a datasheet product.
GATHER_L1_GBS = 214.0 * 64


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)  # frames in flight: the first and last frames of a run have the GPU to themselves
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["render", "trace", "config5"], default="render")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--instanced", action="store_true", help="config5: ten instances of ONE mesh/BVH instead of ten meshes")
    ap.add_argument("--copies", type=int, default=10)
    ap.add_argument("--torch-reduce", action="store_true", help="N>1: reduce with torch.distributed instead of rt_reduce_accum")
    ap.add_argument("--levels", type=int, default=9, help="Loop subdivisions of the tetrahedron (9 = 1 048 576 tris)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--depth", type=int, default=16)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=1)
    ap.add_argument("--max-paths", type=int, default=0, help="paths in flight per batch (0 = library default)")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frames rendered at once in the timed region (rt_pipeline_*); 1 = one synchronous rt_render per step; "
                         "0 = 2 at >= 32 spp per rank, 4 below")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    import ray_tracing_amd as rt
    from ray_tracing_amd import scenes

    # RT_BENCH_FORCE_DIST=1: take the N>1 code path (RCCL init, reduce, max-over-ranks) with one rank --
    # the only rehearsal of it a 1-GPU box allows
    use_dist = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1"
    if use_dist:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    width = args.width
    height = scenes.image_height(width)
    t_setup = time.time()
    if args.workload == "trace":
        desc = scenes.tetrahedron_scene(levels=args.levels)
    elif args.workload == "config5":
        desc = scenes.multi_mesh_scene(levels=args.levels, copies=args.copies, distinct=not args.instanced)
    else:
        desc = scenes.path_scene(levels=args.levels)
    scene = rt.Scene(desc)
    pipe, n_flight = None, 1
    if args.workload != "trace":
        from ray_tracing_amd.distributed import sample_shard
        if args.scaling == "strong":  # the 64-spp frame, sharded by global sample index
            spp_first, spp_count = sample_shard(args.spp, world, rank)
        else:                          # every rank its own args.spp samples
            spp_first, spp_count = rank * args.spp, args.spp
        # the same on every rank (the ranks' shares may differ by one sample): communicators are created slot by slot
        n_flight = args.frames_in_flight or (2 if (args.spp // world if args.scaling == "strong" else args.spp) >= 32 else 4)
        if args.torch_reduce:
            n_flight = 1
    if n_flight > 1:  # one context per frame in flight; slot 0's context also serves the untimed passes
        pipe = rt.FramePipeline(local_rank, n_flight)
        pipe.upload(scene)
        ctx = pipe.context(0)
        if args.max_paths:
            for s in range(n_flight):
                pipe.context(s).set_max_paths_in_flight(args.max_paths)
    else:
        ctx = rt.Context(local_rank)
        ctx.upload(scene)
        if args.max_paths:
            ctx.set_max_paths_in_flight(args.max_paths)
    setup_s = time.time() - t_setup
    n_tris = scene.flat.n_tris

    result = {}
    if args.workload == "trace":
        org, dr = scenes.pixel_center_rays(scene.flat.camera, width, height)
        n = len(org)
        cols = [torch.from_numpy(np.ascontiguousarray(a[:, k])).to(dev) for a in (org, dr) for k in range(3)]
        t = torch.empty(n, dtype=torch.float32, device=dev)
        prim = torch.empty(n, dtype=torch.int32, device=dev)
        obj = torch.empty(n, dtype=torch.int32, device=dev)
        ni = torch.zeros(n, dtype=torch.int32, device=dev)
        nt = torch.zeros(n, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        rays6 = [c.data_ptr() for c in cols]
        hp = dict(t=t.data_ptr(), prim=prim.data_ptr(), obj=obj.data_ptr())
        # untimed counting pass: the timed kernel's own fetches for the algorithmic-bytes figure + shadow-ray set-up
        ctx.set_visit_count_mode(rt.api.COUNT_FETCHES)
        ctx.trace_closest_device(rays6, n, dict(hp, n_int=ni.data_ptr(), n_tri=nt.data_ptr()))
        n_int_total, n_tri_total = int(ni.sum().item()), int(nt.sum().item())
        hit = obj != -1
        O = torch.stack(cols[:3], 1)
        D = torch.stack(cols[3:], 1)
        P = (O[hit] + D[hit] * t[hit, None])
        lp = torch.tensor(list(desc.lights[0].pos), dtype=torch.float32, device=dev)
        L = lp - P
        L = L / torch.sqrt((L * L).sum(1, keepdim=True))
        scols = [P[:, k].contiguous() for k in range(3)] + [L[:, k].contiguous() for k in range(3)]
        ns = len(P)
        occ = torch.empty(ns, dtype=torch.uint8, device=dev)
        srays6 = [c.data_ptr() for c in scols]
        torch.cuda.synchronize()
        rays_per_step = n + ns
        closest_ms, any_ms = [], []

        def step(record):
            ctx.trace_closest_device(rays6, n, hp)
            if record:
                closest_ms.append(ctx.last_kernel_ms())
            ctx.trace_any_device(srays6, ns, occ.data_ptr())
            if record:
                any_ms.append(ctx.last_kernel_ms())

        for _ in range(args.warmup):
            step(False)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(True)
        barrier()
        elapsed = time.perf_counter() - t0
        alg_bytes = 48 * n + 64 * n_int_total + 40 * n_tri_total
        kern_ms = float(np.mean(closest_ms))
        result.update(kernel="k_trace_closest", kernel_ms=kern_ms, any_ms=float(np.mean(any_ms)),
                      alg_bytes=alg_bytes, n_int_mean=n_int_total / n, n_tri_mean=n_tri_total / n,
                      rays_closest=n, rays_shadow=ns)
        workload = f"configs[1]: primary+shadow rays, {n_tris}-triangle subdivided tetrahedron, {width}x{height}, 1 spp"
    else:
        accs = [torch.zeros(height, width, 3, dtype=torch.float32, device=dev) for _ in range(n_flight)]
        acc = accs[0]
        torch.cuda.synchronize()
        # the accumulator reduce: RCCL through the C ABI on the library's own stream (ordered behind rt_render
        # without a host sync); torch.distributed only carries the unique id and the barriers
        comm, reduce_kind = None, "none"
        if use_dist:
            reduce_kind = "torch.distributed"
            if not args.torch_reduce:
                # Every rank first checks that it can reach RCCL through the library at all (rt_comm_unique_id binds
                # librccl and calls ncclGetUniqueId: cheap, not collective) and the ranks vote BEFORE anyone enters the
                # collective ncclCommInitRank: a rank that fails there leaves the others blocked inside it.
                my_id, can = None, 1
                try:
                    my_id = rt.comm_unique_id()
                except rt.RtError as e:
                    can = 0
                    print(f"bench: rank {rank}: RCCL not reachable through the library ({e}); reducing with torch.distributed", file=sys.stderr)
                ok = torch.tensor([can], device=dev)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 1:
                    # rank 0's ids are the communicators': one per frame in flight
                    ids = my_id + b"".join(rt.comm_unique_id() for _ in range(n_flight - 1))
                    uid = torch.tensor(list(ids), dtype=torch.uint8, device=dev)
                    dist.broadcast(uid, src=0)
                    ids = bytes(uid.cpu().tolist())
                    try:
                        if pipe is not None:
                            pipe.comm_create([ids[128 * s:128 * (s + 1)] for s in range(n_flight)], world, rank)
                            comm = pipe
                        else:
                            comm = rt.Comm(ctx, ids, world, rank)
                        reduce_kind = "rt_reduce_accum (RCCL, C ABI)"
                    except rt.RtError as e:  # plumbing fallback for the collective only; labelled in the output
                        print(f"bench: rt_comm_create failed on rank {rank} ({e}); reducing with torch.distributed", file=sys.stderr)
            ok = torch.tensor([1 if (comm is not None or args.torch_reduce) else 0], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:  # some rank fell back: all ranks must, to one synchronous frame per step
                if comm is not None and comm is not pipe:
                    comm.close()
                comm, reduce_kind = None, "torch.distributed"
                if pipe is not None:
                    ctx = rt.Context(local_rank)
                    ctx.upload(scene)
                    pipe.close()
                    pipe, n_flight = None, 1
        # untimed counting passes (also the warm-up of every kernel): the fetches the timed kernels perform
        # (root-box skip and top-level cull active) price the roofline; the reference's visits are listed beside
        _, st_count = ctx.render(width, height, spp_count, args.depth, args.seed, spp_first,
                                 flags=rt.api.RENDER_COUNT_FETCHES, out=acc.data_ptr())
        _, st_ref = ctx.render(width, height, spp_count, args.depth, args.seed, spp_first,
                               flags=rt.api.RENDER_COUNT_VISITS, out=acc.data_ptr())
        last = {}

        ext_ms, ext_launches, mix_ms, mix_launches = [], [], [], []

        def record_stats(st):  # k_extend is timed by HIP events on the library's stream in every render call
            ext_ms.append(st.ms_extend)
            ext_launches.append(st.launches_extend)
            mix_ms.append(st.ms_tail)  # mixed (shadow + next bounce) and tail launches: timed the same way
            mix_launches.append(st.launches_tail)

        def step(record, all_stages=False, reduce=True):
            _, st = ctx.render(width, height, spp_count, args.depth, args.seed, spp_first,
                               flags=rt.api.RENDER_TIME_STAGES if all_stages else 0, out=acc.data_ptr())
            if not reduce or pipe is not None:  # (a pipeline's communicators reduce its own frames only)
                pass
            elif comm is not None:
                comm.reduce_accum(acc.data_ptr(), width, height, 0)  # returns when the sum is complete
            elif use_dist:
                dist.reduce(acc, dst=0, op=dist.ReduceOp.SUM)
                torch.cuda.synchronize()  # the next render overwrites acc on the library's stream
            last["st"] = st
            if record:
                record_stats(st)

        def steps_in_flight(n, record):
            """n frames through the pipeline: all queued at once, up to n_flight of them on the GPU at any time; frame k
            renders into accumulator k % n_flight (the slot's own), followed by its reduce.  Returns when all are done."""
            root = 0 if comm is not None else rt.api.PIPELINE_NO_REDUCE
            tickets = [pipe.submit(width, height, spp_count, args.depth, args.seed, spp_first,
                                   out=accs[k % n_flight].data_ptr(), reduce_root=root) for k in range(n)]
            for t in tickets:
                st = pipe.wait(t)
                if record:
                    record_stats(st)

        if pipe is not None:
            steps_in_flight(args.warmup, False)
            barrier()
            t0 = time.perf_counter()
            steps_in_flight(args.steps, True)
            barrier()
            elapsed = time.perf_counter() - t0
            # with frames in flight an event pair on one stream also spans the other frames' kernels: the kernel times the
            # roofline uses, and the latency of a lone frame, come from frames rendered one at a time right here
            in_flight = {"kernel_ms_extend": float(np.mean(ext_ms)) / max(1, int(np.mean(ext_launches))) if ext_ms else None,
                         "kernel_ms_mixed": float(np.mean(mix_ms)) / max(1, int(np.mean(mix_launches))) if mix_ms else None}
            ext_ms, ext_launches, mix_ms, mix_launches = [], [], [], []
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                step(True, reduce=False)
            ms_frame_alone = (time.perf_counter() - t1) / 3 * 1e3
        else:
            for _ in range(args.warmup):
                step(False)
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step(True)
            barrier()
            elapsed = time.perf_counter() - t0
            in_flight, ms_frame_alone = None, elapsed / args.steps * 1e3
        # one more (untimed) step with HIP events around every stage, for the breakdown in `detail`
        step(False, all_stages=True, reduce=False)
        st = last["st"]
        rays_per_step = st.rays_closest + st.rays_shadow - st.rays_shadow_skipped  # rays actually traced

        def alg_bytes(c, ext):  # 48 + 64 N_int + 40 N_tri per ray of the extend / of the mixed+tail launches
            if ext:
                return (48 * (c.rays_closest - c.rays_closest_tail) + 64 * (c.nodes_visited - c.nodes_visited_tail)
                        + 40 * (c.tris_tested - c.tris_tested_tail))
            return (48 * (c.rays_closest_tail + c.rays_shadow_tail) + 64 * (c.nodes_visited_tail + c.nodes_visited_shadow_tail)
                    + 40 * (c.tris_tested_tail + c.tris_tested_shadow_tail))

        launches = max(1, int(np.mean(ext_launches)))
        ms_extend_timed = float(np.mean(ext_ms))  # average over the K timed frames
        n_mix = max(1, int(np.mean(mix_launches)))
        ms_mix_timed = float(np.mean(mix_ms))
        lean_sfx = "_lean" if st.lean_kernels else ""
        kernels = {
            "k_extend" + lean_sfx: {"kernel_ms": ms_extend_timed / launches, "alg_bytes": alg_bytes(st_count, True) / launches,
                                    "ref_bytes": alg_bytes(st_ref, True) / launches,
                                    "launches": int(launches), "ms_per_frame": ms_extend_timed},
            "k_mixed" + lean_sfx: {"kernel_ms": ms_mix_timed / n_mix, "alg_bytes": alg_bytes(st_count, False) / n_mix,
                                   "ref_bytes": alg_bytes(st_ref, False) / n_mix,
                                   "launches": int(n_mix), "ms_per_frame": ms_mix_timed},
        }
        gbs = lambda v, key="alg_bytes": v[key] / (v["kernel_ms"] * 1e-3) / 1e9 if v["kernel_ms"] > 0 else None
        result["kernels"] = {k: {"fetched_GBs": round(gbs(v), 1) if gbs(v) else None,
                                 "reference_visit_GBs": round(gbs(v, "ref_bytes"), 1) if gbs(v) else None,
                                 "launches": v["launches"], "ms_per_launch": round(v["kernel_ms"], 4),
                                 "ms_per_frame": round(v["ms_per_frame"], 3), "alg_bytes_per_launch": int(v["alg_bytes"]),
                                 "reference_visit_bytes_per_launch": int(v["ref_bytes"])}
                             for k, v in kernels.items()}
        ext = kernels["k_extend" + lean_sfx]
        result.update(kernel="k_extend" + lean_sfx, kernel_ms=ext["kernel_ms"], alg_bytes=ext["alg_bytes"],
                      launches=int(launches), ms_extend=ms_extend_timed, ms_shadow=st.ms_shadow, ms_shade=st.ms_shade,
                      ms_generate=st.ms_generate, ms_publish=st.ms_compact, ms_mixed_tail=st.ms_tail,
                      paths_tail=int(st.paths_tail), ms_device_total=st.ms_total,
                      n_int_mean=st_count.nodes_visited / max(1, st_count.rays_closest),
                      n_tri_mean=st_count.tris_tested / max(1, st_count.rays_closest),
                      n_int_reference=st_ref.nodes_visited / max(1, st_ref.rays_closest),
                      rays_closest=int(st.rays_closest), rays_shadow=int(st.rays_shadow),
                      rays_shadow_not_traced=int(st.rays_shadow_skipped),
                      iterations=int(st.iterations), spp_this_rank=int(spp_count), reduce=reduce_kind,
                      frames_in_flight=int(n_flight), ms_frame_alone=ms_frame_alone,
                      separate_hw_queues=(int(pipe.separate_queues) if pipe is not None else None),
                      kernel_ms_with_frames_in_flight=in_flight)
        if args.workload == "config5":
            workload = (f"configs[4] scene: plane + {args.copies} subdivided tetrahedra, "
                        f"{'instances of one BVH' if args.instanced else 'each its own mesh and BVH'}, {n_tris} triangles, "
                        f"{width}x{height}, {args.spp} spp, depth {args.depth}")
        else:
            workload = (f"configs[2]: path tracing diffuse+dielectric, area-light NEE/MIS, RR, {n_tris} triangles, "
                        f"{width}x{height}, {args.spp} spp, depth {args.depth}")
        if comm is not None and comm is not pipe:
            comm.close()
        if pipe is not None:
            ctx = None
            pipe.close()

    # max over ranks of the timed region
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        rr = torch.tensor([float(rays_per_step)], dtype=torch.float64, device=dev)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
        total_rays_per_step = float(rr.item())
    else:
        total_rays_per_step = float(rays_per_step)

    if rank == 0:
        stream_copy = stream_copy_gbs(torch, dev)
        traffic, traffic_detail = committed_traffic(args, result["kernel"], rt.api.build_id(), world)
        ms_per_step = elapsed / args.steps * 1e3
        value = total_rays_per_step * args.steps / elapsed / 1e6
        requested = result["alg_bytes"] / (result["kernel_ms"] * 1e-3) / 1e9
        # Two figures for one launch.  MEMORY SIDE: the bytes the counters see leave the L2 towards Infinity Cache and HBM
        # (FETCH_SIZE x the factor calibrated for the kernel's access shape + WRITE_SIZE, per launch, from the committed
        # counter profile of THIS build and workload) / this run's kernel time, against the 8 TB/s HBM peak: the headline
        # `frac`.  REQUESTED: SURVEY.md 8(d)'s algorithmic bytes / kernel time -- what the lanes ask for; the vector L1 and
        # the L2 serve most of it, so it exceeds the HBM peak and is no fraction of it; it is compared with the measured
        # rate of the same access shape on L1-resident data (levels.l1_gather_path).
        geom_bytes = scene.flat.n_tris * 48 + max(0, scene.flat.n_nodes - scene.flat.n_bvhs) * 32
        levels = {"l1_gather_path": {"achieved_GBs": round(requested, 1), "peak_GBs": round(GATHER_L1_GBS, 1),
                                     "frac": round(requested / GATHER_L1_GBS, 4),
                                     "bytes": "requested by the lanes (algorithmic)",
                                     "peak_source": "measured: 64 lanes x own 64-B record, L1-resident table (profiles/r03/gather_lanes*.txt, L64/1)"}}
        if traffic is not None:
            t_gbs = traffic / (result["kernel_ms"] * 1e-3) / 1e9
            levels["memory_side"] = {"achieved_GBs": round(t_gbs, 1), "peak_GBs": HBM_PEAK_GBS, "frac": round(t_gbs / HBM_PEAK_GBS, 4),
                                     "frac_of_stream_copy": round(t_gbs / stream_copy, 4),
                                     "bytes": "FETCH_SIZE x calibrated factor + WRITE_SIZE (fabric side of L2: Infinity Cache + HBM)"}
            roof_achieved, roof_note = t_gbs, ("achieved = memory-side counter bytes per launch (traffic) / kernel time of this run; peak = HBM 8 TB/s. "
                                               "requested_GBs is SURVEY 8(d)'s algorithmic figure: requested bytes, cache-served")
        else:  # no counter profile of this build: only SURVEY 8(d)'s literal figure can be given (requested bytes, cache-served)
            roof_achieved, roof_note = requested, ("NO counter profile of this build is committed: achieved = SURVEY 8(d)'s algorithmic bytes / kernel "
                                                   "time, i.e. REQUESTED bytes, mostly served by L1/L2 -- not HBM traffic, frac is not a utilisation")
        out = {
            "metric": "Mrays/s (path-segment rays, whole node)", "value": round(value, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": args.scaling if args.workload != "trace" else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "seed": args.seed,
                       "parallelism": (f"sample-sharded x{world} ({args.scaling}), {result.get('reduce', 'none')}" if world > 1 else "1 GPU"),
                       "ms_per_frame": round(ms_per_step, 3),
                       **({"frames_in_flight": result["frames_in_flight"],
                           "ms_frame_alone": round(result["ms_frame_alone"], 3),
                           "frames_in_flight_note": "ms_per_step / ms_per_frame / value are throughput over the K timed frames, up to "
                                                    "frames_in_flight of them on the GPU at once (rt_pipeline_*); ms_frame_alone is "
                                                    "one frame rendered alone" + (", without its reduce" if world > 1 else "")}
                          if result.get("frames_in_flight", 1) > 1 else {}),
                       "scene_setup_s": round(setup_s, 1),
                       "geometry_bytes": int(geom_bytes), "build": rt.api.build_id()},
            "roofline": {"bound": "hbm", "achieved": round(roof_achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(roof_achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "requested_GBs": round(requested, 1), "requested_over_hbm_peak": round(requested / HBM_PEAK_GBS, 4),
                         "note": roof_note,
                         "levels": levels,
                         "traffic_unit": "memory-side bytes per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, committed profile of this build)",
                         "traffic_detail": traffic_detail,
                         "stream_copy_GBs": round(stream_copy, 1),
                         "kernel": result["kernel"], "kernel_ms": round(result["kernel_ms"], 4),
                         **({"frac_with_frames_in_flight": round(traffic / (result["kernel_ms_with_frames_in_flight"]["kernel_ms_extend"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                            if traffic is not None and (result.get("kernel_ms_with_frames_in_flight") or {}).get("kernel_ms_extend") else {}),
                         **({"kernel_ms_note": "launch duration with ONE frame on the GPU (3 frames rendered one at a time right after the "
                                               "timed region; bench.py --frames-in-flight 1 times exactly those, and the committed rocprofv3 "
                                               "summary is of that command): with frames in flight the launches of different frames share "
                                               "the CUs, and an event pair on one stream spans both (detail.kernel_ms_with_frames_in_flight; frac_with_frames_in_flight "
                                               "divides the same traffic by that stretched duration)"}
                            if result.get("frames_in_flight", 1) > 1 else {}),
                         "alg_bytes_per_launch": int(result["alg_bytes"]),
                         "alg_bytes_note": "48 + 64 N_int + 40 N_tri per ray with N_int / N_tri counted on the TIMED kernel's own fetches "
                                           "(root-box skip and top-level cull active)",
                         "n_int_per_ray": round(result["n_int_mean"], 3), "n_tri_per_ray": round(result["n_tri_mean"], 3),
                         **({"traversal_kernels": result["kernels"]} if "kernels" in result else {})},
            "detail": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in result.items()
                       if k not in ("kernel", "alg_bytes", "kernels", "frames_in_flight")},
        }
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(args, desc, scene, width, height)
        os.write(_JSON_FD, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


def stream_copy_gbs(torch, dev):
    """Measured device-to-device copy rate (read + write bytes) on this GPU: the practical HBM ceiling
    SURVEY.md 8(d) asks to report beside the 8 TB/s spec peak."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty(n, dtype=torch.uint8, device=dev)
    b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def committed_traffic(args, kernel, build, world):
    """Memory-side bytes per launch of `kernel` from the rocprofv3 --pmc passes committed under profiles/
    (scripts/pmc_traffic.sh; counters cannot be read from inside this process).  None unless a committed
    measurement exists for exactly this workload AND this build of the library (a profile of other code says
    nothing about this run)."""
    import glob
    if world != 1:
        return None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "**", "pmc_traffic*.json"), recursive=True), reverse=True):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        w = d.get("workload", {})
        if d.get("build") != build:
            continue
        if (w.get("name") == args.workload and w.get("levels") == args.levels and w.get("width") == args.width
                and w.get("spp") == args.spp and w.get("depth") == args.depth
                and w.get("instanced", False) == bool(getattr(args, "instanced", False)) and kernel in d.get("kernels", {})):
            k = d["kernels"][kernel]
            return k["hbm_bytes_per_launch"], {"fetch_bytes": k["fetch_bytes"], "write_bytes": k["write_bytes"],
                                               "fetch_factor": k.get("fetch_factor"),
                                               "launches": k["launches"], "l2_hit_rate": k.get("l2_hit_rate"),
                                               "source": os.path.relpath(f, ROOT), "build": d.get("build"), "note": d.get("note", "")}
    return None, None


def cpu_baseline(args, desc, scene, width, height):
    """Oracle (C++ restatement of the reference; the reference itself is Rust and cannot be built
    here) on a bounded sample of the same workload, on this box's host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    from ray_tracing_amd import scenes
    orc = ob.OracleScene(desc)
    if args.workload == "trace":
        org, dr = scenes.pixel_center_rays(scene.flat.camera, width, height)
        rows = np.arange(0, height, 4)
        idx = (rows[:, None] * width + np.arange(width)[None, :]).ravel()
        o, d = org[idx], dr[idx]
        t0 = time.perf_counter()
        h = orc.trace_closest(o, d)
        hit = h["obj"] != 0xFFFFFFFF
        p = (o[hit] + d[hit] * h["t"][hit, None]).astype(np.float32)
        lp = np.array(desc.lights[0].pos[:], np.float32)
        l = (lp - p).astype(np.float32)
        l = (l / np.sqrt((l * l).sum(1, dtype=np.float32))[:, None]).astype(np.float32)
        orc.trace_any(p, l)
        dt = time.perf_counter() - t0
        rays = len(o) + len(p)
        return {"value": round(rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": "port",
                "sample": f"every 4th image row of the same ray batch ({rays} rays, {dt:.1f} s)"}
    # bounded sample: 2 bands of 8 image rows (a third and two thirds up the frame) at the full spp -- about 2 M
    # camera paths, ~3 s on one core -- so that the GPU region stays a visible share of the run; the same sample
    # again on all host cores
    bands = [(int(height * (k + 1) / 3) - 4, int(height * (k + 1) / 3) + 4) for k in range(2)]

    pixels = np.concatenate([np.arange(r0 * width, r1 * width, dtype=np.uint32) for r0, r1 in bands])

    def run(threads):  # pixels are dealt to the workers one at a time
        _, st = orc.render_pixels(width, height, pixels, args.spp, args.depth, args.seed, 0, threads=threads)
        return st.rays_closest + st.rays_shadow, st.seconds

    rays, secs = run(1)
    ncores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    rays_all, secs_all = run(ncores)
    return {"value": round(rays / secs / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": "port",
            "sample": f"16 image rows (2 bands of 8) of the same frame at {args.spp} spp ({rays} rays, {secs:.1f} s); "
                      "1 thread = the reference's own execution model",
            "all_cores": {"value": round(rays_all / secs_all / 1e6, 4), "unit": "Mrays/s", "cores": ncores,
                          "seconds": round(secs_all, 1)}}


_JSON_FD = 1


if __name__ == "__main__":
    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner at its first
    # init when NCCL_DEBUG is set on the box): everything but the result line goes to stderr.
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)
    main()
