#!/bin/bash
# bench.py at the driver's K / W with 1, 2 (default at 64 spp) and 4 frames in flight, the forced-distributed rehearsal
# (RCCL through the C ABI with one rank) and the 8-spp share of an 8-GPU run; run on the GPU box
set -e
O=gpurun_out/flight; mkdir -p $O
for f in 1 2 4; do
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --frames-in-flight $f > $O/bench_f$f.json 2> $O/bench_f$f.err
done
RT_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 \
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_dist.json 2> $O/bench_dist.err
for f in 1 2 4; do
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --spp 8 --frames-in-flight $f > $O/bench_spp8_f$f.json 2> $O/bench_spp8_f$f.err
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/flight/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], d["value"], d["ms_per_step"], d["config"].get("frames_in_flight"), d["config"].get("ms_frame_alone"),
          d["roofline"]["kernel_ms"], d["detail"].get("kernel_ms_with_frames_in_flight"), d["detail"].get("reduce"))
PY
