#!/bin/bash
# bench.py with 2 / 3 / 4 frames in flight, the forced-distributed rehearsal and the 8-spp share: what produced
# profiles/r03/flight_priorities.txt (p1_* there came from a throw-away build whose slots had streams of different
# priorities, switched on with RT_PIPELINE_PRIORITIES=1; the library no longer reads that variable)
set -e
O=gpurun_out/flight2; mkdir -p $O
for pr in 0; do
for f in 2 3 4; do
  RT_PIPELINE_PRIORITIES=$pr timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --frames-in-flight $f > $O/bench_p${pr}_f$f.json 2> $O/err.txt
done
RT_PIPELINE_PRIORITIES=$pr RT_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 \
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_p${pr}_dist.json 2> $O/err.txt
RT_PIPELINE_PRIORITIES=$pr timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --spp 8 --frames-in-flight 4 > $O/bench_p${pr}_spp8_f4.json 2> $O/err.txt
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/flight2/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], d["value"], d["ms_per_step"], d["config"].get("frames_in_flight"), d["config"].get("ms_frame_alone"),
          d["detail"].get("kernel_ms_with_frames_in_flight"), d["detail"].get("reduce"), d["detail"].get("separate_hw_queues"))
PY
