#!/bin/bash
# RT_PIPELINE_PACE (fraction of latency / frames_in_flight between frame starts) on bench.py; run on the GPU box
set -e
O=gpurun_out/pace; mkdir -p $O
for g in 0 0.5 0.8 1.0; do
  RT_PIPELINE_PACE=$g timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b64_$g.json 2> $O/err.txt
  RT_PIPELINE_PACE=$g timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --spp 8 > $O/b8_$g.json 2> $O/err.txt
  RT_PIPELINE_PACE=$g timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --spp 16 > $O/b16_$g.json 2> $O/err.txt
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/pace/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], d["value"], d["ms_per_step"], d["config"].get("frames_in_flight"), d["config"].get("ms_frame_alone"))
PY
