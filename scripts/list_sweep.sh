#!/bin/bash
# RT_LIST_PATHS (k_classify + list-driven k_shade from this many paths up; 0 = never): lone frames and frames in flight
O=gpurun_out/list; mkdir -p $O; : > $O/summary.txt
for v in 0 262144 1048576 4194304 16777216; do
  echo "RT_LIST_PATHS=$v" >> $O/summary.txt
  RT_LIST_PATHS=$v timeout -k 10 120 python3 scripts/spp_sweep.py 64 16 8 2>/dev/null | cut -c1-120 >> $O/summary.txt
  for spp in 64 8; do
    RT_LIST_PATHS=$v timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --spp $spp 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   in flight', $spp, 'spp', d['ms_per_step'], 'alone', d['config']['ms_frame_alone'])" >> $O/summary.txt
  done
done
for v in 0 1048576; do
  echo "RT_LIST_PATHS=$v" >> $O/summary.txt
  for c in "config 1" "config 4" "config 2" "config 5 ten distinct meshes"; do RT_LIST_PATHS=$v timeout -k 10 200 python3 scripts/configs_bench.py "$c" 2>/dev/null | cut -c1-190 >> $O/summary.txt; done
done
cat $O/summary.txt
