#!/bin/bash
# frames in flight x samples per frame (scripts/pipeline_probe.py); run on the GPU box
set -e
out=gpurun_out/r03_pipeline_sweep.txt
mkdir -p gpurun_out; : > $out
for spp in ${SPPS:-64 32 16 8}; do
  for n in ${FLIGHT:-2 3 4}; do
    timeout -k 10 120 python -u scripts/pipeline_probe.py $spp 24 $n $MODE 2>/dev/null >> $out
  done
done
cat $out
