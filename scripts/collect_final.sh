#!/bin/bash
# usage (here, after `gpurun -- bash scripts/final_profiles.sh`): bash scripts/collect_final.sh profiles/rNN/final
# Copies what the committed profile directory holds out of gpurun_out/final and derives the summaries.
set -e
F=gpurun_out/final
P=${1:-profiles/r02/final}
for d in render config5_distinct config5_instanced config5_distinct_4k; do
    mkdir -p $P/$d
    cp $F/$d/pmc_traffic.json $F/$d/kernel_stats.csv $P/$d/
done
cp $F/render/kernel_trace.csv $P/render/
cp $F/bench_kernel_stats.csv $F/bench_under_rocprof.json $P/
grep -v amdgpu.ids $F/configs.txt > $P/configs.txt
grep -v amdgpu.ids $F/spp_sweep.txt > $P/spp_sweep.txt
cp $F/pmc_counters/summary.txt $P/pmc_summary.txt
python scripts/pmc_ratios.py $P/pmc_summary.txt > $P/pmc_ratios.txt
python scripts/timeline.py $P/render/kernel_trace.csv > $P/render/timeline.txt 2>&1
grep -h '"build"' $P/*/pmc_traffic.json | sort | uniq -c
