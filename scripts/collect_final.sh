#!/bin/bash
# usage (here, after `gpurun -- bash scripts/final_profiles.sh`): bash scripts/collect_final.sh profiles/rNN/final
# and again after `gpurun -- bash scripts/final_bench.sh` (the bench lines look for this build's traffic under profiles/).
# Copies what the committed profile directory holds out of gpurun_out/final and derives the summaries.
set -e
F=gpurun_out/final
P=${1:-profiles/r02/final}
for d in render config5_distinct config5_instanced config5_distinct_4k; do
    mkdir -p $P/$d
    cp $F/$d/pmc_traffic.json $F/$d/kernel_stats.csv $P/$d/
done
cp $F/render/kernel_trace.csv $P/render/
cp $F/bench_kernel_stats.csv $F/bench_under_rocprof.json $F/bench_in_flight_kernel_stats.csv $F/bench_in_flight_under_rocprof.json $P/
cp $F/pmc_counters/summary.txt $P/pmc_summary.txt
python scripts/pmc_ratios.py $P/pmc_summary.txt > $P/pmc_ratios.txt
python scripts/timeline.py $P/render/kernel_trace.csv > $P/render/timeline.txt 2>&1
grep -h '"build"' $P/*/pmc_traffic.json | sort | uniq -c
[ -f $F/bench.json ] || { echo "no bench lines yet: run scripts/final_bench.sh on the box, then this again"; exit 0; }
cp $F/bench.json $F/bench_one_frame_at_a_time.json $F/bench_trace.json $F/bench_config5_distinct.json $F/bench_config5_distinct_4k.json $F/bench_share_of_*_gpus.json $P/
grep -v amdgpu.ids $F/pipeline_sweep.txt > $P/pipeline_sweep.txt
grep -v amdgpu.ids $F/configs.txt > $P/configs.txt
grep -v amdgpu.ids $F/spp_sweep.txt > $P/spp_sweep.txt
