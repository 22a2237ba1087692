#!/bin/bash
# usage (on the GPU box, once scripts/collect_final.sh has put this build's pmc_traffic.json files under profiles/):
# bash scripts/final_bench.sh -> gpurun_out/final/{configs,spp_sweep,pipeline_sweep}.txt, bench*.json
set -o pipefail
R=$GRAFT_REPO_ROOT
F=$R/gpurun_out/final
mkdir -p $F
cd $R
timeout -k 10 300 python3 scripts/configs_bench.py > $F/configs.txt 2>&1 || { tail -5 $F/configs.txt; exit 1; }
timeout -k 10 300 python3 scripts/spp_sweep.py 64 32 16 8 > $F/spp_sweep.txt 2>&1 || exit 1
SPPS="64 32 16 8" FLIGHT="2 4" MODE= bash scripts/pipeline_sweep.sh > /dev/null 2>&1 || exit 1
cp gpurun_out/r03_pipeline_sweep.txt $F/pipeline_sweep.txt
# the bench lines of this build (the driver's K and W)
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $F/bench.json 2> $F/bench.err || { tail -5 $F/bench.err; exit 1; }
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --frames-in-flight 1 --no-cpu-baseline > $F/bench_one_frame_at_a_time.json 2> $F/bench.err || { tail -5 $F/bench.err; exit 1; }
timeout -k 10 300 python3 bench.py --workload trace --steps 20 --warmup 5 > $F/bench_trace.json 2> $F/bench.err || { tail -5 $F/bench.err; exit 1; }
timeout -k 10 400 python3 bench.py --workload config5 --steps 5 --warmup 2 --no-cpu-baseline > $F/bench_config5_distinct.json 2> $F/bench.err || { tail -5 $F/bench.err; exit 1; }
timeout -k 10 600 python3 bench.py --workload config5 --width 3840 --steps 3 --warmup 1 --no-cpu-baseline > $F/bench_config5_distinct_4k.json 2> $F/bench.err || { tail -5 $F/bench.err; exit 1; }
for n in 2 4 8; do
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --spp $((64 / n)) > $F/bench_share_of_${n}_gpus.json 2> $F/bench.err || { tail -5 $F/bench.err; exit 1; }
done
echo done
