#!/bin/bash
# usage (on the GPU box): bash scripts/final_profiles.sh  -> gpurun_out/final/...  (copied to profiles/rNN/final by hand)
# Everything the committed profile directory holds, from ONE build of the library.
set -o pipefail
R=$GRAFT_REPO_ROOT
F=$R/gpurun_out/final
mkdir -p $F
cd $R
bash scripts/pmc_traffic.sh final/render render 1920 64 > $F/pmc_render.log 2>&1 || { tail -5 $F/pmc_render.log; exit 1; }
bash scripts/pmc_traffic.sh final/config5_distinct config5 1920 64 > $F/pmc_c5d.log 2>&1 || { tail -5 $F/pmc_c5d.log; exit 1; }
bash scripts/pmc_traffic.sh final/config5_instanced config5i 1920 64 > $F/pmc_c5i.log 2>&1 || { tail -5 $F/pmc_c5i.log; exit 1; }
bash scripts/pmc_traffic.sh final/config5_distinct_4k config5 3840 64 > $F/pmc_c5d4k.log 2>&1 || { tail -5 $F/pmc_c5d4k.log; exit 1; }
bash scripts/pmc_render.sh final/pmc_counters 64 > $F/pmc_counters.log 2>&1 || { tail -5 $F/pmc_counters.log; exit 1; }
# bench.py looks for the traffic of its own build under profiles/: give the box's scratch copy of the tree what was just measured
P=$R/profiles/${ROUND:-r03}/final
for d in render config5_distinct config5_instanced config5_distinct_4k; do mkdir -p $P/$d && cp $F/$d/pmc_traffic.json $P/$d/; done
cd /tmp && export TMPDIR=/tmp
# one frame at a time: the launch durations bench.py's roofline uses (HIP events) against the profiler's
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $F/benchkt -o kt --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --frames-in-flight 1 > $F/bench_under_rocprof.json 2> $F/bench_under_rocprof.err || { tail -5 $F/bench_under_rocprof.err; exit 1; }
cp $(find $F/benchkt -name "*kernel_stats.csv" | head -1) $F/bench_kernel_stats.csv
rm -rf $F/benchkt
# the default command (frames in flight): launches of different frames share the GPU, their durations stretch
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $F/benchkt -o kt --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $F/bench_in_flight_under_rocprof.json 2> $F/bench_in_flight_under_rocprof.err || { tail -5 $F/bench_in_flight_under_rocprof.err; exit 1; }
cp $(find $F/benchkt -name "*kernel_stats.csv" | head -1) $F/bench_in_flight_kernel_stats.csv
rm -rf $F/benchkt
echo done
