#!/bin/bash
# usage (on the GPU box): bash scripts/pmc_traffic.sh OUTDIR [WORKLOAD=render|config5|config5i] [WIDTH=1920] [SPP=64]
# rocprofv3 kernel-trace stats + separate --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass) over ONE frame;
# writes gpurun_out/OUTDIR/{kernel_stats.csv,kernel_trace.csv,pmc_traffic.json}.  The JSON records the library's build
# id: bench.py uses a traffic figure only for the build it was measured on.
OUT=$1; WL=${2:-render}; W=${3:-1920}; SPP=${4:-64}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/$OUT
mkdir -p $D
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $D/kt -o kt --output-format csv -- python3 $R/scripts/prof_render.py $SPP 9 $WL $W > $D/kt.log 2>&1 || { echo "kernel trace failed"; tail -5 $D/kt.log; exit 1; }
cp $(find $D/kt -name "*kernel_stats.csv" | head -1) $D/kernel_stats.csv
cp $(find $D/kt -name "*kernel_trace.csv" | head -1) $D/kernel_trace.csv
rm -rf $D/kt
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout -s KILL 300 rocprofv3 --pmc $C -d $D/$N -o $N --output-format csv -- python3 $R/scripts/prof_render.py $SPP 9 $WL $W > $D/$N.log 2>&1 || { echo "pass $N failed"; tail -5 $D/$N.log; exit 1; }
done
BUILD=$(cd $R && python3 -c 'import ray_tracing_amd as rt; print(rt.api.build_id())')
python3 $R/scripts/pmc_traffic_json.py $D $WL $W $SPP $BUILD > $D/pmc_traffic.json && cat $D/pmc_traffic.json
for N in FETCH_SIZE WRITE_SIZE TCC_HIT_sum; do rm -rf $D/$N; done
