#!/bin/bash
# usage (on the GPU box): bash scripts/pmc_traffic.sh OUTDIR
# rocprofv3 kernel-trace stats + two separate --pmc passes (FETCH_SIZE, WRITE_SIZE cannot share a pass)
# over ONE 1080p 64-spp frame of the bench workload; writes gpurun_out/OUTDIR/{kernel_stats.csv,pmc_traffic.json}.
OUT=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/$OUT
mkdir -p $D
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $D/kt -o kt --output-format csv -- python3 $R/scripts/prof_render.py 64 > $D/kt.log 2>&1 || { echo "kernel trace failed"; tail -5 $D/kt.log; exit 1; }
cp $(find $D/kt -name "*kernel_stats.csv" | head -1) $D/kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout -s KILL 200 rocprofv3 --pmc $C -d $D/$N -o $N --output-format csv -- python3 $R/scripts/prof_render.py 64 > $D/$N.log 2>&1 || { echo "pass $N failed"; tail -5 $D/$N.log; exit 1; }
done
python3 $R/scripts/pmc_traffic_json.py $D > $D/pmc_traffic.json && cat $D/pmc_traffic.json && head -12 $D/kernel_stats.csv
