#!/bin/bash
# usage (on the GPU box): bash scripts/trace_frame.sh OUTDIR SPP [WORKLOAD] -> gpurun_out/OUTDIR/kernel_trace.csv of one 1080p frame
OUT=$1; SPP=${2:-64}; WL=${3:-render}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/$OUT
mkdir -p $D
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $D/kt -o kt --output-format csv -- python3 $R/scripts/prof_render.py $SPP 9 $WL > $D/kt.log 2>&1 || { echo "kernel trace failed"; tail -5 $D/kt.log; exit 1; }
cp $(find $D/kt -name "*kernel_trace.csv" | head -1) $D/kernel_trace.csv
cp $(find $D/kt -name "*kernel_stats.csv" | head -1) $D/kernel_stats.csv
rm -rf $D/kt
