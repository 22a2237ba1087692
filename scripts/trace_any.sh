#!/bin/bash
# usage (on the GPU box): bash scripts/trace_any.sh OUTDIR script.py [args]  -> gpurun_out/OUTDIR/{kernel_trace.csv,timeline.txt} of the script's kernels
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/$OUT
mkdir -p $D
timeout -s KILL 200 rocprofv3 --kernel-trace -d $D/kt -o kt --output-format csv -- python3 $R/$1 "${@:2}" > $D/kt.log 2>&1 || { echo "kernel trace failed"; tail -5 $D/kt.log; exit 1; }
cp $(find $D/kt -name "*kernel_trace.csv" | head -1) $D/kernel_trace.csv
rm -rf $D/kt
python3 $R/scripts/timeline.py $D/kernel_trace.csv > $D/timeline.txt
