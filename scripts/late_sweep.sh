#!/bin/bash
# Sweep of the late-round knobs (RT_LATE_PATHS, RT_LATE_BUDGET, RT_TAIL_PATHS_LATE) on the bench scene: frame ms at 64 and 8 spp.
cd "$(dirname "$0")/.."
run() { echo "== late_paths=$1 budget=$2 tail_late=$3"; RT_LATE_PATHS=$1 RT_LATE_BUDGET=$2 RT_TAIL_PATHS_LATE=$3 timeout -k 10 100 python scripts/spp_sweep.py 64 8 2>/dev/null | sed -e 's/{.*ms_tail/ms_tail/' || exit 1; }
run 0 64 4096
for b in 64 128 256; do for t in 4096 32768 131072 262144 524288; do run 4194304 $b $t; done; done
for b in 64 128 1000000; do for t in 131072 262144 524288; do run 1048576 $b $t; done; done
