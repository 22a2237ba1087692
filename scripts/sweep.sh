#!/bin/bash
# usage: scripts/sweep.sh VAR v1 v2 ...   -- runs the short render + trace bench for each value of env VAR
VAR=$1; shift
for V in "$@"; do
  export $VAR=$V
  R=$(timeout -k 10 300 python -u bench.py --steps 1 --warmup 1 --spp ${SPP:-64} --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); x=d['detail']; print('render ms/frame %.1f ext %.1f sh %.1f shade %.1f frac %.3f' % (d['ms_per_step'], x['ms_extend'], x['ms_shadow'], x['ms_shade'], d['roofline']['frac']))")
  T=$(timeout -k 10 300 python -u bench.py --workload trace --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); x=d['detail']; print('trace closest %.3f any %.3f frac %.3f' % (x['kernel_ms'], x['any_ms'], d['roofline']['frac']))")
  echo "$VAR=$V | $R | $T"
done
