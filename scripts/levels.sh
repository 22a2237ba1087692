#!/bin/bash
# node-visit throughput vs mesh size (does the L2-miss path bound the traversal?)
for LV in "$@"; do
  R=$(timeout -k 10 300 python -u bench.py --steps 1 --warmup 1 --spp ${SPP:-64} --levels $LV --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); x=d['detail']; r=d['roofline']; print('ms/frame %.1f ext %.1f sh %.1f shade %.1f  GB/s %.0f n_int %.2f n_tri %.2f rays %d' % (d['ms_per_step'], x['ms_extend'], x['ms_shadow'], x['ms_shade'], r['achieved'], r['n_int_per_ray'], r['n_tri_per_ray'], x['rays_closest']))")
  echo "levels=$LV | $R"
done
