#!/bin/bash
# thresholds of the wavefront loop re-swept with frames in flight (bench.py --steps 20 --warmup 5, 64 and 8 spp); on the GPU box
set -e
O=gpurun_out/envflight; mkdir -p $O; : > $O/summary.txt
run() { # name, env assignments...
  name=$1; shift
  for spp in 64 8; do
    env "$@" timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --spp $spp > $O/$name.$spp.json 2> $O/err.txt
    python3 -c "
import json,sys
d=json.loads(open('$O/$name.$spp.json').read().strip().splitlines()[-1])
print('%-28s %2d spp %8.3f ms/step  alone %8.3f' % ('$name', $spp, d['ms_per_step'], d['config']['ms_frame_alone']))" >> $O/summary.txt
  done
}
run default RT_NOOP=1
for t in ${TAILS:-32768 65536 262144 524288 1048576}; do run tail_$t RT_TAIL_PATHS=$t; done
if [ -z "$TAILS" ]; then
for a in 4194304 67108864; do run async_$a RT_ASYNC_BELOW=$a; done
for m in 67108864 268435456; do run mix_$m RT_MIX_PATHS=$m; done
fi
cat $O/summary.txt
