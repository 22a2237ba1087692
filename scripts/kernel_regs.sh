#!/bin/bash
# VGPRs / scratch / occupancy of the traversal kernels:  scripts/kernel_regs.sh [extra hipcc flags, e.g. -DRT_WAVES_EU_WIDE=5]
cd "$(dirname "$0")/../ray_tracing_amd/csrc" || exit 1
for f in rt_render.hip rt_device.hip; do
/opt/rocm/bin/hipcc -std=c++17 -O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-gpu-flush-denormals-to-zero -fPIC -Rpass-analysis=kernel-resource-usage "$@" -c $f -o /tmp/kernel_regs.o 2>&1 |
  grep -E "Function Name|  VGPRs:|SGPRs Spill|ScratchSize|Occupancy" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | paste - - - - - |
  sed -e 's/Function Name: _ZN12_GLOBAL__N_1[0-9]*//' -e 's/Function Name: _Z[0-9]*//' | grep -E "^k_(extend|shadow|mixed|trace|tail)" | cut -c1-60,100-400 | sed -e 's/N3rtd.*j\t/\t/'
done
