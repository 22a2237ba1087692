#!/bin/bash
# scripts/build_variant.sh NAME "-DFLAG=.. ..."  -> ray_tracing_amd/_lib/librt_amd_NAME.so (A/B experiments;
# select with RT_AMD_LIB=<path>).  Not part of the product build.
set -e
cd "$(dirname "$0")/../ray_tracing_amd/csrc"
NAME=$1; EXTRA=$2
B=build/var_$NAME; mkdir -p $B
HIPFLAGS="-std=c++17 -O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fPIC -Wall -Wno-unused-parameter $EXTRA"
for f in rt_device rt_render rt_comm rt_pipeline; do /opt/rocm/bin/hipcc $HIPFLAGS -c $f.hip -o $B/$f.o & done; wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../_lib/librt_amd_$NAME.so build/mesh.o build/bvh_build.o build/scene_build.o $B/rt_device.o $B/rt_render.o $B/rt_comm.o $B/rt_pipeline.o -ldl
echo built ../_lib/librt_amd_$NAME.so
