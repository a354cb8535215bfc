#!/bin/bash
# Measurement builds that replace ONE approximation of the fast build by the exact operation (PT_EXACT,
# radiance_device.inc), to attribute the fast build's deviation from the strict one:
#   scripts/bias_bisect.sh      ->  path-tracer_amd/variants/libptgpu_exact{1,2,4,8,16,31}.so
# then on the GPU box: PTGPU_LIB=<variant> python scripts/fast_vs_strict.py
set -e
cd "$(dirname "$0")/../path-tracer_amd/csrc"
mkdir -p ../variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wall -Wno-unused-function -fno-slp-vectorize -DPT_WF_WAVES_PER_SIMD=6"
for m in ${*:-1 2 4 8 16 31}; do
  ( /opt/rocm/bin/hipcc $F -DPT_EXACT=$m -ffp-contract=fast-honor-pragmas -c kernels_fast_hyb.hip -o ../variants/exact${m}_hyb.o
    /opt/rocm/bin/hipcc $F -DPT_EXACT=$m -c ptgpu_api.hip -o ../variants/exact${m}_api.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libptgpu_exact$m.so kernels_fast.o kernels_strict.o kernels_fast_bvh.o kernels_strict_bvh.o ../variants/exact${m}_hyb.o kernels_fast_ball.o kernels_strict_ball.o ../variants/exact${m}_api.o ) &
done
wait
ls ../variants/libptgpu_exact*.so
