#!/bin/bash
# A/B builds of the fast hybrid instance with experiment switches: scripts/exp_variants.sh 1 2 3 ...
# -> path-tracer_amd/variants/libptgpu_exp<N>.so  (PT_EXP=N; everything else from the working tree's objects)
set -e
cd "$(dirname "$0")/../path-tracer_amd/csrc"
mkdir -p ../variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wall -Wno-unused-function -fno-slp-vectorize -DPT_WF_WAVES_PER_SIMD=6"
for m in "$@"; do
  ( /opt/rocm/bin/hipcc $F -DPT_EXP=$m -ffp-contract=fast-honor-pragmas -c kernels_fast_hyb.hip -o ../variants/exp${m}_hyb.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libptgpu_exp$m.so kernels_fast.o kernels_strict.o kernels_fast_bvh.o kernels_strict_bvh.o ../variants/exp${m}_hyb.o kernels_fast_ball.o kernels_strict_ball.o ptgpu_api.o ) &
done
wait
ls ../variants/libptgpu_exp*.so
