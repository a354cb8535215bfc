#!/bin/bash
# Where cornell.txt's HBM write traffic beyond the algorithmic 28 B per path comes from: measurement builds of the
# fast hybrid instance  (a) without the 80-VGPR budget (no spilled dwords),  (b) without the per-unit release fence
# (NOT a correct build: a tile's next pass may read stale data - counters only).
#   scripts/traffic_variants.sh   ->  path-tracer_amd/variants/libptgpu_{nospill,norelease}.so
set -e
cd "$(dirname "$0")/../path-tracer_amd/csrc"
mkdir -p ../variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wall -Wno-unused-function -fno-slp-vectorize -ffp-contract=fast-honor-pragmas"
/opt/rocm/bin/hipcc $F -DPT_WF_WAVES_PER_SIMD=4 -c kernels_fast_hyb.hip -o ../variants/nospill_hyb.o &
/opt/rocm/bin/hipcc $F -DPT_WF_WAVES_PER_SIMD=6 -DPT_NO_RELEASE -c kernels_fast_hyb.hip -o ../variants/norelease_hyb.o &
wait
for v in nospill norelease; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libptgpu_$v.so kernels_fast.o kernels_strict.o kernels_fast_bvh.o kernels_strict_bvh.o ../variants/${v}_hyb.o kernels_fast_ball.o kernels_strict_ball.o ptgpu_api.o
done
ls ../variants/libptgpu_nospill.so ../variants/libptgpu_norelease.so
