#!/bin/bash
# Builds measurement-only variants of libptgpu.so (path-tracer_amd/variants/, git-ignored) that drop one
# stage of the path, to attribute time:  ablate1 = shadow rays queued but never tested,
# ablate2 = light samples never ask for a shadow ray, ablate3 = full kernel + event counters
# (ptgpu_debug_stats, scripts/light_phase_stats.py).  Run here (hipcc cross-compiles), then time
# them on the GPU box with PTGPU_LIB=path-tracer_amd/variants/libptgpu_ablateN.so scripts/ablate_time.py.
set -e
cd "$(dirname "$0")/../path-tracer_amd/csrc"
mkdir -p ../variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wall -Wno-unused-function -fno-slp-vectorize"
for n in 1 2 3; do
  /opt/rocm/bin/hipcc $F -DPT_ABLATE=$n -ffp-contract=fast-honor-pragmas -DPT_WF_WAVES_PER_SIMD=6 -c kernels_fast_hyb.hip -o ../variants/kf$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libptgpu_ablate$n.so ../variants/kf$n.o kernels_fast.o kernels_strict.o kernels_fast_bvh.o kernels_strict_bvh.o kernels_fast_ball.o kernels_strict_ball.o ptgpu_api.o
done
ls -la ../variants/*.so
