#!/bin/bash
# A/B builds of the instances that walk a hierarchy for every ray (kernels_fast_bvh, kernels_strict_bvh):
#   scripts/coop_variants.sh name:WAVES_PER_SIMD:"flags" ...  ->  path-tracer_amd/variants/libptgpu_<name>.so
# e.g. scripts/coop_variants.sh percoop:6:-DPT_COOP=0 idle8:6:-DPT_COOP_MIN_IDLE=8 wf4:4:
set -e
cd "$(dirname "$0")/../path-tracer_amd/csrc"
mkdir -p ../variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wall -Wno-unused-function -Wno-unused-but-set-variable -fno-slp-vectorize"
for spec in "$@"; do
  IFS=: read -r name wf flags <<< "$spec"
  ( /opt/rocm/bin/hipcc $F $flags -DPT_WF_WAVES_PER_SIMD=$wf -ffp-contract=fast-honor-pragmas -c kernels_fast_bvh.hip -o ../variants/${name}_fast_bvh.o
    /opt/rocm/bin/hipcc $F $flags -DPT_WF_WAVES_PER_SIMD=$wf -ffp-contract=off -c kernels_strict_bvh.hip -o ../variants/${name}_strict_bvh.o
    /opt/rocm/bin/hipcc $F $flags -c ptgpu_api.hip -o ../variants/${name}_api.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libptgpu_$name.so kernels_fast.o kernels_strict.o ../variants/${name}_fast_bvh.o ../variants/${name}_strict_bvh.o kernels_fast_hyb.o kernels_fast_ball.o kernels_strict_ball.o ../variants/${name}_api.o ) &
done
wait
ls ../variants/
