#!/bin/bash
# A/B builds of the library with experiment switches: scripts/exp_variants.sh name "flags" [name "flags" ...]
# -> path-tracer_amd/variants/libptgpu_<name>.so  (the fast instances rebuilt with the flags; everything else from the
# working tree's objects)
set -e
cd "$(dirname "$0")/../path-tracer_amd/csrc"
mkdir -p ../variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wall -Wno-unused-function -fno-slp-vectorize -DPT_WF_WAVES_PER_SIMD=6"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( for tu in kernels_fast kernels_fast_hyb kernels_fast_bvh; do
      /opt/rocm/bin/hipcc $F $flags -ffp-contract=fast-honor-pragmas -c $tu.hip -o ../variants/${name}_$tu.o &
    done; wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libptgpu_$name.so ../variants/${name}_kernels_fast.o kernels_strict.o \
       ../variants/${name}_kernels_fast_bvh.o kernels_strict_bvh.o ../variants/${name}_kernels_fast_hyb.o kernels_fast_ball.o kernels_strict_ball.o ptgpu_api.o ) &
done
wait
ls -la ../variants/libptgpu_*.so
