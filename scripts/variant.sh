#!/bin/bash
# Builds a measurement variant of libptgpu.so with extra compiler flags on the kernel translation
# units named (default: the fast hybrid one cornell.txt runs):
#   scripts/variant.sh <name> "<flags>" [tu ...]   ->  path-tracer_amd/variants/libptgpu_<name>.so
# Time it on the GPU box with PTGPU_LIB=path-tracer_amd/variants/libptgpu_<name>.so scripts/ablate_time.py.
set -e
NAME=$1; FLAGS=$2; shift 2
TUS=${*:-kernels_fast_hyb}
cd "$(dirname "$0")/../path-tracer_amd/csrc"
mkdir -p ../variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wall -Wno-unused-function -fno-slp-vectorize"
OBJS=""
for tu in kernels_fast kernels_strict kernels_fast_bvh kernels_strict_bvh kernels_fast_hyb kernels_fast_ball kernels_strict_ball; do
  if [[ " $TUS " == *" $tu "* ]]; then
    C="-ffp-contract=fast-honor-pragmas"; [[ $tu == *strict* ]] && C="-ffp-contract=off"
    /opt/rocm/bin/hipcc $F $FLAGS $C -DPT_WF_WAVES_PER_SIMD=6 -c $tu.hip -o ../variants/${NAME}_$tu.o
    OBJS="$OBJS ../variants/${NAME}_$tu.o"
  else OBJS="$OBJS $tu.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libptgpu_$NAME.so $OBJS ptgpu_api.o
ls -la ../variants/libptgpu_$NAME.so
