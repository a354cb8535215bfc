#!/bin/bash
# A/B builds of the fast hybrid instance under different compiler flags:
#   scripts/flag_variants.sh name1 "flags1" name2 "flags2" ...   -> path-tracer_amd/variants/libptgpu_f_<name>.so
set -e
cd "$(dirname "$0")/../path-tracer_amd/csrc"
mkdir -p ../variants
F="--offload-arch=gfx950 -std=c++17 -fPIC -I../../include -I. -Wall -Wno-unused-function -fno-slp-vectorize -DPT_WF_WAVES_PER_SIMD=6 -ffp-contract=fast-honor-pragmas"
while [ $# -ge 2 ]; do
  n=$1; fl=$2; shift 2
  ( /opt/rocm/bin/hipcc $F -O3 $fl -c kernels_fast_hyb.hip -o ../variants/f_${n}_hyb.o 2> ../variants/f_${n}.err &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libptgpu_f_$n.so kernels_fast.o kernels_strict.o kernels_fast_bvh.o kernels_strict_bvh.o ../variants/f_${n}_hyb.o kernels_fast_ball.o kernels_strict_ball.o ptgpu_api.o || echo "FAILED $n: $(tail -2 ../variants/f_${n}.err)" ) &
done
wait
ls ../variants/libptgpu_f_*.so
