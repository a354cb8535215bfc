#!/bin/bash
# Measurement / test builds of libptgpu.so that ship next to it (path-tracer_amd/variants/, git-ignored,
# they travel to the GPU box like the library itself; __graft_entry__.build() runs this):
#   libptgpu_count.so   -DPT_COUNT_WORK=1: every kernel instance counts the work it executes
#                       (ptgpu_work_counters; bench.py's roofline).  Slow; never the timed build.
#   libptgpu_giveup.so  -DPT_QUEUE_SPIN_LIMIT=0: a work-queue wave that has to wait for a tile's previous
#                       chunk gives up at once (tests/test_gpu_synthetic.py checks that ptgpu_finish
#                       reports it and that the context stays usable).
#   libptgpu_probe.so   -DPT_RAY_PROBE=1: ptgpu_probe_rays traces rays the caller names through each code
#                       instance's intersection routines (tests/test_gpu_rays.py: tie rule, hierarchy walks).
set -e
cd "$(dirname "$0")/../path-tracer_amd/csrc"
mkdir -p ../variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wall -Wno-unused-function -fno-slp-vectorize -DPT_WF_WAVES_PER_SIMD=6"
build() { # name flags
  local name=$1 flags=$2 objs="" pids=""
  for tu in kernels_fast kernels_strict kernels_fast_bvh kernels_strict_bvh kernels_fast_hyb kernels_fast_ball kernels_strict_ball ptgpu_api; do
    local c="-ffp-contract=fast-honor-pragmas"; [[ $tu == *strict* ]] && c="-ffp-contract=off"; [[ $tu == ptgpu_api ]] && c=""
    local o=../variants/${name}_$tu.o
    if [[ ! -f $o || -n $(find $tu.hip radiance_device.inc radiance_kernels.inc device_params.h bvh_build.h ../../include/ptgpu.h -newer $o) ]]; then
      /opt/rocm/bin/hipcc $F $flags $c -c $tu.hip -o $o &
      pids="$pids $!"
    fi
    objs="$objs $o"
  done
  for p in $pids; do wait $p; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libptgpu_$name.so $objs
}
build count "-DPT_COUNT_WORK=1"
build giveup "-DPT_QUEUE_SPIN_LIMIT=0"
build probe "-DPT_RAY_PROBE=1"
ls -la ../variants/libptgpu_count.so ../variants/libptgpu_giveup.so ../variants/libptgpu_probe.so
