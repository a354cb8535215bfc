#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of a 2-pass cornell.txt render for each library given (GPU box): scripts/traffic_pmc.sh lib.so ...
export TMPDIR=/tmp
A="--steps 1 --warmup 0 --passes 2 --no-cpu --no-count --no-image-error"
for lib in "$@"; do
  name=$(basename $lib .so); rm -rf gpurun_out/traffic/$name
  PTGPU_LIB=$lib timeout -s KILL 100 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/traffic/$name/r -- python3 bench.py $A > /dev/null 2>&1 || exit 1
  PTGPU_LIB=$lib timeout -s KILL 100 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/traffic/$name/w -- python3 bench.py $A > /dev/null 2>&1 || exit 1
  echo "$name: $(python3 scripts/summarize_pmc.py gpurun_out/traffic/$name $name 2 | grep -E 'bytes_per_pass|scratch' | tr -d '\n')"
done
