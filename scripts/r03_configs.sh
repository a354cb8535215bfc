#!/bin/bash
# BASELINE configs 3, 4, 5 (+ cornell with the other candidate-set modes): one bench line and one rocprofv3
# evidence set each (run on the GPU box via gpurun).  Results under gpurun_out/bench_r03 and gpurun_out/prof_r03.
#   scripts/r03_configs.sh prof    the rocprofv3 evidence (kernel stats + PMC passes) of every config
#   scripts/r03_configs.sh bench   the bench lines - AFTER `python scripts/collect_r03.py` has put the PMC summaries
#                                  of the prof phase under profiles/r03/: a line's issued-FP32 figure is built from them
PHASE=${1:-both}
mkdir -p gpurun_out/bench_r03
run() { # name, bench args...
  local name=$1; shift
  echo "== bench $name"
  timeout -k 10 300 python3 bench.py "$@" > gpurun_out/bench_r03/$name.json 2> gpurun_out/bench_r03/$name.err
  rc=$?; echo "rc=$rc"; case $rc in 124|134|137|139) exit $rc;; esac
}
prof() { echo "== profile $*"; timeout -k 10 400 scripts/profile_r03.sh "$@" > /dev/null 2>&1; rc=$?; echo "rc=$rc"; case $rc in 124|134|137|139) exit $rc;; esac; }
for cfg in "cornell_pt --scene cornell" "materials_pt --scene materials" "materials_dl --scene materials --integrator dl" "caustic_pt --scene caustic" \
           "caustic_dl --scene caustic --integrator dl" "obj-model_pt --scene obj-model" "obj-model-bunny_pt --scene obj-model-bunny" \
           "cornell-large_4k_pt --scene cornell-large --width 3840 --height 2160"; do
  set -- $cfg; name=$1; shift
  [ $PHASE != prof ] && run $name "$@"
  [ $PHASE != bench ] && prof "$@"
done
[ $PHASE = prof ] && exit 0
run cornell_cull_safe --cull safe --no-cpu
run cornell_cull_tight --cull tight --no-cpu
run cornell_cull_ball --cull ball --no-cpu
run cornell_strict --math strict --no-cpu --steps 4 --passes 16
run cornell_strict_safe --math strict --cull safe --no-cpu --steps 4 --passes 16
