#!/bin/bash
# Runs GPU steps one after the other (on the box, via gpurun): `scripts/gpu_steps.sh "cmd1" "cmd2" ...`
# An ordinary failure (exit 1: a failed assertion, a bench that refused its arguments) is reported and the next
# step runs; a time limit, abort, fault or kill (124, 134, 137, 139, 255) ends the script: nothing more is
# started on the GPU after it.
for cmd in "$@"; do
  echo "== $cmd"
  bash -o pipefail -c "$cmd"
  rc=$?
  echo "== rc=$rc"
  case $rc in 124|134|135|137|139|255) echo "stopping: a GPU step died (rc=$rc)"; exit $rc;; esac
done
exit 0
