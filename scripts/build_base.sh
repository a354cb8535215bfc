#!/bin/bash
# Builds libptgpu.so of a git revision (default HEAD) as path-tracer_amd/variants/libptgpu_base.so, for A/B
# timing against the working tree on the same GPU box (scripts/ab_time.py).
set -e
REV=${1:-HEAD}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
rm -rf /tmp/ptbase && mkdir -p /tmp/ptbase "$ROOT/path-tracer_amd/variants"
git -C "$ROOT" archive $REV path-tracer_amd/csrc include | tar -x -C /tmp/ptbase
make -s -C /tmp/ptbase/path-tracer_amd/csrc -j8 OUT="$ROOT/path-tracer_amd/variants/libptgpu_base.so" > /dev/null
ls -la "$ROOT/path-tracer_amd/variants/libptgpu_base.so"
