#!/bin/bash
# PMC passes over one short bench run (each pass its own process; SQ <= 8, TCC <= 4 counters).
# usage: scripts/profile_counters.sh <tag> [bench args...]     (run on the GPU box via gpurun)
set -u
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --passes 2 --no-cpu $*"
run() { # name counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py $ARGS > $OUT/$name.log 2>&1 || { echo "pass $name failed"; return 1; }
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS &&
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM &&
run sq3 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH &&
run sq4 SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_CYCLES SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM &&
run grbm GRBM_GUI_ACTIVE GRBM_COUNT &&
run tccr FETCH_SIZE &&
run tccw WRITE_SIZE
python3 scripts/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
