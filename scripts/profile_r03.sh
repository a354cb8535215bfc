#!/bin/bash
# One workload's rocprofv3 evidence (run on the GPU box via gpurun):
#   scripts/profile_r03.sh [bench args...]      (the workload key comes from bench.py --key-only)
# 1. kernel trace + stats of the bench command (short: 6 steps), 2. PMC passes, each its own process
# (SQ <= 8, TCC <= 4 counters per pass; FETCH_SIZE and WRITE_SIZE apart), 3. the summary JSON.
# Any step that times out or dies ends the script.
set -u
# PMC passes run ONE timed step of the bench (the launch shape the bench line times: 64 passes in one render call),
# not a 2-pass render as in round 2
PMC_PASSES=${PMC_PASSES:-64}
KEY=$(timeout -k 10 120 python3 bench.py --key-only "$@" | tail -1)
[ -n "$KEY" ] || { echo "no workload key"; exit 1; }
OUT=gpurun_out/prof_r03/$KEY
mkdir -p $OUT
export TMPDIR=/tmp
step() { "$@"; rc=$?; if [ $rc -ne 0 ]; then echo "step failed rc=$rc: $*"; exit $rc; fi; }
step timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 6 --warmup 1 --no-cpu --no-count --no-image-error --no-exact "$@" > $OUT/trace_bench.json 2> $OUT/trace.log
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
ARGS="--steps 1 --warmup 0 --passes $PMC_PASSES --no-cpu --no-count --no-image-error --no-exact $*"
run() { local name=$1; shift; step timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc/$name -- python3 bench.py $ARGS > $OUT/pmc_$name.log 2>&1; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM
run sq3 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH
run sq4 SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT SQ_INSTS_SENDMSG
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run tccr FETCH_SIZE
run tccw WRITE_SIZE
python3 scripts/summarize_pmc.py $OUT/pmc $KEY $PMC_PASSES gpurun_out/prof_r03/pmc_summary.json profiles/r03/kernel_resources.json > $OUT/pmc_summary.txt 2>&1
rm -rf $OUT/trace $OUT/pmc
cat $OUT/pmc_summary.txt | head -70
