#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + PMC passes for bench.py's workload.  Outputs under gpurun_out/.
#   tools/profile.sh <name> [trace|full] [extra bench.py args...]
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=${1:-prof}
MODE=${2:-full}
shift; shift
OUT=$R/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline $*"
# the timing pass runs long enough for the clocks to ramp (a 20-step run reads ~4 % slow); the counter passes stay short
TBENCH="python3 $R/bench.py --steps 500 --warmup 20 --no-cpu-baseline $*"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $TBENCH > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
if [ "$MODE" = "full" ]; then
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_insts -- $BENCH > $OUT/pmc_insts.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/pmc_f64 -- $BENCH > $OUT/pmc_f64.log 2>&1
# matrix-core counters (VERDICT r2 row g): instructions, f64 MFMA operations, cycles the MFMA pipe is busy
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_mfma -- $BENCH > $OUT/pmc_mfma.log 2>&1
fi
find $OUT -name "*.csv" | head -30
