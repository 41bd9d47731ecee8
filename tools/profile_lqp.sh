#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + counters of the generic hierarchical-QP kernels on the TOCABI LQP cascade.
#   tools/profile_lqp.sh <name>
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=${1:-lqp}
OUT=$R/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/tools/lqp_cascade.py 1024 20"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/pmc_f64 -- $CMD > $OUT/pmc_f64.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
tail -3 $OUT/trace.log
find $OUT -name "*kernel_stats.csv" | head -2
