#!/bin/bash
# Wave-state counters of one benchmark script (separate --pmc passes, kernel-trace only).
# usage (on the GPU box): bash scripts/pmc_kernel.sh <tag> <script.py> [args...]   -> gpurun_out/<tag>/p*/...
set -e
TAG=$1; shift; SCRIPT=$GRAFT_REPO_ROOT/$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $SCRIPT "$@" > $OUT.p$i.log 2>&1
done
ls $OUT
