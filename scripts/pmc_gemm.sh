#!/bin/bash
# Wave-state / memory-path counters of the GEMM kernels on one shape (separate --pmc passes; kernel-trace only).
# usage (on the GPU box): bash scripts/pmc_gemm.sh <tag> <MxNxK> <variants>
set -e
TAG=${1:-pmcg}; export GEMM_SHAPES=${2:-6368x4096x1024}; VAR=${3:-1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD" \
           "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_BUSY_sum"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/scripts/bench_gemm.py $VAR 1 > $OUT.p$i.log 2>&1
done
ls $OUT
