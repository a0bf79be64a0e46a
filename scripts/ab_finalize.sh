#!/bin/bash
# A/B of the one-launch batched finalize (OCC_DEFER_FINALIZE=1, default) against one finalize launch per producer.
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_finetune.py tests/test_gpu_gemm_p8.py tests/test_gpu_fullsize.py tests/test_gpu_autograd.py -x -q -m gpu > gpurun_out/ab_fin_tests.log 2>&1 || { tail -30 gpurun_out/ab_fin_tests.log; exit 1; }
tail -3 gpurun_out/ab_fin_tests.log
for rep in 1 2; do
  for d in 1 0; do
    OCC_DEFER_FINALIZE=$d python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('defer=$d', j['value'], j['ms_per_step'], j['roofline']['achieved'])" | tee -a gpurun_out/ab_fin.log
  done
done
