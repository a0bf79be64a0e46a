#!/bin/bash
# A/B of the weight-gradient pairs on a side stream (OCC_WGRAD_STREAM=1, default) against everything on one stream.
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_finetune.py tests/test_gpu_fullsize.py tests/test_gpu_autograd.py tests/test_gpu_fp8.py -x -q -m gpu > gpurun_out/ab_wgs_tests.log 2>&1 || { tail -30 gpurun_out/ab_wgs_tests.log; exit 1; }
tail -3 gpurun_out/ab_wgs_tests.log
for rep in 1 2; do
  for d in 1 0; do
    OCC_WGRAD_STREAM=$d python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgrad_stream=$d', j['value'], j['ms_per_step'], j['roofline']['achieved'], j['roofline']['frac'])" | tee -a gpurun_out/ab_wgs.log
  done
done
OCC_WGRAD_STREAM=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --fp8 2>/dev/null | tail -1 | cut -c1-200
OCC_WGRAD_STREAM=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --fp8 2>/dev/null | tail -1 | cut -c1-200
