#!/bin/bash
# fp8 GEMMs on the tile height the rounds ask for (default) against the 256-row tiles they always took before (OCC_GEMM_FP8_ROWS=256)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests/test_gpu_fp8.py tests/test_gpu_gemm_p8.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do
  for v in 0 256; do
    OCC_GEMM_FP8_ROWS=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --fp8 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp8 rows=$v', j['value'], j['ms_per_step'], j['roofline']['achieved'], j['roofline']['frac'])"
  done
done
for v in 0 256; do
  OCC_GEMM_FP8_ROWS=$v python bench.py --no-cpu-baseline --xlsr 1b --backend senet --bs 32 --fp8 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 fp8 rows=$v', j['value'], j['ms_per_step'], j['roofline']['frac'])"
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16', j['value'], j['ms_per_step'], j['roofline']['achieved'], j['roofline']['frac'])"
