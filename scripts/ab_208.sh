#!/bin/bash
# A/B of the 208-row tiles of the eight-phase GEMM (OCC_GEMM_208=1, default) against 224-row tiles, headline bench, two rounds + the other shapes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do
  for v in 1 0; do
    OCC_GEMM_208=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('208=$v', j['value'], j['ms_per_step'], j['roofline']['achieved'], j['roofline']['frac'], j['roofline']['gemm_ms_per_step'])"
  done
done
for v in 1 0; do
  OCC_GEMM_208=$v python bench.py --no-cpu-baseline --frozen 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frozen 208=$v', j['value'], j['ms_per_step'], j['roofline']['frac'])"
  OCC_GEMM_208=$v python bench.py --no-cpu-baseline --xlsr 1b --backend senet --bs 32 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 208=$v', j['value'], j['ms_per_step'], j['roofline']['frac'])"
done
