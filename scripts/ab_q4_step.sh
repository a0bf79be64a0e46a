#!/bin/bash
# A/B of the four-wave GEMM's dispatch policy inside the training step (one process per setting, alternating, one device)
for r in 1 2; do
for q in 0 1 3 7; do
  OCC_GEMM_Q4=$q python bench.py --no-cpu-baseline --steps 15 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('q4=$q', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline'].get('ms_per_step'))"
done
done
