#!/bin/bash
for r in 1 2; do
python bench.py --frozen --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default     ', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
OCC_GEMM_Q4=1 OCC_Q4_ROWS=224 python bench.py --frozen --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('q4 N>=3072  ', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done
