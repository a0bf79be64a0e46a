#!/bin/bash
for r in 1 2; do for gmode in 0 1; do
  OCC_FE_GRAPH=$gmode python bench.py --no-cpu-baseline --steps 20 --warmup 6 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fe_graph=$gmode', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done; done
