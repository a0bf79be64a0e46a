#!/bin/bash
# weight-gradient block order: a tile's pieces side by side (0) vs one reduction piece per XCD across many tiles (1)
for m in 0 1; do echo "== OCC_TN_PIECE_MAJOR=$m"; OCC_TN_PIECE_MAJOR=$m python scripts/bench_tn_p8.py 3 2>&1 | grep "M="; done
for r in 1 2; do for m in 0 1; do
  OCC_TN_PIECE_MAJOR=$m python bench.py --no-cpu-baseline --steps 15 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('piece_major=$m', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done; done
