#!/bin/bash
# A/B of two builds of the library inside one call: scripts/_bin/libocc_base.so (OCC_LIB) against occm_amd/libocc_hip.so, headline bench, two rounds
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B=$PWD/scripts/_bin/libocc_base.so
for rep in 1 2; do
  for lib in new base; do
    if [ $lib = base ]; then export OCC_LIB=$B; else unset OCC_LIB; fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', j['value'], j['ms_per_step'], j['roofline']['achieved'], j['roofline']['gemm_ms_per_step'])"
  done
done
