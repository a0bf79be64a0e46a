#!/bin/bash
# A/B of two builds of the library inside one gpurun call on one device: occm_amd/libocc_hip_base.so (OCC_LIB) against occm_amd/libocc_hip.so,
# alternating, N rounds of the default bench line each.   bash scripts/ab_bench.sh [rounds] [bench flags...]
N=${1:-2}; shift
for i in $(seq $N); do
OCC_LIB=$PWD/occm_amd/libocc_hip_base.so python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new ', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done
