#!/bin/bash
for m in 0 1; do echo "== OCC_CONV0_BWD_MFMA=$m"; OCC_CONV0_BWD_MFMA=$m python scripts/time_conv0_bwd.py 2>&1 | grep -v amdgpu.ids; done
for n in 1 3 4; do echo "== OCC_C0B_WGS_PER_CU=$n"; OCC_C0B_WGS_PER_CU=$n python scripts/time_conv0_bwd.py 2>&1 | grep "backward"; done
