#!/bin/bash
# A/B of the four-wave kernel's start stagger (one process per setting: the env is read once per process)
S=12736x4096x1024,12736x1024x4096,12736x3072x1024,12736x1024x1024
for cfg in "0 0" "1 50" "1 100" "1 200" "2 100"; do
  set -- $cfg
  echo "== OCC_Q4_STAGGER=$1 OCC_Q4_DELAY=$2"
  OCC_Q4_STAGGER=$1 OCC_Q4_DELAY=$2 GEMM_SHAPES=$S python scripts/bench_gemm.py 31,40,41 3 2>&1 | grep "^s"
done
