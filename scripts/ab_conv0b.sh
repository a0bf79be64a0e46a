#!/bin/bash
for fr in 256 512 1024; do echo "== OCC_C0_FRAMES=$fr"; OCC_C0_FRAMES=$fr python scripts/bench_conv0.py 2>&1 | grep "bfloat16"; done
