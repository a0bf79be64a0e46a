#!/bin/bash
for m in 0 1; do echo "== OCC_CONV0_MFMA=$m"; OCC_CONV0_MFMA=$m python scripts/bench_score.py --n 48 --dtypes bf16 --batches 1,16 2>&1 | grep utt_per_s | cut -c1-220; done
