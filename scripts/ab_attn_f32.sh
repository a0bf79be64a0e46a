#!/bin/bash
# f32 attention on the f32 matrix cores (OCC_ATTN_F32_MFMA=1, default) against the VALU kernels: parity tests, then scoring throughput.
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_scoring.py tests/test_gpu_frontend.py -x -q -m gpu > gpurun_out/ab_attn_f32_tests.log 2>&1 || { tail -40 gpurun_out/ab_attn_f32_tests.log; exit 1; }
tail -2 gpurun_out/ab_attn_f32_tests.log
for m in 1 0 1 0; do
  echo "OCC_ATTN_F32_MFMA=$m" | tee -a gpurun_out/ab_attn_f32.log
  OCC_ATTN_F32_MFMA=$m python scripts/bench_score.py --n 96 --dtypes f32 --batches 1,16 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        j=json.loads(l); print('  batch', j['batch_size'], 'masked', j['masked_batches'], 'utt/s', j['utt_per_s'], 'diff', j['max_abs_emb_diff_vs_batch1'])" | tee -a gpurun_out/ab_attn_f32.log
done
