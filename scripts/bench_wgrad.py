"""Accumulate-form (weight-gradient) GEMMs of the fine-tuning path: C[N,Kd] += A[N,M] . B[Kd,M]^T, bf16 operands, f32 C."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd import ops
shapes = [(1024, 1024, 12736), (1024, 4096, 12736), (4096, 1024, 12736), (3072, 1024, 12736), (512, 1536, 409536), (512, 1024, 25536)]
for N, Kd, M in shapes:
    a = torch.randn(N, M, device="cuda").bfloat16(); b = torch.randn(Kd, M, device="cuda").bfloat16()
    C = torch.zeros(N, Kd, device="cuda")
    f = lambda: ops.gemm_raw(N, Kd, M, a, ops.rowmap(N, 0, M), b, M, C, ops.rowmap(N, 0, Kd), ops.OCC_F32, ops.OCC_BF16, R=C, r_map=ops.rowmap(N, 0, Kd), r_dtype=ops.OCC_F32)
    for _ in range(3): f()
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 10)
    us = sorted(ts)[2] * 1e3
    print("N=%5d Kd=%5d M=%7d  %8.1f us  %6.0f TFLOP/s" % (N, Kd, M, us, 2 * N * Kd * M / us / 1e6), flush=True)

# the same products with the operands as they lie in memory ([M, N] row-major): LDS-DMA + transposing-read kernel, no transposes
from occm_amd import backend_ops as K
for with_cs in (True, False):
  print("occ_gemm_tn (bf16 operands, no transposed copies)%s:" % (", fused bias gradient" if with_cs else ""))
  for N, Kd, M in shapes:
      a = torch.randn(M, N, device="cuda").bfloat16(); b = torch.randn(M, Kd, device="cuda").bfloat16()
      C = torch.zeros(N, Kd, device="cuda"); cs = torch.zeros(N, device="cuda") if with_cs else None
      f = lambda: K.gemm_tn(M, N, Kd, a, K.full(M, N), b, K.full(M, Kd), C, Kd, colsum_out=cs, a_bf16=True, b_bf16=True, bf16_mfma=True)
      for _ in range(3): f()
      ts = []
      for r in range(5):
          e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
          e0.record()
          for _ in range(10): f()
          e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 10)
      us = sorted(ts)[2] * 1e3
      print("N=%5d Kd=%5d M=%7d  %8.1f us  %6.0f TFLOP/s" % (N, Kd, M, us, 2 * N * Kd * M / us / 1e6), flush=True)
