"""occ_gemm_tn (bf16 LDS-DMA kernel): in-workgroup row split (OCC_TN_KG) against workgroup-level splits, over the reduction length."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd import backend_ops as K
for N, Kd, M in [(512, 1536, 409536), (512, 1536, 204736), (512, 1536, 102336), (512, 1536, 51136), (512, 1024, 25536), (512, 1024, 12736), (1024, 512, 12736),
                 (1024, 4096, 6368), (1024, 1024, 6368), (3072, 1024, 6368)]:
    a = torch.randn(M, N, device="cuda").bfloat16(); b = torch.randn(M, Kd, device="cuda").bfloat16()
    C = torch.zeros(N, Kd, device="cuda")
    f = lambda: K.gemm_tn(M, N, Kd, a, K.full(M, N), b, K.full(M, Kd), C, Kd, a_bf16=True, b_bf16=True, bf16_mfma=True)
    for _ in range(3): f()
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 10)
    us = sorted(ts)[2] * 1e3
    print("N=%5d Kd=%5d M=%7d  %8.1f us  %6.0f TFLOP/s" % (N, Kd, M, us, 2 * N * Kd * M / us / 1e6), flush=True)
