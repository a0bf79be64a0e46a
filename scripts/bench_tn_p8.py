"""Micro-benchmark of occ_gemm_tn on the fine-tuning weight-gradient shapes (bs 64): the 256x256 eight-phase kernel (OCC_TN_P8=1, default)
against the 128x128 kernels (OCC_TN_P8=0 in a second process), random operands.  usage: bench_tn_p8.py [rounds]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd import backend_ops as K
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
g = torch.Generator().manual_seed(0)
for name, M, N1, N2 in [("fc1", 12736, 4096, 1024), ("fc2", 12736, 1024, 4096), ("qkv", 12736, 3072, 1024), ("out", 12736, 1024, 1024), ("proj", 12736, 1024, 512),
                        ("conv5", 25536, 512, 1024), ("fc1_32", 6368, 4096, 1024), ("out_32", 6368, 1024, 1024)]:
    a = torch.randn(M, N1, generator=g).bfloat16().cuda(); b = torch.randn(M, N2, generator=g).bfloat16().cuda()
    C = torch.zeros(N1, N2, device="cuda")
    ts = []
    for r in range(rounds + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            K.gemm_tn(M, N1, N2, a, K.full(M, N1), b, K.full(M, N2), C, N2, a_bf16=True, b_bf16=True, bf16_mfma=True)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    ts = sorted(ts[1:]); med = ts[len(ts) // 2]
    print("%-7s M=%6d N1=%5d N2=%5d  %8.1f us  %6.0f TF (min %6.0f)" % (name, M, N1, N2, med * 1e3, 2 * M * N1 * N2 / med / 1e9, 2 * M * N1 * N2 / ts[0] / 1e9), flush=True)
