"""fp8 vs bf16 occ_gemm at the fine-tuning shapes (same kernel structure; fp8 = v_mfma_scale_f32_16x16x128_f8f6f4)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd import ops
g = torch.Generator().manual_seed(0)
for name, M, N, K in [("fc1", 12736, 4096, 1024), ("fc2", 12736, 1024, 4096), ("qkv", 12736, 3072, 1024), ("out", 12736, 1024, 1024), ("sq4k", 4096, 4096, 4096), ("sq8k", 8192, 8192, 8192),
                      ("1b_fc1", 6368, 5120, 1280), ("1b_fc2", 6368, 1280, 5120)]:
    xb = (torch.randn(M, K, generator=g) * 0.5).bfloat16().cuda(); wb = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda()
    xq = torch.empty(M, K, device="cuda", dtype=torch.uint8); wq = torch.empty(N, K, device="cuda", dtype=torch.uint8)
    ops.fp8_quantize(xb, xq, 5); ops.fp8_quantize(wb, wq, 5)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = {}
    for tag, a, w, dt in (("bf16", xb, wb, ops.OCC_BF16), ("fp8", xq, wq, 5)):
        f = lambda: ops.gemm_raw(M, N, K, a, ops.rowmap(M, 0, K), w, K, out, ops.rowmap(M, 0, N), ops.OCC_BF16, dt)
        for _ in range(3): f()
        ts = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 10)
        res[tag] = sorted(ts)[2]
    print("%-7s M=%6d N=%5d K=%5d  bf16 %7.1f us %6.0f TF | fp8 %7.1f us %6.0f TF" % (name, M, N, K, res["bf16"] * 1e3, 2 * M * N * K / res["bf16"] / 1e9, res["fp8"] * 1e3, 2 * M * N * K / res["fp8"] / 1e9), flush=True)
