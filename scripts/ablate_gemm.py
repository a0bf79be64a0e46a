"""Ablation timing of the bf16 GEMM kernels (occ_gemm_debug bits): which part of the K loop bounds a variant."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd import ops
from occm_amd._lib import lib
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,5").split(",")]
shapes = [(6368, 4096, 1024), (4096, 4096, 4096)]
g = torch.Generator().manual_seed(0)
for M, N, K in shapes:
    x = (torch.randn(M, K, generator=g) * 0.5).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for v in variants:
        lib().occ_gemm_variant(v)
        line = "M=%d N=%d K=%d v%d:" % (M, N, K, v)
        for bits, tag in [(0, "full"), (1, "no-loads"), (2, "no-mfma"), (4, "no-reads"), (3, "reads-only"), (5, "mfma-only"), (6, "loads-only"), (7, "nothing")]:
            lib().occ_gemm_debug(bits)
            for _ in range(3):
                ops.linear(x, w, None, out=out)
            ts = []
            for r in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    ops.linear(x, w, None, out=out)
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10)
            line += "  %s %.1f us" % (tag, sorted(ts)[2] * 1e3)
        lib().occ_gemm_debug(0)
        print(line, flush=True)
lib().occ_gemm_variant(1)
