import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd import ops
from occm_amd._lib import lib
g = torch.Generator().manual_seed(0)
def t(fn, n=10, rounds=5):
    for _ in range(3): fn()
    ts = []
    for r in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return sorted(ts)[rounds // 2] * 1e3
for M, N, K in [(6368, 4096, 64), (6368, 4096, 1024), (6368, 1024, 1024)]:
    x = (torch.randn(M, K, generator=g) * 0.5).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda()
    b = torch.randn(N, generator=g).cuda()
    ob = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    of = torch.empty(M, N, device="cuda", dtype=torch.float32)
    for v in (4, 1):
        lib().occ_gemm_variant(v)
        line = "M=%d N=%d K=%d v%d:" % (M, N, K, v)
        for bits in (0, 8):
            lib().occ_gemm_debug(bits)
            line += "  bits%d bf16-out %.1f us, f32-out %.1f us, no-bias %.1f us;" % (bits, t(lambda: ops.linear(x, w, b, out=ob)), t(lambda: ops.linear(x, w, b, out=of)), t(lambda: ops.linear(x, w, None, out=ob)))
        lib().occ_gemm_debug(0)
        print(line, flush=True)
    print("  torch copy of the bf16 output: %.1f us; fill: %.1f us" % (t(lambda: ob.copy_(ob2)) if (ob2 := torch.empty_like(ob)) is not None else 0, t(lambda: ob.fill_(1.0))))
lib().occ_gemm_variant(1)
