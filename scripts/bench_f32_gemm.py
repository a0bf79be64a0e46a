"""f32-operand GEMM arithmetic modes at scoring shapes: exact-f32 MFMA vs split-operand bf16 x3 vs operands rounded to bf16."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd import ops
from occm_amd._lib import OCC_F32X3
g = torch.Generator().manual_seed(0)
for M, N, K in ((199, 1024, 1024), (3184, 1024, 1024), (3184, 4096, 1024), (3184, 1024, 4096), (12736, 4096, 1024)):
    x = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) * K ** -0.5).cuda(); b = torch.randn(N, generator=g).cuda()
    out = torch.empty(M, N, device="cuda")
    line = "M=%5d N=%4d K=%4d" % (M, N, K)
    for name, ab in (("f32", ops.OCC_F32), ("x3", OCC_F32X3), ("as_bf16", ops.OCC_F32_AS_BF16)):
        run = lambda: ops.gemm_raw(M, N, K, x, ops.rowmap(M, 0, K), w, K, out, ops.rowmap(M, 0, N), ops.OCC_F32, ab, bias=b)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        line += " | %s %7.1f us %6.1f TF" % (name, us, 2 * M * N * K / us / 1e6)
    # the fast form: [xh | xl | xh] . [wh | wh | wl]^T as ONE bf16 GEMM of depth 3K on the LDS-DMA kernels (weights split once, the activation
    # split timed with the product)
    w3 = ops.split3_bf16(w, mode=1); a3 = torch.empty(M, 3 * K, device="cuda", dtype=torch.bfloat16)
    def run3():
        ops.split3_bf16(x, out=a3, mode=0)
        ops.gemm_raw(M, N, 3 * K, a3, ops.rowmap(M, 0, 3 * K), w3, 3 * K, out, ops.rowmap(M, 0, N), ops.OCC_F32, ops.OCC_BF16, bias=b)
    run3(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run3()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    line += " | split + 3K bf16 GEMM %7.1f us %6.1f TF (of the f32 product)" % (us, 2 * M * N * K / us / 1e6)
    print(line, flush=True)
