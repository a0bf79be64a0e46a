"""Micro-benchmark of occ_gemm on the front-end shapes (bf16).  OCC_GEMM_VARIANT selects the kernel."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd import ops

shapes = [("fc1", 6368, 4096, 1024, True), ("fc2", 6368, 1024, 4096, False), ("qkv", 6368, 3072, 1024, False), ("out", 6368, 1024, 1024, False),
          ("conv1", 204768, 512, 1536, False), ("conv3", 51168, 512, 1536, False), ("sq4k", 4096, 4096, 4096, False)]
g = torch.Generator().manual_seed(0)
print("variant", os.environ.get("OCC_GEMM_VARIANT", "default"))
for name, M, N, K, gelu in shapes:
    x = (torch.randn(M, K, generator=g) * 0.5).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda()
    b = torch.randn(N, generator=g).cuda()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    act = ops.ACT_GELU if gelu else ops.ACT_NONE
    for _ in range(3):
        ops.linear(x, w, b, act=act, out=out)
    ref = torch.nn.functional.linear(x.float(), w.float(), b)
    ref = torch.nn.functional.gelu(ref) if gelu else ref
    err = float((out.float() - ref).abs().max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        ops.linear(x, w, b, act=act, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print("%-6s M=%6d N=%5d K=%5d  %8.1f us  %7.1f TFLOP/s  maxerr %.3g" % (name, M, N, K, ms * 1e3, 2 * M * N * K / ms / 1e9, err))
