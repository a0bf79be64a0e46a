"""What does the |max| of an fp8 side output cost a GEMM?  fc1-shaped launch (M 12736, N 4096, K 1024) in e4m3 with the fp8 copy of its bf16 result,
with and without the |max| pointer (one guarded atomic per WAVE on one address), and the bf16 launch of the same shape."""
import os, sys, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd import ops
from occm_amd._lib import ACT_GELU_KEEP_GRAD, OCC_BF16, OCC_FP8_E4M3

M, N, K = 12736, 4096, 1024
g = torch.Generator().manual_seed(0)
x8 = torch.randint(0, 255, (M, K), generator=g, dtype=torch.uint8).cuda() & 0x77
w8 = torch.randint(0, 255, (N, K), generator=g, dtype=torch.uint8).cuda() & 0x77
xb = torch.randn(M, K, generator=g).bfloat16().cuda(); wb = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda()
bias = torch.randn(N, generator=g).cuda()
C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); aux = torch.empty_like(C)
q = torch.empty(M * N, device="cuda", dtype=torch.uint8); sc = torch.tensor([3.0], device="cuda"); am = torch.zeros(1, device="cuda")
one = torch.tensor([1.0e-3], device="cuda")


class _NullAmax:                       # a c_f8 tuple member whose data_ptr() is NULL: the epilogue then records no |max|
    def data_ptr(self): return 0


junk = torch.empty(150_000_000, device="cuda")
cases = (("bf16", lambda: ops.gemm_raw(M, N, K, xb, ops.rowmap(M, 0, K), wb, K, C, ops.rowmap(M, 0, N), OCC_BF16, OCC_BF16, bias=bias, act=ACT_GELU_KEEP_GRAD, aux=aux)),
         ("fp8, no side output", lambda: ops.gemm_raw(M, N, K, x8, ops.rowmap(M, 0, K), w8, K, C, ops.rowmap(M, 0, N), OCC_BF16, OCC_FP8_E4M3, bias=bias, act=ACT_GELU_KEEP_GRAD, aux=aux, a_dequant=one, w_dequant=one)),
         ("fp8 + fp8 copy", lambda: ops.gemm_raw(M, N, K, x8, ops.rowmap(M, 0, K), w8, K, C, ops.rowmap(M, 0, N), OCC_BF16, OCC_FP8_E4M3, bias=bias, act=ACT_GELU_KEEP_GRAD, aux=aux, a_dequant=one, w_dequant=one, c_f8=(q, sc, _NullAmax(), OCC_FP8_E4M3))),
         ("fp8 + fp8 copy + |max|", lambda: ops.gemm_raw(M, N, K, x8, ops.rowmap(M, 0, K), w8, K, C, ops.rowmap(M, 0, N), OCC_BF16, OCC_FP8_E4M3, bias=bias, act=ACT_GELU_KEEP_GRAD, aux=aux, a_dequant=one, w_dequant=one, c_f8=(q, sc, am, OCC_FP8_E4M3))))
for name, fn in cases:
    for _ in range(3): fn()
    ts = []
    for r in range(16):
        junk.fill_(float(r)); am.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); print("%-26s %7.1f us" % (name, ts[8]), flush=True)
