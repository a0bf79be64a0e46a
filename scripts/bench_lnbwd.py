"""layernorm backward at the fine-tuning shape (12736 x 1024, dy bf16, x f32, residual gradient f32, dx f32 + bf16 copy)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd import ops
M, C = 12736, 1024
dy = torch.randn(M, C, device="cuda").bfloat16(); x = torch.randn(M, C, device="cuda"); g = torch.randn(C, device="cuda")
dres = torch.randn(M, C, device="cuda"); dx = torch.empty(M, C, device="cuda"); dxb = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
f = lambda: ops.layernorm_bwd(dy, x, g, dres, dx, dg, db, dx_bf16=dxb)
for _ in range(3): f()
ts = []
for r in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20)
us = sorted(ts)[2] * 1e3
print("layernorm_bwd %d x %d: %.1f us, %.2f TB/s" % (M, C, us, M * C * (2 + 4 + 4 + 4 + 2) / us / 1e6))
