"""Attention forward micro-benchmark at the bench shape (B=32, T=199, 16 heads of 64)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd import ops
for B, T, H in [(32, 199, 16), (64, 199, 16), (1, 650, 16), (8, 650, 16)]:
    qkv = torch.randn(B * T, 3 * H * 64, device="cuda").bfloat16()
    out = torch.empty(B * T, H * 64, device="cuda", dtype=torch.bfloat16)
    f = lambda: ops.attention(qkv, B, T, H, 64, 0.125, out=out)
    for _ in range(3): f()
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20)
    us = sorted(ts)[2] * 1e3
    print("B=%d T=%d H=%d  %.1f us  %.0f TFLOP/s" % (B, T, H, us, 4 * B * H * T * T * 64 / us / 1e6), flush=True)

print("backward")
for B, T, H, hd in [(64, 199, 16, 64), (32, 199, 16, 64), (8, 650, 16, 64), (32, 199, 16, 80)]:
    D = H * hd
    qkv = torch.randn(B * T, 3 * D, device="cuda").bfloat16(); do = torch.randn(B * T, D, device="cuda").bfloat16()
    lse = torch.empty(B * H, T, device="cuda"); dqkv = torch.empty_like(qkv)
    try:
        out = ops.attention(qkv, B, T, H, hd, hd ** -0.5, lse=lse)
    except Exception as e:
        print("fwd", B, T, H, hd, e); continue
    f = lambda: ops.attention_bwd(qkv, out, do, lse, B, T, H, hd, hd ** -0.5, dqkv=dqkv)
    for _ in range(3): f()
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 10)
    us = sorted(ts)[2] * 1e3
    print("bwd B=%d T=%d H=%d hd=%d  %.1f us  %.0f TFLOP/s" % (B, T, H, hd, us, 10 * B * H * T * T * hd / us / 1e6), flush=True)
