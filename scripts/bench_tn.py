"""Micro-benchmark of occ_gemm_tn on the back-end's conv weight-gradient shapes (f32 operands; exact-f32 vs bf16 MFMA)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd import backend_ops as K
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
B, H, W = 32, 42, 66
for co, ci in [(64, 64), (32, 32), (64, 32)]:
    Wp = W + 2
    M = B * H * W
    dy = torch.randn(M, co, device="cuda")
    img = torch.randn(B, H + 2, Wp, ci, device="cuda")
    C = torch.zeros(co, 6 * ci, device="cuda")
    bmap = K.rowmap(H * W, (H + 2) * Wp * ci, ci, W, Wp * ci)
    for mf in (False, True):
        us = t(lambda: K.gemm_tn(M, co, 6 * ci, dy, K.full(M, co), img, bmap, C, 6 * ci, b_seg=(2, 3 * ci, Wp * ci), bf16_mfma=mf))
        print("conv wgrad M=%d co=%d K=%d  %s  %.1f us  (%.1f TFLOP/s, %.2f TB/s of unique operand bytes)" % (
            M, co, 6 * ci, "bf16-mfma" if mf else "f32-mfma ", us, 2 * M * co * 6 * ci / us / 1e6, (dy.numel() + img.numel()) * 4 / us / 1e6), flush=True)
