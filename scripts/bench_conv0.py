"""Time occ_conv0_ln_gelu at bench size (32 x 64000 samples -> [32, 12799, 512] bf16)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd import ops
B, L, k, st = 32, 64000, 10, 5
T = (L - k) // st + 1
g = torch.Generator().manual_seed(0)
wav = (0.1 * torch.randn(B, L, generator=g)).cuda()
w = (torch.randn(512, k, generator=g) * 0.3).cuda(); b = torch.randn(512, generator=g).cuda() * 0.1
ga = torch.ones(512).cuda(); be = torch.zeros(512).cuda()
for dt in (torch.bfloat16, torch.float32):
    out = ops.conv0_ln_gelu(wav, w, b, ga, be, k, st, out_dtype=dt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv0_ln_gelu(wav, w, b, ga, be, k, st, out_dtype=dt)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    nbytes = wav.numel() * 4 + out.numel() * out.element_size()
    print(dt, "%.1f us  %.2f TB/s" % (us, nbytes / us / 1e6), "mean |y| %.4f" % float(out.float().abs().mean()))
