import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd import ops
g = torch.Generator().manual_seed(0)
w = (torch.randn(512, 10, generator=g) * 0.3).cuda(); b = torch.randn(512, generator=g).cuda() * 0.1
ga = torch.ones(512).cuda(); be = torch.zeros(512).cuda()
for B, L in ((1, 64000), (1, 144000), (16, 64000), (16, 144000), (64, 64000)):
    wav = (0.1 * torch.randn(B, L, generator=g)).cuda()
    T = (L - 10) // 5 + 1
    out = torch.empty(B, T, 512, device="cuda", dtype=torch.bfloat16)
    ops.conv0_ln_gelu(wav, w, b, ga, be, 10, 5, torch.bfloat16, out=out); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv0_ln_gelu(wav, w, b, ga, be, 10, 5, torch.bfloat16, out=out)
    e1.record(); torch.cuda.synchronize()
    print("B=%2d L=%6d  %.1f us" % (B, L, e0.elapsed_time(e1) * 100), flush=True)
