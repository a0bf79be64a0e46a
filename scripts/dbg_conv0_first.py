import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd import ops
g = torch.Generator().manual_seed(0)
w = (torch.randn(512, 10, generator=g) * 0.3).cuda(); b = torch.randn(512, generator=g).cuda() * 0.1
ga = torch.ones(512).cuda(); be = torch.zeros(512).cuda()
big = (0.1 * torch.randn(16, 150000, generator=g)).cuda()
outb = torch.empty(16 * 30000 * 512, device="cuda", dtype=torch.bfloat16)
torch.cuda.synchronize()
for L in (64000, 100000, 100320, 120000, 133333, 140000, 141000, 64000):
    wav = big[:, :L].contiguous()
    T = (L - 10) // 5 + 1
    out = outb[: 16 * T * 512].view(16, T, 512)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ops.conv0_ln_gelu(wav, w, b, ga, be, 10, 5, torch.bfloat16, out=out)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    ops.conv0_ln_gelu(wav, w, b, ga, be, 10, 5, torch.bfloat16, out=out)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("L=%6d first call %.2f ms, second %.2f ms" % (L, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
