"""Time occ_conv0_ln_gelu_bwd at the bench shape (bs 64, 64000 samples: 12799 frames x 512 channels per utterance, bf16 gradient)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd._lib import check, lib, ptr, stream_ptr, OCC_BF16

B, L, C, k, st = 64, 64000, 512, 10, 5
T = (L - k) // st + 1
g = torch.Generator().manual_seed(0)
wav = (0.1 * torch.randn(B, L, generator=g)).cuda()
w = (torch.randn(C, k, generator=g) * 0.3).cuda(); b = (0.05 * torch.randn(C, generator=g)).cuda()
ga = (1 + 0.1 * torch.randn(C, generator=g)).cuda(); be = (0.05 * torch.randn(C, generator=g)).cuda()
dact = torch.randn(B, T, C, generator=g).bfloat16().cuda()
dw, db, dg, dbe = torch.zeros(C, k, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")


def run():
    check(lib().occ_conv0_ln_gelu_bwd(ptr(wav), ptr(w), ptr(b), ptr(ga), ptr(be), ptr(dact), OCC_BF16, ptr(dw), ptr(db), ptr(dg), ptr(dbe), B, L, T, C, k, st, 1e-5,
                                      stream_ptr()), "conv0 bwd")


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record(); torch.cuda.synchronize()
print("conv0 bwd: %.1f us per launch; |dw| %.4e" % (e0.elapsed_time(e1) * 100, float(dw.abs().sum())))
