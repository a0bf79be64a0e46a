"""conv0 block backward (recomputed Conv1d(1->512,10,5) + LayerNorm + GELU) at the bench shape: time and a checksum of the gradients.
    python scripts/time_conv0_bwd.py      # OCC_LIB=<other build> for an A/B in one gpurun call"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd._lib import check, lib, ptr, stream_ptr, OCC_BF16
B, L = 64, 64000
T0 = (L - 10) // 5 + 1
g = torch.Generator().manual_seed(0)
wav = (0.1 * torch.randn(B, L, generator=g)).cuda()
w = (0.3 * torch.randn(512, 10, generator=g)).cuda(); bias = (0.1 * torch.randn(512, generator=g)).cuda()
gam = (1 + 0.1 * torch.randn(512, generator=g)).cuda(); bet = (0.1 * torch.randn(512, generator=g)).cuda()
dact = torch.randn(B * T0, 512, generator=g).bfloat16().cuda()
dw, db, dg, dbe = torch.zeros(512, 10, device="cuda"), torch.zeros(512, device="cuda"), torch.zeros(512, device="cuda"), torch.zeros(512, device="cuda")
run = lambda: check(lib().occ_conv0_ln_gelu_bwd(ptr(wav), ptr(w), ptr(bias), ptr(gam), ptr(bet), ptr(dact), OCC_BF16, ptr(dw), ptr(db), ptr(dg), ptr(dbe), B, L, T0, 512, 10, 5,
                                                1e-5, stream_ptr()), "occ_conv0_ln_gelu_bwd")
run(); torch.cuda.synchronize()
print("checksums (atomics across workgroups: last bits vary)  dw %.6e  db %.6e  dgamma %.6e  dbeta %.6e" % (float(dw.double().abs().sum()), float(db.double().abs().sum()), float(dg.double().abs().sum()), float(dbe.double().abs().sum())))
ts = []
for r in range(20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
print("conv0 backward bs %d: %.1f us (median of 20)   [%s]" % (B, ts[10], os.path.basename(os.environ.get("OCC_LIB", "libocc_hip.so"))))
