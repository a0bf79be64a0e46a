import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd import ops
from occm_amd._lib import lib
torch.manual_seed(0)
V = int(sys.argv[1]) if len(sys.argv) > 1 else 17
for M, N, K in [(128, 128, 128), (256, 256, 128), (300, 520, 128)]:
    x = torch.randn(M, K).bfloat16().cuda(); w = (torch.randn(N, K) * K ** -0.5).bfloat16().cuda()
    b = torch.randn(N).cuda(); r = torch.randn(M, N).cuda()
    base = x.float() @ w.float().t()
    for name, kw, ref in [("plain", {}, base), ("bias", dict(b=b), base + b), ("gelu", dict(act=ops.ACT_GELU), F.gelu(base)),
                          ("res", dict(residual=r), base + r), ("bias+gelu", dict(b=b, act=ops.ACT_GELU), F.gelu(base + b)),
                          ("bias+res", dict(b=b, residual=r), base + b + r), ("all", dict(b=b, act=ops.ACT_GELU, residual=r), F.gelu(base + b) + r)]:
        for odt in (torch.float32, torch.bfloat16):
            lib().occ_gemm_variant(V)
            out = ops.linear(x, w, kw.get("b"), act=kw.get("act", ops.ACT_NONE), residual=kw.get("residual"), out_dtype=odt)
            lib().occ_gemm_variant(1)
            err = (out.float() - ref).abs()
            bad = err > 0.05
            print(M, N, K, "%-10s" % name, str(odt)[6:], "max err %.3g bad %.3f" % (float(err.max()), float(bad.float().mean())),
                  "bad rows", sorted(set((bad.any(1).nonzero().flatten() // 16).tolist()))[:20], "bad cols", sorted(set((bad.any(0).nonzero().flatten() // 16).tolist()))[:40])
