"""Attention backward (with the fused qkv bias sums) at the bench shapes, median of 30 launches with a cache-thrashing fill in between:
    python scripts/time_attn_bwd.py            # OCC_LIB=<other build> for an A/B in one gpurun call"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd import ops

g = torch.Generator().manual_seed(0)
for B, H, hd, T in ((64, 16, 64, 199), (32, 16, 80, 199), (64, 16, 64, 99), (16, 16, 64, 256)):
    D = H * hd
    qkv = torch.randn(B * T, 3 * D, generator=g).bfloat16().cuda()
    lse = torch.empty(B * H, T, device="cuda", dtype=torch.float32)
    att = ops.attention(qkv, B, T, H, hd, hd ** -0.5, lse=lse)
    dout = torch.randn(B * T, D, generator=g).bfloat16().cuda()
    dbias = torch.zeros(3 * D, device="cuda")
    dqkv = torch.empty_like(qkv)
    junk = torch.empty(150_000_000, device="cuda")
    run = lambda: ops.attention_bwd_bias(qkv, att, dout, lse, B, T, H, hd, hd ** -0.5, dbias, dqkv=dqkv)
    for _ in range(3):
        run()
    ts = []
    for r in range(30):
        junk.fill_(float(r))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print("B %3d H %2d hd %2d T %3d   %7.1f us (median of 30; includes the bias finalize)   [%s]" % (B, H, hd, T, ts[15], os.path.basename(os.environ.get("OCC_LIB", "libocc_hip.so"))), flush=True)
    del junk
