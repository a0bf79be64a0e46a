"""One attention-backward launch shape for counter collection (rocprofv3 --pmc ... -- python3 scripts/pmc_attn_bwd.py)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd import ops
B, T, H, hd = 64, 199, 16, 64
D = H * hd
qkv = torch.randn(B * T, 3 * D, device="cuda").bfloat16(); do = torch.randn(B * T, D, device="cuda").bfloat16()
lse = torch.empty(B * H, T, device="cuda"); dqkv = torch.empty_like(qkv)
out = ops.attention(qkv, B, T, H, hd, hd ** -0.5, lse=lse)
for _ in range(5):
    ops.attention_bwd(qkv, out, do, lse, B, T, H, hd, hd ** -0.5, dqkv=dqkv)
torch.cuda.synchronize()
