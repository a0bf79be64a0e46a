"""Yardstick helper: run torch's bf16 addmm (hipBLASLt) on the front-end GEMM shapes so a kernel trace shows which
macro-tile / depth-U the vendor solution picks per shape.  Not part of the product."""
import torch
shapes = [("fc1", 6368, 4096, 1024), ("fc2", 6368, 1024, 4096), ("qkv", 6368, 3072, 1024), ("out", 6368, 1024, 1024),
          ("conv1", 204768, 512, 1536), ("conv3", 51168, 512, 1536), ("conv5", 12768, 512, 1024), ("sq4k", 4096, 4096, 4096)]
for name, M, N, K in shapes:
    x = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        torch.addmm(b, x, w.t(), out=out)
    torch.cuda.synchronize()
