"""LayerNorm forward [12736, 1024] f32 -> bf16, plain and with the e4m3 copy + |max| (occ_layernorm_fp8); OCC_LN8_MODE picks how |max| is recorded."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd import ops
rows, C = 12736, 1024
x = torch.randn(rows, C, device="cuda"); g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
y = torch.empty(rows, C, device="cuda", dtype=torch.bfloat16); q = torch.empty(rows * C, device="cuda", dtype=torch.uint8)
sc = torch.tensor([37.0], device="cuda"); am = torch.zeros(1, device="cuda")
junk = torch.empty(150_000_000, device="cuda")
for name, fn in (("plain", lambda: ops.layernorm(x, g, b, out=y)), ("fp8 copy + amax", lambda: ops.layernorm_fp8(x, g, b, y, q, sc, am))):
    for _ in range(3): fn()
    ts = []
    for r in range(20):
        junk.fill_(float(r)); am.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); print("%-18s %6.1f us  (OCC_LN8_MODE=%s) amax %.3f" % (name, ts[10], os.environ.get("OCC_LN8_MODE", "0"), float(am)), flush=True)
