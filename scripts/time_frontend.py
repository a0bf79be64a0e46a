"""Time the XLS-R-300M front-end forward at bench size (synthetic weights/inputs)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd.models import xlsr

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = xlsr.XlsrConfig.xlsr_300m()
t0 = time.time()
p = xlsr.synthetic_params(cfg, 0)
fe = xlsr.XlsrFrontend(p, cfg, dtype=torch.bfloat16)
print("pack %.1fs" % (time.time() - t0), flush=True)
wav = (0.1 * torch.randn(B, 64000, generator=torch.Generator().manual_seed(1234))).clamp(-1, 1).cuda()
for _ in range(2):
    out = fe.forward(wav)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 5
for _ in range(n):
    out = fe.forward(wav)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print("B=%d fwd %.2f ms  -> %.0f utt/s, %.1f TFLOP/s (147.275 GF/utt)" % (B, ms, B / ms * 1e3, B * 147.275 / ms))
print("finite:", bool(torch.isfinite(out.float()).all()), out.float().abs().mean().item())
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = fe.forward(wav)
g.replay(); torch.cuda.synchronize()
e0.record()
for _ in range(n):
    g.replay()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print("graph replay: %.2f ms -> %.0f utt/s, %.1f TFLOP/s" % (ms, B / ms * 1e3, B * 147.275 / ms))
