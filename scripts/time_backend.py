"""Back-end only (AASIST forward + losses + backward + Adam on cached features, bs 32): wall time per step with eager launches through
ctypes vs the sum of its kernel times -- is the main stream of the pipelined step launch-bound on the host?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd import ops
from occm_amd.models import xlsr
from occm_amd.models.sslassist import AModel
from occm_amd.trainer import OcTrainer

cfg = xlsr.XlsrConfig(dim=1024, ffn=64, heads=16, layers=1)          # tiny front-end: only the feature shape matters here
model = AModel(None, "cuda", ssl_cfg=cfg, ssl_dtype=torch.bfloat16, seed=0, synthetic_ssl=True)
model.train()
tr = OcTrainer(model, lr=1e-5, w_compact=0.0, w_descr=1.0)
be = tr.be
B = 32
feats = torch.randn(B, 199, 1024, device="cuda")
labels = torch.tensor(([0] * 6 + [1] * 6) * 3)[:B].cuda()

def step():
    be.zero_grad()
    emb, logits = be.forward(feats, train=True, masks=None)
    lc, demb = ops.compactness_loss(emb, n_groups=1, group=B, scale=0.0, want_grad=True)
    ld, dlog = ops.ce_loss(logits, labels, scale=1.0, want_grad=True)
    be.backward(demb, dlog)
    tr.opt.step(tr._grads, grad_scale=1.0)

for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 30
for _ in range(n): step()
t_host = (time.perf_counter() - t0) / n                  # host time to ENQUEUE a step (no sync inside)
torch.cuda.synchronize(); t_wall = (time.perf_counter() - t0) / n
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n): step()
e1.record(); torch.cuda.synchronize()
print("back-end step: host enqueue %.2f ms, wall %.2f ms, GPU timeline %.2f ms" % (t_host * 1e3, t_wall * 1e3, e0.elapsed_time(e1) / n))
