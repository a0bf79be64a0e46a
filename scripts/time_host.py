"""Is the fine-tuning step host-bound?  Host time to ENQUEUE a step (no synchronisation inside) against the GPU timeline per step.
usage: python scripts/time_host.py [300m|1b] [aasist|senet] [bs]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd.models import xlsr
from occm_amd.models.sslassist import AModel
from occm_amd.trainer import OcTrainer
size = sys.argv[1] if len(sys.argv) > 1 else "300m"
backend = sys.argv[2] if len(sys.argv) > 2 else "aasist"
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cfg = xlsr.XlsrConfig.xlsr_300m() if size == "300m" else xlsr.XlsrConfig.xlsr_1b()
if backend == "senet":
    from occm_amd.models.senet import ssl_resnet34
    model = ssl_resnet34("cuda", ssl_cfg=cfg, finetune_ssl="full", synthetic_ssl=True)
else:
    model = AModel(None, "cuda", ssl_cfg=cfg, finetune_ssl="full", synthetic_ssl=True)
model.train()
tr = OcTrainer(model, lr=1e-5, train_frontend=True, rawboost_algo=5, group_size=12 if bs % 12 == 0 else None)
wav = (0.1 * torch.randn(bs, 64000)).cuda(); labels = (torch.arange(bs) % 12 >= 6).long().cuda()
for _ in range(4): tr.step(wav, labels, next_wav=wav)
torch.cuda.synchronize()
n = 8
t0 = time.perf_counter()
for _ in range(n): tr.step(wav, labels, next_wav=wav)
t_host = (time.perf_counter() - t0) / n
torch.cuda.synchronize(); t_wall = (time.perf_counter() - t0) / n
print("%s + %s bs %d: host enqueue %.1f ms per step, wall %.1f ms per step" % (size, backend, bs, t_host * 1e3, t_wall * 1e3))
# the same enqueue into an EMPTY queue (back to back the host blocks on the launch queue's depth, so the number above is an upper bound)
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); tr.step(wav, labels, next_wav=wav); ts.append(time.perf_counter() - t0)
ts.sort()
print("   one step enqueued after a synchronize: %.1f ms (median of 5)" % (ts[2] * 1e3))
