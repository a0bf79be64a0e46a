"""BASELINE configs[0] (the reference's CPU-runnable case): SE-ResNet34 on LFCC [B,1,266,13], loss 0.1*c + 0.9*d, Adam.
Times the GPU training step (LFCC + forward + backward + Adam, inputs resident) and the CPU oracle (numpy LFCC + torch-CPU
forward/backward/Adam) on the same synthetic batch.  usage: bench_config0.py [bs=8] [steps=50]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd.models import senet
from occm_amd.trainer import OcTrainer

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
g = torch.Generator().manual_seed(1234)
wav = (0.1 * torch.randn(bs, 64000, generator=g)).clamp(-1, 1)
labels = torch.tensor(([0] * 6 + [1] * 6) * ((bs + 11) // 12))[:bs]
SKIP_GPU = os.environ.get("CONFIG0_CPU_ONLY")
model = senet.lfcc_resnet34("cuda", seed=3)
model.train()
tr = OcTrainer(model, lr=1e-5, w_compact=0.1, w_descr=0.9, group_size=12 if bs % 12 == 0 else None)
w, l = wav.cuda(), labels.cuda()
for _ in range(5 if not SKIP_GPU else 1):
    tr.step(w, l)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps if not SKIP_GPU else 1):
    tr.step(w, l)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print("GPU  bs=%d  %.3f ms/step  %.0f utt/s" % (bs, dt * 1e3, bs / dt))

from oracle import lfcc_ref, losses_ref, senet_ref
import bench                                     # the cores the job may really use (cgroup quota), as the headline bench does
torch.set_num_threads(bench.usable_cores())
p = {k: v.detach().cpu().float().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k and "num_batches" not in k)
     for k, v in model.resnet34.state_dict().items()}
opt = torch.optim.Adam([v for v in p.values() if v.requires_grad], lr=1e-5)
nb = min(bs, 8)
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 10.0 and n < 20:
    feats = torch.from_numpy(np.stack([lfcc_ref.extract_lfcc(x.numpy().astype(np.float64)) for x in wav[:nb]])).float().unsqueeze(1)
    com, des = senet_ref.senet34_forward(feats, p, train=True)
    loss = 0.1 * losses_ref.compactness_loss(com) + 0.9 * losses_ref.descriptiveness_loss(des, labels[:nb])
    opt.zero_grad(); loss.backward(); opt.step(); n += 1
dtc = (time.perf_counter() - t0) / n
print("CPU oracle  bs=%d  %.1f ms/step  %.1f utt/s  (%d threads)" % (nb, dtc * 1e3, nb / dtc, torch.get_num_threads()))
