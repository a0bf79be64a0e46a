"""End-to-end training soak: configs[2]'s step (XLS-R-300M fine-tuned + AASIST, RawBoost 5, bf16 or fp8) for a few hundred optimizer steps on a
synthetic task WITH a learnable signal -- "spoof" utterances carry a faint 3.1 kHz tone on top of the same coloured noise -- and fresh
waveforms every step.  Reports the loss curve (descriptiveness loss must fall well below ln 2), NaN / Inf checks on every loss, and the
allocator's high-water mark at the start and the end (a leak shows up as growth).
    python scripts/soak_train.py [steps] [bs] [--fp8]   -> one JSON line (also written to gpurun_out/soak_train*.json)"""
import json, math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd.models import xlsr
from occm_amd.models.sslassist import AModel
from occm_amd.trainer import OcTrainer

args = [a for a in sys.argv[1:] if not a.startswith("--")]
steps = int(args[0]) if len(args) > 0 else 240
bs = int(args[1]) if len(args) > 1 else 48
fp8 = "--fp8" in sys.argv
L = 64000
model = AModel(None, "cuda", ssl_cfg=xlsr.XlsrConfig.xlsr_300m(), finetune_ssl="full", synthetic_ssl=True)
if fp8:
    model.ssl_model.model.enable_fp8()
model.train()
tr = OcTrainer(model, lr=1e-5, w_compact=0.0, w_descr=1.0, train_frontend=True, rawboost_algo=5, group_size=12)
labels = (torch.arange(bs) % 12 >= 6).long().cuda()
g = torch.Generator(device="cuda").manual_seed(1)
tt = torch.arange(L, device="cuda", dtype=torch.float32) / 16000.0
tone = 0.02 * torch.sin(2 * math.pi * 3100.0 * tt)


def batch():
    x = torch.randn(bs, L, device="cuda", generator=g)
    x = 0.05 * (x + 0.9 * torch.roll(x, 1, dims=1))                  # coloured noise, new every step
    return x + labels[:, None].float() * tone


losses, t0 = [], time.perf_counter()
mem0 = None
nxt = batch()
for i in range(steps):
    cur, nxt = nxt, batch()
    lc, ld = tr.step(cur, labels, next_wav=nxt)
    losses.append((lc, ld))
    if i == min(20, steps - 1):
        torch.cuda.synchronize(); mem0 = torch.cuda.max_memory_allocated()
    if i % 40 == 39:
        print("step %d  loss_d %.4f" % (i + 1, float(ld)), flush=True)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
ld = [float(b) for _, b in losses]
lc = [float(a) for a, _ in losses]
k = max(1, steps // 10)
out = {"steps": steps, "bs": bs, "fp8": fp8, "all_finite": all(math.isfinite(v) for v in ld + lc),
       "loss_d_first_tenth": sum(ld[:k]) / k, "loss_d_last_tenth": sum(ld[-k:]) / k, "loss_d_min": min(ld), "ln2": math.log(2.0),
       "utt_per_s_incl_data_gen": steps * bs / wall, "max_mem_GB_after_20_steps": mem0 / 2 ** 30, "max_mem_GB_end": torch.cuda.max_memory_allocated() / 2 ** 30,
       "loss_d_every_20": [round(v, 4) for v in ld[::20]]}
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/soak_train%s.json" % ("_fp8" if fp8 else ""), "w").write(json.dumps(out) + "\n")
sys.exit(0 if out["all_finite"] and out["loss_d_last_tenth"] < 0.8 * out["loss_d_first_tenth"] else 1)
