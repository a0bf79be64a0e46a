"""Uninitialised-read screen: every torch.empty / empty_like / new_empty buffer is filled with NaN (floats) or 0x7f bytes (integers) before
the library sees it, then the training step, the frozen step and the scoring forward run and their outputs are checked for NaN.
A kernel that reads a buffer element nobody wrote shows up as NaN in the losses or the embeddings.
    python scripts/poison_check.py            # on the GPU box"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

_real = dict(empty=torch.empty, empty_like=torch.empty_like, new_empty=torch.Tensor.new_empty)
POISON = [False]


def _fill(t):
    if POISON[0] and t.is_cuda and t.numel():
        if t.dtype.is_floating_point:
            t.fill_(float("nan"))
        elif t.dtype in (torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64):
            t.fill_(0x7f)
    return t


torch.empty = lambda *a, **k: _fill(_real["empty"](*a, **k))
torch.empty_like = lambda *a, **k: _fill(_real["empty_like"](*a, **k))
torch.Tensor.new_empty = lambda self, *a, **k: _fill(_real["new_empty"](self, *a, **k))

from occm_amd.models import xlsr
from occm_amd.models.sslassist import AModel
from occm_amd.trainer import OcTrainer


def run(poison, finetune, steps=3, dim=256, heads=4, T=16000, fp8=False):
    POISON[0] = poison
    cfg = xlsr.XlsrConfig(dim=dim, ffn=2 * dim, heads=heads, layers=2)
    g = torch.Generator().manual_seed(2)
    wavs = [(0.1 * torch.randn(12, T, generator=g)).cuda() for _ in range(steps)]
    labels = (torch.arange(12) >= 6).long().cuda()
    model = AModel(None, "cuda", ssl_cfg=cfg, seed=0, synthetic_ssl=True, finetune_ssl="full" if finetune else None)
    if fp8:
        model.ssl_model.model.enable_fp8()
    model.train()
    tr = OcTrainer(model, lr=1e-6, w_compact=0.1, w_descr=0.9, train_frontend=finetune, seed=3, group_size=12, dropout_masks={}, graph_backend=False)
    out = []
    for w in wavs:
        lc, ld = tr.step(w, labels)
        out.append((float(lc), float(ld)))
    model.eval()
    with torch.no_grad():
        emb, logits = model(wavs[0])
    POISON[0] = False
    return out, emb.float().cpu(), logits.float().cpu()


import math
bad = 0
for finetune in (True, False):
    for dim, heads, fp8 in ((256, 4, False), (640, 8, False), (256, 4, True)):
        # (values are not compared with an unpoisoned run: with random-initialised AASIST weights two clean runs already differ by percents
        # from the second step on -- top-k graph pooling amplifies bf16-ulp differences; NaN is the signal)
        if fp8 and not finetune:
            continue                                                   # fp8 exists in the fine-tuning front-end only
        out, emb, logits = run(True, finetune, dim=dim, heads=heads, fp8=fp8)
        ok = all(math.isfinite(v) for pair in out for v in pair) and bool(torch.isfinite(emb).all()) and bool(torch.isfinite(logits).all())
        print("finetune=%s dim=%d hd=%d fp8=%s  poisoned losses %s  -> %s" % (finetune, dim, dim // heads, fp8, out, "ok" if ok else "NaN: UNINITIALISED READ"), flush=True)
        bad += not ok
sys.exit(1 if bad else 0)
