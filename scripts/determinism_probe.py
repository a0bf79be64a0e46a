"""Where does run-to-run variation of the second training step come from?  Two fresh trainers take the same first step; then parameters,
front-end features, back-end outputs of the second batch are compared between them."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occm_amd.models import xlsr
from occm_amd.models.sslassist import AModel
from occm_amd.trainer import OcTrainer

cfg = xlsr.XlsrConfig(dim=256, ffn=512, heads=4, layers=2)
g = torch.Generator().manual_seed(2)
wavs = [(0.1 * torch.randn(12, 16000, generator=g)).cuda() for _ in range(2)]
labels = (torch.arange(12) >= 6).long().cuda()
lr = float(os.environ.get("LR", "1e-6"))
runs = []
for r in range(2):
    model = AModel(None, "cuda", ssl_cfg=cfg, seed=0, synthetic_ssl=True, finetune_ssl="full")
    model.train()
    fe, be = model.ssl_model.model, model
    p0 = fe.P.clone()
    tr = OcTrainer(model, lr=lr, w_compact=0.1, w_descr=0.9, train_frontend=True, seed=3, group_size=12, dropout_masks={}, graph_backend=False)
    l0 = [float(v) for v in tr.step(wavs[0], labels)]
    torch.cuda.synchronize()
    rec = {"l0": l0, "P": fe.P.clone(), "dP": fe.P - p0, "G": fe.G.clone() if hasattr(fe, "G") else None}
    bp = getattr(tr.be, "P", None)
    rec["bP"] = bp.clone() if torch.is_tensor(bp) else None
    feats = fe.forward_train(wavs[1])
    rec["feats"] = feats.float().clone()
    emb, logits = tr.be.forward(feats, train=True, masks={})
    rec["emb"], rec["logits"] = emb.float().clone(), logits.float().clone()
    runs.append(rec)
a, b = runs
print("step-0 losses", a["l0"], b["l0"])
for k in ("P", "dP", "G", "bP", "feats", "emb", "logits"):
    if a[k] is None:
        print(k, "n/a"); continue
    d = (a[k] - b[k]).abs()
    print("%-6s numel %9d  max|a| %.3e  max|d| %.3e  differing %.4f  nan %d" % (k, d.numel(), float(a[k].abs().max()), float(d.max()), float((d > 0).float().mean()), int(torch.isnan(a[k]).sum())))
print("dP abs mean", float(a["dP"].abs().mean()), "max", float(a["dP"].abs().max()))
