"""Debug: which buffer of the conv-stack backward is not run-to-run bit-identical?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import xlsr_ref
from oracle.fill import fill_like
from occm_amd.models import xlsr

kw = dict(dim=1024, ffn=512, heads=16, layers=1)
px = fill_like(xlsr_ref.param_shapes(xlsr_ref.XlsrConfig(**kw)), seed=5)
ssl = xlsr.SSLModel("cuda", state_dict=px, cfg=xlsr.XlsrConfig(**kw), finetune=True)
ssl.train()
fe = ssl.model
inputs = 0.1 * torch.randn(12, 16000, generator=torch.Generator().manual_seed(1)).cuda()
df = torch.randn(12, 49, 1024, generator=torch.Generator().manual_seed(2)).cuda() * 1e-3
snaps = []
for rep in range(3):
    fe.zero_grad()
    f = fe.forward_train(inputs)
    fe.backward(df)
    ws = fe._workspace(12, 16000, slot=0)
    tr = ws["tr"]; cv = tr["conv"]
    s = {"dx": tr["dx"].clone(), "dxb": tr["dxb"].clone(), "dln": cv["dln"].clone(), "dupad": cv["dupad"].clone(), "pos_dw": fe.pos_dw.clone()}
    for i in range(7):
        s["dact%d" % i] = cv["dact"][i].clone()
        if i > 0:
            s["dpre%d" % i] = cv["dpre"][i].clone()
    for name, (o, shp, nel) in fe.tslots.items():
        s["G." + name] = fe.G[o:o + nel].clone()
    snaps.append(s)
for k in snaps[0]:
    d = [int((snaps[r][k].float() != snaps[0][k].float()).sum()) for r in (1, 2)]
    if any(d):
        print(k, "elements differing from run 0:", d, "of", snaps[0][k].numel())
