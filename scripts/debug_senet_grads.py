import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from oracle import senet_ref
from oracle.fill import fill_like
from occm_amd.models.senet import se_resnet34
shape, seed = (2, 1, 199, 1024), 8
p = fill_like(senet_ref.param_shapes(), seed=1)
x = torch.randn(*shape, generator=torch.Generator().manual_seed(seed))
B = shape[0]
dcom = torch.randn(B, 128, generator=torch.Generator().manual_seed(20))
ddes = torch.randn(B, 2, generator=torch.Generator().manual_seed(21))
def run(dt):
    q = {k: (v.clone().to(dt).requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else (v.clone().to(dt) if v.dtype.is_floating_point else v.clone())) for k, v in p.items()}
    com, des = senet_ref.senet34_forward(x.to(dt), q, train=True)
    ((com * dcom.to(dt)).sum() + (des * ddes.to(dt)).sum()).backward()
    return {k: v.grad for k, v in q.items() if torch.is_tensor(v) and v.requires_grad}
g64 = run(torch.float64)
g32 = run(torch.float32)
net = se_resnet34(state_dict=p); net.train()
net(x.cuda()); net.backend.zero_grad(); net.backward(dcom.cuda(), ddes.cuda())
gg = net.backend.grad_dict()
for k in g64:
    m = float(g64[k].abs().max()) + 1e-30
    e32 = float((g32[k].double() - g64[k]).abs().max()) / m
    eg = float((gg[k].cpu().double() - g64[k]).abs().max()) / m
    flag = "  <<<" if eg > 2e-3 else ""
    print("%-36s max %.3e  cpu32 %.2e  hip %.2e%s" % (k, m, e32, eg, flag))
