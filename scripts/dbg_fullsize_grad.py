"""Diagnostic: per-tensor gradient agreement of the full-size fine-tune step with the CPU oracle (end to end, and with the oracle's
feature gradient injected into the front-end backward)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import aasist_ref, losses_ref, xlsr_ref
from oracle.fill import fill_like
from occm_amd import ops
from occm_amd.models import xlsr
from occm_amd.models.sslassist import AModel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
g = torch.Generator().manual_seed(31)
wav = (0.1 * torch.randn(B, 64000, generator=g)).clamp_(-1, 1)
rcfg = xlsr_ref.XlsrConfig.xlsr_300m()
p = fill_like(xlsr_ref.param_shapes(rcfg), seed=0); pb = fill_like(aasist_ref.param_shapes(), seed=0)
labels = (torch.arange(B) % 2).long()
pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
feats = xlsr_ref.extract_feat(wav, pr, rcfg)
feats.retain_grad()
emb, out = aasist_ref.backend_forward(feats, pb, train=True)
loss = losses_ref.descriptiveness_loss(out, labels); loss.backward()
model = AModel(None, "cuda", ssl_cfg=xlsr.XlsrConfig.xlsr_300m(), ssl_state_dict=p, backend_state_dict=pb, finetune_ssl="full", backend_compute="f32")
model.train(); fe, be = model.ssl_model.model, model.backend
f = fe.forward_train(wav.cuda())
print("feature err max %.3g mean %.3g" % (float((f.cpu() - feats.detach()).abs().max()), float((f.cpu() - feats.detach()).abs().mean())))
be.zero_grad(); fe.zero_grad()
e, lg = be.forward(f, train=True, masks={})
ld, dlog = ops.ce_loss(lg, labels.cuda(), scale=1.0, want_grad=True)
dfe = be.backward(None, dlog, want_dfeats=True)
r = feats.grad
d = dfe.cpu()
print("loss %.5f vs %.5f; dfeats cos %.5f rel %.4f" % (float(ld), float(loss), float((d * r).sum() / (d.norm() * r.norm())), float((d - r).abs().max() / r.abs().max())))
def report(tag):
    gd = fe.grad_dict()
    for k in ["encoder.layer_norm.bias", "encoder.layers.23.fc1.bias", "encoder.layers.23.self_attn.out_proj.weight", "encoder.layers.17.fc2.weight", "encoder.layers.11.self_attn.q_proj.weight",
              "encoder.layers.5.self_attn_layer_norm.weight", "encoder.layers.0.fc1.weight", "encoder.pos_conv.0.weight_v", "post_extract_proj.weight",
              "feature_extractor.conv_layers.3.0.weight", "feature_extractor.conv_layers.0.0.weight"]:
        a, b = gd[k].cpu().reshape(-1), pr[k].grad.reshape(-1)
        print("%s %-46s cos %.5f  rel %.4f  |ref| %.3g" % (tag, k, float((a * b).sum() / (a.norm() * b.norm() + 1e-30)), float((a - b).abs().max() / (b.abs().max() + 1e-30)), float(b.norm())))
fe.backward(dfe); report("e2e   ")
f = fe.forward_train(wav.cuda()); fe.zero_grad(); fe.backward(r.cuda().contiguous()); report("inject")
