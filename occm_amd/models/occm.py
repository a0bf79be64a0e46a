"""Dual-branch OCCM on MI355X -- drop-in for ``models/occm.py`` (OCCM :48-67): one XLS-R front-end feeding an SE-ResNet34 branch
(``(com [B,128], des [B,2])``, models/senet.py) and an LCNN branch (logits [B,2], models/lcnn.py) on the same features.

The reference runs the front-end once and hands ``x.unsqueeze(1)`` to both branches (occm.py:55-67); so does this module, and the
backward pass adds the two branches' feature gradients before the front-end's own backward.
"""
import torch

from .lcnn import lcnn_net
from .senet import se_resnet34


class OCCM(torch.nn.Module):
    def __init__(self, device="cuda", ssl_cfg=None, ssl_dtype=torch.bfloat16, ssl_state_dict=None, senet_state_dict=None, lcnn_state_dict=None,
                 finetune_ssl=False, backend_compute=None, ssl_cp_path=None, synthetic_ssl=False, seed=1):
        super().__init__()
        from .xlsr import SSLModel
        self.frontend = SSLModel(device, cp_path=ssl_cp_path, state_dict=ssl_state_dict, cfg=ssl_cfg, dtype=ssl_dtype, finetune=finetune_ssl,
                                 synthetic=synthetic_ssl)
        if backend_compute is None:
            backend_compute = "bf16" if ssl_dtype == torch.bfloat16 else "f32"
        self.senet34_branch = se_resnet34(state_dict=senet_state_dict, device=device, seed=seed, compute=backend_compute)
        self.lcnn_branch = lcnn_net(state_dict=lcnn_state_dict, device=device, seed=seed + 3, compute=backend_compute, asoftmax=False)
        self.ssl_model = self.frontend

    def forward_features(self, feats, masks=None):
        """Both branches on given front-end features [B,T,1024] (f32): ((com, des), lcnn logits)."""
        self.senet34_branch.train(self.training); self.lcnn_branch.train(self.training)
        x = feats.unsqueeze(1)
        return self.senet34_branch(x), self.lcnn_branch(x, masks=masks)

    def forward(self, x, masks=None):
        """x f32 [B,L] waveforms -> (senet34_output, lcnn_output) as occm.py:55-67."""
        return self.forward_features(self.frontend.extract_feat(x), masks=masks)

    def backward(self, dcom, ddes, dlcnn, want_dfeats=False):
        """Parameter gradients of both branches; with want_dfeats the summed gradient wrt the shared features [B,T,1024]."""
        d1 = self.senet34_branch.backward(dcom, ddes, want_dfeats=want_dfeats)
        d2 = self.lcnn_branch.backward(dlcnn, want_dfeats=want_dfeats)
        if not want_dfeats:
            return None
        from .. import backend_ops as K
        return K.axpby(d1.view(-1), d2.view(-1), d1.view(-1)).view(d1.shape)

    def state_dict(self, *a, **kw):
        """Keys as the reference module tree gives them: ``frontend.model.*``, ``senet34_branch.*``, ``lcnn_branch.*``."""
        sd = {}
        for k, v in self.frontend.full_state_dict().items():
            sd["frontend.model." + k] = v.detach().clone() if torch.is_tensor(v) else v
        sd.update({"senet34_branch." + k: v for k, v in self.senet34_branch.state_dict().items()})
        sd.update({"lcnn_branch." + k: v for k, v in self.lcnn_branch.state_dict().items()})
        return sd

    def load_state_dict(self, sd, strict=True):
        self.senet34_branch.load_state_dict({k[len("senet34_branch."):]: v for k, v in sd.items() if k.startswith("senet34_branch.")})
        self.lcnn_branch.load_state_dict({k[len("lcnn_branch."):]: v for k, v in sd.items() if k.startswith("lcnn_branch.")})
        ssl = {k[len("frontend.model."):]: v for k, v in sd.items() if k.startswith("frontend.model.")}
        if ssl or strict:
            self.frontend.load_params(ssl, strict=strict)
        return self
