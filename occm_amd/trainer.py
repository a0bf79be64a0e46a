"""One training step of the hot path: (RawBoost ->) XLS-R front-end -> AASIST back-end -> losses -> backward ->
gradient all-reduce -> Adam.  Mirrors the loop body of oc_training.py:363-385."""
import torch

from . import ops
from .parallel import FlatGradAllReducer


class OcTrainer:
    """model: occm_amd.models.sslassist.AModel.  Loss weights default to the committed 0.0 / 1.0 (oc_training.py:380-381);
    the SE-ResNet script uses 0.1 / 0.9 (test_dataloader_v2.py:127)."""

    def __init__(self, model, lr=1e-5, w_compact=0.0, w_descr=1.0, train_frontend=False, group_size=None):
        if train_frontend:
            raise NotImplementedError("fine-tuning the XLS-R front-end (backward through the transformer) is not built yet")
        self.model = model
        self.be = model.backend
        self.w_c, self.w_d = w_compact, w_descr
        self.group_size = group_size
        self.opt = ops.AdamMulti([self.be.P], lr=lr)
        self._grads = [self.be.G]
        self.reducer = FlatGradAllReducer(self.be.G)
        self.last = None

    def step(self, wav, labels):
        """wav f32 [B,L] cuda, labels i64 [B] cuda.  Returns device tensors (loss_c, loss_d); no host sync."""
        be = self.be
        feats = self.model.ssl_model.model.forward(wav, out_dtype=torch.float32)
        be.zero_grad()
        emb, logits = be.forward(feats, train=True)
        B = emb.shape[0]
        ng = 1 if not self.group_size else B // self.group_size
        lc, demb = ops.compactness_loss(emb, n_groups=ng, group=self.group_size or B, scale=self.w_c, want_grad=True)
        ld, dlog = ops.ce_loss(logits, labels, scale=self.w_d, want_grad=True)
        be.backward(demb, dlog)
        self.reducer.all_reduce()
        self.opt.step(self._grads, grad_scale=self.reducer.grad_scale)
        self.last = (lc, ld)
        return lc, ld
