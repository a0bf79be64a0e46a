"""One training step of the hot path: (RawBoost ->) XLS-R front-end -> AASIST or SE-ResNet34 back-end -> losses ->
backward -> gradient all-reduce -> Adam.  Mirrors the loop bodies of oc_training.py:363-385 and
test_dataloader_v2.py:107-130."""
import torch

from . import ops
from .parallel import FlatGradAllReducer


class OcTrainer:
    """model: occm_amd.models.sslassist.AModel or occm_amd.models.senet.ssl_resnet34 (anything with ``.ssl_model.model`` and a
    ``.backend`` engine offering forward/backward/zero_grad and flat P/G buffers).  Loss weights default to the committed
    0.0 / 1.0 (oc_training.py:380-381); the SE-ResNet script uses 0.1 / 0.9 (test_dataloader_v2.py:127)."""

    def __init__(self, model, lr=1e-5, w_compact=0.0, w_descr=1.0, train_frontend=False, group_size=None, dropout_masks=None, rawboost_algo=0,
                 rawboost_args=None, seed=0):
        self.model = model
        self.rawboost_algo, self.rawboost_args, self.seed, self.nstep = rawboost_algo, rawboost_args, seed, 0
        self.dropout_masks = dropout_masks      # None: draw masks on the device (normal training); {}: no dropout; dict: injected keep-masks
        self.be = model.backend
        self.fe = model.ssl_model.model
        self.train_frontend = train_frontend
        if train_frontend and not hasattr(self.fe, "forward_train"):
            raise ValueError("train_frontend=True needs a model built with finetune_ssl=True")
        self.w_c, self.w_d = w_compact, w_descr
        self.group_size = group_size
        params, self._grads = [self.be.P], [self.be.G]
        if train_frontend:                     # transformer encoder of XLS-R is trained; the conv stack stays frozen this round
            params.append(self.fe.P); self._grads.append(self.fe.G)
        self.opt = ops.AdamMulti(params, lr=lr)
        self.reducers = [FlatGradAllReducer(g) for g in self._grads]
        self.reducer = self.reducers[0]
        self.last = None

    def step(self, wav, labels):
        """wav f32 [B,L] cuda, labels i64 [B] cuda.  Returns device tensors (loss_c, loss_d); no host sync."""
        be = self.be
        if self.rawboost_algo:          # on-GPU RawBoost (data_utils_SSL.py:111-173; the call the reference leaves commented at oc_training.py:221)
            from .RawBoost import rawboost_batch_device
            from .oc_training import rawboost_args
            wav = rawboost_batch_device(wav, self.rawboost_args or rawboost_args(), self.rawboost_algo, seed=self.seed, step=self.nstep)
        self.nstep += 1
        if self.train_frontend:
            return self._step_finetune(wav, labels)
        feats = self.model.ssl_model.model.forward(wav, out_dtype=torch.float32)
        be.zero_grad()
        emb, logits = be.forward(feats, train=True, masks=self.dropout_masks)
        B = emb.shape[0]
        ng = 1 if not self.group_size else B // self.group_size
        lc, demb = ops.compactness_loss(emb, n_groups=ng, group=self.group_size or B, scale=self.w_c, want_grad=True)
        ld, dlog = ops.ce_loss(logits, labels, scale=self.w_d, want_grad=True)
        be.backward(demb, dlog)
        self.reducer.all_reduce()
        self.opt.step(self._grads, grad_scale=self.reducer.grad_scale)
        self.last = (lc, ld)
        return lc, ld

    def _step_finetune(self, wav, labels):
        be, fe = self.be, self.fe
        feats = fe.forward_train(wav)
        be.zero_grad(); fe.zero_grad()
        emb, logits = be.forward(feats, train=True, masks=self.dropout_masks)
        B = emb.shape[0]
        ng = 1 if not self.group_size else B // self.group_size
        lc, demb = ops.compactness_loss(emb, n_groups=ng, group=self.group_size or B, scale=self.w_c, want_grad=True)
        ld, dlog = ops.ce_loss(logits, labels, scale=self.w_d, want_grad=True)
        dfeats = be.backward(demb, dlog, want_dfeats=True)
        fe.backward(dfeats)
        for r in self.reducers:
            r.all_reduce()
        self.opt.step(self._grads, grad_scale=self.reducer.grad_scale)
        fe.refresh_operands()
        self.last = (lc, ld)
        return lc, ld
