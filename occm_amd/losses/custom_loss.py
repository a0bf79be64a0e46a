"""Losses of the reference's ``losses/custom_loss.py`` (:4-99) on HIP kernels, same names and call signatures.

Each function returns a 0-d tensor (shape ``[1]`` for triplet / euclidean, as in the reference) that takes part in
torch autograd, so a reference-style loop (``loss.backward()``) works unchanged; the fused fast path used by
``occm_amd.trainer.OcTrainer`` calls the same kernels directly and skips autograd.
"""
import torch

from .. import ops


class _Compactness(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb):
        loss, demb = ops.compactness_loss(emb.contiguous().float(), n_groups=1, group=emb.shape[0], scale=1.0, want_grad=True)
        ctx.save_for_backward(demb)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (demb,) = ctx.saved_tensors
        return demb * g


class _Descriptiveness(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        loss, dl = ops.ce_loss(logits.contiguous().float(), labels.contiguous().long(), scale=1.0, want_grad=True)
        ctx.save_for_backward(dl)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None


def compactness_loss(batch_embeddings):
    """custom_loss.py:4-29: mean leave-one-out distance over the first six rows."""
    return _Compactness.apply(batch_embeddings)


def descriptiveness_loss(batch_embeddings, labels):
    """custom_loss.py:78-99: summed cross-entropy divided by the number of rows."""
    return _Descriptiveness.apply(batch_embeddings, labels)


def _dist(a, b):
    return ops.pairwise_dist(a.contiguous().float().reshape(-1), b.contiguous().float().reshape(1, -1))


def triplet_loss(batch_embeddings, margin=9.0):
    """custom_loss.py:32-57 (forward only: the reference never trains with it, oc_training.py:379)."""
    e = batch_embeddings.detach()
    return torch.relu(_dist(e[0], e[1]) - _dist(e[0], e[2]) + margin)


def euclidean_distance_loss(batch_embeddings):
    """custom_loss.py:59-74 (forward only)."""
    e = batch_embeddings.detach()
    pairs = [(0, 1), (0, 2), (0, 3), (2, 1), (2, 3)]
    loss = 0.0
    for i, j in pairs:
        loss = loss + _dist(e[i], e[j])
    return loss / len(pairs)
