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


class _PairDist(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, pairs, weights, bias, relu):
        loss, demb = ops.pair_dist_loss(emb.contiguous().float(), pairs, weights, bias=bias, relu=relu, want_grad=True)
        ctx.save_for_backward(demb)
        return loss                                              # shape [1], as F.pairwise_distance of two [1,E] rows gives

    @staticmethod
    def backward(ctx, g):
        (demb,) = ctx.saved_tensors
        return demb * g, None, None, None, None


def triplet_loss(batch_embeddings, margin=9.0):
    """custom_loss.py:32-57: relu(||bona1 - bona2|| - ||bona1 - spoof1|| + margin) on rows [0, 1, 2]; value and gradient from
    occ_pair_dist_loss."""
    return _PairDist.apply(batch_embeddings, ((0, 1), (0, 2)), (1.0, -1.0), float(margin), True)


def euclidean_distance_loss(batch_embeddings):
    """custom_loss.py:59-74: mean distance over the pairs (0,1), (0,2), (0,3), (2,1), (2,3)."""
    pairs = ((0, 1), (0, 2), (0, 3), (2, 1), (2, 3))
    return _PairDist.apply(batch_embeddings, pairs, (1.0 / len(pairs),) * len(pairs), 0.0, False)
