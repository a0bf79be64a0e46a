"""torch-CPU restatement of the reference's losses and distance scoring (TEST ORACLE).

Follows losses/custom_loss.py:4-99 and the scoring arithmetic of
oc_classifier.py:189-197, 261.
"""
import torch

PAIRWISE_EPS = 1e-6     # F.pairwise_distance adds eps to the *difference* (custom_loss.py:25)


def pairwise_l2(a, b, eps=PAIRWISE_EPS):
    """||a - b + eps||_2 over the last dim."""
    d = a - b + eps
    return torch.sqrt(torch.sum(d * d, dim=-1))


def compactness_loss(emb):
    """custom_loss.py:4-29 -- first six rows only (:15); leave-one-out mean distance."""
    e = emb[:6]
    n = e.shape[0]
    total = e.sum(dim=0, keepdim=True)
    others_mean = (total - e) / (n - 1)
    return pairwise_l2(e, others_mean).mean()


def descriptiveness_loss(logits, labels):
    """custom_loss.py:78-99 -- sum of per-row CE divided by the row count."""
    lse = torch.logsumexp(logits, dim=1)
    picked = logits.gather(1, labels.view(-1, 1)).squeeze(1)
    return (lse - picked).sum() / logits.shape[0]


def triplet_loss(emb, margin=9.0):
    """custom_loss.py:32-57 -- rows [bona1, bona2, spoof1]; shape [1]."""
    d_pos = pairwise_l2(emb[0:1], emb[1:2])
    d_neg = pairwise_l2(emb[0:1], emb[2:3])
    return torch.relu(d_pos - d_neg + margin)


def euclidean_distance_loss(emb):
    """custom_loss.py:59-74 -- mean of five fixed pair distances; shape [1]."""
    pairs = [(0, 1), (0, 2), (0, 3), (2, 1), (2, 3)]
    tot = 0.0
    for i, j in pairs:
        tot = tot + pairwise_l2(emb[i:i + 1], emb[j:j + 1])
    return tot / len(pairs)


def reference_embedding_and_threshold(embs):
    """oc_classifier.py:189-197: embs [N,1,E] -> mean embedding [1,E], max distance."""
    ref = embs.mean(dim=0)
    dist = pairwise_l2(ref.unsqueeze(0), embs).reshape(-1)
    return ref, dist.max(), dist
