"""numpy restatement of the reference's EER metric (TEST ORACLE).

Follows evaluate_metrics.py:3-40.
"""
import numpy as np


def det_curve(target, nontarget):
    """evaluate_metrics.py:3-21 -- stable sort of the pooled scores, cumulative error rates."""
    nt, nn = target.size, nontarget.size
    scores = np.concatenate((target, nontarget))
    is_tar = np.concatenate((np.ones(nt), np.zeros(nn)))
    order = np.argsort(scores, kind='mergesort')
    is_tar = is_tar[order]
    tar_below = np.cumsum(is_tar)
    non_above = nn - (np.arange(1, nt + nn + 1) - tar_below)
    frr = np.concatenate(([0.0], tar_below / nt))
    far = np.concatenate(([1.0], non_above / nn))
    thr = np.concatenate(([scores[order[0]] - 0.001], scores[order]))
    return frr, far, thr


def compute_eer(target, nontarget):
    """evaluate_metrics.py:34-40."""
    frr, far, thr = det_curve(target, nontarget)
    i = int(np.argmin(np.abs(frr - far)))
    return float((frr[i] + far[i]) / 2.0), float(thr[i])


def confusion(target, nontarget, threshold):
    """evaluate_metrics.py:23-32 -> tp, tn, fp, fn."""
    return (int(np.sum(target > threshold)), int(np.sum(nontarget <= threshold)),
            int(np.sum(nontarget > threshold)), int(np.sum(target <= threshold)))
