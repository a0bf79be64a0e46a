"""EER scoring with the call surface of the reference's ``evaluate_metrics.py`` (compute_det_curve :3-21,
compute_eer :34-40).  Host numpy: at most a few hundred thousand f64 scores per evaluation, sorted once
 -- not a GPU workload.

Formulation: one stable ascending sort of [targets ; nontargets]; a sorted slot came from a target iff its source
index is below ``n_tgt``.  Running counts of each class up to every slot give the miss rate (targets at or below the
slot) and false-accept rate (nontargets above it); one extra operating point in front ("accept everything") sits 0.001
below the smallest score, as in the reference.  All counts are exact integers, so the rates are bit-identical to the
reference's float cumsum form."""
import numpy as np


def _sorted_classes(target_scores, nontarget_scores):
    tgt = np.asarray(target_scores).ravel()
    non = np.asarray(nontarget_scores).ravel()
    pooled = np.concatenate((tgt, non))
    order = np.argsort(pooled, kind="stable")           # ties keep targets-before-nontargets, input order within a class
    return tgt.size, non.size, pooled[order], order < tgt.size


def compute_det_curve(target_scores, nontarget_scores):
    """-> (frr, far, thresholds), each of length n_tgt + n_non + 1."""
    n_tgt, n_non, ranked, from_tgt = _sorted_classes(target_scores, nontarget_scores)
    missed = np.cumsum(from_tgt)                         # targets scoring <= ranked[k]
    rejected = np.cumsum(~from_tgt)                      # nontargets scoring <= ranked[k]
    frr = np.empty(ranked.size + 1)
    far = np.empty(ranked.size + 1)
    thresholds = np.empty(ranked.size + 1, dtype=ranked.dtype)
    frr[0], far[0], thresholds[0] = 0.0, 1.0, ranked[0] - 0.001
    frr[1:] = missed / n_tgt
    far[1:] = (n_non - rejected) / n_non
    thresholds[1:] = ranked
    return frr, far, thresholds


def calculate_confusion_matrix(target_scores, nontarget_scores, threshold):
    """-> (tp, tn, fp, fn) with "accept" meaning score > threshold."""
    accepted_tgt = int(np.count_nonzero(np.asarray(target_scores) > threshold))
    accepted_non = int(np.count_nonzero(np.asarray(nontarget_scores) > threshold))
    return accepted_tgt, np.size(nontarget_scores) - accepted_non, accepted_non, np.size(target_scores) - accepted_tgt


def compute_eer(target_scores, nontarget_scores):
    """-> (eer, threshold): the operating point where |frr - far| is smallest (first one on ties)."""
    frr, far, thresholds = compute_det_curve(target_scores, nontarget_scores)
    k = int(np.abs(frr - far).argmin())
    return 0.5 * (frr[k] + far[k]), thresholds[k]
