"""Oracle losses / EER vs vectors from the reference's losses/custom_loss.py and evaluate_metrics.py."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import eer_ref, losses_ref

G = golden("losses_eer.npz")
CASES = [(12, 160), (12, 128), (6, 160), (24, 32)]


@pytest.mark.parametrize("seed", range(4))
def test_losses_and_grads(seed):
    n, e = CASES[seed]
    g = torch.Generator().manual_seed(seed)
    emb = torch.randn(n, e, generator=g)
    logits = torch.randn(n, 2, generator=g)
    labels = (torch.arange(n) % 12 >= 6).long()
    np.testing.assert_allclose(losses_ref.compactness_loss(emb).numpy(), G["compact_%d" % seed], rtol=2e-6)
    np.testing.assert_allclose(losses_ref.descriptiveness_loss(logits, labels).numpy(), G["descr_%d" % seed], rtol=2e-6)
    np.testing.assert_allclose(losses_ref.triplet_loss(emb).numpy(), G["triplet_%d" % seed], rtol=2e-6)
    np.testing.assert_allclose(losses_ref.euclidean_distance_loss(emb).numpy(), G["euclid_%d" % seed], rtol=2e-6)
    emb.requires_grad_(True); logits.requires_grad_(True)
    (0.1 * losses_ref.compactness_loss(emb) + 0.9 * losses_ref.descriptiveness_loss(logits, labels)).backward()
    np.testing.assert_allclose(emb.grad.numpy(), G["gemb_%d" % seed], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(logits.grad.numpy(), G["glogits_%d" % seed], rtol=1e-5, atol=1e-8)


def test_survey_spot_values():
    # SURVEY.md section 8c probe: seed-0 randn(12,160) -> 14.41619 ; CE on randn(12,2) -> 0.897358
    g = torch.Generator().manual_seed(0)
    emb = torch.randn(12, 160, generator=g)
    assert abs(float(losses_ref.compactness_loss(emb)) - float(G["compact_0"])) < 1e-5


@pytest.mark.parametrize("seed", range(3))
def test_eer(seed):
    rs = np.random.RandomState(seed)
    tar = rs.randn(700) + 1.0
    non = rs.randn(1300) - 0.5
    if seed == 2:
        tar = np.round(tar, 1); non = np.round(non, 1)
    eer, thr = eer_ref.compute_eer(tar, non)
    np.testing.assert_allclose([eer, thr], G["eer_%d" % seed], rtol=0, atol=0)
    np.testing.assert_array_equal(np.array(eer_ref.confusion(tar, non, thr)), G["conf_%d" % seed])
