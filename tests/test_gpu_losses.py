"""HIP losses / scoring / Adam (through the C ABI) vs the reference's outputs and torch.optim.Adam."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu
G = golden("losses_eer.npz")
CASES = [(12, 160), (12, 128), (6, 160), (24, 32)]


@pytest.mark.parametrize("seed", range(4))
def test_losses_and_grads_match_reference(seed):
    from occm_amd import ops
    n, e = CASES[seed]
    g = torch.Generator().manual_seed(seed)
    emb = torch.randn(n, e, generator=g).cuda()
    logits = torch.randn(n, 2, generator=g).cuda()
    labels = (torch.arange(n) % 12 >= 6).long().cuda()
    c, demb = ops.compactness_loss(emb, scale=0.1, want_grad=True)
    d, dlog = ops.ce_loss(logits, labels, scale=0.9, want_grad=True)
    np.testing.assert_allclose(c.item(), G["compact_%d" % seed], rtol=1e-5)
    np.testing.assert_allclose(d.item(), G["descr_%d" % seed], rtol=1e-5)
    np.testing.assert_allclose(demb.cpu().numpy(), G["gemb_%d" % seed], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(dlog.cpu().numpy(), G["glogits_%d" % seed], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("seed", range(4))
def test_triplet_and_euclidean_losses_match_reference_and_differentiate(seed):
    """custom_loss.py:32-74 through occ_pair_dist_loss: values vs the reference's own outputs (losses_eer.npz), gradients vs the
    oracle's autograd; the losses take part in torch autograd like the reference's (shape [1])."""
    from occm_amd.losses import custom_loss as cl
    from oracle import losses_ref
    n, e = CASES[seed]
    emb = torch.randn(n, e, generator=torch.Generator().manual_seed(seed))
    for name, fn, ref_fn, kw in (("triplet", cl.triplet_loss, losses_ref.triplet_loss, {}), ("euclid", cl.euclidean_distance_loss, losses_ref.euclidean_distance_loss, {}),
                                 ("triplet_shut", cl.triplet_loss, losses_ref.triplet_loss, {"margin": -50.0})):
        x = emb.clone().cuda().requires_grad_(True)
        loss = fn(x, **kw)
        assert tuple(loss.shape) == (1,)
        if not kw:
            np.testing.assert_allclose(loss.detach().cpu().numpy(), G["%s_%d" % (name, seed)], rtol=1e-5)
        (loss * 0.7).sum().backward()
        xr = emb.clone().requires_grad_(True)
        lr = ref_fn(xr, **kw)
        (lr * 0.7).sum().backward()
        np.testing.assert_allclose(loss.detach().cpu().numpy(), lr.detach().numpy(), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(x.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-7)
    if "shut" in name:
        assert float(loss) == 0.0 and float(x.grad.abs().max()) == 0.0


def test_grouped_compactness_is_mean_of_group_values():
    from occm_amd import ops
    from oracle import losses_ref
    emb = torch.randn(36, 160, generator=torch.Generator().manual_seed(1))
    ref = torch.stack([losses_ref.compactness_loss(emb[g * 12:(g + 1) * 12]) for g in range(3)]).mean()
    c, _ = ops.compactness_loss(emb.cuda(), n_groups=3, group=12)
    np.testing.assert_allclose(c.item(), ref.item(), rtol=1e-5)


def test_pairwise_dist_and_threshold():
    from occm_amd import ops
    from oracle import losses_ref
    embs = torch.randn(50, 1, 160, generator=torch.Generator().manual_seed(2))
    ref, thr, dist = losses_ref.reference_embedding_and_threshold(embs)
    d = ops.pairwise_dist(ref.reshape(-1).cuda(), embs.reshape(50, 160).cuda())
    np.testing.assert_allclose(d.cpu().numpy(), dist.numpy(), rtol=1e-5)
    np.testing.assert_allclose(float(d.max()), float(thr), rtol=1e-5)


def test_adam_multi_matches_torch_adam():
    from occm_amd import ops
    g = torch.Generator().manual_seed(0)
    shapes = [(128, 1024), (64,), (2, 160), (1, 42, 64), (5,)]
    ps = [torch.randn(s, generator=g) for s in shapes]
    ref = [p.clone().requires_grad_(True) for p in ps]
    opt = torch.optim.Adam(ref, lr=1e-3)
    dev = [p.clone().cuda() for p in ps]
    # tensors 0 and 3 also get a bf16 mirror written by the same launch (the GEMM operands of a fine-tuned front-end); 4 is odd-sized
    mirrors = [torch.zeros(p.numel(), device="cuda", dtype=torch.bfloat16) if i in (0, 3, 4) else None for i, p in enumerate(ps)]
    mine = ops.AdamMulti(dev, lr=1e-3, bf16_copies=mirrors)
    for step in range(5):
        grads = [torch.randn(s, generator=g) for s in shapes]
        grads_dev = [x.cuda() for x in grads]
        if step == 2:
            grads[1] = None; grads_dev[1] = None            # parameter without a gradient (dead bn1)
        for p, gr in zip(ref, grads):
            p.grad = gr
        opt.step()
        mine.step(grads_dev)
    for p, q in zip(ref, dev):
        torch.testing.assert_close(q.cpu(), p.detach(), rtol=2e-5, atol=2e-6)
    for q, m in zip(dev, mirrors):
        if m is not None:
            assert torch.equal(m.view(q.shape), q.bfloat16())
