"""Oracle AASIST / SE-ResNet34 back-ends vs vectors from the reference's models/sslassist.py
(AModel with the fairseq wrapper stubbed out) and models/senet.py."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import aasist_ref, losses_ref, senet_ref
from oracle.fill import fill_like

GA = golden("aasist.npz")
GS = golden("senet.npz")
CASES = {"a": (12, 199, 100), "b": (3, 201, 101), "c": (1, 650, 102)}


def _feats(tag):
    B, T, s = CASES[tag]
    return torch.randn(B, T, 1024, generator=torch.Generator().manual_seed(s))


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_aasist_eval(tag):
    p = fill_like(aasist_ref.param_shapes(), seed=0)
    with torch.no_grad():
        emb, out = aasist_ref.backend_forward(_feats(tag), p, train=False)
    np.testing.assert_allclose(emb.numpy(), GA["eval_emb_" + tag], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(out.numpy(), GA["eval_out_" + tag], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_aasist_train_and_grads(tag):
    p = fill_like(aasist_ref.param_shapes(), seed=0)
    for k, v in p.items():
        if v.dtype.is_floating_point and not k.split(".")[-1].startswith("running"):
            v.requires_grad_(True)
    B = CASES[tag][0]
    emb, out = aasist_ref.backend_forward(_feats(tag), p, train=True, masks=None)
    labels = (torch.arange(B) % 12 >= 6).long()
    loss = 0.1 * losses_ref.compactness_loss(emb) + 0.9 * losses_ref.descriptiveness_loss(out, labels)
    loss.backward()
    np.testing.assert_allclose(emb.detach().numpy(), GA["train_emb_" + tag], rtol=2e-4, atol=5e-5)
    np.testing.assert_allclose(out.detach().numpy(), GA["train_out_" + tag], rtol=2e-4, atol=5e-5)
    np.testing.assert_allclose(loss.item(), GA["train_loss_" + tag], rtol=1e-5)
    for key in GA.files:
        if key.startswith("rs_%s_" % tag):
            np.testing.assert_allclose(p[key[len("rs_%s_" % tag):]].detach().numpy(), GA[key], rtol=1e-4, atol=1e-6)
    names = list(GA["gradnames_" + tag]); norms = GA["gradnorms_" + tag]
    for n, ref in zip(names, norms):
        g = p[n].grad
        if ref < 0:        # reference produced no gradient (bn1 of encoder.1..5: quirk 1)
            assert g is None or float(g.abs().max()) == 0.0, n
        else:
            assert g is not None, n
            assert abs(float(g.norm()) - ref) <= 2e-3 * ref + 2e-5, (n, float(g.norm()), ref)
    for key in GA.files:
        if key.startswith("grad_%s_" % tag) and "rows0_4" not in key:
            n = key[len("grad_%s_" % tag):]
            ref = GA[key]
            np.testing.assert_allclose(p[n].grad.numpy(), ref, rtol=2e-3, atol=max(2e-3 * np.abs(ref).max(), 1e-6))
    ref = GA["grad_%s_LL.weight_rows0_4" % tag]
    np.testing.assert_allclose(p["LL.weight"].grad[:4].numpy(), ref, rtol=2e-3, atol=max(2e-3 * np.abs(ref).max(), 1e-6))


def test_aasist_no_grad_for_dead_bn1():
    names = list(GA["gradnames_a"]); norms = GA["gradnorms_a"]
    dead = sorted(n for n, v in zip(names, norms) if v < 0)
    assert dead == sorted("encoder.%d.0.bn1.%s" % (i, w) for i in range(1, 6) for w in ("weight", "bias"))


@pytest.mark.parametrize("tag,shape,seed", [("lfcc", (4, 1, 266, 13), 7), ("ssl", (2, 1, 199, 1024), 8)])
def test_senet(tag, shape, seed):
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(seed))
    p = fill_like(senet_ref.param_shapes(), seed=1)
    with torch.no_grad():
        com, des = senet_ref.senet34_forward(x, p, train=False)
    np.testing.assert_allclose(com.numpy(), GS["eval_com_" + tag], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(des.numpy(), GS["eval_des_" + tag], rtol=1e-4, atol=1e-5)
    p = fill_like(senet_ref.param_shapes(), seed=1)
    with torch.no_grad():
        com, des = senet_ref.senet34_forward(x, p, train=True)
    np.testing.assert_allclose(com.numpy(), GS["train_com_" + tag], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(des.numpy(), GS["train_des_" + tag], rtol=2e-4, atol=2e-5)


def test_oracle_training_steps_match_reference_loop():
    """tests/golden/train_steps.npz: three steps of the reference's loop body (zero_grad, forward, 0.1 c + 0.9 d, backward, Adam 1e-4)
    run by models/sslassist.py + losses/custom_loss.py + torch.optim.Adam themselves; the oracle must retrace them."""
    GT = golden("train_steps.npz")
    p = fill_like(aasist_ref.param_shapes(), seed=0)
    q = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in p.items()}
    opt = torch.optim.Adam([v for v in q.values() if torch.is_tensor(v) and v.requires_grad], lr=1e-4)
    labels = (torch.arange(12) >= 6).long()
    for step in range(3):
        feats = torch.randn(12, 199, 1024, generator=torch.Generator().manual_seed(200 + step))
        opt.zero_grad()
        emb, logit = aasist_ref.backend_forward(feats, q, train=True, masks={})
        lc, ld = losses_ref.compactness_loss(emb), losses_ref.descriptiveness_loss(logit, labels)
        (0.1 * lc + 0.9 * ld).backward()
        opt.step()
        # step 0 is a pure forward of identical parameters.  Adam's first updates are sign-like (g / (|g| + 1e-8)): parameters whose
        # gradient is rounding noise (biases in front of a train-mode BatchNorm, ...) move by +-lr in an implementation-dependent
        # direction, so later losses agree to percent level only and a parameter may differ by up to 2*lr per step.
        rt = 2e-4 if step == 0 else 2e-2
        np.testing.assert_allclose(float(lc.detach()), GT["loss_c"][step], rtol=rt)
        np.testing.assert_allclose(float(ld.detach()), GT["loss_d"][step], rtol=rt)
    for k in GT.files:
        if k.startswith("p_") and GT[k].dtype.kind == "f":
            np.testing.assert_allclose(q[k[2:]].detach().numpy(), GT[k], rtol=0, atol=6.1e-4)
    assert int(q["first_bn.num_batches_tracked"]) == int(GT["p_first_bn.num_batches_tracked"]) == 3


@pytest.mark.parametrize("tag,shape,seed", [("a", (3, 1, 48, 1024), 21), ("b", (2, 1, 199, 1024), 22)])
def test_lcnn_oracle_matches_reference_vectors(tag, shape, seed):
    """oracle/lcnn_ref.py vs tests/golden/lcnn.npz (models/lcnn.py lcnn_net(asoftmax=False) run by gen_golden.py): eval logits, train-mode
    logits with the dropouts at p = 0, BatchNorm running statistics, every parameter-gradient norm (the never-applied ``group.bn`` has
    none) and three whole gradients."""
    from oracle import lcnn_ref
    GL = golden("lcnn.npz")
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*shape, generator=g)
    p = fill_like(lcnn_ref.param_shapes(), seed=4)
    with torch.no_grad():
        y = lcnn_ref.lcnn_forward(x, p, train=False)
    np.testing.assert_allclose(y.numpy(), GL["eval_" + tag], rtol=1e-4, atol=1e-5)
    q = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in p.items()}
    y = lcnn_ref.lcnn_forward(x, q, train=True, masks={})
    np.testing.assert_allclose(y.detach().numpy(), GL["train_" + tag], rtol=2e-4, atol=2e-5)
    wgt = torch.randn(y.shape, generator=g)
    np.testing.assert_array_equal(wgt.numpy(), GL["train_wgt_" + tag])
    (y * wgt).sum().backward()
    for k in ("layer2.2.running_mean", "layer2.2.running_var", "layer3.2.running_mean", "layer3.2.running_var"):
        np.testing.assert_allclose(q[k].numpy(), GL["rs_%s_%s" % (tag, k)], rtol=1e-4, atol=1e-6)
    names, norms = list(GL["gradnames_" + tag]), GL["gradnorms_" + tag]
    dead = sorted(n for n, v in zip(names, norms) if v < 0)
    assert dead == sorted("layer%d.0.bn.%s" % (i, w) for i in (2, 3) for w in ("weight", "bias"))
    for n, v in zip(names, norms):
        if v >= 0:
            assert abs(float(q[n].grad.norm()) - v) <= 2e-3 * v + 1e-6, (n, float(q[n].grad.norm()), v)
    for k in ("layer1.0.filter.weight", "layer3.0.conv.filter.weight", "fc0.0.filter.0.weight"):
        ref = GL["grad_%s_%s" % (tag, k)]
        np.testing.assert_allclose(q[k].grad.numpy(), ref, rtol=2e-3, atol=max(2e-3 * np.abs(ref).max(), 1e-7))
