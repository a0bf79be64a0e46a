"""LCNN back-end and the dual-branch OCCM head on the GPU through the C ABI: against tests/golden/lcnn.npz (vectors
produced by the reference's own models/lcnn.py, oracle/gen_golden.py::gen_lcnn) and against the CPU oracle's autograd."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu


def _params():
    from oracle import lcnn_ref
    from oracle.fill import fill_like
    return fill_like(lcnn_ref.param_shapes(), seed=4)


def test_mfm_kernels_match_torch_maximum_and_maxpool_including_ties():
    import ctypes
    from occm_amd._lib import check, lib, ptr, stream_ptr
    from occm_amd.ops import rowmap
    g = torch.Generator().manual_seed(0)
    B, H, W, C = 2, 7, 10, 4                                   # odd H: the last row belongs to no pooling window
    x = torch.randint(-2, 3, (B, H, W, 2 * C), generator=g).float()      # small integers: plenty of ties between halves and inside windows
    xr = x.clone().requires_grad_(True)
    a, b = xr[..., :C], xr[..., C:]
    m = torch.maximum(a, b)
    pooled = torch.nn.functional.max_pool2d(m.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    dy = torch.randn(pooled.shape, generator=g)
    (pooled * dy).sum().backward()
    xd = x.cuda()
    y = torch.empty(B, H // 2, W // 2, C, device="cuda"); idx = torch.empty(y.numel(), device="cuda", dtype=torch.uint8)
    full = rowmap(B * (H // 2) * (W // 2), 0, C)
    check(lib().occ_mfm_pool2_fwd(ptr(xd), ptr(y), ctypes.byref(full), ptr(idx), B, H, W, C, stream_ptr()), "fwd")
    assert torch.equal(y.cpu(), pooled.detach())
    dx = torch.full((B, H, W, 2 * C), 7.0, device="cuda")
    check(lib().occ_mfm_pool2_bwd(ptr(dy.cuda()), ctypes.byref(full), ptr(idx), ptr(dx), ctypes.byref(rowmap(B * H * W, 0, 2 * C)), B, H, W, C, stream_ptr()), "bwd")
    assert torch.equal(dx.cpu(), xr.grad)
    # plain MFM
    xr = x.clone().requires_grad_(True)
    m = torch.maximum(xr[..., :C], xr[..., C:])
    dm = torch.randn(m.shape, generator=g)
    (m * dm).sum().backward()
    R = B * H * W
    y = torch.empty(R, C, device="cuda")
    check(lib().occ_mfm_fwd(ptr(xd), ptr(y), ctypes.byref(rowmap(R, 0, C)), R, C, stream_ptr()), "mfm")
    assert torch.equal(y.cpu().view(m.shape), m.detach())
    dx = torch.empty(R, 2 * C, device="cuda")
    check(lib().occ_mfm_bwd(ptr(dm.cuda()), ctypes.byref(rowmap(R, 0, C)), ptr(xd), ptr(dx), R, C, stream_ptr()), "mfm bwd")
    assert torch.equal(dx.cpu().view(x.shape), xr.grad)
    # adaptive pooling, a width that does not divide (overlapping bins)
    for Wd in (128, 100):
        f = torch.randn(2, 5, Wd, 16, generator=g)
        fr = f.clone().requires_grad_(True)
        ref = torch.nn.functional.adaptive_avg_pool2d(fr.permute(0, 3, 1, 2), (1, 64)).reshape(2, -1)
        dr = torch.randn(ref.shape, generator=g)
        (ref * dr).sum().backward()
        out = torch.empty(2, 1024, device="cuda")
        check(lib().occ_adaptive_avgpool_1xw_fwd(ptr(f.cuda()), ptr(out), 2, 5, Wd, 16, 64, stream_ptr()), "ap")
        torch.testing.assert_close(out.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
        dxp = torch.empty(2, 5, Wd, 16, device="cuda")
        check(lib().occ_adaptive_avgpool_1xw_bwd(ptr(dr.cuda()), ptr(dxp), 2, 5, Wd, 16, 64, stream_ptr()), "ap bwd")
        torch.testing.assert_close(dxp.cpu(), fr.grad, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("tag,shape,seed", [("a", (3, 1, 48, 1024), 21), ("b", (2, 1, 199, 1024), 22)])
def test_lcnn_matches_reference_vectors(tag, shape, seed):
    """Eval logits, train-mode logits (dropout off, as the vectors were made), BatchNorm running statistics, every parameter-gradient norm
    and three whole gradients of sum(logits * w) -- all produced by the reference's LCNN itself."""
    from occm_amd.models.lcnn import lcnn_net
    GL = golden("lcnn.npz")
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*shape, generator=g)
    m = lcnn_net(state_dict=_params())
    m.eval()
    y = m(x.cuda())
    np.testing.assert_allclose(y.cpu().numpy(), GL["eval_" + tag], rtol=2e-4, atol=2e-5)
    m.train()
    m.backend.zero_grad()
    y = m(x.cuda(), masks={})
    np.testing.assert_allclose(y.detach().cpu().numpy(), GL["train_" + tag], rtol=5e-4, atol=5e-5)
    m.backward(torch.tensor(GL["train_wgt_" + tag]).cuda())
    sd = m.state_dict()
    for k in ("layer2.2.running_mean", "layer2.2.running_var", "layer3.2.running_mean", "layer3.2.running_var"):
        np.testing.assert_allclose(sd[k].cpu().numpy(), GL["rs_%s_%s" % (tag, k)], rtol=1e-4, atol=1e-6)
    assert int(sd["layer2.2.num_batches_tracked"]) == 1 and int(sd["layer2.0.bn.num_batches_tracked"]) == 0       # group.bn is never applied
    gd = m.backend.grad_dict()
    for n, v in zip(GL["gradnames_" + tag], GL["gradnorms_" + tag]):
        got = float(gd[str(n)].norm())
        if v < 0:
            assert got == 0.0, n
        else:
            assert abs(got - v) <= 3e-3 * v + 1e-6, (n, got, v)
    for k in ("layer1.0.filter.weight", "layer3.0.conv.filter.weight", "fc0.0.filter.0.weight"):
        ref = GL["grad_%s_%s" % (tag, k)]
        np.testing.assert_allclose(gd[k].cpu().numpy(), ref, rtol=3e-3, atol=max(3e-3 * np.abs(ref).max(), 1e-7))


def test_lcnn_dropout_masks_feature_gradient_and_state_dict_against_oracle():
    """Injected fc0 / fc1 keep-masks: logits, all parameter gradients and the gradient wrt the input features against the oracle's autograd;
    the state_dict carries exactly the reference's keys, in its order, and round-trips."""
    from oracle import lcnn_ref
    from occm_amd.models.lcnn import lcnn_net
    p = _params()
    g = torch.Generator().manual_seed(9)
    B, T = 4, 33                                                # odd sizes down the pooling chain: 33 -> 16 -> 8 -> 4
    x = torch.randn(B, 1, T, 1024, generator=g)
    masks = {"fc0": (torch.rand(B, 64, generator=g) > 0.75).to(torch.uint8), "fc1": (torch.rand(B, 64, generator=g) > 0.75).to(torch.uint8)}
    q = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in p.items()}
    xr = x.clone().requires_grad_(True)
    ref = lcnn_ref.lcnn_forward(xr, q, train=True, masks=masks)
    w = torch.randn(ref.shape, generator=g)
    (ref * w).sum().backward()
    m = lcnn_net(state_dict=p)
    assert list(m.state_dict().keys()) == list(lcnn_ref.param_shapes().keys())          # the reference's keys, in its order (gen_golden loaded them strict)
    m.train(); m.backend.zero_grad()
    y = m(x.cuda(), masks=masks)
    torch.testing.assert_close(y.detach().cpu(), ref.detach(), rtol=5e-4, atol=5e-5)
    dx = m.backward(w.cuda(), want_dfeats=True)
    gd = m.backend.grad_dict()
    gmax = max(float(v.grad.abs().max()) for v in q.values() if torch.is_tensor(v) and v.grad is not None)
    for k, v in q.items():
        if not (torch.is_tensor(v) and v.requires_grad):
            continue
        r = v.grad if v.grad is not None else torch.zeros_like(v)
        torch.testing.assert_close(gd[k].cpu(), r, rtol=3e-3, atol=3e-3 * float(r.abs().max()) + 1e-6 * gmax, msg=lambda s, k=k: k + ": " + s)
    torch.testing.assert_close(dx.cpu(), xr.grad[:, 0], rtol=3e-3, atol=3e-3 * float(xr.grad.abs().max()))
    sd = m.state_dict()
    m2 = lcnn_net(state_dict=sd)
    for k, v in m2.state_dict().items():
        assert torch.equal(v, sd[k]), k
    # device-drawn masks: a second train step differs from the first, eval is deterministic
    m.train()
    a, b = m(x.cuda()).clone(), m(x.cuda()).clone()
    assert not torch.equal(a, b)
    m.eval()
    assert torch.equal(m(x.cuda()), m(x.cuda()))


def test_occm_dual_branch_forward_backward_against_oracle():
    """OCCM (occm.py:48-67) on given front-end features: both heads against the oracle composition, and the summed feature gradient of
    the two branches against autograd; bf16-compute mode stays within the stated back-end bound of the f32 result."""
    from oracle import lcnn_ref, senet_ref
    from oracle.fill import fill_like
    from occm_amd.models import xlsr
    from occm_amd.models.occm import OCCM
    ps, pl = fill_like(senet_ref.param_shapes(), seed=1), _params()
    g = torch.Generator().manual_seed(3)
    B, T = 3, 40
    feats = torch.randn(B, T, 1024, generator=g)
    qs = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in ps.items()}
    ql = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in pl.items()}
    fr = feats.clone().requires_grad_(True)
    (com, des), lo = lcnn_ref.occm_forward(fr, qs, ql, train=True, masks={})
    wc, wd, wl = torch.randn(com.shape, generator=g), torch.randn(des.shape, generator=g), torch.randn(lo.shape, generator=g)
    ((com * wc).sum() + (des * wd).sum() + (lo * wl).sum()).backward()
    cfg = xlsr.XlsrConfig(dim=1024, ffn=512, heads=16, layers=1)
    m = OCCM("cuda", ssl_cfg=cfg, ssl_dtype=torch.float32, synthetic_ssl=True, senet_state_dict=ps, lcnn_state_dict=pl)
    m.train()
    m.senet34_branch.backend.zero_grad(); m.lcnn_branch.backend.zero_grad()
    (c2, d2), l2 = m.forward_features(feats.cuda(), masks={})
    torch.testing.assert_close(c2.detach().cpu(), com.detach(), rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(d2.detach().cpu(), des.detach(), rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(l2.detach().cpu(), lo.detach(), rtol=1e-3, atol=1e-4)
    dfe = m.backward(wc.cuda(), wd.cuda(), wl.cuda(), want_dfeats=True)
    torch.testing.assert_close(dfe.cpu(), fr.grad, rtol=5e-3, atol=5e-3 * float(fr.grad.abs().max()))
    keys = set(m.state_dict())
    assert {"senet34_branch.conv1.weight", "lcnn_branch.fc3.bias", "lcnn_branch.layer2.0.bn.running_var"} <= keys and any(k.startswith("frontend.model.") for k in keys)
    # whole module from waveforms, eval: shapes and determinism
    m.eval()
    wav = 0.1 * torch.randn(2, 16000, generator=g)
    (c3, d3), l3 = m(wav.cuda())
    assert c3.shape == (2, 128) and d3.shape == (2, 2) and l3.shape == (2, 2) and torch.isfinite(l3).all()
