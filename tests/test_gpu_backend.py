"""HIP AASIST back-end (C ABI) vs the reference's own outputs (tests/golden/aasist.npz, produced by
models/sslassist.py with the fairseq wrapper stubbed) and vs the torch-CPU oracle for train mode with
injected dropout masks.  Tolerance: 1e-3 (north-star fp32 bar) on O(1) outputs; gradients 2e-3 relative
to the tensor's max."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden

pytestmark = pytest.mark.gpu
GA = golden("aasist.npz")
CASES = {"a": (12, 199, 100), "b": (3, 201, 101), "c": (1, 650, 102)}
# Absolute floor for gradient comparisons.  Several gradients (BatchNorm scale/shift of the 1-channel stem, biases in
# front of a train-mode BatchNorm or a softmax) are sums of ~1e5 cancelling f32 terms: an f64 run of the oracle gives
# first_bn.weight.grad = 5.73e-3 where the reference's own f32 run gives 5.98e-3, so f32 implementations of the same
# arithmetic differ by a few 1e-4 there.  5e-4 is half the north-star 1e-3 bar.
GRAD_FLOOR = 5e-4


def _r(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _feats(tag):
    B, T, s = CASES[tag]
    return torch.randn(B, T, 1024, generator=torch.Generator().manual_seed(s))


def _params():
    from oracle import aasist_ref
    from oracle.fill import fill_like
    return fill_like(aasist_ref.param_shapes(), seed=0)


# ------------------------------------------------------------------------------- kernel level ---
@pytest.mark.parametrize("M,N1,N2", [(1000, 64, 64), (88704, 32, 192), (517, 128, 1024), (33, 4, 24), (4096, 2, 160) if False else (4096, 4, 160)])
def test_gemm_tn_and_colsum(M, N1, N2):
    from occm_amd import backend_ops as K
    a, b = _r(M, N1, seed=1), _r(M, N2, seed=2)
    C = torch.zeros(N1, N2).cuda()
    K.gemm_tn(M, N1, N2, a.cuda(), K.full(M, N1), b.cuda(), K.full(M, N2), C, N2)
    ref = a.double().T @ b.double()
    tol = 2e-6 * float(ref.abs().max()) + 1e-4
    torch.testing.assert_close(C.cpu().double(), ref, rtol=1e-4, atol=tol)
    s = torch.zeros(N1).cuda()
    K.colsum(a.cuda(), K.full(M, N1), M, N1, s)
    torch.testing.assert_close(s.cpu().double(), a.double().sum(0), rtol=1e-4, atol=1e-3)
    C2, s2 = torch.zeros(N1, N2).cuda(), torch.zeros(N1).cuda()         # bias gradient fused into the same launch
    K.gemm_tn(M, N1, N2, a.cuda(), K.full(M, N1), b.cuda(), K.full(M, N2), C2, N2, colsum_out=s2)
    torch.testing.assert_close(C2.cpu().double(), ref, rtol=1e-4, atol=tol)
    torch.testing.assert_close(s2.cpu().double(), a.double().sum(0), rtol=1e-4, atol=1e-3)


def test_conv2d_2x3_fwd_dgrad_wgrad_as_gemms():
    """One (2,3) pad (1,1) conv over a zero-bordered channels-last image: forward, input grad, weight grad."""
    from occm_amd import backend_ops as K, ops
    B, H, W, ci, co = 2, 42, 66, 32, 64
    Wp = W + 2
    x = _r(B, ci, H, W, seed=3).requires_grad_(True)
    w = _r(co, ci, 2, 3, seed=4, scale=0.1).requires_grad_(True)
    bias = _r(co, seed=5)
    y = F.conv2d(x, w, bias, padding=(1, 1))                      # [B,co,43,W]
    gy = _r(B, co, 43, W, seed=6)
    y.backward(gy)
    X = torch.zeros(B, 44, Wp, ci); X[:, 1:43, 1:W + 1] = x.detach().permute(0, 2, 3, 1)
    X = X.cuda()
    wi = w.detach().permute(0, 2, 3, 1).contiguous().cuda()       # [co,2,3,ci]
    R1 = B * 43 * W
    out = torch.empty(R1, co, device="cuda")
    amap = ops.rowmap(43 * W, 44 * Wp * ci, ci, W, Wp * ci)
    ops.gemm_raw(R1, co, 6 * ci, X, amap, wi, 6 * ci, out, ops.rowmap(R1, 0, co), ops.OCC_F32, ops.OCC_F32, bias=bias.cuda(), a_seg=(2, 3 * ci, Wp * ci))
    torch.testing.assert_close(out.cpu().view(B, 43, W, co).permute(0, 3, 1, 2), y.detach(), rtol=1e-4, atol=1e-4)
    # weight gradient
    D1 = torch.zeros(B, 43, Wp, co); D1[:, :, 1:W + 1] = gy.permute(0, 2, 3, 1)
    D1 = D1.cuda()
    dmap = ops.rowmap(43 * W, 43 * Wp * co, co, W, Wp * co)
    gw = torch.zeros(co, 2, 3, ci, device="cuda")
    K.gemm_tn(R1, co, 6 * ci, D1.data_ptr() + co * 4, dmap, X, amap, gw, 6 * ci, b_seg=(2, 3 * ci, Wp * ci))
    torch.testing.assert_close(gw.cpu().permute(0, 3, 1, 2), w.grad, rtol=1e-3, atol=1e-3 * float(w.grad.abs().max()))
    gb = torch.zeros(co, device="cuda")
    K.colsum(D1.data_ptr() + co * 4, dmap, R1, co, gb)
    torch.testing.assert_close(gb.cpu(), gy.sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
    # input gradient: flipped / transposed weights over the padded output gradient
    wd = torch.empty(ci, 2, 3, co, device="cuda")
    K.copy_strided(wi, wd, 3 * ci + 2 * ci, (ci, 2, 3, co), (1, -3 * ci, -ci, 6 * ci))
    R = B * 42 * W
    gx = torch.empty(R, ci, device="cuda")
    ops.gemm_raw(R, ci, 6 * co, D1, ops.rowmap(42 * W, 43 * Wp * co, co, W, Wp * co), wd, 6 * co, gx, ops.rowmap(R, 0, ci), ops.OCC_F32, ops.OCC_F32,
                 a_seg=(2, 3 * co, Wp * co))
    torch.testing.assert_close(gx.cpu().view(B, 42, W, ci).permute(0, 3, 1, 2), x.grad, rtol=1e-3, atol=1e-3 * float(x.grad.abs().max()))


@pytest.mark.parametrize("C,act", [(64, "selu"), (1, "selu"), (128, "none"), (32, "relu")])
def test_batchnorm_train_fwd_bwd_and_running_stats(C, act):
    from occm_amd import backend_ops as K
    from occm_amd._lib import ACT_NONE, ACT_RELU, ACT_SELU
    code = {"selu": ACT_SELU, "none": ACT_NONE, "relu": ACT_RELU}[act]
    fn = {"selu": F.selu, "none": lambda t: t, "relu": F.relu}[act]
    rows = 1777
    x = (_r(rows, C, seed=7) * 2 + 0.5).requires_grad_(True)
    g, b = (1 + 0.1 * _r(C, seed=8)).requires_grad_(True), (0.1 * _r(C, seed=9)).requires_grad_(True)
    rm, rv = 0.1 * _r(C, seed=10), 0.5 + _r(C, seed=11).abs()
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = fn(F.batch_norm(x, rm_ref, rv_ref, g, b, training=True, momentum=0.1, eps=1e-5))
    gy = _r(rows, C, seed=12)
    y.backward(gy)
    dev = lambda t: t.detach().clone().cuda()
    ws, sums = torch.empty(512 * 256 * 2, dtype=torch.float64, device="cuda"), torch.empty(512, device="cuda")
    mean, rstd = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    rmd, rvd, nbt = dev(rm), dev(rv), torch.zeros(1, dtype=torch.int64, device="cuda")
    xd = dev(x)
    K.bn_stats(xd, K.full(rows, C), rows, C, ws, mean, rstd, rmd, rvd, nbt, True)
    out = torch.empty(rows, C, device="cuda")
    K.bn_act_fwd(xd, K.full(rows, C), mean, rstd, dev(g), dev(b), code, out, K.full(rows, C), rows, C)
    torch.testing.assert_close(out.cpu(), y.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(rmd.cpu(), rm_ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rvd.cpu(), rv_ref, rtol=1e-5, atol=1e-6)
    assert int(nbt) == 1
    dx, dg, db = torch.empty(rows, C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    K.bn_act_bwd(dev(gy), K.full(rows, C), xd, K.full(rows, C), mean, rstd, dev(g), dev(b), code, dx, K.full(rows, C), dg, db, ws, sums, rows, C)
    torch.testing.assert_close(dx.cpu(), x.grad, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(dg.cpu(), g.grad, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(db.cpu(), b.grad, rtol=1e-3, atol=1e-3)


def test_softmax_weighted_sums_fwd_bwd():
    from occm_amd import backend_ops as K
    B, H, W, C = 2, 42, 66, 64
    x = _r(B, C, H, W, seed=13).requires_grad_(True)
    w = _r(B, C, H, W, seed=14).requires_grad_(True)
    pos = _r(1, H, C, seed=15)
    eS = (x * torch.softmax(w, dim=-1)).sum(-1).transpose(1, 2) + pos
    eT = (x * torch.softmax(w, dim=-2)).sum(-2).transpose(1, 2)
    gS, gT = _r(B, H, C, seed=16), _r(B, W, C, seed=17)
    ((eS * gS).sum() + (eT * gT).sum()).backward()
    cl = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().cuda()
    xc, wc = cl(x), cl(w)
    oS, oT = torch.empty(B, H, C, device="cuda"), torch.empty(B, W, C, device="cuda")
    K.softmax_wsum_fwd(xc, wc, B * H, 1, W * C, 0, W, C, C, pos.cuda(), H, oS)
    K.softmax_wsum_fwd(xc, wc, B * W, W, H * W * C, C, H, W * C, C, None, 1, oT)
    torch.testing.assert_close(oS.cpu(), eS.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(oT.cpu(), eT.detach(), rtol=1e-4, atol=1e-5)
    dx, dw = torch.empty_like(xc), torch.empty_like(wc)
    K.softmax_wsum_bwd(xc, wc, B * H, 1, W * C, 0, W, C, C, gS.cuda(), dx, dw, 0)
    K.softmax_wsum_bwd(xc, wc, B * W, W, H * W * C, C, H, W * C, C, gT.cuda(), dx, dw, 1)
    torch.testing.assert_close(dx.cpu().permute(0, 3, 1, 2), x.grad, rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(dw.cpu().permute(0, 3, 1, 2), w.grad, rtol=1e-3, atol=1e-5)


def test_dropout_mask_statistics_and_reuse():
    from occm_amd import backend_ops as K
    x = torch.ones(1 << 18, device="cuda")
    y, m = torch.empty_like(x), torch.empty(1 << 18, dtype=torch.uint8, device="cuda")
    K.dropout(x, y, m, 0.3, 11, 5, True)
    keep = float(m.float().mean())
    assert abs(keep - 0.7) < 5e-3 and torch.allclose(y, m.float() / 0.7)
    y2 = torch.empty_like(x)
    K.dropout(2 * x, y2, m, 0.3, 0, 0, False)
    assert torch.allclose(y2, 2 * y)


# ------------------------------------------------------------------------------- model level ---
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_backend_eval_matches_reference(tag):
    from occm_amd.models.sslassist import AasistBackend
    be = AasistBackend(_params())
    emb, out = be.forward(_feats(tag).cuda(), train=False)
    np.testing.assert_allclose(emb.cpu().numpy(), GA["eval_emb_" + tag], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(out.cpu().numpy(), GA["eval_out_" + tag], rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_backend_train_no_dropout_matches_reference_incl_grads(tag):
    from occm_amd import ops
    from occm_amd.models.sslassist import AasistBackend
    B = CASES[tag][0]
    be = AasistBackend(_params())
    be.zero_grad()
    emb, out = be.forward(_feats(tag).cuda(), train=True, masks={})
    np.testing.assert_allclose(emb.cpu().numpy(), GA["train_emb_" + tag], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(out.cpu().numpy(), GA["train_out_" + tag], rtol=1e-3, atol=1e-3)
    labels = (torch.arange(B) % 12 >= 6).long().cuda()
    lc, demb = ops.compactness_loss(emb, scale=0.1, want_grad=True)
    ld, dlog = ops.ce_loss(out, labels, scale=0.9, want_grad=True)
    np.testing.assert_allclose(0.1 * lc.item() + 0.9 * ld.item(), GA["train_loss_" + tag], rtol=1e-4)
    be.backward(demb, dlog)
    sd = be.state_dict()
    for key in GA.files:
        if key.startswith("rs_%s_" % tag):
            np.testing.assert_allclose(sd[key[len("rs_%s_" % tag):]].cpu().numpy(), GA[key], rtol=1e-3, atol=1e-5)
    grads = be.grad_dict()
    names, norms = list(GA["gradnames_" + tag]), GA["gradnorms_" + tag]
    bad = []
    for n, ref in zip(names, norms):
        got = float(grads[str(n)].norm())
        if ref < 0:
            if got != 0.0:
                bad.append((n, got, ref))
        elif abs(got - ref) > 3e-3 * ref + 5e-4:       # 5e-4 floor: see GRAD_FLOOR note below
            bad.append((n, got, ref))
    assert not bad, bad[:12]
    for key in GA.files:
        if key.startswith("grad_%s_" % tag) and "rows0_4" not in key:
            ref = GA[key]
            np.testing.assert_allclose(grads[key[len("grad_%s_" % tag):]].cpu().numpy(), ref, rtol=3e-3, atol=max(3e-3 * np.abs(ref).max(), GRAD_FLOOR))
    ref = GA["grad_%s_LL.weight_rows0_4" % tag]
    np.testing.assert_allclose(grads["LL.weight"][:4].cpu().numpy(), ref, rtol=3e-3, atol=3e-3 * np.abs(ref).max())


def test_backend_train_with_injected_dropout_masks_matches_oracle():
    """Train mode WITH dropout: the same keep-masks drive the oracle and the HIP path."""
    from oracle import aasist_ref, losses_ref
    from occm_amd import ops
    from occm_amd.models.sslassist import AasistBackend
    B, T = 6, 199
    W = T // 3
    g = torch.Generator().manual_seed(77)
    shapes = {"GAT_layer_S": (B, 42, 64), "GAT_layer_T": (B, W, 64), "pool_S": (B, 42, 64), "pool_T": (B, W, 64),
              "HtrgGAT_layer_ST11": (B, 54, 64), "HtrgGAT_layer_ST21": (B, 54, 64), "HtrgGAT_layer_ST12": (B, 26, 32), "HtrgGAT_layer_ST22": (B, 26, 32),
              "pool_hS1": (B, 21, 32), "pool_hT1": (B, 33, 32), "pool_hS2": (B, 21, 32), "pool_hT2": (B, 33, 32),
              "way_T1": (B, 16, 32), "way_T2": (B, 16, 32), "way_S1": (B, 10, 32), "way_S2": (B, 10, 32), "way_M1": (B, 32), "way_M2": (B, 32),
              "last": (B, 160)}
    ps = {"GAT": 0.2, "Htr": 0.2, "poo": 0.3, "way": 0.2, "las": 0.5}
    masks = {k: (torch.rand(s, generator=g) >= ps[k[:3]]).to(torch.uint8) for k, s in shapes.items()}
    feats = torch.randn(B, T, 1024, generator=g)
    p = _params()
    for k, v in p.items():
        if v.dtype.is_floating_point and not k.split(".")[-1].startswith("running"):
            v.requires_grad_(True)
    m_or = {k: (v.unsqueeze(1) if k.startswith("way_M") else v) for k, v in masks.items()}
    emb_r, out_r = aasist_ref.backend_forward(feats, p, train=True, masks=m_or)
    labels = (torch.arange(B) % 12 >= 3).long()
    (0.1 * losses_ref.compactness_loss(emb_r) + 0.9 * losses_ref.descriptiveness_loss(out_r, labels)).backward()
    be = AasistBackend(_params())
    be.zero_grad()
    emb, out = be.forward(feats.cuda(), train=True, masks=masks)
    torch.testing.assert_close(emb.cpu(), emb_r.detach(), rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(out.cpu(), out_r.detach(), rtol=1e-3, atol=1e-3)
    assert float((emb == 0).float().mean()) > 0.3            # quirk: the returned emb is the dropped one
    _, demb = ops.compactness_loss(emb, scale=0.1, want_grad=True)
    _, dlog = ops.ce_loss(out, labels.cuda(), scale=0.9, want_grad=True)
    be.backward(demb, dlog)
    grads = be.grad_dict()
    bad = []
    for k, v in p.items():
        if v.grad is None or k not in grads:
            continue
        ref = v.grad
        err = float((grads[k].cpu() - ref).abs().max())
        if err > 3e-3 * float(ref.abs().max()) + GRAD_FLOOR:
            bad.append((k, err, float(ref.abs().max())))
    assert not bad, bad[:12]


def test_state_dict_round_trip_uses_reference_keys_and_layouts():
    from oracle import aasist_ref
    from occm_amd.models.sslassist import AasistBackend
    p = _params()
    be = AasistBackend(p)
    sd = be.state_dict()
    assert set(sd.keys()) == set(aasist_ref.param_shapes().keys())
    for k, v in p.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
        torch.testing.assert_close(sd[k].cpu().to(v.dtype), v)


def test_backend_bf16_compute_mode_tracks_the_f32_mode():
    """bench dtype: Linear / Conv2d fwd and dgrad GEMMs round operands to bf16 (f32 accumulate).  The graph pools make
    discrete top-k choices, so element-wise agreement is only meaningful BEFORE the first pool; after it the check is on the
    loss and on gradient directions.  Stated bounds vs the exact-f32 mode on the same inputs:
      spectral/temporal node features e_S, e_T: max |d| < 0.06 (values O(1));  |d loss| < 0.05;
      gradient cosine >= 0.9 for every tensor with a non-negligible gradient, >= 0.95 for the flat gradient as a whole
    (measured on MI355X: 0.912 worst tensor, 0.970 flat -- top-k flips in the pools move whole nodes)."""
    from occm_amd import ops
    from occm_amd.models.sslassist import AasistBackend
    feats = _feats("a").cuda()
    labels = (torch.arange(12) % 12 >= 6).long().cuda()
    res = {}
    for mode in ("f32", "bf16"):
        be = AasistBackend(_params(), compute=mode)
        be.zero_grad()
        emb, out = be.forward(feats, train=True, masks={})
        taps = (be.ctx["eS"].cpu(), be.ctx["eT"].cpu())
        lc, demb = ops.compactness_loss(emb, scale=0.1, want_grad=True)
        ld, dlog = ops.ce_loss(out, labels, scale=0.9, want_grad=True)
        be.backward(demb, dlog)
        res[mode] = (taps, 0.1 * lc.item() + 0.9 * ld.item(), {k: v.cpu() for k, v in be.grad_dict().items()}, be.G.cpu().clone())
    dS = float((res["bf16"][0][0] - res["f32"][0][0]).abs().max())
    dT = float((res["bf16"][0][1] - res["f32"][0][1]).abs().max())
    dloss = abs(res["bf16"][1] - res["f32"][1])
    worst, wname = 1.0, ""
    for k, g32 in res["f32"][2].items():
        if float(g32.norm()) < 1e-2:
            continue
        cos = float((g32 * res["bf16"][2][k]).sum() / (g32.norm() * res["bf16"][2][k].norm() + 1e-30))
        if cos < worst:
            worst, wname = cos, k
    flat = float((res["f32"][3] * res["bf16"][3]).sum() / (res["f32"][3].norm() * res["bf16"][3].norm()))
    print("bf16-vs-f32 back-end: max|d e_S| %.4f  max|d e_T| %.4f  |d loss| %.4f  worst grad cosine %.4f (%s)  flat cosine %.5f"
          % (dS, dT, dloss, worst, wname, flat))
    assert dS < 0.06 and dT < 0.06 and dloss < 0.05
    assert worst > 0.9 and flat > 0.95


@pytest.mark.parametrize("M,N1,N2,seg", [(1000, 64, 64, None), (88704, 64, 384, "conv"), (517, 128, 1024, None), (33, 4, 24, None), (4099, 32, 192, None)])
def test_gemm_tn_bf16_mfma(M, N1, N2, seg):
    """compute = OCC_BF16 form of the weight-gradient GEMM (transposing LDS reads): equals the f64 product of the bf16-rounded
    operands to f32 summation accuracy; the fused bias gradient stays a sum of the un-rounded values."""
    from occm_amd import backend_ops as K
    a = _r(M, N1, seed=1)
    if seg == "conv":            # B rows are 2x3 windows over a [*, 68, 64] channels-last image: 2 K-segments of 192
        Wp, ci = 68, 64
        img = _r(M // 66 + 3, Wp, ci, seed=2)
        imgd = img.cuda()
        rows = torch.arange(M)
        h, w = rows // 66, rows % 66
        win = torch.stack([img[h + dh, w + dw] for dh in range(2) for dw in range(3)], 1).reshape(M, 6 * ci)
        b_ref, Bm, b_seg = win, imgd, (2, 3 * ci, Wp * ci)
        b_map = K.rowmap(66 * (M // 66 + 1), 0, ci, 66, Wp * ci)     # one "batch" of lines: row r -> line r // 66, column r % 66
    else:
        b = _r(M, N2, seed=2)
        b_ref, Bm, b_map, b_seg = b, b.cuda(), K.full(M, N2), None
    C, s = torch.zeros(N1, N2).cuda(), torch.zeros(N1).cuda()
    K.gemm_tn(M, N1, N2, a.cuda(), K.full(M, N1), Bm, b_map, C, N2, b_seg=b_seg, colsum_out=s, bf16_mfma=True)
    ref = a.bfloat16().double().T @ b_ref.bfloat16().double()
    torch.testing.assert_close(C.cpu().double(), ref, rtol=1e-4, atol=2e-6 * float(ref.abs().max()) + 1e-4)
    torch.testing.assert_close(s.cpu().double(), a.double().sum(0), rtol=1e-4, atol=1e-3)
    exact = a.double().T @ b_ref.double()                      # and it is close to the un-rounded product (bf16 products, f32 sums)
    assert float((C.cpu().double() - exact).abs().max()) <= 2e-2 * float(exact.abs().max()) + 1e-3


def test_trainer_prefetch_pipeline_equals_sequential_steps():
    """OcTrainer.step(..., next_wav=...) computes the next batch's frozen-front-end features on a side stream during the current
    back-end update.  The features every step trains on must be bit-identical to the sequential loop's (they do not depend on the
    optimizer update), the first step's losses identical, later ones equal up to the run-to-run noise of the f32 atomic adds in
    the weight-gradient kernel (two sequential runs differ by ~1e-3 relative after one update at this learning rate)."""
    from occm_amd.models import xlsr
    from occm_amd.models.sslassist import AModel
    from occm_amd.trainer import OcTrainer
    cfg = xlsr.XlsrConfig(dim=1024, ffn=512, heads=16, layers=2)
    wavs = [(0.1 * _r(12, 16000, seed=40 + i)).cuda() for i in range(5)]
    labels = (torch.arange(12) >= 6).long().cuda()

    def run(pipelined):
        model = AModel(None, "cuda", ssl_cfg=cfg, seed=0, synthetic_ssl=True)
        model.train()
        tr = OcTrainer(model, lr=1e-4, w_compact=0.1, w_descr=0.9, dropout_masks={})
        seen, fwd = [], tr.be.forward

        def spy(feats, **kw):
            seen.append(feats.clone())
            return fwd(feats, **kw)
        tr.be.forward = spy
        out, p1 = [], None
        for i, w in enumerate(wavs):
            nxt = wavs[i + 1] if pipelined and i + 1 < len(wavs) else None
            lc, ld = tr.step(w, labels, next_wav=nxt)
            out.append((float(lc), float(ld)))
            if i == 0:
                p1 = tr.be.P.clone()
        return out, seen, p1

    seq, f_seq, p_seq = run(False)
    pip, f_pip, p_pip = run(True)
    assert len(f_seq) == len(f_pip) == len(wavs)
    for a, b in zip(f_seq, f_pip):
        assert torch.equal(a, b)
    assert seq[0] == pip[0]
    # the sharp statement after one update: both orders applied Adam to gradients that differ only by float-atomic noise, so the parameters
    # agree except where a noise-level gradient flips the sign of Adam's first (sign-like, +-lr) update -- a pipeline that trained on a
    # wrong or stale tensor would move most elements by up to 2 lr
    dp = (p_seq - p_pip).abs()
    frac = float((dp > 0.1 * 1e-4).float().mean())
    assert float(dp.max()) <= 2.1e-4 and frac < 0.02, (float(dp.max()), frac)
    (a, b), (c, d) = seq[1], pip[1]             # after one update; further steps of this tiny model amplify the atomics' noise
    # (run-to-run spread of the second step's losses, measured over this round's runs: 0.2-2.1 % -- a one-ulp difference flips a top-k
    # graph-pooling choice in the random-initialised back-end; a wrong feature tensor would already have failed the bit-equality above)
    assert abs(a - c) <= 0.05 * abs(a) and abs(b - d) <= 0.05 * abs(b), (seq, pip)


def test_backend_section_replayed_from_hip_graph_equals_eager():
    """OcTrainer replays zero_grad -> AASIST forward -> losses -> backward from a HIP graph from the third time a batch shape is seen.
    With fixed parameters the replay must reproduce the eager section for new inputs: losses, every parameter gradient and the feature
    gradient (to the run-to-run noise two eager runs show as well), with dropout masks keyed by the device-side step counter the graph itself advances -- the same
    counter value gives the same masks, the next value different ones."""
    from occm_amd.models import xlsr
    from occm_amd.models.sslassist import AModel
    from occm_amd.trainer import OcTrainer
    cfg = xlsr.XlsrConfig(dim=1024, ffn=512, heads=16, layers=1)
    model = AModel(None, "cuda", ssl_cfg=cfg, seed=0, synthetic_ssl=True)
    model.train()
    tr = OcTrainer(model, lr=1e-4, w_compact=0.1, w_descr=0.9, group_size=12)
    be = tr.be
    labels = (torch.arange(12) >= 6).long().cuda()

    def run(feats, graph, step):
        be.rng_step_dev.fill_(step)
        tr.graph_backend = graph
        lc, ld, df = tr._backend_section(feats, labels, True)
        assert int(be.rng_step_dev) == step + 1
        return float(lc), float(ld), df.clone(), be.G.clone()

    key = None
    for i in range(4):
        feats = _r(12, 199, 1024, seed=70 + i).cuda()
        e = run(feats, False, 100 + i)
        g = run(feats, True, 100 + i)
        key = key or next(iter(tr._graphs))
        assert (tr._graphs[key]["graph"] is not None) == (i >= 1)            # captured on the second sighting, replayed from the third
        assert abs(e[0] - g[0]) <= 1e-5 * abs(e[0]) + 1e-7 and abs(e[1] - g[1]) <= 1e-5 * abs(e[1]) + 1e-7, (i, e[:2], g[:2])
        # Bounds: two EAGER runs of one input are not bit-equal either.  The master-node / pooling kernels add with float atomics
        # (~1e-8 relative noise on a few gradients); in the bf16 compute mode such a difference now and then flips the bf16 rounding
        # of one GEMM operand element, and the BatchNorm / SELU backward chain (differences of large terms) amplifies that single
        # ulp to ~5e-3 of the largest feature gradient and ~1e-3 of the largest parameter gradient (measured over 40 runs of one
        # input: 16 differed, all by exactly this pattern, eagerly and replayed alike).  A replay that used a wrong buffer or a
        # stale mask would be off by O(1).
        assert float((g[2] - e[2]).abs().max()) <= 2e-2 * float(e[2].abs().max())
        assert float((g[3] - e[3]).abs().max()) <= 1e-2 * float(e[3].abs().max())
    a = run(feats, True, 500)
    b = run(feats, True, 500)
    c = run(feats, True, 501)
    assert abs(a[1] - b[1]) <= 1e-5 * abs(a[1]) and abs(a[1] - c[1]) > 1e-6 * abs(a[1]), (a[:2], b[:2], c[:2])     # masks: same step same draw, next step another
    # a different batch shape falls back to eager and gets its own record
    run(_r(12, 150, 1024, seed=9).cuda(), True, 7)
    assert len(tr._graphs) == 2


def test_training_steps_match_reference_loop_golden():
    """tests/golden/train_steps.npz (three steps of the reference's own loop body on fixed features, dropout off) retraced through
    the C ABI: back-end forward/backward (exact-f32 MFMA), loss kernels, occ_adam_multi.  Tolerances as in the oracle's test."""
    from oracle import aasist_ref
    from oracle.fill import fill_like
    from occm_amd import ops
    from occm_amd.models.sslassist import AasistBackend
    GT = golden("train_steps.npz")
    be = AasistBackend(fill_like(aasist_ref.param_shapes(), seed=0), device="cuda", compute="f32")
    opt = ops.AdamMulti([be.P], lr=1e-4)
    labels = (torch.arange(12) >= 6).long().cuda()
    for step in range(3):
        feats = torch.randn(12, 199, 1024, generator=torch.Generator().manual_seed(200 + step)).cuda()
        be.zero_grad()
        emb, logits = be.forward(feats, train=True, masks={})
        lc, demb = ops.compactness_loss(emb, n_groups=1, group=12, scale=0.1, want_grad=True)
        ld, dlog = ops.ce_loss(logits, labels, scale=0.9, want_grad=True)
        be.backward(demb, dlog)
        opt.step([be.G])
        rt = 2e-4 if step == 0 else 2e-2
        np.testing.assert_allclose(float(lc), GT["loss_c"][step], rtol=rt)
        np.testing.assert_allclose(float(ld), GT["loss_d"][step], rtol=rt)
    sd = be.state_dict()
    for k in GT.files:
        if k.startswith("p_") and GT[k].dtype.kind == "f":
            np.testing.assert_allclose(sd[k[2:]].cpu().numpy().reshape(GT[k].shape), GT[k], rtol=0, atol=6.1e-4)
    assert int(sd["first_bn.num_batches_tracked"]) == 3


@pytest.mark.parametrize("M,N1,N2", [(12736, 1024, 1024), (4099, 256, 384), (300, 128, 136), (25536, 512, 1024)])
def test_gemm_tn_dma_large_weight_gradients(M, N1, N2):
    """bf16 operands, bf16 MFMA, big outputs: the LDS-DMA / transposing-read kernel (no transposed copies), incl. a ragged last slab,
    ragged column tiles and the fused bias gradient; accumulates onto what C holds."""
    from occm_amd import backend_ops as K
    a = _r(M, N1, seed=1).bfloat16(); b = _r(M, N2, seed=2).bfloat16()
    c0 = _r(N1, N2, seed=3)
    C, s = c0.clone().cuda(), torch.zeros(N1).cuda()
    K.gemm_tn(M, N1, N2, a.cuda(), K.full(M, N1), b.cuda(), K.full(M, N2), C, N2, colsum_out=s, a_bf16=True, b_bf16=True, bf16_mfma=True)
    ref = c0.double() + a.double().T @ b.double()
    torch.testing.assert_close(C.cpu().double(), ref, rtol=1e-4, atol=2e-6 * float(ref.abs().max()) + 2e-4)
    torch.testing.assert_close(s.cpu().double(), a.double().sum(0), rtol=1e-4, atol=2e-3)


def test_gemm_tn_dma_conv_windows():
    """B rows are k=3, stride-2 windows of a channels-last [B, Tin, 512] signal (three K-segments of 512), A rows the conv output
    gradient: the conv-stack weight gradient of the fine-tuning path without materialising the windows."""
    from occm_amd import backend_ops as K
    Bn, Tin, Cc, k, st = 3, 401, 512, 3, 2
    Tout = (Tin - k) // st + 1
    M = Bn * Tout
    x = _r(Bn, Tin, Cc, seed=5).bfloat16()
    dy = _r(M, Cc, seed=6).bfloat16()
    win = torch.stack([x[:, t * st: t * st + k].reshape(Bn, k * Cc) for t in range(Tout)], 1).reshape(M, k * Cc)
    C = torch.zeros(Cc, k * Cc).cuda()
    K.gemm_tn(M, Cc, k * Cc, dy.cuda(), K.full(M, Cc), x.cuda(), K.rowmap(Tout, Tin * Cc, st * Cc), C, k * Cc, b_seg=(k, Cc, Cc), a_bf16=True, b_bf16=True,
              bf16_mfma=True)
    ref = dy.double().T @ win.double()
    torch.testing.assert_close(C.cpu().double(), ref, rtol=1e-4, atol=2e-6 * float(ref.abs().max()) + 2e-4)


@pytest.mark.parametrize("seed", range(8))
def test_gemm_tn_fuzz(seed):
    """Random weight-gradient problems over the three occ_gemm_tn kernels (exact f32, bf16 MFMA, LDS-DMA bf16): 3-level row maps on
    both operands, B K-segments, ragged M / N1 / N2, accumulation onto existing C, fused bias gradient."""
    from occm_amd import backend_ops as K
    rs = np.random.RandomState(500 + seed)
    for _ in range(4):
        kind = rs.choice(["f32", "bf16_mfma", "dma"])
        unit = 8 if kind == "dma" else 4
        nb, nl, rpl = int(rs.randint(1, 4)), int(rs.randint(1, 5)), int(rs.randint(8, 60))
        M = nb * nl * rpl
        if kind == "dma":
            M = max(M, 256); nb, nl, rpl = 1, 1, M
        N1 = int(rs.randint(16, 40)) * unit if kind == "dma" else int(rs.randint(1, 40)) * unit
        nseg = int(rs.choice([1, 1, 2, 3]))
        seg_len = (128 * int(rs.randint(1, 3))) if (kind == "dma" and nseg > 1) else int(rs.randint(2, 24)) * unit
        if kind == "dma" and nseg == 1:
            seg_len = max(seg_len, 128)
        N2 = nseg * seg_len
        a_rs = N1 + unit * int(rs.randint(0, 3))
        a_ls = rpl * a_rs + unit * int(rs.randint(0, 3))
        a_bs = nl * a_ls + unit * int(rs.randint(0, 3))
        b_rs = int(rs.choice([seg_len, unit, seg_len + unit])) if nseg == 1 else seg_len + unit * int(rs.randint(0, 2))
        seg_stride = rpl * b_rs + seg_len + unit * int(rs.randint(0, 3)) if nseg > 1 else 0
        b_ls = (rpl - 1) * b_rs + N2 + seg_stride * nseg + unit * int(rs.randint(0, 3))
        b_ls = (b_ls + unit - 1) // unit * unit
        b_bs = nl * b_ls + unit * int(rs.randint(0, 3))
        a_buf = torch.from_numpy(rs.randn(nb * a_bs + N1).astype("float32"))
        b_buf = torch.from_numpy(rs.randn(nb * b_bs + N2 + seg_stride * nseg).astype("float32"))
        bf = kind == "dma"
        a_dev, b_dev = (a_buf.bfloat16(), b_buf.bfloat16()) if bf else (a_buf, b_buf)
        a_ref, b_ref = (a_buf.bfloat16().float(), b_buf.bfloat16().float()) if kind != "f32" else (a_buf, b_buf)
        a_src = a_buf.bfloat16().float() if bf else a_buf
        ar, br, ar_src = [], [], []
        for m in range(M):
            b, rem = divmod(m, nl * rpl); l, r = divmod(rem, rpl)
            ao = b * a_bs + l * a_ls + r * a_rs
            bo = b * b_bs + l * b_ls + r * b_rs
            ar.append(a_ref[ao: ao + N1]); ar_src.append(a_src[ao: ao + N1])
            br.append(torch.cat([b_ref[bo + s * seg_stride: bo + s * seg_stride + seg_len] for s in range(nseg)]))
        A, Bm = torch.stack(ar).double(), torch.stack(br).double()
        ldc = N2 + 4 * int(rs.randint(0, 3))
        c0 = torch.from_numpy(rs.randn(N1, ldc).astype("float32"))
        C, cs = c0.clone().cuda(), torch.zeros(N1).cuda()
        alpha = float(rs.choice([1.0, 0.25]))
        K.gemm_tn(M, N1, N2, a_dev.cuda(), K.rowmap(nl * rpl, a_bs, a_rs, rpl, a_ls), b_dev.cuda(), K.rowmap(nl * rpl, b_bs, b_rs, rpl, b_ls), C, ldc,
                  b_seg=(nseg, seg_len, seg_stride) if nseg > 1 else None, alpha=alpha, colsum_out=cs, a_bf16=bf, b_bf16=bf, bf16_mfma=kind != "f32")
        ref = c0.double()
        ref[:, :N2] += alpha * (A.T @ Bm)
        tol = 2e-5 * float(ref.abs().max()) + 3e-4
        assert float((C.cpu().double() - ref).abs().max()) <= tol, (kind, M, N1, N2, nseg, float((C.cpu().double() - ref).abs().max()), tol)
        sref = alpha * torch.stack(ar_src).double().sum(0)          # the bias gradient sums the operand as stored (un-rounded f32, or the bf16 values)
        torch.testing.assert_close(cs.cpu().double(), sref, rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("pre,N,D,Do,n1", [("GAT_layer_T", 66, 64, 64, 66), ("GAT_layer_S", 42, 64, 64, 42), ("HtrgGAT_layer_ST11", 54, 64, 32, 33),
                                           ("HtrgGAT_layer_ST12", 26, 32, 32, 16), ("GAT_layer_T", 93, 64, 64, 93), ("HtrgGAT_layer_ST21", 7, 64, 32, 4)])
def test_fused_graph_attention_core_matches_the_unfused_kernels_and_f64(pre, N, D, Do, n1):
    """csrc/gat_fused.hip (one kernel forward, one backward, no [B,N,N,D] tensors) against (a) the unfused kernel chain it replaces in
    the bf16-compute mode and (b) an f64 torch evaluation of sslassist.py:102-130 / 271-300 with autograd: attention map, aggregated
    features, and the gradients wrt the node features, att_proj and the (typed) attention weights.  Both device paths round the same
    operands to bf16, so they must sit equally close to the f64 values."""
    from occm_amd.models.sslassist import AasistBackend, TEMPS
    B = 5
    g = torch.Generator().manual_seed(N + D)
    xd = torch.randn(B, N, D, generator=g)
    dh = torch.randn(B, N, D, generator=g)
    res = {}
    for fused in (False, True):
        be = AasistBackend(device="cuda", seed=0, compute="bf16")
        be.fuse_gat = fused
        be.zero_grad()
        c = {pre + ".xd": xd.cuda()}
        h = be._att_core_fwd(pre, c[pre + ".xd"], B, N, D, Do, n1, c)
        assert (c[pre + ".P"] is None) == fused
        dxd = torch.full((B, N, D), 0.5, device="cuda")
        be._att_core_bwd(pre, dh.cuda(), dxd, B, N, D, Do, n1, c)
        res[fused] = dict(alpha=c[pre + ".alpha"].cpu().double(), h=h.cpu().double(), dxd=dxd.cpu().double() - 0.5, gW=be.g[pre + ".att_proj.weight"].cpu().double(),
                          gb=be.g[pre + ".att_proj.bias"].cpu().double(), gaw=be.g[pre + ".aw3"].cpu().double())
        W, bb, aw3 = be.p[pre + ".att_proj.weight"].cpu().double(), be.p[pre + ".att_proj.bias"].cpu().double(), be.p[pre + ".aw3"].cpu().double()
    # f64 reference
    x = xd.double().requires_grad_(True)
    W_, b_, aw_ = W.clone().requires_grad_(True), bb.clone().requires_grad_(True), aw3.clone().requires_grad_(True)
    z = torch.tanh(torch.einsum("bijd,od->bijo", x.unsqueeze(2) * x.unsqueeze(1), W_) + b_)
    idx = torch.arange(N)
    ty = torch.where(idx[:, None] < n1, torch.where(idx[None, :] < n1, 0, 2), torch.where(idx[None, :] < n1, 2, 1))
    score = (z * aw_[ty]).sum(-1) / TEMPS[pre]
    alpha = torch.softmax(score, dim=-1)
    h = alpha @ x
    (h * dh.double()).sum().backward()
    ref = dict(alpha=alpha.detach(), h=h.detach(), dxd=x.grad, gW=W_.grad, gb=b_.grad, gaw=aw_.grad)
    for k in ref:
        scale = float(ref[k].abs().max()) + 1e-12
        e_un, e_fu = float((res[False][k] - ref[k]).abs().max()) / scale, float((res[True][k] - ref[k]).abs().max()) / scale
        assert e_fu < max(2.5e-2, 2.0 * e_un), (k, e_fu, e_un)
        assert float((res[True][k] - res[False][k]).abs().max()) / scale < 4e-2, k
    if n1 == N:
        assert float(res[True]["gaw"][1:].abs().max()) == 0.0            # homogeneous layers touch w11 only
