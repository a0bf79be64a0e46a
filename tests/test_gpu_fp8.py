"""OCP fp8 path: occ_fp8_quantize (e4m3fn / e5m2, saturating, per-tensor scale, |max| tracking) against torch's
float8 casts, and occ_gemm with fp8 operands (v_mfma_scale_f32_16x16x128_f8f6f4 in the eight-phase kernel) -- exact on integer-valued
operands, to f32 round-off on random fp8 bit patterns, per-tensor dequantisation scales and epilogues included."""
import pytest
import torch

pytestmark = pytest.mark.gpu
F8 = {5: torch.float8_e4m3fn, 6: torch.float8_e5m2}


@pytest.mark.parametrize("fmt", [5, 6])
@pytest.mark.parametrize("src_dtype", [torch.float32, torch.bfloat16])
def test_fp8_quantize_matches_torch_cast_and_tracks_amax(fmt, src_dtype):
    from occm_amd import ops
    g = torch.Generator().manual_seed(fmt)
    n = 100003                                              # odd tail
    x = (torch.randn(n, generator=g) * torch.logspace(-3, 2.5, n)[torch.randperm(n, generator=g)]).to(src_dtype)
    x[:4] = torch.tensor([0.0, -0.0, 1e6, -1e6]).to(src_dtype)        # saturation
    scale = torch.tensor([0.37], device="cuda")
    amax = torch.zeros(1, device="cuda")
    q = torch.empty(n, device="cuda", dtype=torch.uint8)
    ops.fp8_quantize(x.cuda(), q, fmt, scale=scale, amax=amax)
    lim = ops.FP8_MAX[fmt]
    ref = (x.float() * 0.37).clamp(-lim, lim).to(F8[fmt])
    got = q.cpu().view(F8[fmt])
    assert torch.equal(got.float(), ref.float())
    assert float(amax) == float(x.float().abs().max())
    amax2 = torch.zeros(1, device="cuda")
    ops.fp8_amax(x.cuda(), amax2)
    assert float(amax2) == float(amax)
    sc, inv = torch.empty(1, device="cuda"), torch.empty(1, device="cuda")
    ops.fp8_update_scales(amax, sc, inv, fmt, margin=1.0)
    assert abs(float(sc) * float(x.float().abs().max()) - lim) < 1e-3 * lim and abs(float(sc) * float(inv) - 1) < 1e-6 and float(amax) == 0.0


def _int_fp8(rows, cols, seed, fmt, lo=-4, hi=4):
    g = torch.Generator().manual_seed(seed)
    v = torch.randint(lo, hi + 1, (rows, cols), generator=g).float()
    return v, v.to(F8[fmt]).view(torch.uint8)


@pytest.mark.parametrize("a_fmt", [5, 6])
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 256, 384), (1000, 512, 1024), (12736, 1024, 1024), (2049, 260, 256)])
def test_fp8_gemm_exact_integer_products(M, N, K, a_fmt):
    from occm_amd import ops
    xv, xq = _int_fp8(M, K, 1, a_fmt)
    wv, wq = _int_fp8(N, K, 2, 5)
    bias = torch.randint(-3, 4, (N,)).float()
    out = torch.full((M, N), 7777.0, device="cuda")
    ops.gemm_raw(M, N, K, xq.cuda(), ops.rowmap(M, 0, K), wq.cuda(), K, out, ops.rowmap(M, 0, N), ops.OCC_F32, a_fmt, bias=bias.cuda())
    ref = xv.double() @ wv.double().T + bias.double()
    assert torch.equal(out.cpu().double(), ref), float((out.cpu().double() - ref).abs().max())
    from occm_amd._lib import lib
    assert lib().occ_gemm_last_kernel() == 9               # OCC_GEMM_KERNEL_P8_FP8


@pytest.mark.parametrize("a_fmt", [5, 6])
def test_fp8_gemm_random_bit_patterns_scales_and_epilogue(a_fmt):
    """Every finite fp8 value on both operands, per-tensor dequantisation scalars, bias + GELU + bf16 output."""
    from occm_amd import ops
    M, N, K = 1531, 1024, 512
    g = torch.Generator().manual_seed(3)
    xb = torch.randint(0, 256, (M, K), generator=g, dtype=torch.uint8)
    wb = torch.randint(0, 256, (N, K), generator=g, dtype=torch.uint8)
    xf, wf = xb.view(F8[a_fmt]).float(), wb.view(F8[5]).float()
    xb[~torch.isfinite(xf)] = 0; wb[~torch.isfinite(wf)] = 0            # NaN / inf encodings -> +0
    xf, wf = xb.view(F8[a_fmt]).float(), wb.view(F8[5]).float()
    xf = xf.clamp(-64, 64); xb = xf.to(F8[a_fmt]).view(torch.uint8); xf = xb.view(F8[a_fmt]).float()      # keep f32 sums far from overflow (e5m2 reaches 57344)
    dqa, dqw = torch.tensor([1.0 / 37.0], device="cuda"), torch.tensor([1.0 / 5.0], device="cuda")
    bias = torch.randn(N, generator=g)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm_raw(M, N, K, xb.cuda(), ops.rowmap(M, 0, K), wb.cuda(), K, out, ops.rowmap(M, 0, N), ops.OCC_BF16, a_fmt, bias=bias.cuda(), act=ops.ACT_GELU,
                 a_dequant=dqa, w_dequant=dqw)
    ref = torch.nn.functional.gelu((xf.double() @ wf.double().T).float() / (37.0 * 5.0) + bias)
    torch.testing.assert_close(out.cpu().float(), ref, rtol=1e-2, atol=1e-2 * float(ref.abs().max()) / 50 + 1e-2)
    out32 = torch.empty(M, N, device="cuda")
    ops.gemm_raw(M, N, K, xb.cuda(), ops.rowmap(M, 0, K), wb.cuda(), K, out32, ops.rowmap(M, 0, N), ops.OCC_F32, a_fmt, a_dequant=dqa, w_dequant=dqw)
    ref32 = (xf.double() @ wf.double().T) / (37.0 * 5.0)
    # the 128-deep block product aligns its terms to the largest one before adding (measured: differences up to 7e-5 of the largest
    # output when operands span the whole e4m3 / e5m2 range; exact on narrow-range operands, see the integer test)
    torch.testing.assert_close(out32.cpu().double(), ref32, rtol=1e-3, atol=3e-4 * float(ref32.abs().max()))


def test_finetuner_fp8_path_against_bf16_path_and_oracle():
    """XLS-R fine-tuner with the transformer's forward / input-gradient GEMMs in fp8 (e4m3 activations and weights, e5m2 gradients,
    delayed per-tensor scaling): output and every gradient against (a) the same fine-tuner in bf16 and (b) the CPU oracle's autograd.
    One e4m3 x e4m3 linear layer with per-tensor scales carries ~3.7 % mean relative error by itself (3 mantissa bits on both operands;
    the same experiment in torch float8 on the CPU gives 0.037), so the stated bounds are: features within 0.5 max / 8e-2 mean of the
    bf16 path (LayerNorm-ed, O(1) values; measured 0.24 / 4.0e-2 on this 2-layer model), gradient cosine >= 0.95 vs the oracle on
    every tensor that has a gradient above noise level; a second step uses the delayed scales of the first."""
    from oracle import xlsr_ref
    from oracle.fill import fill_like
    from occm_amd.models import xlsr
    kw = dict(dim=256, ffn=512, heads=4, layers=2)
    rcfg, cfg = xlsr_ref.XlsrConfig(**kw), xlsr.XlsrConfig(**kw)
    p = fill_like(xlsr_ref.param_shapes(rcfg), seed=3)
    for v in p.values():
        v.requires_grad_(True)
    g = torch.Generator().manual_seed(5)
    wav = 0.1 * torch.randn(4, 16000, generator=g)
    ref = xlsr_ref.extract_feat(wav, p, rcfg)
    dfe = torch.randn(ref.shape, generator=g)
    (ref * dfe).sum().backward()
    pd = {k: v.detach() for k, v in p.items()}
    ft16 = xlsr.XlsrFullFineTuner(pd, cfg)
    o16 = ft16.forward_train(wav.cuda()).clone()
    ft16.zero_grad(); ft16.backward(dfe.cuda())
    g16 = ft16.grad_dict()
    ft8 = xlsr.XlsrFullFineTuner(pd, cfg)
    ft8.enable_fp8()
    o8 = ft8.forward_train(wav.cuda()).clone()
    ft8.zero_grad(); ft8.backward(dfe.cuda())
    g8 = ft8.grad_dict()
    d = (o8 - o16).abs()
    assert float(d.max()) < 0.5 and float(d.mean()) < 8e-2, (float(d.max()), float(d.mean()))
    gmax = max(float(v.grad.abs().max()) for v in p.values())
    worst = 1.0
    for k, v in p.items():
        r = v.grad.reshape(-1)
        if float(r.abs().max()) < 1e-4 * gmax:
            continue
        a, b = g8[k].cpu().reshape(-1), g16[k].cpu().reshape(-1)
        c_or = float((a * r).sum() / (a.norm() * r.norm() + 1e-30))
        c_16 = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
        worst = min(worst, c_or)
        assert c_or > 0.95 and c_16 > 0.95, (k, c_or, c_16)
    print("fp8 fine-tuner: worst gradient cosine vs oracle %.4f; feature diff vs bf16 max %.3g mean %.3g" % (worst, float(d.max()), float(d.mean())))
    # delayed scaling: after a refresh (what the trainer does after the optimizer step) the scales are those measured in step 1
    s_before = ft8.f8["scale4"].clone()
    ft8.refresh_operands(cast=True)
    assert ft8.f8["warm"] is False and float(ft8.f8["amax4"].abs().max()) == 0.0
    assert bool((ft8.f8["scale4"] > 0).all()) and not torch.equal(ft8.f8["scale4"], torch.ones_like(s_before))
    # a second refresh with no step in between (load_params on resume, a layer skipped by layerdrop): nothing was measured, so the
    # activation / gradient sites keep their delayed scales instead of dropping to 1 (small e5m2 gradients would flush to zero)
    s_mid = ft8.f8["scale4"].clone(); s5_mid = ft8.f8["scale5"].clone()
    ft8.refresh_operands(cast=True)
    assert torch.equal(ft8.f8["scale4"], s_mid) and torch.equal(ft8.f8["scale5"], s5_mid) and not bool((ft8.f8["scale5"] == 1).all())
    o8b = ft8.forward_train(wav.cuda())
    d2 = (o8b - o8).abs()
    assert float(d2.max()) < 0.3, float(d2.max())                 # same input, same weights: only the (now delayed) scales may differ slightly
    assert float(ft8.f8["amax4"].max()) > 0                        # and this step's |max| is being recorded for the next one


@pytest.mark.parametrize("kind", ["gelu_aux_e4m3", "gelu_grad_e5m2", "plain_e4m3"])
def test_gemm_epilogue_writes_the_fp8_copy_of_its_bf16_result(kind):
    """occ_gemm c_f8: the 256-row kernel's row epilogue also writes fp8(bf16 result * scale) and raises |max| -- bit-identical to
    occ_fp8_quantize run over the bf16 result afterwards (same scale), for the fc1 form (bias + GELU + saved pre-activation, e4m3), the
    fc2-input-gradient form (GELU' from the saved pre-activation, e5m2) and a plain one; fp8 operands as on the configs[4] path."""
    from occm_amd import ops
    from occm_amd._lib import ACT_GELU, ACT_GELU_GRAD, ACT_NONE, OCC_BF16, OCC_FP8_E4M3, OCC_FP8_E5M2
    M, N, K = 1531, 1024, 512
    g = torch.Generator().manual_seed(9)
    a8 = (torch.randn(M, K, generator=g) * 0.5).to(torch.float8_e4m3fn).view(torch.uint8).cuda()
    w8 = (torch.randn(N, K, generator=g) * 0.3).to(torch.float8_e4m3fn).view(torch.uint8).cuda()
    dq = torch.tensor([0.5], device="cuda")
    bias = torch.randn(N, generator=g).cuda()
    fmt = OCC_FP8_E5M2 if kind.endswith("e5m2") else OCC_FP8_E4M3
    kw = {}
    if kind == "gelu_aux_e4m3":
        kw = dict(bias=bias, act=ACT_GELU, aux=torch.empty(M, N, device="cuda", dtype=torch.bfloat16))
    elif kind == "gelu_grad_e5m2":
        kw = dict(act=ACT_GELU_GRAD, aux=torch.randn(M, N, generator=g).bfloat16().cuda())
    else:
        kw = dict(bias=bias)
    C0, C1 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16), torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    run = lambda C, **extra: ops.gemm_raw(M, N, K, a8, ops.rowmap(M, 0, K), w8, K, C, ops.rowmap(M, 0, N), OCC_BF16, OCC_FP8_E4M3, a_dequant=dq, w_dequant=dq, **kw, **extra)
    run(C0)
    scale, amax0, amax1 = torch.tensor([23.0], device="cuda"), torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    q0, q1 = torch.empty(M * N, device="cuda", dtype=torch.uint8), torch.zeros(M * N, device="cuda", dtype=torch.uint8)
    ops.fp8_quantize(C0, q0, fmt, scale=scale, amax=amax0)
    run(C1, c_f8=(q1, scale, amax1, fmt))
    assert torch.equal(C1, C0) and torch.equal(q1, q0) and float(amax1) == float(amax0) > 0
    # a launch that cannot take the 256-row kernel refuses instead of silently skipping the copy
    from occm_amd._lib import OccError
    small = torch.empty(64, N, device="cuda", dtype=torch.bfloat16)
    x16 = torch.randn(64, K, generator=g).bfloat16().cuda(); w16 = torch.randn(N, K, generator=g).bfloat16().cuda()
    with pytest.raises(OccError):
        ops.gemm_raw(64, N, K, x16, ops.rowmap(64, 0, K), w16, K, small, ops.rowmap(64, 0, N), OCC_BF16, OCC_BF16, c_f8=(q1, scale, amax1, fmt))
