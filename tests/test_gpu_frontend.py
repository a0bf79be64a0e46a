"""HIP front-end kernels (C ABI) vs torch-CPU fp32 math and the XLS-R oracle.

Tolerances: f32 MFMA path <= 1e-3 absolute on O(1) activations (north-star bar); bf16 path is the
bench dtype and is checked against the same oracle with a looser, stated bound."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _r(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (200, 64, 40), (1, 4, 8), (6368, 1024, 512), (333, 132, 1536), (257, 3072, 64)])
@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_linear_bias_act_residual(M, N, K, dt):
    from occm_amd import ops
    x, w, b, r = _r(M, K, seed=1), _r(N, K, seed=2, scale=K ** -0.5), _r(N, seed=3), _r(M, N, seed=4)
    if dt == "bf16":
        if K % 8:
            pytest.skip("bf16 needs K % 8 == 0")
        xq, wq = x.bfloat16(), w.bfloat16()
        ref = F.gelu(xq.float() @ wq.float().T + b) + r
        out = ops.linear(xq.cuda(), wq.cuda(), b.cuda(), act=ops.ACT_GELU, residual=r.cuda(), out_dtype=torch.float32)
        tol = 2e-3
    else:
        ref = F.gelu(x @ w.T + b) + r
        out = ops.linear(x.cuda(), w.cuda(), b.cuda(), act=ops.ACT_GELU, residual=r.cuda())
        tol = 2e-5 * max(1.0, K / 256)
    torch.testing.assert_close(out.cpu(), ref, rtol=tol, atol=tol)


def test_linear_bf16_output_and_alpha():
    from occm_amd import ops
    x, w = _r(300, 256, seed=5).bfloat16(), _r(192, 256, seed=6, scale=1 / 16).bfloat16()
    out = ops.linear(x.cuda(), w.cuda(), alpha=0.5)
    assert out.dtype == torch.bfloat16
    ref = (0.5 * (x.float() @ w.float().T)).bfloat16()
    torch.testing.assert_close(out.cpu().float(), ref.float(), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("M,N,K", [(300, 48, 128), (128, 64, 64), (6368, 64, 8192), (1, 8, 64), (777, 20, 192), (12736, 1024, 4096), (6368, 512, 1024), (400, 1024, 512)])
def test_bf16_dispatch_classes_narrow_and_under_filled(M, N, K):
    """The default bf16 dispatch has three kernels: 128x64 tiles for N <= 64 (ragged N, clamped W rows), the in-workgroup split-K
    form for launches of at most two tiles per CU with an even slab count >= 8, and the plain 128x128 kernel.  All of them with bias,
    GELU and an f32 residual against an f64 reference of the bf16 operands."""
    from occm_amd import ops
    x, w = _r(M, K, seed=11).bfloat16(), _r(N, K, seed=12, scale=K ** -0.5).bfloat16()
    b, r = _r(N, seed=13), _r(M, N, seed=14)
    ref = (F.gelu(x.double() @ w.double().T + b.double()) + r.double()).float()
    out = ops.linear(x.cuda(), w.cuda(), b.cuda(), act=ops.ACT_GELU, residual=r.cuda(), out_dtype=torch.float32)
    torch.testing.assert_close(out.cpu(), ref, rtol=2e-3, atol=2e-3)
    out2 = ops.linear(x.cuda(), w.cuda(), b.cuda(), act=ops.ACT_GELU, residual=r.cuda(), out_dtype=torch.float32)
    assert torch.equal(out, out2)                                           # no atomics on these paths: run-to-run bit-identical


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("k,s,Tin", [(3, 2, 101), (2, 2, 64), (3, 2, 12799)])
def test_conv1d_as_window_gemm(dt, k, s, Tin):
    """Conv1d(512->512) over channels-last input == GEMM with overlapping row windows."""
    from occm_amd import ops
    from occm_amd._lib import dtype_code
    B, C = 2, 512
    x = _r(B, Tin, C, seed=7).to(dt)
    w = _r(C, C, k, seed=8, scale=(C * k) ** -0.5).to(dt)
    bias = _r(C, seed=9)
    Tout = (Tin - k) // s + 1
    ref = F.conv1d(x.float().transpose(1, 2), w.float(), bias, stride=s).transpose(1, 2)
    wp = w.permute(0, 2, 1).reshape(C, k * C).contiguous().cuda()
    out = torch.empty(B * Tout, C, device="cuda", dtype=torch.float32)
    code = dtype_code(x)
    ops.gemm_raw(B * Tout, C, k * C, x.cuda(), ops.rowmap(Tout, Tin * C, s * C), wp, k * C, out, ops.rowmap(B * Tout, 0, C),
                 ops.OCC_F32, code, bias=bias.cuda())
    tol = 3e-5 if dt == torch.float32 else 3e-3
    torch.testing.assert_close(out.cpu().view(B, Tout, C), ref, rtol=tol, atol=tol)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_grouped_pos_conv_gelu_residual(dt):
    """weight-normed grouped Conv1d(k=128, groups=16, pad=64) + SamePad + GELU + residual."""
    from occm_amd import ops
    from occm_amd._lib import dtype_code
    B, T, D, G, Kp = 2, 50, 256, 16, 128
    cg = D // G
    x = _r(B, T, D, seed=10).to(dt)
    w = _r(D, cg, Kp, seed=11, scale=(cg * Kp) ** -0.5).to(dt)
    bias = _r(D, seed=12)
    pc = F.conv1d(x.float().transpose(1, 2), w.float(), bias, padding=Kp // 2, groups=G)[:, :, :-1]
    ref = x.float() + F.gelu(pc).transpose(1, 2)
    xpad = torch.zeros(B, T + Kp, D, dtype=dt)
    xpad[:, Kp // 2: Kp // 2 + T] = x
    xpad = xpad.cuda()
    wp = w.reshape(G, cg, cg, Kp).permute(0, 1, 3, 2).reshape(G, cg, Kp * cg).contiguous().cuda()
    out = torch.empty(B * T, D, device="cuda", dtype=torch.float32)
    pmap = ops.rowmap(T, (T + Kp) * D, D)
    inner = xpad.data_ptr() + (Kp // 2) * D * xpad.element_size()
    code = dtype_code(xpad)
    ops.gemm_raw(B * T, cg, Kp * cg, xpad, pmap, wp, Kp * cg, out, ops.rowmap(B * T, 0, D), ops.OCC_F32, code, bias=bias.cuda(),
                 act=ops.ACT_GELU, R=inner, r_map=pmap, r_dtype=code, a_seg=(Kp, cg, D), groups=(G, cg, cg * Kp * cg, cg))
    tol = 3e-5 if dt == torch.float32 else 4e-3
    torch.testing.assert_close(out.cpu().view(B, T, D), ref, rtol=tol, atol=tol)


@pytest.mark.parametrize("C", [512, 1024, 1280, 256])
@pytest.mark.parametrize("gelu", [False, True])
def test_layernorm(C, gelu):
    from occm_amd import ops
    x, g, b = _r(37, C, seed=13, scale=3.0) + 0.7, 1 + 0.1 * _r(C, seed=14), 0.1 * _r(C, seed=15)
    ref = F.layer_norm(x, (C,), g, b)
    ref = F.gelu(ref) if gelu else ref
    out = ops.layernorm(x.cuda(), g.cuda(), b.cuda(), gelu=gelu)
    torch.testing.assert_close(out.cpu(), ref, rtol=2e-5, atol=2e-5)
    outb = ops.layernorm(x.bfloat16().cuda(), g.cuda(), b.cuda(), gelu=gelu, out_dtype=torch.float32)
    refb = F.layer_norm(x.bfloat16().float(), (C,), g, b)
    refb = F.gelu(refb) if gelu else refb
    torch.testing.assert_close(outb.cpu(), refb, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("L", [400, 16000, 64000])
def test_conv0_ln_gelu(L):
    from occm_amd import ops
    B = 2
    wav = _r(B, L, seed=16, scale=0.1)
    w, b, g, be = _r(512, 1, 10, seed=17, scale=0.3), 0.05 * _r(512, seed=18), 1 + 0.1 * _r(512, seed=19), 0.1 * _r(512, seed=20)
    ref = F.conv1d(wav.unsqueeze(1), w, b, stride=5).transpose(1, 2)
    ref = F.gelu(F.layer_norm(ref, (512,), g, be))
    out = ops.conv0_ln_gelu(wav.cuda(), w.reshape(512, 10).contiguous().cuda(), b.cuda(), g.cuda(), be.cuda(), 10, 5, torch.float32)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=1e-4)
    outb = ops.conv0_ln_gelu(wav.cuda(), w.reshape(512, 10).contiguous().cuda(), b.cuda(), g.cuda(), be.cuda(), 10, 5, torch.bfloat16)
    torch.testing.assert_close(outb.cpu().float(), ref, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("B,T,H,hd", [(2, 199, 16, 64), (1, 7, 2, 64), (3, 65, 4, 80), (2, 201, 4, 16), (2, 201, 3, 64), (1, 64, 2, 64), (2, 256, 2, 64), (1, 100, 5, 64), (1, 290, 2, 64),
                                        (2, 150, 3, 64), (2, 161, 2, 64), (2, 192, 3, 64), (40, 199, 16, 64), (1, 224, 2, 64), (1, 225, 2, 64)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_attention(B, T, H, hd, dt):
    from occm_amd import ops
    D = H * hd
    qkv = _r(B * T, 3 * D, seed=21).to(dt)
    q, k, v = [t.float().view(B, T, H, hd).transpose(1, 2) for t in qkv.split(D, dim=1)]
    ref = (torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, dim=-1) @ v).transpose(1, 2).reshape(B * T, D)
    out = ops.attention(qkv.cuda(), B, T, H, hd, hd ** -0.5)
    tol = 2e-5 if dt == torch.float32 else 1e-2
    torch.testing.assert_close(out.cpu().float(), ref, rtol=tol, atol=tol)


def _small_cfgs():
    from oracle import xlsr_ref
    from occm_amd.models import xlsr
    kw = dict(dim=256, ffn=512, heads=4, layers=2)
    return xlsr_ref.XlsrConfig(**kw), xlsr.XlsrConfig(**kw)


@pytest.mark.parametrize("L", [16000, 4000])
def test_xlsr_frontend_f32_matches_oracle(L):
    from oracle import xlsr_ref
    from oracle.fill import fill_like
    from occm_amd.models import xlsr
    rcfg, cfg = _small_cfgs()
    p = fill_like(xlsr_ref.param_shapes(rcfg), seed=3)
    wav = 0.1 * _r(2, L, seed=5)
    rt, taps = {}, {}
    with torch.no_grad():
        ref = xlsr_ref.extract_feat(wav, p, rcfg, rt)
    fe = xlsr.XlsrFrontend(p, cfg, dtype=torch.float32)
    out = fe.forward(wav.cuda(), taps=taps)
    assert out.shape == ref.shape
    torch.testing.assert_close(taps["conv"].cpu(), rt["conv"], rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(taps["pos"].cpu(), rt["pos"], rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-3, atol=1e-3)


def test_xlsr_frontend_bf16_close_to_oracle():
    """bf16 operands / f32 accumulate: stated bound 6e-2 max, 1e-2 mean absolute on unit-variance (LayerNormed) outputs."""
    from oracle import xlsr_ref
    from oracle.fill import fill_like
    from occm_amd.models import xlsr
    rcfg, cfg = _small_cfgs()
    p = fill_like(xlsr_ref.param_shapes(rcfg), seed=3)
    wav = 0.1 * _r(2, 16000, seed=5)
    with torch.no_grad():
        ref = xlsr_ref.extract_feat(wav, p, rcfg)
    out = xlsr.XlsrFrontend(p, cfg, dtype=torch.bfloat16).forward(wav.cuda(), out_dtype=torch.float32).cpu()
    err = (out - ref).abs()
    assert float(err.max()) < 6e-2 and float(err.mean()) < 1e-2, (float(err.max()), float(err.mean()))


def test_sslmodel_dropin_surface():
    from occm_amd.models.xlsr import SSLModel, XlsrConfig
    m = SSLModel("cuda", cfg=XlsrConfig(dim=256, ffn=512, heads=4, layers=1), dtype=torch.float32, synthetic=True)
    x = 0.1 * _r(2, 4000, seed=1).cuda()
    y = m.extract_feat(x)
    y3 = m.extract_feat(x.unsqueeze(-1))          # sslassist.py:42-43: [B,L,1] accepted
    assert m.out_dim == 256 and y.shape == (2, 12, 256) and torch.equal(y, y3)


@pytest.mark.parametrize("variant", [3, 14, 22, 30])
def test_bf16_gemm_kernel_family_agrees(variant):
    """occ_gemm_variant forces one kernel of the bf16 family (256x128 tile, half-slab pipeline, in-workgroup split-K, 256x256 eight-phase).
    They are tuning alternatives of the default and must give the same results on ragged tiles, conv windows and the grouped,
    K-segmented positional conv."""
    from occm_amd._lib import lib
    prev = lib().occ_gemm_variant(variant)
    try:
        for M, N, K in [(128, 128, 128), (6368, 1024, 512), (333, 132, 1536), (257, 3072, 64), (700, 260, 192)]:
            test_linear_bias_act_residual(M, N, K, "bf16")
        test_linear_bf16_output_and_alpha()
        test_conv1d_as_window_gemm(torch.bfloat16, 3, 2, 12799)
        test_grouped_pos_conv_gelu_residual(torch.bfloat16)
    finally:
        lib().occ_gemm_variant(prev)
    assert lib().occ_gemm_variant(-1) == prev


@pytest.mark.parametrize("N,Kd,M", [(128, 128, 8192), (512, 1536, 20480), (1024, 1024, 12736), (200, 136, 4096)])
def test_gemm_split_k_accumulate(N, Kd, M):
    """Weight-gradient form C[N,Kd] += A[N,M] . B[Kd,M]^T with few output tiles and a long reduction: occ_gemm splits K over
    workgroups and adds the pieces with f32 atomics (R aliases C = accumulate).  Must equal the one-workgroup result."""
    from occm_amd import ops
    a = _r(N, M, seed=1).bfloat16(); b = _r(Kd, M, seed=2).bfloat16()
    c0 = _r(N, Kd, seed=3)
    C = c0.clone().cuda()
    ops.gemm_raw(N, Kd, M, a.cuda(), ops.rowmap(N, 0, M), b.cuda(), M, C, ops.rowmap(N, 0, Kd), ops.OCC_F32, ops.OCC_BF16, R=C, r_map=ops.rowmap(N, 0, Kd),
                 r_dtype=ops.OCC_F32, alpha=0.5)
    ref = c0.double() + 0.5 * (a.double() @ b.double().T)
    torch.testing.assert_close(C.cpu().double(), ref, rtol=1e-4, atol=2e-5 * float(ref.abs().max()) + 1e-4)


@pytest.mark.parametrize("B,T,H", [(1, 650, 16), (2, 1001, 2), (1, 257, 3), (1, 384, 1), (1, 3000, 1)])
def test_attention_long_sequences(B, T, H):
    """T > 256 (evaluation utterances of any length): keys streamed in blocks of 128 with the online-softmax recurrence."""
    from occm_amd import ops
    hd, D = 64, H * 64
    qkv = _r(B * T, 3 * D, seed=31).bfloat16()
    q, k, v = [t.float().view(B, T, H, hd).transpose(1, 2) for t in qkv.split(D, dim=1)]
    ref = (torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, dim=-1) @ v).transpose(1, 2).reshape(B * T, D)
    out = ops.attention(qkv.cuda(), B, T, H, hd, hd ** -0.5)
    torch.testing.assert_close(out.cpu().float(), ref, rtol=1e-2, atol=1e-2)


def test_attention_long_forced_rescale():
    """A key far into the sequence that dominates one query's softmax forces the running-max rescale branch (the accumulators
    built from the first blocks must be scaled down by 2^(m_old - m_new)); random data alone rarely moves the max by much."""
    from occm_amd import ops
    B, T, H, hd = 1, 700, 2, 64
    D = H * hd
    qkv = _r(B * T, 3 * D, seed=33).view(T, 3, H, hd)
    qkv[5, 0, 0] = 3.0 * torch.ones(hd)                   # query 5 of head 0 ...
    qkv[600, 1, 0] = 4.0 * torch.ones(hd)                 # ... matches key 600 (5th block): score 768 * scale = 96 >> the rest
    qkv[300, 1, 1] = -qkv[9, 0, 1] * 2.0                  # and a strongly negative one elsewhere
    qkv = qkv.reshape(T, 3 * D).bfloat16()
    q, k, v = [t.float().view(B, T, H, hd).transpose(1, 2) for t in qkv.split(D, dim=1)]
    ref = (torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, dim=-1) @ v).transpose(1, 2).reshape(B * T, D)
    lse = torch.empty(B * H * T, device="cuda")
    out = ops.attention(qkv.cuda(), B, T, H, hd, hd ** -0.5, lse=lse)
    torch.testing.assert_close(out.cpu().float(), ref, rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(out.cpu().float()[5, :hd], v[0, 0, 600], rtol=1e-2, atol=1e-2)   # that query's output is (almost) exactly v[600]
    s2 = q @ k.transpose(-1, -2) * hd ** -0.5 * 1.4426950408889634          # scores in the log2 domain, as the kernel keeps them
    ref_lse = torch.log2(torch.exp2(s2 - s2.amax(-1, keepdim=True)).sum(-1)) + s2.amax(-1)
    torch.testing.assert_close(lse.cpu().view(B, H, T), ref_lse, rtol=1e-3, atol=2e-2)


def test_frontend_long_utterance_matches_oracle():
    """13 s of audio (T = 649 frames) through the bf16 front-end vs the fp32 oracle (same bar as the 4 s bf16 test)."""
    from oracle import xlsr_ref
    from oracle.fill import fill_like
    from occm_amd.models import xlsr
    rcfg, cfg = _small_cfgs()
    p = fill_like(xlsr_ref.param_shapes(rcfg), seed=3)
    wav = 0.1 * _r(1, 208000, seed=5)
    with torch.no_grad():
        ref = xlsr_ref.extract_feat(wav, p, rcfg)
    out = xlsr.XlsrFrontend(p, cfg, dtype=torch.bfloat16).forward(wav.cuda(), out_dtype=torch.float32).cpu()
    assert out.shape == ref.shape and out.shape[1] == 649
    err = (out - ref).abs()
    assert float(err.max()) < 6e-2 and float(err.mean()) < 1e-2, (float(err.max()), float(err.mean()))


@pytest.mark.parametrize("B,T,H", [(2, 199, 16), (1, 650, 3), (3, 65, 4), (1, 7, 2)])
def test_attention_head_dim_80(B, T, H):
    """XLS-R-1B geometry (1280 / 16 heads): head_dim 80 on the streaming MFMA kernel (third k-step half zero, 5 output blocks)."""
    from occm_amd import ops
    hd, D = 80, H * 80
    qkv = _r(B * T, 3 * D, seed=41).bfloat16()
    q, k, v = [t.float().view(B, T, H, hd).transpose(1, 2) for t in qkv.split(D, dim=1)]
    ref = (torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, dim=-1) @ v).transpose(1, 2).reshape(B * T, D)
    lse = torch.empty(B * H * T, device="cuda")
    out = ops.attention(qkv.cuda(), B, T, H, hd, hd ** -0.5, lse=lse)
    torch.testing.assert_close(out.cpu().float(), ref, rtol=1e-2, atol=1e-2)
    s2 = q @ k.transpose(-1, -2) * hd ** -0.5 * 1.4426950408889634
    ref_lse = torch.log2(torch.exp2(s2 - s2.amax(-1, keepdim=True)).sum(-1)) + s2.amax(-1)
    torch.testing.assert_close(lse.cpu().view(B, H, T), ref_lse, rtol=1e-3, atol=2e-2)


def test_xlsr_1b_geometry_frontend_matches_oracle():
    """Two encoder layers with XLS-R-1B's dimensions (d = 1280, ffn = 5120, 16 heads of 80, 80-channel positional-conv groups) against
    the fp32 oracle: bf16 bound as for the 300M geometry."""
    from oracle import xlsr_ref
    from oracle.fill import fill_like
    from occm_amd.models import xlsr
    kw = dict(dim=1280, ffn=5120, heads=16, layers=2)
    rcfg, cfg = xlsr_ref.XlsrConfig(**kw), xlsr.XlsrConfig(**kw)
    p = fill_like(xlsr_ref.param_shapes(rcfg), seed=3)
    wav = 0.1 * _r(2, 16000, seed=5)
    with torch.no_grad():
        ref = xlsr_ref.extract_feat(wav, p, rcfg)
    out = xlsr.XlsrFrontend(p, cfg, dtype=torch.bfloat16).forward(wav.cuda(), out_dtype=torch.float32).cpu()
    assert out.shape == ref.shape == (2, 49, 1280)
    err = (out - ref).abs()
    assert float(err.max()) < 6e-2 and float(err.mean()) < 1e-2, (float(err.max()), float(err.mean()))


def _fuzz_case(rs):
    """One random occ_gemm problem: 3-level row maps on A / C / R, K-segments, groups, every operand / output / residual dtype."""
    from occm_amd import ops
    mode = rs.choice(["f32", "bf16", "f32_as_bf16", "af32_wbf16"])
    ce = 4 if mode == "f32" else 8
    G = int(rs.choice([1, 1, 1, 2, 3]))
    nseg = int(rs.choice([1, 1, 2, 3]))
    seg_len = int(rs.randint(1, 9)) * ce * (8 if rs.rand() < 0.3 else 1)
    K = nseg * seg_len
    N = int(rs.randint(1, 70)) * 4
    nb, nl, rpl = int(rs.randint(1, 4)), int(rs.randint(1, 6)), int(rs.randint(1, 40))
    M = nb * nl * rpl
    # A: rows in lines in batches, row stride >= seg_len (overlapping windows allowed when nseg == 1), padded strides
    rstride = int(rs.choice([seg_len, seg_len + ce, ce])) if nseg == 1 else seg_len + ce * int(rs.randint(0, 3))
    seg_stride = (rpl * rstride + seg_len + ce * int(rs.randint(0, 3))) if nseg > 1 else 0
    lstride = (rpl - 1) * rstride + K + seg_stride * (nseg - 1) + ce * int(rs.randint(0, 4))
    lstride = (lstride + ce - 1) // ce * ce
    bstride = nl * lstride + ce * int(rs.randint(0, 4)) + seg_stride * nseg
    gstride_a = (nb * bstride + 2 * K + ce - 1) // ce * ce
    a_len = G * gstride_a + K + seg_stride * nseg
    a_f = torch.from_numpy(rs.randn(a_len).astype("float32"))
    w_f = torch.from_numpy((rs.randn(G, N, K) / K ** 0.5).astype("float32"))
    bias = torch.from_numpy(rs.randn(G * N).astype("float32")) if rs.rand() < 0.7 else None
    act = rs.choice([ops.ACT_NONE, ops.ACT_GELU, ops.ACT_RELU, ops.ACT_TANH])
    alpha = float(rs.choice([1.0, 1.0, 0.5]))
    c_bf, r_kind = rs.rand() < 0.5, rs.choice(["none", "f32", "bf16"])
    ldc = G * N + 4 * int(rs.randint(0, 3))
    c_rows = nb * nl * rpl
    r_f = torch.from_numpy(rs.randn(c_rows, ldc).astype("float32"))
    a_dev = a_f.bfloat16() if mode == "bf16" else a_f
    w_dev = w_f.bfloat16() if mode in ("bf16", "af32_wbf16") else w_f
    # reference operands as the kernel sees them
    a_ref = a_f.bfloat16().float() if mode != "f32" else a_f
    w_ref = w_f.bfloat16().float() if mode != "f32" else w_f
    rows = []
    for g in range(G):
        for m in range(M):
            b, rem = divmod(m, nl * rpl); l, r = divmod(rem, rpl)
            base = g * gstride_a + b * bstride + l * lstride + r * rstride
            rows.append(torch.cat([a_ref[base + sgi * seg_stride: base + sgi * seg_stride + seg_len] for sgi in range(nseg)]))
    A = torch.stack(rows).view(G, M, K).double()
    ref = alpha * torch.einsum("gmk,gnk->gmn", A, w_ref.double())
    if bias is not None:
        ref = ref + bias.view(G, 1, N).double()
    ref = {ops.ACT_NONE: lambda v: v, ops.ACT_GELU: lambda v: F.gelu(v), ops.ACT_RELU: torch.relu, ops.ACT_TANH: torch.tanh}[act](ref)
    ref = ref.permute(1, 0, 2).reshape(M, G * N)
    R = None
    if r_kind != "none":
        R = (r_f.bfloat16() if r_kind == "bf16" else r_f)
        ref = ref + R[:, :G * N].double()
    C = torch.full((c_rows, ldc), 7.0, dtype=torch.bfloat16 if c_bf else torch.float32).cuda()
    codes = {"f32": ops.OCC_F32, "bf16": ops.OCC_BF16, "f32_as_bf16": ops.OCC_F32_AS_BF16, "af32_wbf16": ops.OCC_AF32_WBF16}
    Rd = R.cuda() if R is not None else None
    ad, wd = a_dev.cuda(), w_dev.contiguous().cuda()
    ops.gemm_raw(M, N, K, ad, ops.rowmap(nl * rpl, bstride, rstride, rpl, lstride), wd, K, C, ops.rowmap(c_rows, 0, ldc), ops.OCC_BF16 if c_bf else ops.OCC_F32,
                 codes[mode], bias=bias.cuda() if bias is not None else None, act=act, alpha=alpha, R=Rd, r_map=None if Rd is None else ops.rowmap(c_rows, 0, ldc),
                 r_dtype=ops.OCC_BF16 if r_kind == "bf16" else ops.OCC_F32, a_seg=(nseg, seg_len, seg_stride) if nseg > 1 else None,
                 groups=(G, gstride_a, N * K, N) if G > 1 else None)
    got = C.cpu().double()
    tol = (2e-5 if mode == "f32" else 3e-3) * max(1.0, float(ref.abs().max())) + (1e-2 * float(ref.abs().max()) if c_bf else 0.0)
    assert float((got[:, :G * N] - ref).abs().max()) <= tol, (mode, M, N, K, G, nseg, act, c_bf, r_kind, float((got[:, :G * N] - ref).abs().max()), tol)
    if ldc > G * N:
        assert bool((got[:, G * N:] == 7.0).all()), "columns outside N were written"


@pytest.mark.parametrize("seed", range(12))
def test_gemm_fuzz_row_maps_segments_groups_dtypes(seed):
    rs = np.random.RandomState(1000 + seed)
    for _ in range(6):
        _fuzz_case(rs)


@pytest.mark.parametrize("M,N,K", [(199, 1024, 1024), (3184, 4096, 1024), (777, 260, 136), (12736, 512, 1536), (64, 64, 4096)])
def test_f32x3_split_operand_gemm_is_f32_grade(M, N, K):
    """OCC_F32X3: f32 operands split into bf16 hi + lo while staged, Wh.Xh + Wl.Xh + Wh.Xl on the bf16 MFMA.  Against f64: the error must
    sit in the exact-f32 kernel's class (here: within 2e-5 of the largest output, ~100x below a bf16-operand product), with bias, GELU,
    an f32 residual, ragged N and a K that is not a multiple of the 32-element slab."""
    from occm_amd import ops
    from occm_amd._lib import OCC_F32X3
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g); r = torch.randn(M, N, generator=g)
    ref = x.double() @ w.double().T + b.double()
    errs = {}
    for name, ab in (("x3", OCC_F32X3), ("f32", ops.OCC_F32), ("as_bf16", ops.OCC_F32_AS_BF16)):
        out = torch.empty(M, N, device="cuda")
        ops.gemm_raw(M, N, K, x.cuda(), ops.rowmap(M, 0, K), w.cuda(), K, out, ops.rowmap(M, 0, N), ops.OCC_F32, ab, bias=b.cuda())
        errs[name] = float((out.cpu().double() - ref).abs().max()) / float(ref.abs().max())
    assert errs["x3"] < 2e-5 and errs["x3"] < 0.02 * errs["as_bf16"] and errs["f32"] < 2e-6, errs
    out = torch.empty(M, N, device="cuda")
    ops.gemm_raw(M, N, K, x.cuda(), ops.rowmap(M, 0, K), w.cuda(), K, out, ops.rowmap(M, 0, N), ops.OCC_F32, OCC_F32X3, bias=b.cuda(), act=ops.ACT_GELU,
                 R=r.cuda(), r_map=ops.rowmap(M, 0, N), r_dtype=ops.OCC_F32)
    ref2 = torch.nn.functional.gelu(ref) + r.double()
    assert float((out.cpu().double() - ref2).abs().max()) < 3e-5 * float(ref2.abs().max())


@pytest.mark.parametrize("M,N,K", [(199, 1024, 1024), (3184, 4096, 1024), (3184, 1024, 4096), (12736, 3072, 1024)])
def test_split3_panels_give_an_f32_grade_product_as_one_bf16_gemm(M, N, K):
    """occ_split3_bf16: activations [xh | xl | xh], weights [wh | wh | wl]; ONE bf16 GEMM of depth 3K on the LDS-DMA kernels then equals
    xh.wh + xl.wh + xh.wl.  Panels bit-exact against torch; the product within 2e-5 of the largest f64 output (the exact-f32 kernel's
    class, ~100x below a bf16-operand product), with bias, GELU and an f32 residual that aliases the output (the out-proj / fc2 form)."""
    from occm_amd import ops
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g); r = torch.randn(M, N, generator=g)
    a3, w3 = ops.split3_bf16(x.cuda(), mode=0), ops.split3_bf16(w.cuda(), mode=1)
    xh = x.bfloat16(); xl = (x - xh.float()).bfloat16()
    wh = w.bfloat16(); wl = (w - wh.float()).bfloat16()
    assert torch.equal(a3.cpu(), torch.cat([xh, xl, xh], 1)) and torch.equal(w3.cpu(), torch.cat([wh, wh, wl], 1))
    ref = x.double() @ w.double().T + b.double()
    out = torch.empty(M, N, device="cuda")
    ops.gemm_raw(M, N, 3 * K, a3, ops.rowmap(M, 0, 3 * K), w3, 3 * K, out, ops.rowmap(M, 0, N), ops.OCC_F32, ops.OCC_BF16, bias=b.cuda())
    assert float((out.cpu().double() - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    c = r.cuda().clone()
    ops.gemm_raw(M, N, 3 * K, a3, ops.rowmap(M, 0, 3 * K), w3, 3 * K, c, ops.rowmap(M, 0, N), ops.OCC_F32, ops.OCC_BF16, bias=b.cuda(), R=c, r_map=ops.rowmap(M, 0, N),
                 r_dtype=ops.OCC_F32)
    assert float((c.cpu().double() - (ref + r.double())).abs().max()) < 2e-5 * float(ref.abs().max())
    o2 = torch.empty(M, N, device="cuda")
    ops.gemm_raw(M, N, 3 * K, a3, ops.rowmap(M, 0, 3 * K), w3, 3 * K, o2, ops.rowmap(M, 0, N), ops.OCC_F32, ops.OCC_BF16, bias=b.cuda(), act=ops.ACT_GELU)
    assert float((o2.cpu().double() - torch.nn.functional.gelu(ref)).abs().max()) < 3e-5 * float(ref.abs().max())


@pytest.mark.parametrize("case,gemm", [("a", "exact"), ("b", "exact"), ("b", "x3")])
def test_xlsr_f32_path_matches_huggingface_proxy_fixtures(case, gemm):
    """HIP f32-MFMA front-end vs outputs of HuggingFace ``Wav2Vec2Model`` at the XLS-R-300M geometry (tests/golden/xlsr_hf.npz, written in
    the build container by oracle/gen_golden_hf.py from seeds): an implementation nobody in this repository wrote.  Case "a": 2 layers,
    16000 samples, every tap; case "b": all 24 layers, 64000 samples, final output and three intermediate layers.  The reference's own
    front-end (fairseq) cannot run anywhere here, so this does not pin parity with the reference -- it removes the builder's
    restatement from the comparison."""
    from conftest import golden
    from oracle import xlsr_ref
    from oracle.fill import fill_like
    from occm_amd.models import xlsr
    G = golden("xlsr_hf.npz")
    layers, B, L, wseed, xseed, st, ost = [int(v) for v in G[case + "_meta"]]
    p = fill_like(xlsr_ref.param_shapes(xlsr_ref.XlsrConfig(dim=1024, ffn=4096, heads=16, layers=layers)), seed=wseed)
    wav = 0.1 * torch.randn(B, L, generator=torch.Generator().manual_seed(xseed))
    fe = xlsr.XlsrFrontend(p, xlsr.XlsrConfig(dim=1024, ffn=4096, heads=16, layers=layers), dtype=torch.float32, f32_gemm=gemm)
    taps = {}
    out = fe.forward(wav.cuda(), out_dtype=torch.float32, taps=taps).cpu()
    normed = F.layer_norm(taps["conv"].cpu(), (512,), p["layer_norm.weight"], p["layer_norm.bias"])
    worst = {}
    worst["extract_features"] = float((normed[:, ::st] - torch.from_numpy(G[case + "_extract_features"])).abs().max())
    worst["pos"] = float((taps["pos"].cpu()[:, ::st] - torch.from_numpy(G[case + "_pos"])).abs().max())
    for k in G.files:
        if k.startswith(case + "_layer"):
            ref = torch.from_numpy(G[k])
            got = taps[k[len(case) + 1:]].cpu()[:, ::st]
            worst[k] = float((got - ref).abs().max()) / float(ref.abs().max())       # the residual stream grows with depth: relative to its largest value
    worst["out"] = float((out[:, ::ost] - torch.from_numpy(G[case + "_out"])).abs().max())
    print("HIP f32 (%s GEMMs) vs HF proxy, case %s: %s" % (gemm, case, worst))
    assert worst["extract_features"] < 1e-3 and worst["pos"] < 1e-3 and worst["out"] < 1e-3, worst
    assert all(v < 2e-4 for k, v in worst.items() if "layer" in k), worst
