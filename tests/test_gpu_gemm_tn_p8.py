"""The 256x256 eight-phase weight-gradient kernel (csrc/gemm_tn_p8.hip) through occ_gemm_tn: exact-integer checks (small-integer bf16
operands: every f32 partial sum is exact, so the slab / ticket reduction is order-independent and any misplaced row, column block
or k-slot is an integer difference), reduction pieces with and without a workspace, a row count that is not a multiple of 64 (the
tail goes through the small-tile kernel), conv windows through row maps, accumulation onto a non-zero C, a run-to-run race screen."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ints(rows, cols, seed, lo=-2, hi=2):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, (rows, cols), generator=g).float()


@pytest.mark.parametrize("M,N1,N2", [(1024, 256, 256), (12736, 1024, 1024), (12736, 4096, 1024), (6368, 1024, 3072), (2000, 512, 256), (1088, 256, 768)])
@pytest.mark.parametrize("use_ws", [True, False])
def test_tn_p8_exact_integer_products(M, N1, N2, use_ws):
    from occm_amd import backend_ops as K
    a, b = _ints(M, N1, 1), _ints(M, N2, 2)
    c0 = _ints(N1, N2, 3, -50, 50)
    C = c0.clone().cuda()
    s = torch.zeros(N1, device="cuda")
    if not use_ws:
        K._TN_WS.clear()
        saved, K.TN_WORKSPACE_BYTES = K.TN_WORKSPACE_BYTES, 16            # too small for any slab: one piece per tile
    try:
        K.gemm_tn(M, N1, N2, a.bfloat16().cuda(), K.full(M, N1), b.bfloat16().cuda(), K.full(M, N2), C, N2, alpha=0.5, colsum_out=s, a_bf16=True, b_bf16=True,
                  bf16_mfma=True)
    finally:
        if not use_ws:
            K.TN_WORKSPACE_BYTES = saved
            K._TN_WS.clear()
    ref = c0.double() + 0.5 * (a.double().T @ b.double())
    assert torch.equal(C.cpu().double(), ref), float((C.cpu().double() - ref).abs().max())
    assert torch.equal(s.cpu().double(), 0.5 * a.double().sum(0))


def test_tn_p8_conv_windows_row_maps():
    """Conv1d(k=3, s=2) weight gradient: X rows are overlapping windows of the channels-last activation (row stride s*C, one batch
    stride per utterance), dY rows sit inside a buffer with one pad row in front of every utterance."""
    from occm_amd import backend_ops as K
    B, Tin, Cc, k, st = 4, 1025, 512, 3, 2
    Tout = (Tin - k) // st + 1                      # 512 -> M = 2048
    M = B * Tout
    x = _ints(B * Tin, Cc, 5)
    dy_pad = torch.zeros(B, Tout + 2, 512)
    dy_pad[:, 1:-1] = _ints(M, 512, 6).view(B, Tout, 512)
    C = torch.zeros(512, k * Cc, device="cuda")
    dyd = dy_pad.bfloat16().cuda()
    K.gemm_tn(M, 512, k * Cc, dyd.data_ptr() + 512 * 2, K.rowmap(Tout, (Tout + 2) * 512, 512), x.bfloat16().cuda(), K.rowmap(Tout, Tin * Cc, st * Cc), C, k * Cc,
              a_bf16=True, b_bf16=True, bf16_mfma=True)
    win = x.view(B, Tin, Cc).unfold(1, k, st).permute(0, 1, 3, 2).reshape(M, k * Cc)
    ref = dy_pad[:, 1:-1].reshape(M, 512).double().T @ win.double()
    assert torch.equal(C.cpu().double(), ref)


def test_tn_p8_random_data_and_race_screen():
    from occm_amd import backend_ops as K
    g = torch.Generator().manual_seed(7)
    probs = []
    for M, N1, N2 in [(12736, 1024, 4096), (12736, 3072, 1024), (6368, 1024, 1024)]:
        a = torch.randn(M, N1, generator=g).bfloat16().cuda(); b = torch.randn(M, N2, generator=g).bfloat16().cuda()
        probs.append((M, N1, N2, a, b, torch.empty(N1, N2, device="cuda")))
    first = {}
    for it in range(12):
        for i, (M, N1, N2, a, b, C) in enumerate(probs):
            C.zero_()
            K.gemm_tn(M, N1, N2, a, K.full(M, N1), b, K.full(M, N2), C, N2, a_bf16=True, b_bf16=True, bf16_mfma=True)
            if it == 0:
                ref = a.float().T @ b.float()
                torch.testing.assert_close(C, ref, rtol=2e-3, atol=2e-3 * float(ref.abs().max()))
                first[i] = C.clone()
            else:
                assert torch.equal(C, first[i]), (it, M, N1, N2)         # fixed summation order: bit-reproducible


@pytest.mark.parametrize("M,shapes", [(12736, ((1024, 1024), (3072, 1024))), (12736, ((1024, 4096), (4096, 1024))), (2048, ((256, 256), (512, 256))),
                                      (1000, ((256, 256), (256, 512))), (6368, ((1280, 1280), (3840, 1280))), (2388, ((1024, 4096), (4096, 1024)))])
def test_tn_p8_pair_launch_exact(M, shapes):
    """occ_gemm_tn_pair: two weight (+ bias) gradients with the same reduction rows in one launch (out-proj with qkv, fc2 with fc1 at bs 64;
    a small pair; a row count under 1024, which runs as two single launches; row counts that are not multiples of 64 -- the paired launch
    covers the whole 64-row K-tiles and the M % 64 tail rows go through the small-tile kernel per product).  Integer operands: exact, with
    accumulation onto non-zero C and bias buffers."""
    from occm_amd import backend_ops as K
    ops_in, refs, outs = [], [], []
    for p, (N1, N2) in enumerate(shapes):
        a, b = _ints(M, N1, 10 + p), _ints(M, N2, 20 + p)
        c0, s0 = _ints(N1, N2, 30 + p, -50, 50), _ints(1, N1, 40 + p, -9, 9)[0]
        C, s = c0.clone().cuda(), s0.clone().cuda()
        ad, bd = a.bfloat16().cuda(), b.bfloat16().cuda()
        ops_in.append((N1, N2, ad, K.full(M, N1), bd, K.full(M, N2), C, N2, s))
        refs.append((c0.double() + a.double().T @ b.double(), s0.double() + a.double().sum(0)))
        outs.append((C, s))
    K.gemm_tn_pair(M, ops_in[0], ops_in[1])
    for (C, s), (rc, rs) in zip(outs, refs):
        assert torch.equal(C.cpu().double(), rc), float((C.cpu().double() - rc).abs().max())
        assert torch.equal(s.cpu().double(), rs)


@pytest.mark.parametrize("B,T,G,cg,Kp", [(3, 70, 4, 64, 8), (2, 199, 16, 64, 128)])
def test_grouped_conv_weight_gradient_in_one_launch_exact_integers(B, T, G, cg, Kp):
    """The positional conv's weight gradient (grouped Conv1d, k = Kp, groups = G: fairseq make_conv_pos) as ONE grouped occ_gemm_tn:
    group g multiplies column slice g of dY [B*T, D] with the Kp shifted windows of column slice g of the zero-padded input
    [B, T + Kp, D] (K-segments of cg channels, one per tap).  Small-integer operands: every f32 sum is exact, so any misplaced group,
    tap segment or row is an integer difference.  Checked against the 16 per-group products of round 2 and an f64 einsum."""
    from occm_amd import backend_ops as K
    from occm_amd.ops import rowmap
    D, M, Tp = G * cg, B * T, T + Kp
    g = torch.Generator().manual_seed(7)
    dy = torch.randint(-2, 3, (B, Tp, D), generator=g).float()            # stored in the padded layout, interior rows [Kp/2 - 1, Kp/2 - 1 + T)
    x = torch.randint(-2, 3, (B, Tp, D), generator=g).float()
    lo = Kp // 2 - 1
    dyb, xb = dy.bfloat16().cuda(), x.bfloat16().cuda()
    dmap = rowmap(T, Tp * D, D)
    dy_in = dyb.data_ptr() + lo * D * 2
    C = torch.zeros(G, cg, Kp * cg, device="cuda")
    K.gemm_tn(M, cg, Kp * cg, dy_in, dmap, xb, dmap, C, Kp * cg, b_seg=(Kp, cg, D), a_bf16=True, b_bf16=True, bf16_mfma=True, groups=(G, cg, cg, cg * Kp * cg))
    # reference: dW[g, co, tap, ci] = sum_{b,t} dy[b, lo + t, g*cg + co] * x[b, t + tap, g*cg + ci]
    dyi = dy[:, lo:lo + T].reshape(B, T, G, cg).double()
    win = torch.stack([x[:, tap:tap + T] for tap in range(Kp)], 2).reshape(B, T, Kp, G, cg).double()
    ref = torch.einsum("btgo,btkgi->goki", dyi, win).reshape(G, cg, Kp * cg)
    assert torch.equal(C.cpu().double(), ref), float((C.cpu().double() - ref).abs().max())
    C1 = torch.zeros_like(C)
    for gi in range(min(G, 3)):                                           # the per-group launches it replaces give the same integers
        K.gemm_tn(M, cg, Kp * cg, dy_in + gi * cg * 2, dmap, xb.data_ptr() + gi * cg * 2, dmap, C1[gi], Kp * cg, b_seg=(Kp, cg, D), a_bf16=True, b_bf16=True, bf16_mfma=True)
        assert torch.equal(C1[gi], C[gi])
    s = torch.zeros(D, device="cuda")
    K.colsum(dy_in, dmap, M, D, s, a_dtype=1)
    assert torch.equal(s.cpu().double(), dy[:, lo:lo + T].reshape(M, D).double().sum(0))


@pytest.mark.parametrize("M,shapes", [(12736, ((1024, 1024), (3072, 1024))), (12736, ((1024, 4096), (4096, 1024))), (6368, ((1280, 1280), (3840, 1280)))])
def test_tn_p8_c_is_zero_hint_stores_instead_of_accumulating(M, shapes):
    """occ_gemm_tn_desc.c_is_zero: with C cleared the result is the product, exactly as with accumulation onto zeros; with a C that is NOT
    zero the hinted call overwrites it (which shows the store path is the one that ran) while the plain call accumulates.  The bias sums
    and the M % 64 tail rows, which are added after the slab reduce, are unaffected."""
    from occm_amd import backend_ops as K
    prods, ops_in = [], {}
    for hint, c_init in ((True, 0.0), (False, 0.0), (True, 7.0)):
        cur = []
        for p, (N1, N2) in enumerate(shapes):
            a, b = _ints(M, N1, 10 + p), _ints(M, N2, 20 + p)
            C, s = torch.full((N1, N2), c_init, device="cuda"), torch.zeros(N1, device="cuda")
            cur.append((N1, N2, a.bfloat16().cuda(), K.full(M, N1), b.bfloat16().cuda(), K.full(M, N2), C, N2, s))
            if len(prods) <= p:
                prods.append((a.double().T @ b.double(), a.double().sum(0)))
        K.gemm_tn_pair(M, cur[0], cur[1], c_is_zero=hint)
        for (N1, N2, _, _, _, _, C, _, s), (rc, rs) in zip(cur, prods):
            assert torch.equal(C.cpu().double(), rc), (hint, c_init, float((C.cpu().double() - rc).abs().max()))
            assert torch.equal(s.cpu().double(), rs)
