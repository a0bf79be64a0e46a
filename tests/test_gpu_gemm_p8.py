"""The 256x256 eight-phase bf16 GEMM (csrc/gemm_p8.hip) through occ_gemm: exact-integer checks of the whole tile/fragment/DMA
addressing (small-integer bf16 operands make every f32 sum exact, so any misplaced row, chunk or k-step shows as an integer
difference), ragged tile edges, odd / one / two K-tiles, conv windows through row maps, epilogues, and a run-to-run race screen."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _force_p8():
    from occm_amd._lib import lib
    prev = lib().occ_gemm_variant(30)
    yield
    lib().occ_gemm_variant(prev)


def _ints(rows, cols, seed, lo=-3, hi=3):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, (rows, cols), generator=g).float()


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (256, 256, 128), (512, 256, 192), (300, 256, 320), (1000, 512, 1024), (257, 768, 1536),
                                   (12736, 1024, 1024), (6368, 4096, 1024), (2049, 260, 448)])
def test_p8_exact_integer_products(M, N, K):
    from occm_amd import ops
    x, w = _ints(M, K, 1), _ints(N, K, 2)                 # asymmetric operands: a swapped row/column map cannot cancel
    bias = _ints(1, N, 3)[0]
    ref = x.double() @ w.double().T + bias.double()
    out = torch.full((M, N), 7777.0, device="cuda")
    ops.gemm_raw(M, N, K, x.bfloat16().cuda(), ops.rowmap(M, 0, K), w.bfloat16().cuda(), K, out, ops.rowmap(M, 0, N), ops.OCC_F32, ops.OCC_BF16, bias=bias.cuda())
    assert torch.equal(out.cpu().double(), ref), (M, N, K, float((out.cpu().double() - ref).abs().max()))
    from occm_amd._lib import lib
    assert lib().occ_gemm_last_kernel() == 8               # OCC_GEMM_KERNEL_P8: the launch really went to the eight-phase kernel


def test_default_dispatch_sends_well_filled_bf16_launches_to_p8():
    """Heuristic dispatch (variant 1): the bs-64 encoder shapes go to the eight-phase kernel, an under-filled launch does not."""
    from occm_amd import ops
    from occm_amd._lib import lib
    prev = lib().occ_gemm_variant(1)
    try:
        for (M, N, K), want in (((12736, 1024, 1024), 11), ((12736, 4096, 1024), 11), ((12736, 3072, 1024), 11), ((12736, 1024, 4096), 11), ((4096, 4096, 4096), 8), ((512, 512, 1024), 0)):
            x = torch.zeros(M, K, device="cuda", dtype=torch.bfloat16); w = torch.zeros(N, K, device="cuda", dtype=torch.bfloat16)
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            ops.gemm_raw(M, N, K, x, ops.rowmap(M, 0, K), w, K, out, ops.rowmap(M, 0, N), ops.OCC_BF16, ops.OCC_BF16)
            assert lib().occ_gemm_last_kernel() == want, (M, N, K)
    finally:
        lib().occ_gemm_variant(prev)


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (448, 256, 128), (1000, 512, 1024), (12736, 1024, 1024), (2049, 260, 448)])
def test_p8_224_row_tiles_exact(M, N, K):
    """Variant 31: seven 16-row blocks per wave.  Exact integer products, rows past a wave's 112 untouched (the output buffer is
    pre-filled and compared whole), bf16 output with GELU + saved pre-activation against the 256-row kernel bit for bit."""
    from occm_amd import ops
    from occm_amd._lib import lib
    x, w = _ints(M, K, 1), _ints(N, K, 2)
    bias = _ints(1, N, 3)[0]
    ref = x.double() @ w.double().T + bias.double()
    xb, wb = x.bfloat16().cuda(), w.bfloat16().cuda()
    res = []
    for variant in (31, 30):
        lib().occ_gemm_variant(variant)
        out = torch.full((M, N), 7777.0, device="cuda")
        ops.gemm_raw(M, N, K, xb, ops.rowmap(M, 0, K), wb, K, out, ops.rowmap(M, 0, N), ops.OCC_F32, ops.OCC_BF16, bias=bias.cuda())
        assert lib().occ_gemm_last_kernel() == (11 if variant == 31 else 8)
        assert torch.equal(out.cpu().double(), ref), (variant, float((out.cpu().double() - ref).abs().max()))
        o16 = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16); aux = torch.zeros_like(o16)
        ops.gemm_raw(M, N, K, xb, ops.rowmap(M, 0, K), wb, K, o16, ops.rowmap(M, 0, N), ops.OCC_BF16, ops.OCC_BF16, bias=bias.cuda(), act=ops.ACT_GELU, aux=aux, alpha=1.0 / 64)
        res.append((o16, aux))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (417, 256, 128), (624, 256, 128), (416, 512, 192), (1000, 512, 1024), (12736, 1024, 1024), (2049, 260, 448), (6368, 1280, 1280)])
def test_p8_208_row_tiles_exact(M, N, K):
    """Variant 32: 208-row tiles -- the upper wave row's 7 blocks (112 rows) and 6 of the lower one's (96 rows).  Exact integer products in
    a pre-filled output compared whole (a row written twice, skipped, or taken from the neighbouring tile shows), bf16 output with GELU +
    saved pre-activation and the GELU' column sums against the 256-row kernel bit for bit."""
    from occm_amd import ops
    from occm_amd._lib import lib
    x, w = _ints(M, K, 1), _ints(N, K, 2)
    bias = _ints(1, N, 3)[0]
    ref = x.double() @ w.double().T + bias.double()
    xb, wb = x.bfloat16().cuda(), w.bfloat16().cuda()
    u = _ints(M, N, 4, -2, 2).bfloat16().cuda()
    res = []
    prev = lib().occ_gemm_variant(-1)
    try:
        for variant in (32, 30):
            lib().occ_gemm_variant(variant)
            out = torch.full((M, N), 7777.0, device="cuda")
            ops.gemm_raw(M, N, K, xb, ops.rowmap(M, 0, K), wb, K, out, ops.rowmap(M, 0, N), ops.OCC_F32, ops.OCC_BF16, bias=bias.cuda())
            assert lib().occ_gemm_last_kernel() == (11 if variant == 32 else 8)
            assert torch.equal(out.cpu().double(), ref), (variant, float((out.cpu().double() - ref).abs().max()))
            o16 = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16); aux = torch.zeros_like(o16)
            ops.gemm_raw(M, N, K, xb, ops.rowmap(M, 0, K), wb, K, o16, ops.rowmap(M, 0, N), ops.OCC_BF16, ops.OCC_BF16, bias=bias.cuda(), act=ops.ACT_GELU, aux=aux, alpha=1.0 / 64)
            cs = torch.full((N,), 0.5, device="cuda")
            du = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
            if N % 8 == 0:
                ops.gemm_raw(M, N, K, xb, ops.rowmap(M, 0, K), wb, K, du, ops.rowmap(M, 0, N), ops.OCC_BF16, ops.OCC_BF16, act=ops.ACT_MUL_AUX, aux=u, c_colsum=cs)
            res.append((o16, aux, du, cs))
    finally:
        lib().occ_gemm_variant(prev)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    if N % 8 == 0:                         # the same bf16 values summed over other row ranges: equal to f32 round-off
        torch.testing.assert_close(res[0][3], res[1][3], rtol=1e-5, atol=1e-2)
        assert float((res[0][3] - 0.5).abs().max()) > 0


def test_tail_split_matches_single_launch_exactly():
    """800 tiles on 256 CUs.  With a row-layout epilogue the heuristic takes 224-row tiles; without one (bf16 residual) it runs 48 row
    tiles in the eight-phase kernel and rows 12288.. through the small-tile kernels.  Integer operands make every form exact, so the
    results must equal the forced single launch bit for bit -- bf16 output with bias, GELU and the saved pre-activation, and f32
    output with a bf16 residual in a padded buffer."""
    from occm_amd import ops
    from occm_amd._lib import lib
    M, N, K = 12736, 4096, 1024
    x, w, bias = _ints(M, K, 21, -2, 2).bfloat16().cuda(), _ints(N, K, 22, -2, 2).bfloat16().cuda(), _ints(1, N, 23)[0].cuda()
    res = _ints(M, N + 8, 24).bfloat16().cuda()
    outs = []
    for variant in (30, 1):
        lib().occ_gemm_variant(variant)
        o16 = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16); aux = torch.zeros_like(o16)
        ops.gemm_raw(M, N, K, x, ops.rowmap(M, 0, K), w, K, o16, ops.rowmap(M, 0, N), ops.OCC_BF16, ops.OCC_BF16, bias=bias, act=ops.ACT_GELU, aux=aux, alpha=1.0 / 64)
        k1 = lib().occ_gemm_last_kernel()
        o32 = torch.zeros(M, N + 8, device="cuda")
        ops.gemm_raw(M, N, K, x, ops.rowmap(M, 0, K), w, K, o32, ops.rowmap(M, 0, N + 8), ops.OCC_F32, ops.OCC_BF16, R=res, r_map=ops.rowmap(M, 0, N + 8), r_dtype=ops.OCC_BF16)
        assert (k1, lib().occ_gemm_last_kernel()) == ((8, 8) if variant == 30 else (11, 10))
        outs.append((o16, aux, o32))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert float(outs[0][2][:, :N].abs().max()) > 100 and float(outs[0][2][:, N:].abs().max()) == 0


def test_p8_conv_windows_and_padded_output_rows():
    """Conv1d(k=3, s=2) as overlapping row windows (row stride s*C < K) into an output whose rows sit inside a padded buffer."""
    from occm_amd import ops
    B, Tin, C, k, s = 3, 701, 512, 3, 2
    Tout = (Tin - k) // s + 1
    x = _ints(B * Tin, C, 5, -2, 2)
    w = _ints(512, k * C, 6, -2, 2)
    xb = x.bfloat16().cuda()
    out = torch.zeros(B, Tout + 2, 512, device="cuda")
    ops.gemm_raw(B * Tout, 512, k * C, xb, ops.rowmap(Tout, Tin * C, s * C), w.bfloat16().cuda(), k * C,
                 out.data_ptr() + 512 * 4, ops.rowmap(Tout, (Tout + 2) * 512, 512), ops.OCC_F32, ops.OCC_BF16)
    win = x.view(B, Tin, C).unfold(1, k, s).permute(0, 1, 3, 2).reshape(B * Tout, k * C)      # [b, t, tap, c]
    ref = (win.double() @ w.double().T).view(B, Tout, 512)
    assert torch.equal(out[:, 1:-1].cpu().double(), ref)
    assert float(out[:, 0].abs().max()) == 0 and float(out[:, -1].abs().max()) == 0           # the pad rows were not touched


@pytest.mark.parametrize("gelu,res,cbf", [(True, False, True), (False, True, False), (True, True, True), (False, False, True)])
def test_p8_epilogues_random_data(gelu, res, cbf):
    from occm_amd import ops
    M, N, K = 1531, 1024, 512
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(M, K, generator=g) * 0.5).bfloat16(); w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(N, generator=g); r = torch.randn(M, N, generator=g)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16 if cbf else torch.float32)
    ops.gemm_raw(M, N, K, x.cuda(), ops.rowmap(M, 0, K), w.cuda(), K, out, ops.rowmap(M, 0, N), ops.OCC_BF16 if cbf else ops.OCC_F32, ops.OCC_BF16,
                 bias=b.cuda(), act=ops.ACT_GELU if gelu else ops.ACT_NONE, R=r.cuda() if res else None, r_map=ops.rowmap(M, 0, N), r_dtype=ops.OCC_F32)
    ref = x.float() @ w.float().T + b
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    if res:
        ref = ref + r
    torch.testing.assert_close(out.cpu().float(), ref, rtol=1e-2 if cbf else 1e-4, atol=2e-2 if cbf else 1e-4)


def test_p8_gelu_aux_and_gelu_grad():
    from occm_amd import ops
    M, N, K = 777, 512, 256
    g = torch.Generator().manual_seed(12)
    x = (torch.randn(M, K, generator=g) * 0.5).bfloat16(); w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(N, generator=g)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); aux = torch.empty_like(out)
    ops.gemm_raw(M, N, K, x.cuda(), ops.rowmap(M, 0, K), w.cuda(), K, out, ops.rowmap(M, 0, N), ops.OCC_BF16, ops.OCC_BF16, bias=b.cuda(), act=ops.ACT_GELU, aux=aux)
    pre = x.float() @ w.float().T + b
    torch.testing.assert_close(aux.cpu().float(), pre, rtol=1e-2, atol=2e-2)
    torch.testing.assert_close(out.cpu().float(), torch.nn.functional.gelu(pre), rtol=1e-2, atol=2e-2)
    dy = (torch.randn(M, K, generator=g)).bfloat16()
    du = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm_raw(M, N, K, dy.cuda(), ops.rowmap(M, 0, K), w.cuda(), K, du, ops.rowmap(M, 0, N), ops.OCC_BF16, ops.OCC_BF16, act=ops.ACT_GELU_GRAD, aux=aux)
    u = aux.cpu().float().requires_grad_(True)
    torch.nn.functional.gelu(u).backward(dy.float() @ w.float().T)
    torch.testing.assert_close(du.cpu().float(), u.grad, rtol=2e-2, atol=3e-2)


def test_p8_race_screen_bitwise_repeatable():
    """A fragment read ahead of its DMA (or a DMA over a buffer still being read) would show as run-to-run differences: 30 launches per
    shape, interleaved with launches of other shapes that change the memory timing, must be bit-identical."""
    from occm_amd import ops
    g = torch.Generator().manual_seed(13)
    shapes = [(6368, 1024, 4096), (12736, 3072, 1024), (4096, 4096, 4096), (1111, 512, 1536)]
    ops_ = []
    for M, N, K in shapes:
        x = (torch.randn(M, K, generator=g) * 0.5).bfloat16().cuda(); w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda()
        ops_.append((M, N, K, x, w, torch.empty(M, N, device="cuda")))
    first = {}
    for it in range(30):
        for i, (M, N, K, x, w, out) in enumerate(ops_):
            out.fill_(float("nan"))
            ops.gemm_raw(M, N, K, x, ops.rowmap(M, 0, K), w, K, out, ops.rowmap(M, 0, N), ops.OCC_F32, ops.OCC_BF16)
            if it == 0:
                first[i] = out.clone()
                ref = x.float() @ w.float().T
                torch.testing.assert_close(out, ref, rtol=2e-3, atol=2e-3 * float(ref.abs().max()))
            else:
                assert torch.equal(out, first[i]), (it, M, N, K)


@pytest.mark.parametrize("M,N,K", [(12736, 4096, 1024), (6368, 5120, 1280), (12100, 1024, 512)])
def test_gelu_grad_epilogue_also_gives_the_column_sums(M, N, K):
    """occ_gemm c_colsum: the bias gradient of fc1 (column sums of du = dY.W2 * GELU'(u)) from the epilogue that writes du, against
    occ_colsum over the bf16 result; accumulates onto what the buffer holds, twice gives twice (fixed summation order: bit-equal runs)."""
    from occm_amd import backend_ops as K_, ops
    from occm_amd._lib import ACT_GELU_GRAD, OCC_BF16
    g = torch.Generator().manual_seed(4)
    x = (torch.randn(M, K, generator=g) * 0.5).bfloat16().cuda(); w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda()
    u = torch.randn(M, N, generator=g).bfloat16().cuda()
    C0, C1 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16), torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    run = lambda C, **kw: ops.gemm_raw(M, N, K, x, ops.rowmap(M, 0, K), w, K, C, ops.rowmap(M, 0, N), OCC_BF16, OCC_BF16, act=ACT_GELU_GRAD, aux=u, **kw)
    run(C0)
    ref = torch.full((N,), 0.5, device="cuda")
    K_.colsum(C0, ops.rowmap(M, 0, N), M, N, ref)
    cs = torch.full((N,), 0.5, device="cuda")
    run(C1, c_colsum=cs)
    assert torch.equal(C1, C0)
    torch.testing.assert_close(cs, ref, rtol=1e-4, atol=1e-2)
    exact = 0.5 + C0.double().sum(0).float()
    torch.testing.assert_close(cs, exact, rtol=1e-4, atol=1e-2)
    cs2 = torch.full((N,), 0.5, device="cuda")
    run(C1, c_colsum=cs2)
    assert torch.equal(cs2, cs)


@pytest.mark.parametrize("force", [30, 0])                    # the eight-phase kernel's row epilogue, then whatever the dispatcher picks for the shape
@pytest.mark.parametrize("M,N,K", [(2048, 512, 256), (448, 260, 128), (12736, 4096, 1024)])
def test_gelu_keep_grad_and_mul_aux_epilogues(M, N, K, force):
    """OCC_ACT_GELU_KEEP_GRAD: C = gelu(x W^T + b), aux = bf16(gelu'(x W^T + b)) (one exponential for both); OCC_ACT_MUL_AUX: C = (dy W2) * aux.
    Against f64 torch on the same bf16 operands: both outputs are bf16 roundings of the exact values up to the 1.5e-7 of the erf approximation."""
    from occm_amd import ops
    from occm_amd._lib import lib
    if force == 0:
        lib().occ_gemm_variant(1)
    g = torch.Generator().manual_seed(M + N)
    x = (0.5 * torch.randn(M, K, generator=g)).bfloat16(); w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16(); b = 0.3 * torch.randn(N, generator=g)
    pre = (x.double() @ w.double().T + b.double())
    pre.requires_grad_(True)
    y = torch.nn.functional.gelu(pre)
    y.sum().backward()
    C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"); aux = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    cmap = ops.rowmap(M, 0, N)
    ops.gemm_raw(M, N, K, x.cuda(), ops.rowmap(M, 0, K), w.cuda(), K, C, cmap, ops.OCC_BF16, ops.OCC_BF16, bias=b.cuda(), act=ops.ACT_GELU_KEEP_GRAD, aux=aux)
    ey = (C.cpu().double() - y.detach()).abs(); eg = (aux.cpu().double() - pre.grad).abs()
    assert float((ey / (y.detach().abs() + 1e-2)).max()) < 6e-3 and float((eg / (pre.grad.abs() + 1e-2)).max()) < 6e-3, (float(ey.max()), float(eg.max()))
    # backward form: du = (dy W2^T-operand) * aux, with aux the tensor just written
    dy = (0.5 * torch.randn(M, K, generator=g)).bfloat16(); w2t = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16()
    du = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm_raw(M, N, K, dy.cuda(), ops.rowmap(M, 0, K), w2t.cuda(), K, du, cmap, ops.OCC_BF16, ops.OCC_BF16, act=ops.ACT_MUL_AUX, aux=aux)
    ref = (dy.double() @ w2t.double().T) * aux.cpu().double()
    err = (du.cpu().double() - ref).abs()
    assert float((err / (ref.abs() + 1e-2)).max()) < 6e-3, float(err.max())
