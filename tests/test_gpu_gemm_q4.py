"""The four-wave 256x128 bf16 GEMM with two workgroups per CU (csrc/gemm_q4.hip) through occ_gemm (variants 40 / 41): exact-integer
checks of the tile / fragment / DMA-ring addressing (small-integer bf16 operands make every f32 sum exact, so a misplaced row, chunk,
k-step or ring slot shows as an integer difference) over every K-tile count modulo the five-slot ring, ragged edges, conv windows through
row maps, every row-epilogue form bit for bit against the eight-phase kernel, and a run-to-run race screen."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _force_q4():
    from occm_amd._lib import lib
    prev = lib().occ_gemm_variant(40)
    yield
    lib().occ_gemm_variant(prev)


def _ints(rows, cols, seed, lo=-3, hi=3):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, (rows, cols), generator=g).float()


@pytest.mark.parametrize("variant", [40, 41])
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (256, 256, 128), (512, 256, 192), (300, 256, 256), (300, 384, 320), (1000, 512, 384), (449, 256, 448), (640, 256, 512),
                                   (257, 768, 1536), (1000, 512, 1024), (12736, 1024, 1024), (6368, 4096, 1024), (2049, 260, 704)])
def test_q4_exact_integer_products(M, N, K, variant):
    from occm_amd import ops
    from occm_amd._lib import lib
    lib().occ_gemm_variant(variant)
    x, w = _ints(M, K, 1), _ints(N, K, 2)                 # asymmetric operands: a swapped row/column map cannot cancel
    bias = _ints(1, N, 3)[0]
    ref = x.double() @ w.double().T + bias.double()
    out = torch.full((M, N), 7777.0, device="cuda")
    ops.gemm_raw(M, N, K, x.bfloat16().cuda(), ops.rowmap(M, 0, K), w.bfloat16().cuda(), K, out, ops.rowmap(M, 0, N), ops.OCC_F32, ops.OCC_BF16, bias=bias.cuda())
    assert lib().occ_gemm_last_kernel() == 12              # OCC_GEMM_KERNEL_Q4
    assert torch.equal(out.cpu().double(), ref), (M, N, K, float((out.cpu().double() - ref).abs().max()))


@pytest.mark.parametrize("M,N,K", [(448, 256, 128), (12736, 4096, 1024), (2049, 264, 448)])
def test_q4_row_epilogues_equal_the_eight_phase_kernel_bit_for_bit(M, N, K):
    """Same wave tile, same accumulation order per output element, same epilogue code: every form of the row epilogue the training step
    uses (bf16 + bias; GELU + saved pre-activation; gelu / gelu' pair; multiply by the side tensor + column sums; f32 + f32 residual) must
    reproduce the eight-phase kernel's bits on random data, for 256- and 224-row tiles."""
    from occm_amd import ops
    from occm_amd._lib import lib
    g = torch.Generator().manual_seed(M)
    x = (torch.randn(M, K, generator=g) * 0.5).bfloat16().cuda(); w = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda()
    b = torch.randn(N, generator=g).cuda(); r = torch.randn(M, N, generator=g).cuda(); side = torch.randn(M, N, generator=g).bfloat16().cuda()
    xm, cm = ops.rowmap(M, 0, K), ops.rowmap(M, 0, N)

    def run(variant):
        lib().occ_gemm_variant(variant)
        o = {}
        o["bf16"] = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        ops.gemm_raw(M, N, K, x, xm, w, K, o["bf16"], cm, ops.OCC_BF16, ops.OCC_BF16, bias=b)
        o["gelu"], o["pre"] = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16), torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        ops.gemm_raw(M, N, K, x, xm, w, K, o["gelu"], cm, ops.OCC_BF16, ops.OCC_BF16, bias=b, act=ops.ACT_GELU, aux=o["pre"])
        o["g2"], o["dg"] = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16), torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        ops.gemm_raw(M, N, K, x, xm, w, K, o["g2"], cm, ops.OCC_BF16, ops.OCC_BF16, bias=b, act=ops.ACT_GELU_KEEP_GRAD, aux=o["dg"])
        o["mul"] = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        o["cs"] = torch.zeros(N, device="cuda")
        ops.gemm_raw(M, N, K, x, xm, w, K, o["mul"], cm, ops.OCC_BF16, ops.OCC_BF16, act=ops.ACT_MUL_AUX, aux=side, **({"c_colsum": o["cs"]} if N % 8 == 0 else {}))
        o["res"] = torch.zeros(M, N, device="cuda")
        ops.gemm_raw(M, N, K, x, xm, w, K, o["res"], cm, ops.OCC_F32, ops.OCC_BF16, bias=b, R=r, r_map=cm, r_dtype=ops.OCC_F32)
        return o

    for variant, same_rows in ((40, 30), (41, 31)):           # (the column sums add per-tile partial rows in tile order: compare equal row tilings)
        ref = run(same_rows)
        got = run(variant)
        assert lib().occ_gemm_last_kernel() == 12
        for k in ref:
            assert torch.equal(got[k], ref[k]), (variant, k, float((got[k].float() - ref[k].float()).abs().max()))


def test_q4_conv_windows_and_padded_output_rows():
    """Conv1d(k=3, s=2) as overlapping row windows (row stride s*C < K) into an output whose rows sit inside a padded buffer."""
    from occm_amd import ops
    B, Tin, C, k, s = 3, 701, 512, 3, 2
    Tout = (Tin - k) // s + 1
    x = _ints(B * Tin, C, 5, -2, 2)
    w = _ints(512, k * C, 6, -2, 2)
    out = torch.zeros(B, Tout + 2, 512, device="cuda")
    ops.gemm_raw(B * Tout, 512, k * C, x.bfloat16().cuda(), ops.rowmap(Tout, Tin * C, s * C), w.bfloat16().cuda(), k * C,
                 out.data_ptr() + 512 * 4, ops.rowmap(Tout, (Tout + 2) * 512, 512), ops.OCC_F32, ops.OCC_BF16)
    win = x.view(B, Tin, C).unfold(1, k, s).permute(0, 1, 3, 2).reshape(B * Tout, k * C)
    ref = (win.double() @ w.double().T).view(B, Tout, 512)
    assert torch.equal(out[:, 1:-1].cpu().double(), ref)
    assert float(out[:, 0].abs().max()) == 0 and float(out[:, -1].abs().max()) == 0


@pytest.mark.parametrize("variant", [40, 41])
def test_q4_race_screen_bitwise_repeatable(variant):
    """A fragment read ahead of its DMA (or a DMA into a ring slot still being read) would show as run-to-run differences: 30 launches per
    shape, two workgroups per CU contending for the same memory system, interleaved with other shapes, must be bit-identical."""
    from occm_amd import ops
    from occm_amd._lib import lib
    lib().occ_gemm_variant(variant)
    g = torch.Generator().manual_seed(13)
    shapes = [(6368, 1024, 4096), (12736, 3072, 1024), (4096, 4096, 4096), (1111, 512, 1536), (12736, 1024, 448)]
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
