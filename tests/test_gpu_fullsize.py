"""Size-independent properties at BASELINE.json's full sizes (XLS-R-300M, bs 32, 64000 samples), where the CPU oracle is too slow
to be the checker: per-utterance independence and permutation equivariance of the front-end, the same for the back-end in eval
mode, conservation laws of a training step.  All through the C ABI (the same objects bench.py drives)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _wav(B, L=64000, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (0.1 * torch.randn(B, L, generator=g)).clamp_(-1, 1).cuda()


@pytest.fixture(scope="module")
def fe300():
    from occm_amd.models import xlsr
    cfg = xlsr.XlsrConfig.xlsr_300m()
    return xlsr.XlsrFrontend(xlsr.synthetic_params(cfg, 0), cfg, dtype=torch.bfloat16)


def test_frontend_300m_utterances_are_independent_and_order_equivariant(fe300):
    wav = _wav(8)
    full = fe300.forward(wav, out_dtype=torch.float32).clone()
    assert full.shape == (8, 199, 1024) and bool(torch.isfinite(full).all())
    one = fe300.forward(wav[3:4], out_dtype=torch.float32).clone()
    # A different batch size can select a different GEMM kernel (under-filled launches split K inside the workgroup), i.e. another
    # summation order: equal to bf16 round-off carried through 24 layers, not bit-equal.  Same shape = same kernels = bit-equal (below).
    d = (one[0] - full[3]).abs()
    assert float(d.max()) < 8e-2 and float(d.mean()) < 4e-3, (float(d.max()), float(d.mean()))
    again = fe300.forward(wav[3:4], out_dtype=torch.float32)
    torch.testing.assert_close(again, one, rtol=0, atol=0)                 # run-to-run: bit-identical (no atomics on this path)
    perm = torch.tensor([5, 2, 7, 0, 1, 6, 3, 4], device="cuda")
    shuf = fe300.forward(wav[perm].contiguous(), out_dtype=torch.float32)
    torch.testing.assert_close(shuf, full[perm], rtol=0, atol=0)
    # LayerNorm is the last op of the encoder: every frame has ~zero mean / unit variance over the 1024 channels (affine = 1/0 + noise)
    assert float(full.mean(-1).abs().max()) < 0.2 and 0.5 < float(full.std(-1).mean()) < 1.5


def test_frontend_300m_batch32_matches_two_halves(fe300):
    wav = _wav(32, seed=1)
    full = fe300.forward(wav, out_dtype=torch.float32).clone()
    lo = fe300.forward(wav[:16].contiguous(), out_dtype=torch.float32).clone()
    hi = fe300.forward(wav[16:].contiguous(), out_dtype=torch.float32)
    d = (torch.cat([lo, hi]) - full).abs()                                 # bs 16 and bs 32 may take different GEMM kernels: round-off level, see above
    assert float(d.max()) < 8e-2 and float(d.mean()) < 4e-3, (float(d.max()), float(d.mean()))


def test_backend_bs32_eval_is_order_equivariant():
    from occm_amd.models.sslassist import AasistBackend
    be = AasistBackend(device="cuda", seed=0, compute="bf16")
    feats = torch.randn(32, 199, 1024, generator=torch.Generator().manual_seed(2)).cuda()
    emb, out = be.forward(feats, train=False)
    emb, out = emb.clone(), out.clone()
    perm = torch.randperm(32, generator=torch.Generator().manual_seed(3)).cuda()
    emb2, out2 = be.forward(feats[perm].contiguous(), train=False)
    torch.testing.assert_close(emb2, emb[perm], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(out2, out[perm], rtol=1e-5, atol=1e-5)


def test_training_step_bs32_conservation():
    """One bench-sized step: CE gradient rows sum to zero over the two classes, so the out_layer bias gradient sums to zero;
    Adam moves every back-end parameter that has a gradient by at most lr * (1 + eps slack)."""
    from occm_amd.models import xlsr
    from occm_amd.models.sslassist import AModel
    from occm_amd.trainer import OcTrainer
    cfg = xlsr.XlsrConfig(dim=1024, ffn=1024, heads=16, layers=2)       # full-size back-end and batch, short encoder (front-end covered above)
    model = AModel(None, "cuda", ssl_cfg=cfg, seed=0, synthetic_ssl=True)
    model.train()
    lr = 1e-3
    tr = OcTrainer(model, lr=lr, w_compact=0.0, w_descr=1.0)
    before = model.backend.P.clone()
    labels = torch.tensor(([0] * 6 + [1] * 6) * 3, device="cuda")[:32]
    lc, ld = tr.step(_wav(32, seed=4), labels)
    assert bool(torch.isfinite(ld)) and float(ld) > 0
    g = model.backend.grad_dict()
    assert abs(float(g["out_layer.bias"].sum())) < 1e-5
    delta = (model.backend.P - before).abs()
    assert float(delta.max()) <= lr * 1.001 and float(delta.max()) > 0.5 * lr
