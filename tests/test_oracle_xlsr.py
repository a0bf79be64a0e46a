"""oracle/xlsr_ref.py vs HuggingFace Wav2Vec2Model (independent restatement of the same
architecture).  This does NOT pin equality with the reference's fairseq checkpoint path --
fairseq is absent from the reference tree and the image -- see oracle/xlsr_ref.py header."""
import numpy as np
import pytest
import torch

from oracle import xlsr_ref
from oracle.fill import fill_like

transformers = pytest.importorskip("transformers")


from oracle.hf_proxy import hf_model as _hf_model  # noqa: E402


@pytest.mark.parametrize("L,layers", [(16000, 2), (4000, 3)])
def test_xlsr_oracle_matches_hf_proxy(L, layers):
    cfg = xlsr_ref.XlsrConfig(dim=256, ffn=512, heads=4, layers=layers, pos_k=128, pos_groups=16)
    p = fill_like(xlsr_ref.param_shapes(cfg), seed=3)
    wav = 0.1 * torch.randn(2, L, generator=torch.Generator().manual_seed(5))
    taps = {}
    with torch.no_grad():
        mine = xlsr_ref.extract_feat(wav, p, cfg, taps)
        hf = _hf_model(cfg, p)
        out = hf(wav)
    assert mine.shape == (2, xlsr_ref.n_frames(L), cfg.dim)
    # HF's ``extract_features`` is the conv stack output AFTER feature_projection.layer_norm
    normed = torch.nn.functional.layer_norm(taps["conv"], (512,), p["layer_norm.weight"], p["layer_norm.bias"])
    np.testing.assert_allclose(normed.numpy(), out.extract_features.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(mine.numpy(), out.last_hidden_state.numpy(), rtol=1e-3, atol=2e-4)


def _compare_all_taps(cfg, L, B, seed, rtol, atol):
    """xlsr_ref vs the HF proxy at every intermediate tap: conv stack (after the projection LayerNorm), every transformer layer's
    output (HF hidden_states[i+1] of the stable-layer-norm encoder = the residual stream after layer i) and the final LayerNorm."""
    p = fill_like(xlsr_ref.param_shapes(cfg), seed=seed)
    wav = 0.1 * torch.randn(B, L, generator=torch.Generator().manual_seed(seed + 2))
    taps = {}
    with torch.no_grad():
        mine = xlsr_ref.extract_feat(wav, p, cfg, taps)
        hf = _hf_model(cfg, p)
        out = hf(wav, output_hidden_states=True)
    assert mine.shape == (B, xlsr_ref.n_frames(L), cfg.dim)
    normed = torch.nn.functional.layer_norm(taps["conv"], (512,), p["layer_norm.weight"], p["layer_norm.bias"])
    np.testing.assert_allclose(normed.numpy(), out.extract_features.numpy(), rtol=rtol, atol=atol)
    hs = out.hidden_states
    assert len(hs) == cfg.layers + 1
    worst = 0.0
    np.testing.assert_allclose(taps["pos"].numpy(), hs[0].numpy(), rtol=rtol, atol=atol)          # projection + positional conv
    for i in range(cfg.layers - 1):                  # HF applies the final LayerNorm to its last hidden state before returning it
        d = float((taps["layer%d" % i] - hs[i + 1]).abs().max()) / (float(hs[i + 1].abs().max()) + 1e-9)
        worst = max(worst, d)
    assert worst < 5e-5, worst                       # relative to each layer's largest activation (f32 summation-order noise only)
    np.testing.assert_allclose(mine.numpy(), out.last_hidden_state.numpy(), rtol=rtol, atol=atol)


def test_xlsr_oracle_matches_hf_proxy_at_300m_geometry_all_taps():
    """The real XLS-R-300M geometry (d 1024, 24 layers, 16 heads, ffn 4096): the whole depth, all taps, 1 s and 4 s of audio."""
    _compare_all_taps(xlsr_ref.XlsrConfig.xlsr_300m(), 16000, 1, seed=11, rtol=2e-3, atol=5e-4)
    _compare_all_taps(xlsr_ref.XlsrConfig.xlsr_300m(), 64000, 1, seed=12, rtol=2e-3, atol=5e-4)


def test_xlsr_oracle_matches_hf_proxy_at_1b_geometry():
    """XLS-R-1B geometry (d 1280, heads of 80, ffn 5120) at 4 of its 48 layers."""
    cfg = xlsr_ref.XlsrConfig(dim=1280, ffn=5120, heads=16, layers=4)
    _compare_all_taps(cfg, 16000, 2, seed=13, rtol=2e-3, atol=5e-4)


def test_flop_formula_matches_survey():
    f = xlsr_ref.flops_forward(64000, xlsr_ref.XlsrConfig.xlsr_300m())
    assert xlsr_ref.n_frames(64000) == 199 and xlsr_ref.n_frames(64600) == 201
    assert abs(f["total"] / 1e9 - 147.275) < 0.01 and abs(f["fe"] / 1e9 - 19.626) < 0.001
    assert abs(xlsr_ref.flops_forward(64000, xlsr_ref.XlsrConfig.xlsr_1b())["total"] / 1e9 - 410.462) < 0.01
