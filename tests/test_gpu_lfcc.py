"""LFCC front-end on HIP (occm_amd.utils.extract_lfcc, utils.py:127-138 of the reference) against the float64 restatement in
oracle/lfcc_ref.py.  Parity UNPINNED: the reference delegates to spafe, which is not available; both sides follow spafe's
published algorithm.  f32 DFT-by-GEMM against a float64 FFT: tolerances in the log / cepstral domain are stated per stage."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sig(n, seed):
    g = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    return (0.05 * g.standard_normal(n) + 0.3 * np.sin(2 * np.pi * (300 + 50 * seed) * t) * np.exp(-t)).astype(np.float32)


@pytest.mark.parametrize("n", [64000, 64600, 16000, 721, 480])
def test_lfcc_matches_oracle(n):
    from oracle import lfcc_ref
    from occm_amd import utils
    x = _sig(n, 3)
    ref_n, st = lfcc_ref.extract_lfcc(x.astype(np.float64), return_stages=True)
    raw = utils.extract_lfcc_batch(torch.from_numpy(x[None]).cuda(), 16000, normalize=False)[0].cpu().numpy()
    assert raw.shape == st["ceps"].shape == (utils.n_frames(n), 13)
    np.testing.assert_allclose(raw, st["ceps"], rtol=0, atol=2e-3)         # cepstra are O(10..100); f32 DFT + log
    if raw.shape[0] > 1:
        got = utils.extract_lfcc(x, 16000)                                   # the reference's call shape: numpy in, numpy out
        assert got.dtype == np.float64 and got.shape == ref_n.shape
        np.testing.assert_allclose(got, ref_n, rtol=0, atol=2e-3)            # unit-variance columns


def test_lfcc_batch_and_tensor_input_and_silence():
    from oracle import lfcc_ref
    from occm_amd import utils
    xs = np.stack([_sig(64000, s) for s in range(8)])                       # BASELINE configs[0] batch: 8 utterances of 4 s
    out = utils.extract_lfcc_batch(torch.from_numpy(xs).cuda())
    assert out.shape == (8, 266, 13)
    for b in (0, 5, 7):
        np.testing.assert_allclose(out[b].cpu().numpy(), lfcc_ref.extract_lfcc(xs[b].astype(np.float64)), atol=2e-3)
    t = utils.extract_lfcc(torch.from_numpy(xs[2]), 16000)                   # CPU tensor in -> CPU tensor out
    assert torch.is_tensor(t) and t.device.type == "cpu" and torch.allclose(t, out[2].cpu(), atol=1e-6)
    # digital silence: every filter output is exactly zero -> log(eps) everywhere, as the restatement gives before normalisation
    z = utils.extract_lfcc_batch(torch.zeros(1, 4000, device="cuda"), normalize=False)[0].cpu().numpy()
    zr = lfcc_ref.extract_lfcc(np.zeros(4000), normalize=False)
    np.testing.assert_allclose(z, zr, rtol=1e-5, atol=1e-3)
    with pytest.raises(ValueError):
        utils.extract_lfcc_batch(torch.zeros(1, 4000))                       # host tensor: no CPU fallback


def test_config0_composition_lfcc_into_se_resnet34():
    """BASELINE configs[0]: waveform [8, 64000] -> LFCC [8, 1, 266, 13] -> se_resnet34 -> (com [8,128], des [8,2]); forward parity of the
    composed path against the oracle pieces (SE-ResNet34 itself is pinned by tests/golden, the LFCC stage is not)."""
    from oracle import lfcc_ref, senet_ref
    from occm_amd import utils
    from occm_amd.models import senet
    xs = np.stack([_sig(64000, 10 + s) for s in range(8)])
    feats = utils.extract_lfcc_batch(torch.from_numpy(xs).cuda()).unsqueeze(1)             # [8,1,266,13]
    params = senet.synthetic_senet_params(seed=0)
    model = senet.se_resnet34(state_dict=params, device="cuda")
    model.eval()
    com, des = model(feats)
    assert com.shape == (8, 128) and des.shape == (8, 2)
    ref_feats = torch.from_numpy(np.stack([lfcc_ref.extract_lfcc(x.astype(np.float64)) for x in xs])).float().unsqueeze(1)
    rcom, rdes = senet_ref.senet34_forward(ref_feats, {k: v.detach().cpu().float() for k, v in model.state_dict().items()}, train=False)
    torch.testing.assert_close(com.cpu(), rcom, rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(des.cpu(), rdes, rtol=2e-3, atol=2e-3)


def test_config0_training_steps_lfcc_se_resnet34():
    """BASELINE configs[0] as a training step on the GPU: bs 8, LFCC -> SE-ResNet34, loss 0.1*compactness + 0.9*descriptiveness
    (test_dataloader_v2.py:127), Adam.  Step-0 losses equal the oracle's on the same weights and batch; the fixed batch is then fitted."""
    from oracle import lfcc_ref, losses_ref, senet_ref
    from occm_amd.models import senet
    from occm_amd.trainer import OcTrainer
    xs = np.stack([_sig(64000, 20 + s) for s in range(8)])
    labels = torch.tensor([0, 0, 0, 0, 0, 0, 1, 1])
    model = senet.lfcc_resnet34("cuda", seed=3)
    model.train()
    sd0 = {k: v.detach().cpu().float().clone() for k, v in model.resnet34.state_dict().items()}
    tr = OcTrainer(model, lr=1e-3, w_compact=0.1, w_descr=0.9)
    wav = torch.from_numpy(xs).cuda()
    losses = []
    for _ in range(6):
        lc, ld = tr.step(wav, labels.cuda())
        losses.append((float(lc), float(ld)))
    feats = torch.from_numpy(np.stack([lfcc_ref.extract_lfcc(x.astype(np.float64)) for x in xs])).float().unsqueeze(1)
    com, des = senet_ref.senet34_forward(feats, sd0, train=True)
    rc, rd = float(losses_ref.compactness_loss(com)), float(losses_ref.descriptiveness_loss(des, labels))
    # OcTrainer.step returns the unweighted terms (the weights only scale the gradients)
    assert abs(losses[0][0] - rc) < 1e-3 * max(1.0, abs(rc)) and abs(losses[0][1] - rd) < 1e-3, (losses[0], rc, rd)
    assert all(np.isfinite(v) for l in losses for v in l)
    assert losses[-1][1] < losses[0][1]                                      # the descriptiveness term of the fixed batch goes down
