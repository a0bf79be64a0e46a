"""The LFCC restatement (oracle/lfcc_ref.py, parity unpinned: spafe is not available) is at least internally what it says it is."""
import numpy as np
import scipy.fft

from oracle import lfcc_ref


def test_frame_count_and_padding_rule():
    assert lfcc_ref.n_frames(64000) == 266            # [8, 1, 266, 13]
    assert lfcc_ref.n_frames(64600) == 269
    assert lfcc_ref.n_frames(480) == 1 and lfcc_ref.n_frames(720) == 2 and lfcc_ref.n_frames(721) == 3
    x = np.random.default_rng(0).standard_normal(1000)
    fr = lfcc_ref.frames_of(x)
    assert fr.shape == (lfcc_ref.n_frames(1000), 480)
    pe = np.append(x[0], x[1:] - 0.97 * x[:-1])
    np.testing.assert_allclose(fr[1], pe[240:720] * np.hamming(480))
    assert np.all(fr[-1][1000 - 240 * (fr.shape[0] - 1):] == 0)           # zero padding past the signal


def test_dct_matrix_is_scipy_ortho_dct2():
    x = np.random.default_rng(1).standard_normal((5, 128))
    np.testing.assert_allclose(x @ lfcc_ref.dct2_ortho(128, 13).T, scipy.fft.dct(x, type=2, norm="ortho", axis=1)[:, :13], atol=1e-12)


def test_filter_bank_shape_and_partition():
    fb = lfcc_ref.linear_filter_banks()
    assert fb.shape == (128, 1025) and fb.min() >= 0 and abs(fb.max() - 1.0) < 0.02
    # neighbouring triangles overlap by half: between the first and last centre the bank sums to one
    s = fb.sum(0)
    f = np.linspace(0, 8000, 1025)
    inner = (f > 8000 / 129) & (f < 8000 * 128 / 129)
    np.testing.assert_allclose(s[inner], 1.0, atol=1e-9)


def test_lfcc_of_a_tone_and_of_noise():
    t = np.arange(64000) / 16000.0
    tone = 0.5 * np.sin(2 * np.pi * 1000.0 * t)
    out, st = lfcc_ref.extract_lfcc(tone, return_stages=True)
    assert out.shape == (266, 13) and np.all(np.isfinite(out))
    k = int(np.argmax(st["power"][10]))
    assert abs(k * 16000 / 2048 - 1000.0) < 8.0                            # spectral peak at the tone
    noise = np.random.default_rng(2).standard_normal(64000) * 0.1
    o = lfcc_ref.extract_lfcc(noise)
    np.testing.assert_allclose(o.mean(0), 0.0, atol=1e-9)
    np.testing.assert_allclose(o.std(0), 1.0, atol=1e-9)
