"""Oracle (oracle/rawboost_np.py) vs vectors produced by the reference's RawBoost.py /
data_utils_SSL.py in the build container (oracle/gen_golden.py)."""
import numpy as np
import pytest

from conftest import golden, synth_wave
from oracle import rawboost_np as rb

G = golden("rawboost.npz")
L = 8000


@pytest.mark.parametrize("algo", list(range(9)))
@pytest.mark.parametrize("seed", [11, 12])
def test_process_rawboost_matches_reference(algo, seed):
    x = synth_wave(1000 + seed, L)
    if algo == 3 and seed == 12:
        x = x * 12.0
    np.random.seed(seed)
    y = rb.process_rawboost(x, 16000, rb.RawBoostArgs(), algo)
    ref = G["algo%d_seed%d" % (algo, seed)]
    assert y.shape == ref.shape and y.dtype == ref.dtype
    np.testing.assert_allclose(y, ref, rtol=1e-9, atol=1e-12)


def test_loud_input_takes_peak_normalisation():
    x = synth_wave(77, L) * 8.0
    np.random.seed(5)
    y = rb.process_rawboost(x, 16000, rb.RawBoostArgs(), 5)
    np.testing.assert_allclose(y, G["algo5_loud"], rtol=1e-9, atol=1e-12)
    assert np.max(np.abs(y)) <= 1.0 + 1e-12


@pytest.mark.parametrize("seed,gains", [(1, (0, 0)), (2, (0, 0)), (3, (0, 0)), (4, (-5, -20))])
def test_notch_coeffs(seed, gains):
    np.random.seed(seed)
    bands, Gd = rb.draw_notch_params(5, 20, 8000, 100, 1000, 10, 100, gains[0], gains[1])
    b = rb.notch_coeffs(bands, Gd, 16000)
    key = "notch_seed%d" % seed if gains == (0, 0) else "notch_gain_seed4"
    assert b.shape == G[key].shape
    np.testing.assert_allclose(b, G[key], rtol=1e-9, atol=1e-13)
    assert b.shape[0] % 2 == 1


@pytest.mark.parametrize("nt", [11, 101, 501])
def test_filter_fir(nt):
    b = None
    rs = np.random.RandomState(9)
    for n in (11, 101, 501):
        bb = rs.randn(n) / n
        if n == nt:
            b = bb
    y = rb.filter_fir(synth_wave(90 + nt, L), b)
    np.testing.assert_allclose(y, G["fir_%d" % nt], rtol=1e-9, atol=1e-13)
    assert y.shape[0] == L


def test_pad_tile():
    np.testing.assert_array_equal(rb.pad_tile(synth_wave(3, 1000), 2600), G["pad_short"])
    np.testing.assert_array_equal(rb.pad_tile(synth_wave(3, 3000), 2600), G["pad_long"])
