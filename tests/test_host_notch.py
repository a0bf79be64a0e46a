"""Host-only C++ filter design (occ_notch_coeffs_host, no GPU) vs the reference's genNotchCoeffs vectors."""
import numpy as np
import pytest

from conftest import golden
from occm_amd import ops
from oracle import rawboost_np as rb

G = golden("rawboost.npz")


@pytest.mark.parametrize("seed,gains,key", [(1, (0, 0), "notch_seed1"), (2, (0, 0), "notch_seed2"), (3, (0, 0), "notch_seed3"),
                                            (4, (-5, -20), "notch_gain_seed4")])
def test_notch_host_matches_reference(seed, gains, key):
    np.random.seed(seed)
    bands, Gd = rb.draw_notch_params(5, 20, 8000, 100, 1000, 10, 100, gains[0], gains[1])
    coef, nt = ops.notch_coeffs_host(bands, Gd, 16000, 512)
    assert nt == G[key].shape[0]
    np.testing.assert_allclose(coef[:nt], G[key], rtol=1e-9, atol=1e-13)
    assert np.all(coef[nt:] == 0)


def test_dropin_genNotchCoeffs_consumes_the_same_random_stream():
    from occm_amd import RawBoost as RB
    np.random.seed(2)
    b = RB.genNotchCoeffs(5, 20, 8000, 100, 1000, 10, 100, 0, 0, 16000)
    after = np.random.uniform()
    np.testing.assert_allclose(b, G["notch_seed2"], rtol=1e-9, atol=1e-13)
    np.random.seed(2)
    rb.draw_notch_params(5, 20, 8000, 100, 1000, 10, 100, 0, 0)
    assert after == np.random.uniform()


def test_too_many_taps_is_an_error():
    from occm_amd._lib import OccError
    with pytest.raises(OccError):
        ops.notch_coeffs_host([(4000.0, 500.0, 101)] * 5, 0.0, 16000, 128)
