"""Host FLAC decoder of the input pipeline (occ_flac_info / occ_flac_decode through occm_amd.data_utils_SSL) against streams written by the
test-only encoder tests/flac_writer.py: bit-exact PCM for every subframe type and stereo mode, CRC / MD5 failures are reported.
No GPU needed (the decoder is host code inside libocc_hip.so).  libFLAC-written known answers: the RFC 9639 Appendix D example streams (last test)."""
import os

import numpy as np
import pytest

try:
    from tests import flac_writer as fw
except ImportError:                                   # run as plain `pytest`: tests/ itself is on sys.path
    import flac_writer as fw
from occm_amd import data_utils_SSL as du


def _speechlike(n, nch=1, seed=0, bps=16):
    g = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    x = np.stack([0.3 * np.sin(2 * np.pi * (220 + 40 * c) * t) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t)) + 0.02 * g.standard_normal(n) for c in range(nch)], 1)
    return np.round(x * (1 << (bps - 1)) * 0.9).astype(np.int64)


@pytest.mark.parametrize("kind", ["verbatim", "fixed0", "fixed1", "fixed2", "fixed3", "fixed4", "lpc"])
@pytest.mark.parametrize("method,porder", [(0, 0), (0, 3), (1, 2)])
def test_subframe_types_and_rice_partitions_roundtrip(kind, method, porder):
    pcm = _speechlike(9000, 1, seed=1)
    lpc = ([1876, -1012, 205, -61, 17, 9, -4, 2], 12, 10) if kind == "lpc" else None        # any coefficients give a valid stream
    raw = fw.encode(pcm, blocksize=4096, kinds=(kind,), method=method, porder=porder, lpc=lpc)
    out, fs, bps = du.decode_flac_bytes(raw)
    assert (fs, bps) == (16000, 16) and out.shape == (9000, 1)
    np.testing.assert_array_equal(out[:, 0], pcm[:, 0])


@pytest.mark.parametrize("stereo", ["independent", "left_side", "side_right", "mid_side"])
def test_stereo_decorrelation_modes(stereo):
    pcm = _speechlike(5000, 2, seed=2)
    pcm[:, 1] = pcm[:, 0] // 2 + pcm[:, 1] // 3                                # correlated channels, odd/even sums for the mid/side LSB
    raw = fw.encode(pcm, blocksize=1152, kinds=("fixed2", "fixed1"), stereo=stereo, porder=2)
    out, fs, bps = du.decode_flac_bytes(raw)
    np.testing.assert_array_equal(out, pcm)


def test_escape_partitions_constant_frames_wasted_bits_and_odd_block_sizes():
    pcm = _speechlike(3 * 700 + 123, 1, seed=3)
    raw = fw.encode(pcm, blocksize=700, kinds=("fixed2",), method=0, porder=0, force_escape=True)       # 16-bit explicit block size, raw partitions
    np.testing.assert_array_equal(du.decode_flac_bytes(raw)[0][:, 0], pcm[:, 0])
    raw = fw.encode(pcm[:600], blocksize=200, kinds=("fixed1",), variable=True)                           # 8-bit explicit size, sample-numbered frames
    np.testing.assert_array_equal(du.decode_flac_bytes(raw)[0][:, 0], pcm[:600, 0])
    z = np.zeros((5000, 1), dtype=np.int64); z[4096:] = -7
    raw = fw.encode(z, blocksize=4096, kinds=("constant",))                                               # digital silence, then a constant tail
    np.testing.assert_array_equal(du.decode_flac_bytes(raw)[0], z)
    w = (_speechlike(4096, 1, seed=4) >> 3) << 3                                                          # three wasted bits per sample
    raw = fw.encode(w, kinds=("fixed2+wasted",))
    np.testing.assert_array_equal(du.decode_flac_bytes(raw)[0], w)
    big = _speechlike(4096, 2, seed=5, bps=24)                                                            # 24-bit: 25-bit side channel
    raw = fw.encode(big, bps=24, kinds=("fixed2",), stereo="mid_side", method=1, porder=1)
    out, fs, bps = du.decode_flac_bytes(raw)
    assert bps == 24
    np.testing.assert_array_equal(out, big)


def test_corruption_is_reported_not_returned():
    pcm = _speechlike(6000, 1, seed=6)
    raw = bytearray(fw.encode(pcm, blocksize=1024, kinds=("fixed2",), porder=1))
    bad = bytearray(raw); bad[len(bad) // 2] ^= 0x10                                                      # a flipped bit inside a frame
    with pytest.raises(Exception) as e:
        du.decode_flac_bytes(bytes(bad))
    assert "CRC" in str(e.value) or "subframe" in str(e.value) or "sync" in str(e.value)
    wrong = bytearray(raw); wrong[8 + 18] ^= 0xff                                                         # first MD5 byte of STREAMINFO
    with pytest.raises(ValueError, match="MD5"):
        du.decode_flac_bytes(bytes(wrong))
    with pytest.raises(Exception, match="fLaC"):
        du.decode_flac_bytes(b"RIFF" + bytes(100))
    with pytest.raises(Exception):
        du.decode_flac_bytes(bytes(raw[:len(raw) // 2]))                                                  # truncated file: fewer samples than STREAMINFO says


def test_load_audio_reads_flac_like_wav(tmp_path):
    import wave
    pcm = _speechlike(16000, 1, seed=7)
    p = tmp_path / "LA_T_1000137.flac"
    p.write_bytes(fw.encode(pcm, blocksize=4096, kinds=("lpc", "fixed2"), porder=3, lpc=([1520, -640, 90], 12, 10)))
    x, fs = du.load_audio(str(p))
    assert fs == 16000 and x.dtype == np.float32 and x.shape == (16000,)
    wp = tmp_path / "same.wav"
    with wave.open(str(wp), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.astype("<i2").tobytes())
    y, fs2 = du.load_audio(str(wp))
    assert fs2 == 16000
    np.testing.assert_array_equal(x, y)
    st = _speechlike(3000, 2, seed=8)
    p2 = tmp_path / "stereo.flac"
    p2.write_bytes(fw.encode(st, blocksize=1024, stereo="left_side"))
    m, _ = du.load_audio(str(p2))
    np.testing.assert_allclose(m, st.mean(1) / 32768.0, atol=1e-7)


@pytest.mark.parametrize("k", [1, 2, 3])
def test_rfc9639_appendix_example_streams_known_answers(k):
    """The three example files of RFC 9639, Appendix D (written by reference libFLAC 1.3.3, as the vendor string inside example 2 says):
    tests/golden/flac_rfc9639_example{1,2,3}.flac are the RFC's byte listings -- 1: one stereo 16-bit verbatim frame with wasted bits;
    2: seek table + Vorbis comment + padding blocks, two frames, fixed predictors with partitioned Rice coding, side-channel stereo; 3: 8-bit
    mono LPC.  These are streams this repository's encoder (tests/flac_writer.py) did not write.  Known answers, none of them the decoder's
    own arithmetic: the MD5 signature in each STREAMINFO (computed by libFLAC over the PCM it encoded) re-computed here with hashlib, the
    sample counts of the STREAMINFO blocks and the decoded values the RFC tabulates (example 3's 24 samples; the first samples of 1 and 2)."""
    import hashlib
    from conftest import GOLDEN, golden
    raw = open(os.path.join(GOLDEN, "flac_rfc9639_example%d.flac" % k), "rb").read()
    out, fs, bps = du.decode_flac_bytes(raw)
    E = golden("flac_rfc9639_expected.npz")
    np.testing.assert_array_equal(out, E["pcm%d" % k])
    assert (fs, bps) == (int(E["fs%d" % k]), int(E["bps%d" % k]))
    # STREAMINFO: 4 bytes "fLaC", 4 bytes block header, then 18 bytes of fields followed by the 16-byte MD5 of the interleaved little-endian PCM
    si = raw[8:8 + 34]
    total = ((si[13] & 0x0f) << 32) | int.from_bytes(si[14:18], "big")
    assert total == out.shape[0] == {1: 1, 2: 19, 3: 24}[k]
    pcm = out.astype({8: "<i1", 16: "<i2"}[bps]).tobytes()
    assert hashlib.md5(pcm).digest() == si[18:34]
    if k == 1:
        assert out.tolist() == [[25588, 10416]]
    if k == 2:
        assert out[0].tolist() == [10372, 6070] and out[16].tolist() == [-15486, -9072]          # first sample of either frame (16 + 3 samples)
    if k == 3:
        assert out[:, 0].tolist() == [0, 79, 111, 78, 8, -61, -90, -68, -13, 42, 67, 53, 13, -27, -46, -38, -12, 14, 24, 19, 6, -4, -5, 0]
