"""numpy (float64) restatement of the reference's LFCC helper (TEST ORACLE) -- PARITY UNPINNED.

The reference's ``utils.extract_lfcc`` (utils.py:127-138) is one call into the third-party package ``spafe``:

    lfcc(y, fs=sr, pre_emph=1, pre_emph_coeff=0.97, window=SlidingWindow(0.03, 0.015, "hamming"),
         nfilts=128, nfft=2048, low_freq=0, high_freq=8000, normalize="mvn")

spafe is not under /root/reference, is pinned in neither requirement.txt nor environment.yml, and is not installed in this
container, and the reference holds no test or golden vector for it.  What follows restates the algorithm spafe 0.3.x publishes for
``spafe.features.lfcc.lfcc`` with its defaults (num_ceps=13, scale="constant", dct_type=2 with norm="ortho", no energy, no lifter):

    pre-emphasis  y[0] = x[0], y[n] = x[n] - 0.97 x[n-1]
    framing       480-sample frames every 240 samples; the signal is zero-padded by (hop - rest) samples when
                  rest = (len - (frame - hop)) % hop is not 0 (64000 samples -> 266 frames)
    window        numpy.hamming(480)
    spectrum      (1 / nfft) * |FFT_2048(frame)|^2, bins 0..1024
    filter bank   128 triangles with edges evenly spaced on [0, 8000] Hz (130 points), evaluated at the 1025 bin frequencies
                  linspace(0, fs/2, 1025); unit peak ("constant" scale)
    log           natural log, exact zeros replaced by float64 eps first
    DCT-II        orthonormal, over the 128 filter outputs; first 13 coefficients
    mvn           per coefficient over the frames of the utterance: (c - mean) / std (population std)

Nothing here is checked against spafe output; tests/ use it to check the HIP implementation of the same algorithm.
"""
import numpy as np

FRAME_LEN_S, HOP_S, PRE_EMPH = 0.03, 0.015, 0.97
NFILTS, NFFT, NUM_CEPS, LOW_HZ, HIGH_HZ = 128, 2048, 13, 0.0, 8000.0


def n_frames(n_samples, fs=16000):
    flen, hop = int(FRAME_LEN_S * fs), int(HOP_S * fs)
    rest = abs(n_samples - (flen - hop)) % hop
    padded = n_samples + (hop - rest if rest else 0)
    return (padded - flen) // hop + 1


def linear_filter_banks(fs=16000, nfilts=NFILTS, nfft=NFFT, low=LOW_HZ, high=HIGH_HZ):
    """[nfilts, nfft//2 + 1] triangles, unit peak."""
    edges = low + (high - low) / (nfilts + 1.0) * np.arange(nfilts + 2)
    freqs = np.linspace(0.0, fs / 2.0, nfft // 2 + 1)
    fb = np.zeros((nfilts, nfft // 2 + 1))
    for j in range(nfilts):
        lo, c, hi = edges[j], edges[j + 1], edges[j + 2]
        up = (freqs >= lo) & (freqs <= c)
        fb[j, up] = (freqs[up] - lo) / (c - lo)
        dn = (freqs >= c) & (freqs <= hi)
        fb[j, dn] = (hi - freqs[dn]) / (hi - c)
    return fb


def dct2_ortho(n_in, n_out):
    """[n_out, n_in] matrix of the orthonormal DCT-II (scipy.fftpack.dct(type=2, norm='ortho') along the last axis)."""
    k = np.arange(n_out)[:, None]
    n = np.arange(n_in)[None, :]
    m = np.cos(np.pi * k * (2 * n + 1) / (2.0 * n_in)) * np.sqrt(2.0 / n_in)
    m[0] *= np.sqrt(0.5)
    return m


def frames_of(sig, fs=16000):
    sig = np.asarray(sig, dtype=np.float64)
    pe = np.append(sig[0], sig[1:] - PRE_EMPH * sig[:-1])
    flen, hop = int(FRAME_LEN_S * fs), int(HOP_S * fs)
    rest = abs(len(pe) - (flen - hop)) % hop
    if rest:
        pe = np.append(pe, np.zeros(hop - rest))
    nf = (len(pe) - flen) // hop + 1
    idx = np.arange(flen)[None, :] + hop * np.arange(nf)[:, None]
    return pe[idx] * np.hamming(flen)[None, :]


def extract_lfcc(sig, fs=16000, normalize=True, return_stages=False):
    """utils.py:127-138 -> [n_frames, 13]."""
    fr = frames_of(sig, fs)
    power = (1.0 / NFFT) * np.abs(np.fft.rfft(fr, NFFT, axis=1)) ** 2
    feats = power @ linear_filter_banks(fs).T
    feats = np.where(feats == 0, np.finfo(float).eps, feats)
    logf = np.log(feats)
    ceps = logf @ dct2_ortho(NFILTS, NUM_CEPS).T
    with np.errstate(invalid="ignore", divide="ignore"):          # a one-frame utterance has zero variance: nan, as numpy gives spafe
        out = (ceps - ceps.mean(axis=0)) / ceps.std(axis=0) if normalize else ceps
    if return_stages:
        return out, dict(frames=fr, power=power, logfb=logf, ceps=ceps)
    return out
