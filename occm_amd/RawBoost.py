"""RawBoost augmentation on MI355X -- drop-in for the reference's ``RawBoost.py`` (:14-97).

Same function names and argument order as the reference.  Random *parameters* (band centres, widths,
tap counts, gains, ISD positions, SSI noise and SNR) are drawn on the host from the legacy global
``np.random`` stream in exactly the reference's order, so ``np.random.seed(s)`` reproduces the
reference's augmentation; the filter design runs in the library's host C++ (occ_notch_coeffs_host) and
all per-sample work (5-branch FIR bank, mean/peak normalisation, scatter, noise mixing) runs in HIP
kernels.  Inputs may be numpy arrays (a float64/float32 numpy array comes back, like the reference) or
CUDA tensors (a CUDA tensor comes back, nothing leaves the GPU).  ``*_batch`` variants take ``[B,L]``.
"""
import numpy as np
import torch

from . import ops
from ._lib import require_gpu

MAX_TAPS = 512          # 5 bands x <=101 taps - 4 = 501 (maxCoeff default 100, RawBoost.py:33-36)


def randRange(x1, x2, integer):
    """RawBoost.py:14-18."""
    y = np.random.uniform(low=x1, high=x2, size=(1,))
    if integer:
        y = int(y[0])
    return y


def _draw_notch(nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG):
    bands = []
    for _ in range(nBands):
        fc = float(randRange(minF, maxF, 0)[0])
        bw = float(randRange(minBW, maxBW, 0)[0])
        c = randRange(minCoeff, maxCoeff, 1)
        bands.append((fc, bw, c))
    G = float(randRange(minG, maxG, 0)[0])
    return bands, G


def genNotchCoeffs(nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG, fs):
    """RawBoost.py:28-48 (host C++ filter design, numpy draws)."""
    bands, G = _draw_notch(nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG)
    coef, nt = ops.notch_coeffs_host(bands, G, fs, max(MAX_TAPS, 5 * (int(maxCoeff) + 1)))
    return coef[:nt]


def normWav(x, always):
    """RawBoost.py:20-25 (host helper kept for API parity; the device path fuses it)."""
    peak = np.amax(np.abs(x))
    if always or peak > 1:
        x = x / peak
    return x


def _to_dev(x):
    require_gpu()
    if isinstance(x, torch.Tensor):
        t = x if x.is_cuda else x.cuda()
        was_np = False
    else:
        t = torch.from_numpy(np.ascontiguousarray(x)).cuda()
        was_np = True
    squeeze = t.dim() == 1
    if squeeze:
        t = t.unsqueeze(0)
    return t.contiguous(), was_np, squeeze


def _from_dev(y, was_np, squeeze, np_dtype=None):
    if squeeze:
        y = y[0]
    if was_np:
        a = y.cpu().numpy()
        return a.astype(np_dtype) if np_dtype is not None else a
    return y


def _coef_bank(B, n_filt, draw, fs, max_taps):
    """Host: draw + design n_filt filters for each of B utterances -> (coef[B,F,max_taps] f64, ntaps[B,F])."""
    coef = np.zeros((B, n_filt, max_taps), dtype=np.float64)
    ntaps = np.zeros((B, n_filt), dtype=np.int32)
    for b in range(B):
        for f, (bands, G) in enumerate(draw(b)):
            c, nt = ops.notch_coeffs_host(bands, G, fs, max_taps)
            coef[b, f] = c
            ntaps[b, f] = nt
    return torch.from_numpy(coef).cuda(), torch.from_numpy(ntaps).cuda()


def _lnl_dev(x, N_f, nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG, minB, maxB, fs):
    B = x.shape[0]
    max_taps = max(MAX_TAPS, nBands * (int(maxCoeff) + 1))

    def draw(_b):
        g0, g1 = minG, maxG
        sets = []
        for i in range(N_f):
            if i == 1:                       # RawBoost.py:62-64: lowered once, stays lowered
                g0 = g0 - minB
                g1 = g1 - maxB
            sets.append(_draw_notch(nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, g0, g1))
        return sets

    coef, ntaps = _coef_bank(B, N_f, draw, fs, max_taps)
    y = ops.rawboost_fir_bank(x, coef, ntaps, powers=True)
    return ops.rawboost_center_norm(y, subtract_mean=True, norm_mode=1)


def _isd_dev(y64, P, g_sd):
    B, L = y64.shape
    pos = np.zeros((B, max(int(L * P / 100) + 1, 1)), dtype=np.int32)
    fr = np.zeros(pos.shape, dtype=np.float64)
    n = np.zeros(B, dtype=np.int32)
    for b in range(B):
        beta = float(randRange(0, P, 0)[0])
        nb = int(L * (beta / 100))
        p = np.random.permutation(L)[:nb]
        f_r = np.multiply(((2 * np.random.rand(p.shape[0])) - 1), ((2 * np.random.rand(p.shape[0])) - 1))
        pos[b, :nb] = p
        fr[b, :nb] = f_r
        n[b] = nb
    ops.rawboost_isd_scatter(y64, torch.from_numpy(pos).cuda(), torch.from_numpy(fr).cuda(), torch.from_numpy(n).cuda(), g_sd)
    return ops.rawboost_center_norm(y64, subtract_mean=False, norm_mode=1)


def _ssi_dev(x64, SNRmin, SNRmax, nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG, fs):
    B, L = x64.shape
    max_taps = max(MAX_TAPS, nBands * (int(maxCoeff) + 1))
    noise = np.zeros((B, L), dtype=np.float64)
    coef = np.zeros((B, 1, max_taps), dtype=np.float64)
    ntaps = np.zeros((B, 1), dtype=np.int32)
    snr = np.zeros(B, dtype=np.float64)
    for b in range(B):                       # RawBoost.py:90-94 draw order: noise, filter, SNR
        noise[b] = np.random.normal(0, 1, L)
        bands, G = _draw_notch(nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG)
        coef[b, 0], ntaps[b, 0] = ops.notch_coeffs_host(bands, G, fs, max_taps)
        snr[b] = float(randRange(SNRmin, SNRmax, 0)[0])
    nz = ops.rawboost_fir_bank(torch.from_numpy(noise).cuda(), torch.from_numpy(coef).cuda(), torch.from_numpy(ntaps).cuda(), powers=False)
    nz = ops.rawboost_center_norm(nz, subtract_mean=False, norm_mode=2)
    return ops.rawboost_ssi_mix(x64, nz, torch.from_numpy(snr).cuda())


def _as64(t):
    return t if t.dtype == torch.float64 else ops.cast(t, torch.float64)


def LnL_convolutive_noise(x, N_f, nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG,
                          minBiasLinNonLin, maxBiasLinNonLin, fs):
    """RawBoost.py:59-69."""
    t, was_np, sq = _to_dev(x)
    y = _lnl_dev(t, N_f, nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG, minBiasLinNonLin, maxBiasLinNonLin, fs)
    return _from_dev(y, was_np, sq)


def ISD_additive_noise(x, P, g_sd):
    """RawBoost.py:73-84 (dtype of the input is kept, as in the reference)."""
    t, was_np, sq = _to_dev(x)
    in_dtype = t.dtype
    y = _isd_dev(_as64(t).clone() if t.dtype == torch.float64 else _as64(t), P, g_sd)
    if in_dtype != torch.float64:
        y = ops.cast(y, in_dtype)
    return _from_dev(y, was_np, sq)


def SSI_additive_noise(x, SNRmin, SNRmax, nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG, fs):
    """RawBoost.py:89-97."""
    t, was_np, sq = _to_dev(x)
    y = _ssi_dev(_as64(t), SNRmin, SNRmax, nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG, fs)
    return _from_dev(y, was_np, sq)


# ------------------------------------------------------------------------------------------------------------------------
# Device mode: everything per-sample happens on the GPU, including the filter design and the random draws that scale with
# the signal length (ISD positions, SSI noise).  Only the O(100) scalar parameters per utterance come from a host
# ``np.random.Generator`` (vectorised).  Same distributions as the reference, different random stream -- use the functions
# above (legacy np.random order) when bit-compatibility with the reference's augmentation is wanted.
def _design_bank_device(rng, B, n_filt, a, fs, gain_lo, gain_hi, max_taps):
    """gain_lo / gain_hi: length-n_filt sequences (the LnL bias lowers the gain range for the non-linear branches)."""
    from ._lib import check, lib, ptr, stream_ptr
    fc = rng.uniform(a.minF, a.maxF, size=(B, n_filt, a.nBands))
    bw = rng.uniform(a.minBW, a.maxBW, size=(B, n_filt, a.nBands))
    c = rng.uniform(a.minCoeff, a.maxCoeff, size=(B, n_filt, a.nBands)).astype(np.int32)
    G = np.stack([rng.uniform(gain_lo[i], gain_hi[i], size=B) if gain_hi[i] > gain_lo[i] else np.full(B, float(gain_lo[i])) for i in range(n_filt)], axis=1)
    dev = lambda t: torch.from_numpy(np.ascontiguousarray(t)).cuda()
    fc_d, bw_d, c_d, g_d = dev(fc), dev(bw), dev(c), dev(G)
    coef = torch.empty(B, n_filt, max_taps, device="cuda", dtype=torch.float64)
    ntaps = torch.empty(B, n_filt, device="cuda", dtype=torch.int32)
    check(lib().occ_notch_coeffs(ptr(fc_d), ptr(bw_d), ptr(c_d), ptr(g_d), B * n_filt, a.nBands, float(fs), ptr(coef), ptr(ntaps), max_taps, stream_ptr()),
          "occ_notch_coeffs")
    return coef, ntaps


def rawboost_batch_device(x, args, algo, rng=None, seed=0, step=0, fs=16000):
    """process_Rawboost_feature (data_utils_SSL.py:111-173) for a CUDA batch x f32 [B,L] -> f32 [B,L], randomness on the device."""
    from ._lib import check, lib, ptr, stream_ptr
    require_gpu()
    if algo not in (1, 2, 3, 4, 5, 6, 7, 8):
        return x
    a = args
    rng = rng or np.random.default_rng(seed * 1000003 + step)
    B, L = x.shape
    max_taps = max(MAX_TAPS, a.nBands * (int(a.maxCoeff) + 1))
    sid = [(step << 4) + 1]

    def lnl(v):
        lo = [a.minG] + [a.minG - a.minBiasLinNonLin] * (a.N_f - 1)             # RawBoost.py:62-64
        hi = [a.maxG] + [a.maxG - a.maxBiasLinNonLin] * (a.N_f - 1)
        coef, ntaps = _design_bank_device(rng, B, a.N_f, a, fs, lo, hi, max_taps)
        y = ops.rawboost_fir_bank(v, coef, ntaps, powers=True)
        return ops.rawboost_center_norm(y, subtract_mean=True, norm_mode=1)

    def isd(v):
        y = _as64(v) if v.dtype != torch.float64 else v
        if y is v:
            y = v.clone()
        beta = rng.uniform(0, a.P, size=B)
        n = torch.from_numpy((L * (beta / 100)).astype(np.int32)).cuda()
        thr = torch.empty(B, device="cuda", dtype=torch.int32)
        sid[0] += 1
        check(lib().occ_rawboost_isd_device(ptr(y), ptr(n), ptr(thr), None, B, L, float(a.g_sd), int(seed), int(sid[0]), stream_ptr()), "occ_rawboost_isd_device")
        return ops.rawboost_center_norm(y, subtract_mean=False, norm_mode=1)

    def ssi(v):
        sid[0] += 1
        noise = ops.philox_fill((B, L), torch.float64, seed, sid[0], normal=True)
        coef, ntaps = _design_bank_device(rng, B, 1, a, fs, [a.minG], [a.maxG], max_taps)
        nz = ops.rawboost_fir_bank(noise, coef, ntaps, powers=False)
        nz = ops.rawboost_center_norm(nz, subtract_mean=False, norm_mode=2)
        snr = torch.from_numpy(rng.uniform(a.SNRmin, a.SNRmax, size=B)).cuda()
        return ops.rawboost_ssi_mix(_as64(v), nz, snr)

    t = x.contiguous()
    if algo == 1:
        y = lnl(t)
    elif algo == 2:
        y = isd(t)
    elif algo == 3:
        y = ssi(t)
    elif algo == 4:
        y = ssi(isd(lnl(t)))
    elif algo == 5:
        y = isd(lnl(t))
    elif algo == 6:
        y = ssi(lnl(t))
    elif algo == 7:
        y = ssi(isd(t))
    else:
        y = ops.rawboost_center_norm(ops.add_f64(lnl(t), isd(t)), subtract_mean=False, norm_mode=1)
    return ops.cast(y, torch.float32) if y.dtype != torch.float32 else y
