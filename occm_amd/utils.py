"""``utils.extract_lfcc`` of the reference (utils.py:127-138) on the GPU.

The reference helper is one call into ``spafe.features.lfcc.lfcc`` (30 ms / 15 ms Hamming frames, pre-emphasis 0.97, 2048-point
spectrum, 128 linear triangles on 0-8 kHz, log, DCT-II, 13 coefficients, per-utterance mean/variance normalisation).  spafe is not
vendored or pinned by the reference, so this follows the algorithm restated in ``oracle/lfcc_ref.py`` (parity unpinned: checked
against that restatement, not against spafe).  The other helpers of the reference's utils.py (wavelet / synchrosqueezing demos,
dense padding helpers) are not on any path and are not mirrored.

Device pipeline, f32 throughout: ``occ_lfcc_frames`` (pre-emphasis + framing + window) -> ``occ_gemm`` with the [2052 x 480] DFT
matrix (a 480-sample frame has only 480 non-zero inputs of the 2048-point transform, so the transform is a small dense product on
the f32 MFMA path rather than an FFT) -> ``occ_lfcc_power`` -> ``occ_gemm`` (filter bank) -> ``occ_log_eps`` -> ``occ_gemm`` (DCT)
-> ``occ_mvn_frames``.  No CPU fallback: without the library or a GPU these raise.
"""
import numpy as np
import torch

from . import ops
from ._lib import check, lib, ptr, require_gpu, stream_ptr

FRAME_LEN_S, HOP_S, PRE_EMPH = 0.03, 0.015, 0.97
NFILTS, NFFT, NUM_CEPS, LOW_HZ, HIGH_HZ = 128, 2048, 13, 0.0, 8000.0
_EPS64 = float(np.finfo(np.float64).eps)
_consts = {}


def n_frames(n_samples, fs=16000):
    flen, hop = int(FRAME_LEN_S * fs), int(HOP_S * fs)
    rest = abs(n_samples - (flen - hop)) % hop
    return (n_samples + (hop - rest if rest else 0) - flen) // hop + 1


def _constants(fs, device):
    key = (int(fs), str(device))
    if key not in _consts:
        flen = int(FRAME_LEN_S * fs)
        nb = NFFT // 2 + 1
        ang = 2.0 * np.pi * np.outer(np.arange(nb), np.arange(flen)) / NFFT
        dft = np.zeros(((2 * nb + 3) // 4 * 4, flen))                        # rows: re(0..nb-1) | im(0..nb-1) | zero padding to N % 4 == 0
        dft[:nb], dft[nb:2 * nb] = np.cos(ang), -np.sin(ang)
        edges = LOW_HZ + (HIGH_HZ - LOW_HZ) / (NFILTS + 1.0) * np.arange(NFILTS + 2)
        freqs = np.linspace(0.0, fs / 2.0, nb)
        fb = np.zeros((NFILTS, (nb + 3) // 4 * 4))                           # K padded to a multiple of 4 with zero columns
        for j in range(NFILTS):
            lo, c, hi = edges[j], edges[j + 1], edges[j + 2]
            up = (freqs >= lo) & (freqs <= c)
            fb[j, :nb][up] = (freqs[up] - lo) / (c - lo)
            dn = (freqs >= c) & (freqs <= hi)
            fb[j, :nb][dn] = (hi - freqs[dn]) / (hi - c)
        k, n = np.arange(16)[:, None], np.arange(NFILTS)[None, :]
        dct = np.cos(np.pi * k * (2 * n + 1) / (2.0 * NFILTS)) * np.sqrt(2.0 / NFILTS)
        dct[0] *= np.sqrt(0.5)
        dct[NUM_CEPS:] = 0.0                                                 # N padded 13 -> 16
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
        _consts[key] = dict(window=to(np.hamming(flen)), dft=to(dft), fbank=to(fb), dct=to(dct), nb=nb, flen=flen, hop=int(HOP_S * fs))
    return _consts[key]


def extract_lfcc_batch(wav, sr=16000, normalize=True):
    """wav f32 [B, L] on the GPU -> f32 [B, n_frames, 13] (every utterance of the batch has L samples)."""
    require_gpu()
    if not (torch.is_tensor(wav) and wav.is_cuda and wav.dtype == torch.float32 and wav.dim() == 2 and wav.is_contiguous()):
        raise ValueError("extract_lfcc_batch: wav must be a contiguous float32 [B, L] CUDA tensor")
    B, L = wav.shape
    c = _constants(sr, wav.device)
    F, flen, nb = n_frames(L, sr), c["flen"], c["nb"]
    if F < 1 or L < 2:
        raise ValueError("extract_lfcc_batch: %d samples give no frame" % L)
    dev, s = wav.device, stream_ptr()
    frames = torch.empty(B * F, flen, device=dev, dtype=torch.float32)
    check(lib().occ_lfcc_frames(ptr(wav), ptr(c["window"]), ptr(frames), B, L, F, flen, c["hop"], flen, PRE_EMPH, s), "occ_lfcc_frames")
    spec = ops.linear(frames, c["dft"])                                              # [B*F, 2052]
    ldp = c["fbank"].shape[1]
    power = torch.empty(B * F, ldp, device=dev, dtype=torch.float32)
    check(lib().occ_lfcc_power(ptr(spec), ptr(power), B * F, nb, spec.shape[1], ldp, 1.0 / NFFT, s), "occ_lfcc_power")
    logfb = ops.linear(power, c["fbank"])                                            # [B*F, 128]
    check(lib().occ_log_eps(ptr(logfb), logfb.numel(), _EPS64, s), "occ_log_eps")
    ceps = ops.linear(logfb, c["dct"])                                               # [B*F, 16], columns 13.. are zero
    out = torch.empty(B, F, NUM_CEPS, device=dev, dtype=torch.float32)
    if normalize:
        check(lib().occ_mvn_frames(ptr(ceps), ptr(out), B, F, NUM_CEPS, 16, NUM_CEPS, s), "occ_mvn_frames")
    else:
        out.copy_(ceps.view(B, F, 16)[:, :, :NUM_CEPS])
    return out


def extract_lfcc(y, sr):
    """Same call as the reference's helper (utils.py:127): one utterance, numpy in -> numpy [n_frames, 13] out (a torch tensor in gives
    a tensor on the same device back)."""
    if torch.is_tensor(y):
        x = y.detach().to(device="cuda", dtype=torch.float32).reshape(1, -1).contiguous()
        return extract_lfcc_batch(x, sr)[0].to(y.device)
    x = torch.from_numpy(np.ascontiguousarray(np.asarray(y, dtype=np.float32).reshape(1, -1))).cuda()
    return extract_lfcc_batch(x, sr)[0].cpu().numpy().astype(np.float64)
