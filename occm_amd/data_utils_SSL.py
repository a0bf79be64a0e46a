"""Dataset API of the reference's ``data_utils_SSL.py`` (:17-173) on top of the HIP RawBoost path.

``genSpoof_list``, ``pad``, ``Dataset_ASVspoof2019_train``, ``Dataset_ASVspoof2021_eval`` and
``process_Rawboost_feature`` keep their names, arguments and return structure.  Audio decoding is file
I/O, out of the hot path: ``load_audio`` reads 16-bit PCM WAV with the standard library and, when the
optional ``soundfile``/``librosa`` packages exist, FLAC through them.
"""
import os
import wave

import numpy as np
import torch
from torch.utils.data import Dataset

from . import ops
from . import RawBoost as RB
from .RawBoost import ISD_additive_noise, LnL_convolutive_noise, SSI_additive_noise, normWav  # noqa: F401


def genSpoof_list(dir_meta, is_train=False, is_eval=False):
    """data_utils_SSL.py:17-43: bonafide -> 1, spoof -> 0 (opposite polarity to PFDataset)."""
    d_meta, file_list = {}, []
    with open(dir_meta, "r") as f:
        l_meta = f.readlines()
    if is_eval and not is_train:
        return [line.strip() for line in l_meta]
    for line in l_meta:
        _, key, _, _, label = line.strip().split()
        file_list.append(key)
        d_meta[key] = 1 if label == "bonafide" else 0
    return d_meta, file_list


def pad(x, max_len=64600):
    """data_utils_SSL.py:47-54: truncate, or tile-repeat (not zero-pad) up to max_len."""
    x_len = x.shape[0]
    if x_len >= max_len:
        return x[:max_len]
    num_repeats = int(max_len / x_len) + 1
    return np.tile(x, num_repeats)[:max_len]


def load_audio(path, sr=16000):
    if path.endswith(".wav") and os.path.exists(path):
        with wave.open(path, "rb") as w:
            assert w.getsampwidth() == 2, "only 16-bit PCM wav is handled by the built-in reader"
            data = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float32) / 32768.0
            if w.getnchannels() > 1:
                data = data.reshape(-1, w.getnchannels()).mean(axis=1)
            return data, w.getframerate()
    if path.lower().endswith(".flac") and os.path.exists(path):
        return read_flac(path)
    try:
        import soundfile as sf
        data, fs = sf.read(path, dtype="float32")
        return (data.mean(axis=1) if data.ndim > 1 else data), fs
    except ImportError:
        import librosa
        return librosa.load(path, sr=sr)


def decode_flac_bytes(raw, verify_md5=True):
    """FLAC file contents -> (int32 [n, channels], sample rate, bits per sample) through the library's host decoder
    (occ_flac_info / occ_flac_decode: frame CRCs checked there).  The STREAMINFO MD5 of the decoded PCM is checked here,
    so a stream the decoder misreads raises instead of returning noise."""
    import ctypes
    import hashlib
    from ._lib import check, lib
    buf = (ctypes.c_uint8 * len(raw)).from_buffer_copy(raw)
    info = (ctypes.c_int32 * 4)(); total = ctypes.c_int64(0); md5 = (ctypes.c_uint8 * 16)()
    check(lib().occ_flac_info(buf, len(raw), info, ctypes.byref(total), md5), "occ_flac_info")
    fs, nch, bps, has_md5 = info[0], info[1], info[2], info[3]
    cap = total.value if total.value > 0 else max(1, len(raw) * 8)          # unknown length: a sample takes at least one bit
    out = np.empty((cap, nch), dtype=np.int32)
    got = ctypes.c_int64(0)
    check(lib().occ_flac_decode(buf, len(raw), out.ctypes.data_as(ctypes.c_void_p), cap, ctypes.byref(got)), "occ_flac_decode")
    out = out[:got.value]
    if total.value and got.value != total.value:
        raise ValueError("FLAC stream ended after %d of %d samples" % (got.value, total.value))
    if verify_md5 and has_md5:
        nbytes = (bps + 7) // 8                                              # signature = MD5 of the interleaved little-endian samples
        le = out.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :nbytes]
        if hashlib.md5(np.ascontiguousarray(le).tobytes()).digest() != bytes(md5):
            raise ValueError("FLAC MD5 signature mismatch: the decoded audio is not what the encoder saw")
    return out, fs, bps


def read_flac(path):
    """-> (float32 mono in [-1, 1), sample rate), like the wav branch above (channels averaged, no resampling)."""
    with open(path, "rb") as f:
        pcm, fs, bps = decode_flac_bytes(f.read())
    x = pcm.astype(np.float32) / float(1 << (bps - 1))
    return (x.mean(axis=1) if x.shape[1] > 1 else x[:, 0]), fs


def process_Rawboost_feature(feature, sr, args, algo):
    """data_utils_SSL.py:111-173: algo 1/2/3 single, 4 = 1->2->3, 5 = 1->2, 6 = 1->3, 7 = 2->3,
    8 = normWav(1 + 2), otherwise identity.  numpy in -> numpy out, CUDA tensor in -> CUDA tensor out;
    intermediates stay on the GPU in float64 like the reference's host arrays."""
    if algo not in (1, 2, 3, 4, 5, 6, 7, 8):
        return feature
    t, was_np, sq = RB._to_dev(feature)
    a = args

    def lnl(v):
        return RB._lnl_dev(v, a.N_f, a.nBands, a.minF, a.maxF, a.minBW, a.maxBW, a.minCoeff, a.maxCoeff, a.minG, a.maxG,
                           a.minBiasLinNonLin, a.maxBiasLinNonLin, sr)

    def isd(v):
        return RB._isd_dev(RB._as64(v), a.P, a.g_sd)

    def ssi(v):
        return RB._ssi_dev(RB._as64(v), a.SNRmin, a.SNRmax, a.nBands, a.minF, a.maxF, a.minBW, a.maxBW, a.minCoeff, a.maxCoeff,
                           a.minG, a.maxG, sr)

    out_dtype = None
    if algo == 1:
        y = lnl(t)
    elif algo == 2:
        y = isd(t); out_dtype = t.dtype
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
        f1 = lnl(t)
        f2 = isd(t)
        if t.dtype != torch.float64:                      # the reference's ISD output keeps the input dtype
            f2 = ops.cast(ops.cast(f2, t.dtype), torch.float64)
        y = ops.rawboost_center_norm(ops.add_f64(f1, f2), subtract_mean=False, norm_mode=1)
    if out_dtype is not None and y.dtype != out_dtype:
        y = ops.cast(y, out_dtype)
    return RB._from_dev(y, was_np, sq)


class Dataset_ASVspoof2019_train(Dataset):
    """data_utils_SSL.py:57-82."""

    def __init__(self, args, list_IDs, labels, base_dir, algo):
        self.list_IDs, self.labels, self.base_dir, self.algo, self.args = list_IDs, labels, base_dir, algo, args
        self.cut = 64600

    def __len__(self):
        return len(self.list_IDs)

    def __getitem__(self, index):
        utt_id = self.list_IDs[index]
        X, fs = load_audio(self.base_dir + "flac/" + utt_id + ".flac", sr=16000)
        Y = process_Rawboost_feature(X, fs, self.args, self.algo)
        x_inp = torch.Tensor(pad(Y, self.cut))
        return x_inp, self.labels[utt_id]


class Dataset_ASVspoof2021_eval(Dataset):
    """data_utils_SSL.py:85-104."""

    def __init__(self, list_IDs, base_dir):
        self.list_IDs, self.base_dir = list_IDs, base_dir
        self.cut = 64600

    def __len__(self):
        return len(self.list_IDs)

    def __getitem__(self, index):
        utt_id = self.list_IDs[index]
        X, fs = load_audio(self.base_dir + "flac/" + utt_id + ".flac", sr=16000)
        return torch.Tensor(pad(X, self.cut)), utt_id
