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
    try:
        import soundfile as sf
        data, fs = sf.read(path, dtype="float32")
        return (data.mean(axis=1) if data.ndim > 1 else data), fs
    except ImportError:
        import librosa
        return librosa.load(path, sr=sr)


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
