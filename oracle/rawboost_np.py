"""numpy restatement of the reference's RawBoost augmentation (TEST ORACLE).

Follows RawBoost.py:14-97 and the dispatcher data_utils_SSL.py:111-173.
The reference leans on scipy.signal (pinned scipy 1.11.3, requirement.txt:108)
for ``firwin``/``freqz``/``lfilter``; their published algorithms are restated
here in plain numpy so the HIP/C++ product code has a dependency-free twin to
be compared with.

All random parameters are *explicit arguments*: ``draw_*`` helpers reproduce
the reference's draw order on the legacy global ``np.random`` stream so that
``np.random.seed(s)`` followed by these helpers consumes the stream exactly as
the reference does.
"""
import numpy as np


# ----------------------------------------------------------------------------
# parameter draws (reference order)
# ----------------------------------------------------------------------------
def rand_range(x1, x2, integer):
    """RawBoost.py:14-18 (int() of a 1-element array, truncating)."""
    y = np.random.uniform(low=x1, high=x2, size=(1,))
    return int(y[0]) if integer else float(y[0])


def draw_notch_params(nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG):
    """Draw order of genNotchCoeffs, RawBoost.py:30-33, 45: (fc, bw, c) per band, then G."""
    bands = []
    for _ in range(nBands):
        fc = rand_range(minF, maxF, 0)
        bw = rand_range(minBW, maxBW, 0)
        c = rand_range(minCoeff, maxCoeff, 1)
        bands.append((fc, bw, c))
    G = rand_range(minG, maxG, 0)
    return bands, G


def draw_lnl_params(N_f, nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG,
                    minBias, maxBias):
    """RawBoost.py:61-66: gains are lowered once at i==1 and stay lowered."""
    out = []
    for i in range(N_f):
        if i == 1:
            minG = minG - minBias
            maxG = maxG - maxBias
        out.append(draw_notch_params(nBands, minF, maxF, minBW, maxBW, minCoeff, maxCoeff, minG, maxG))
    return out


def draw_isd_params(x_len, P):
    """RawBoost.py:74-81: beta, permutation prefix, two uniform vectors."""
    beta = rand_range(0, P, 0)
    n = int(x_len * (beta / 100))
    p = np.random.permutation(x_len)[:n]
    u1 = np.random.rand(p.shape[0])
    u2 = np.random.rand(p.shape[0])
    return p, u1, u2


# ----------------------------------------------------------------------------
# filter design (scipy.signal.firwin / freqz restated)
# ----------------------------------------------------------------------------
def _sinc(x):
    x = np.asarray(x, dtype=np.float64)
    y = np.pi * np.where(x == 0, 1.0e-20, x)
    return np.sin(y) / y


def firwin_bandstop_hamming(numtaps, f1, f2, fs):
    """scipy.signal.firwin(numtaps, [f1, f2], window='hamming', fs=fs) (band-stop, scaled at DC)."""
    nyq = 0.5 * fs
    lo, hi = f1 / nyq, f2 / nyq
    alpha = 0.5 * (numtaps - 1)
    m = np.arange(0, numtaps) - alpha
    h = lo * _sinc(lo * m)                      # pass band [0, lo]
    h = h + (1.0 * _sinc(1.0 * m) - hi * _sinc(hi * m))   # pass band [hi, 1]
    n = np.arange(numtaps)
    if numtaps == 1:
        win = np.ones(1)
    else:
        win = 0.54 - 0.46 * np.cos(2.0 * np.pi * n / (numtaps - 1))
    h = h * win
    return h / np.sum(h)                        # unity gain at f=0


def freqz_mag_max(b, worN=512):
    """max |H(e^{jw})| on scipy.signal.freqz's default grid w = pi*k/512, k=0..511."""
    n = np.arange(b.shape[0])
    w = np.pi * np.arange(worN) / worN
    H = np.exp(-1j * np.outer(w, n)) @ b
    return np.max(np.abs(H))


def notch_coeffs(bands, G, fs):
    """genNotchCoeffs, RawBoost.py:28-48, with the draws passed in."""
    b = np.ones(1)
    for fc, bw, c in bands:
        if c / 2 == int(c / 2):
            c = c + 1
        f1 = fc - bw / 2
        f2 = fc + bw / 2
        if f1 <= 0:
            f1 = 1 / 1000
        if f2 >= fs / 2:
            f2 = fs / 2 - 1 / 1000
        b = np.convolve(firwin_bandstop_hamming(c, float(f1), float(f2), fs), b)
    return pow(10, G / 20) * b / freqz_mag_max(b)


# ----------------------------------------------------------------------------
# signal path
# ----------------------------------------------------------------------------
def norm_wav(x, always):
    """normWav, RawBoost.py:20-25."""
    peak = np.amax(np.abs(x))
    if always or peak > 1:
        x = x / peak
    return x


def filter_fir(x, b):
    """filterFIR, RawBoost.py:51-56: causal FIR on x padded with N=len(b)+1 zeros,
    then the slice [N/2, len-N/2) -- one sample later than a zero-phase 'same'."""
    N = b.shape[0] + 1
    xpad = np.pad(x, (0, N), 'constant')
    y = np.convolve(xpad.astype(np.float64), b)[:xpad.shape[0]]
    return y[int(N / 2):int(y.shape[0] - N / 2)]


def lnl_convolutive_noise(x, coeff_sets):
    """LnL_convolutive_noise, RawBoost.py:59-69 with coefficient sets b_1..b_Nf given."""
    y = np.zeros(x.shape[0], dtype=np.float64)
    for i, b in enumerate(coeff_sets):
        y = y + filter_fir(np.power(x, (i + 1)), b)
    y = y - np.mean(y)
    return norm_wav(y, 0)


def isd_additive_noise(x, p, u1, u2, g_sd):
    """ISD_additive_noise, RawBoost.py:73-84 with (p, u1, u2) given."""
    y = x.copy()
    f_r = np.multiply((2 * u1) - 1, (2 * u2) - 1)
    r = g_sd * x[p] * f_r
    y[p] = x[p] + r
    return norm_wav(y, 0)


def ssi_additive_noise(x, noise, b, SNR):
    """SSI_additive_noise, RawBoost.py:89-97 with (noise, b, SNR) given."""
    noise = filter_fir(noise, b)
    noise = norm_wav(noise, 1)
    noise = noise / np.linalg.norm(noise, 2) * np.linalg.norm(x, 2) / 10.0 ** (0.05 * SNR)
    return x + noise


class RawBoostArgs:
    """Defaults of the 17 flags, oc_training.py:79-119."""
    algo = 3
    nBands = 5; minF = 20; maxF = 8000; minBW = 100; maxBW = 1000
    minCoeff = 10; maxCoeff = 100; minG = 0; maxG = 0
    minBiasLinNonLin = 5; maxBiasLinNonLin = 20; N_f = 5
    P = 10; g_sd = 2
    SNRmin = 10; SNRmax = 40


def process_rawboost(x, fs, args, algo):
    """process_Rawboost_feature, data_utils_SSL.py:111-173, drawing from np.random in reference order."""
    a = args

    def lnl(v):
        sets = draw_lnl_params(a.N_f, a.nBands, a.minF, a.maxF, a.minBW, a.maxBW, a.minCoeff,
                               a.maxCoeff, a.minG, a.maxG, a.minBiasLinNonLin, a.maxBiasLinNonLin)
        # coefficients are produced band set by band set *interleaved* with the draws in
        # the reference; the draws do not depend on the signal so drawing first is identical.
        return lnl_convolutive_noise(v, [notch_coeffs(b, G, fs) for b, G in sets])

    def isd(v):
        p, u1, u2 = draw_isd_params(v.shape[0], a.P)
        return isd_additive_noise(v, p, u1, u2, a.g_sd)

    def ssi(v):
        noise = np.random.normal(0, 1, v.shape[0])
        bands, G = draw_notch_params(a.nBands, a.minF, a.maxF, a.minBW, a.maxBW, a.minCoeff,
                                     a.maxCoeff, a.minG, a.maxG)
        b = notch_coeffs(bands, G, fs)
        SNR = rand_range(a.SNRmin, a.SNRmax, 0)
        return ssi_additive_noise(v, noise, b, SNR)

    if algo == 1:
        return lnl(x)
    if algo == 2:
        return isd(x)
    if algo == 3:
        return ssi(x)
    if algo == 4:
        return ssi(isd(lnl(x)))
    if algo == 5:
        return isd(lnl(x))
    if algo == 6:
        return ssi(lnl(x))
    if algo == 7:
        return ssi(isd(x))
    if algo == 8:
        f1 = lnl(x)
        f2 = isd(x)
        return norm_wav(f1 + f2, 0)
    return x


def pad_tile(x, max_len=64600):
    """pad, data_utils_SSL.py:47-54: truncate or tile-repeat to max_len."""
    n = x.shape[0]
    if n >= max_len:
        return x[:max_len]
    reps = int(max_len / n) + 1
    return np.tile(x, reps)[:max_len]
