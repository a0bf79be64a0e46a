// RawBoost signal path on gfx950: FIR bank (LnL / SSI colouring), mean/peak normalisation,
// ISD scatter, SSI mix, Philox fill.  Reference: RawBoost.py:20-97 (numpy/scipy on the host).
//
// Layout: waveforms are [B, L] row-major; the FIR kernel stages an audio tile plus its filter
// halo in LDS as f64 (the reference filters in float64 -- scipy.signal.lfilter on a float64 b),
// each thread owns 4 consecutive output samples and slides a 7-sample register window over the
// taps, filter coefficients are wave-uniform and come through the scalar cache.
#include "occ_common.h"

namespace {

constexpr int FIR_THREADS = 256;
constexpr int FIR_R = 4;                       // consecutive outputs per thread
constexpr int FIR_TILE = FIR_THREADS * FIR_R;  // 1024 outputs per workgroup
constexpr int FIR_MAXT = 1024;                 // max taps
constexpr int FIR_SEG = FIR_TILE + FIR_MAXT + 8;

template <typename TX>
__global__ __launch_bounds__(FIR_THREADS) void fir_bank_kernel(
    const TX* __restrict__ x, double* __restrict__ y, const double* __restrict__ coef,
    const int32_t* __restrict__ ntaps, int L, int n_filt, int max_taps, int powers) {
    __shared__ __attribute__((aligned(16))) double seg[FIR_SEG];
    const int b = blockIdx.y;
    const int j0 = blockIdx.x * FIR_TILE;
    const int tid = threadIdx.x;
    const TX* xb = x + (size_t)b * L;
    double acc[FIR_R] = {0.0, 0.0, 0.0, 0.0};

    for (int f = 0; f < n_filt; ++f) {
        const int nt = ntaps[b * n_filt + f];                 // uniform
        const double* cf = coef + ((size_t)b * n_filt + f) * max_taps;
        const int half = (nt + 1) / 2;                        // N/2 with N = nt + 1 (RawBoost.py:52,55)
        const int lo = j0 + half - (nt - 1);                  // global index held in seg[0]
        const int need = FIR_TILE + nt - 1;
        const int need4 = (need + 7) & ~3;                    // window reads up to 4 past the end
        __syncthreads();                                      // previous filter's reads are done
        for (int s = tid; s < need4; s += FIR_THREADS) {
            const int g = lo + s;
            double v = 0.0;
            if (s < need && g >= 0 && g < L) {
                const double p = (double)xb[g];
                v = p;
                if (powers) {
                    for (int e = 0; e < f; ++e) v *= p;
                    if (sizeof(TX) == 4) v = (double)(float)v;   // np.power on float32 stays float32
                }
            }
            seg[s] = v;
        }
        __syncthreads();
        // acc[r] += sum_kk cf[nt-1-kk] * seg[t0 + r + kk]
        const int t0 = tid * FIR_R;
        double w0 = seg[t0], w1 = seg[t0 + 1], w2 = seg[t0 + 2], w3 = seg[t0 + 3];
        int kk = 0;
        for (; kk + 4 <= nt; kk += 4) {
            const double w4 = seg[t0 + kk + 4], w5 = seg[t0 + kk + 5], w6 = seg[t0 + kk + 6], w7 = seg[t0 + kk + 7];
            const double c0 = cf[nt - 1 - kk], c1 = cf[nt - 2 - kk], c2 = cf[nt - 3 - kk], c3 = cf[nt - 4 - kk];
            acc[0] = fma(c0, w0, acc[0]); acc[1] = fma(c0, w1, acc[1]); acc[2] = fma(c0, w2, acc[2]); acc[3] = fma(c0, w3, acc[3]);
            acc[0] = fma(c1, w1, acc[0]); acc[1] = fma(c1, w2, acc[1]); acc[2] = fma(c1, w3, acc[2]); acc[3] = fma(c1, w4, acc[3]);
            acc[0] = fma(c2, w2, acc[0]); acc[1] = fma(c2, w3, acc[1]); acc[2] = fma(c2, w4, acc[2]); acc[3] = fma(c2, w5, acc[3]);
            acc[0] = fma(c3, w3, acc[0]); acc[1] = fma(c3, w4, acc[1]); acc[2] = fma(c3, w5, acc[2]); acc[3] = fma(c3, w6, acc[3]);
            w0 = w4; w1 = w5; w2 = w6; w3 = w7;
        }
        for (; kk < nt; ++kk) {                               // tail (nt is odd)
            const double c = cf[nt - 1 - kk];
#pragma unroll
            for (int r = 0; r < FIR_R; ++r) acc[r] = fma(c, seg[t0 + r + kk], acc[r]);
        }
    }
    const int j = j0 + tid * FIR_R;
    double* yb = y + (size_t)b * L;
#pragma unroll
    for (int r = 0; r < FIR_R; ++r)
        if (j + r < L) yb[j + r] = acc[r];
}

// ---- per-utterance statistics: partials[b][tile] = {sum, min, max, sumsq} ------------------------
constexpr int ST_THREADS = 256;
constexpr int ST_TILE = 4096;

__device__ __forceinline__ void block_reduce4(double& s, double& mn, double& mx, double& q) {
    __shared__ double red[4][ST_THREADS / OCC_WAVE];
    s = wave_sum(s); q = wave_sum(q); mn = wave_min(mn); mx = wave_max(mx);
    const int w = threadIdx.x / OCC_WAVE, l = threadIdx.x % OCC_WAVE;
    if (l == 0) { red[0][w] = s; red[1][w] = mn; red[2][w] = mx; red[3][w] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < ST_THREADS / OCC_WAVE; ++i) {
            s += red[0][i]; q += red[3][i];
            mn = red[1][i] < mn ? red[1][i] : mn; mx = red[2][i] > mx ? red[2][i] : mx;
        }
    }
}

__global__ __launch_bounds__(ST_THREADS) void stats_kernel(const double* __restrict__ y, const double* __restrict__ y2,
                                                           double* __restrict__ partials, int L) {
    // y2 == nullptr: {sum y, min y, max y, 0};  else (SSI): {sum y^2, -, -, sum y2^2}
    const int b = blockIdx.y, tile = blockIdx.x, ntile = gridDim.x;
    const double* yb = y + (size_t)b * L;
    double s = 0.0, q = 0.0, mn = 1e300, mx = -1e300;
    for (int i = threadIdx.x; i < ST_TILE; i += ST_THREADS) {
        const int j = tile * ST_TILE + i;
        if (j < L) {
            const double v = yb[j];
            if (y2) {
                const double u = y2[(size_t)b * L + j];
                s += v * v; q += u * u;
            } else {
                s += v; mn = v < mn ? v : mn; mx = v > mx ? v : mx;
            }
        }
    }
    block_reduce4(s, mn, mx, q);
    if (threadIdx.x == 0) {
        double* p = partials + ((size_t)b * ntile + tile) * 4;
        p[0] = s; p[1] = mn; p[2] = mx; p[3] = q;
    }
}

__global__ __launch_bounds__(ST_THREADS) void center_norm_apply_kernel(double* __restrict__ y, const double* __restrict__ partials,
                                                                       int L, int subtract_mean, int norm_mode) {
    const int b = blockIdx.y, tile = blockIdx.x, ntile = gridDim.x;
    double s = 0.0, mn = 1e300, mx = -1e300;
    for (int t = 0; t < ntile; ++t) {                      // fixed order: bitwise reproducible
        const double* p = partials + ((size_t)b * ntile + t) * 4;
        s += p[0]; mn = p[1] < mn ? p[1] : mn; mx = p[2] > mx ? p[2] : mx;
    }
    const double mean = subtract_mean ? s / (double)L : 0.0;
    const double hi = mx - mean, lo = mean - mn;
    const double peak = hi > lo ? hi : lo;                 // max |y - mean|
    const bool scale = (norm_mode == 2) || (norm_mode == 1 && peak > 1.0);
    double* yb = y + (size_t)b * L;
    for (int i = threadIdx.x; i < ST_TILE; i += ST_THREADS) {
        const int j = tile * ST_TILE + i;
        if (j < L) {
            double v = yb[j] - mean;
            if (scale) v = v / peak;
            yb[j] = v;
        }
    }
}

__global__ __launch_bounds__(ST_THREADS) void ssi_mix_apply_kernel(const double* __restrict__ x, const double* __restrict__ noise,
                                                                   const double* __restrict__ snr, double* __restrict__ out,
                                                                   const double* __restrict__ partials, int L) {
    const int b = blockIdx.y, tile = blockIdx.x, ntile = gridDim.x;
    double nn = 0.0, xx = 0.0;
    for (int t = 0; t < ntile; ++t) {
        const double* p = partials + ((size_t)b * ntile + t) * 4;
        nn += p[0]; xx += p[3];
    }
    const double n_norm = sqrt(nn), x_norm = sqrt(xx);
    const double den = pow(10.0, 0.05 * snr[b]);
    for (int i = threadIdx.x; i < ST_TILE; i += ST_THREADS) {
        const int j = tile * ST_TILE + i;
        if (j < L) {
            const size_t o = (size_t)b * L + j;
            out[o] = x[o] + noise[o] / n_norm * x_norm / den;      // RawBoost.py:95-96 operation order
        }
    }
}

__global__ void isd_scatter_kernel(double* __restrict__ y, const int32_t* __restrict__ pos, const double* __restrict__ fr,
                                   const int32_t* __restrict__ n, int L, int max_n, double g_sd) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n[b] || i >= max_n) return;
    const int p = pos[(size_t)b * max_n + i];
    if (p < 0 || p >= L) return;
    double* yp = y + (size_t)b * L + p;
    const double v = *yp;
    *yp = v + g_sd * v * fr[(size_t)b * max_n + i];               // RawBoost.py:81-82
}

template <typename TS, typename TD>
__global__ void cast_kernel(const TS* __restrict__ s, TD* __restrict__ d, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        d[i] = (TD)s[i];
}
__global__ void cast_f32_bf16_kernel(const float* __restrict__ s, unsigned short* __restrict__ d, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        d[i] = f32_to_bf16_bits(s[i]);
}
__global__ void cast_bf16_f32_kernel(const unsigned short* __restrict__ s, float* __restrict__ d, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        d[i] = bf16_bits_to_f32(s[i]);
}

__global__ void add_f64_kernel(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ o, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) o[i] = a[i] + b[i];
}

// ---- Philox4x32-10 -------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox4x32_10(uint64_t ctr, uint64_t stream_id, uint64_t seed, uint32_t (&out)[4]) {
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

template <typename T>
__global__ void philox_fill_kernel(T* __restrict__ dst, int64_t n, uint64_t seed, uint64_t stream_id, int normal) {
    const int64_t nquad = (n + 3) / 4;
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < nquad; q += (int64_t)gridDim.x * blockDim.x) {
        uint32_t r[4];
        philox4x32_10((uint64_t)q, stream_id, seed, r);
        double v[4];
        if (normal) {                                   // Box-Muller on two pairs
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const double u1 = ((double)r[2 * h] + 1.0) * (1.0 / 4294967296.0);      // (0,1]
                const double u2 = (double)r[2 * h + 1] * (1.0 / 4294967296.0);
                const double rad = sqrt(-2.0 * log(u1));
                v[2 * h] = rad * cos(6.283185307179586476925 * u2);
                v[2 * h + 1] = rad * sin(6.283185307179586476925 * u2);
            }
        } else {
#pragma unroll
            for (int h = 0; h < 4; ++h) v[h] = (double)r[h] * (1.0 / 4294967296.0);
        }
#pragma unroll
        for (int h = 0; h < 4; ++h)
            if (q * 4 + h < n) dst[q * 4 + h] = (T)v[h];
    }
}


// ---- on-device filter design: genNotchCoeffs (RawBoost.py:28-48) for a batch of filters -----------------------------
// One workgroup per filter: 5 Hamming-windowed band-stops (scipy firwin) convolved together in LDS, |H| on freqz's
// 512-point grid by a phasor recurrence, scaled to 10^(G/20) / max|H|.  f64 throughout, like the host version.
constexpr int NC_THREADS = 512, NC_MAXT = 1024;
__global__ __launch_bounds__(NC_THREADS) void notch_coeffs_kernel(const double* __restrict__ fc, const double* __restrict__ bw, const int32_t* __restrict__ cc,
                                                                  const double* __restrict__ gain, int n_bands, double fs, double* __restrict__ coef,
                                                                  int32_t* __restrict__ ntaps, int max_taps) {
    __shared__ double b0[NC_MAXT], b1[NC_MAXT], h[128];
    __shared__ double red[NC_THREADS / 64];
    __shared__ double bc;
    const int f = blockIdx.x, tid = threadIdx.x;
    double* cur = b0; double* nxt = b1;
    int len = 1;
    if (tid == 0) cur[0] = 1.0;
    const double nyq = 0.5 * fs, PI = 3.14159265358979323846;
    for (int i = 0; i < n_bands; ++i) {
        int c = cc[f * n_bands + i];
        if ((c & 1) == 0) c += 1;
        if (c > 127) c = 127;
        double f1 = fc[f * n_bands + i] - bw[f * n_bands + i] / 2, f2 = fc[f * n_bands + i] + bw[f * n_bands + i] / 2;
        if (f1 <= 0) f1 = 1.0 / 1000;
        if (f2 >= fs / 2) f2 = fs / 2 - 1.0 / 1000;
        const double lo = f1 / nyq, hi = f2 / nyq, alpha = 0.5 * (c - 1);
        double v = 0.0;
        if (tid < c) {
            const double m = tid - alpha;
            auto sinc = [&](double x) { const double y = PI * (x == 0.0 ? 1.0e-20 : x); return sin(y) / y; };
            v = lo * sinc(lo * m);
            v = v + (1.0 * sinc(1.0 * m) - hi * sinc(hi * m));
            v *= c == 1 ? 1.0 : 0.54 - 0.46 * cos(2.0 * PI * tid / (c - 1));
        }
        double s = wave_sum(v);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = s;
        __syncthreads();
        double tot = 0.0;
        for (int w = 0; w < NC_THREADS / 64; ++w) tot += red[w];
        if (tid < c) h[tid] = v / tot;
        __syncthreads();
        const int nl = len + c - 1;                       // np.convolve(h, cur)
        for (int p = tid; p < nl && p < NC_MAXT; p += NC_THREADS) {
            double acc = 0.0;
            const int j0 = p >= len - 1 ? p - (len - 1) : 0;
            for (int j = j0; j < c && j <= p; ++j) acc += h[j] * cur[p - j];
            nxt[p] = acc;
        }
        __syncthreads();
        double* t = cur; cur = nxt; nxt = t;
        len = nl < NC_MAXT ? nl : NC_MAXT;
    }
    // max |H(e^{jw})|, w = pi*k/512
    double mag = 0.0;
    {
        const int k = tid;                                 // 512 threads = 512 frequencies
        const double w = PI * k / 512.0, cw = cos(w), sw = sin(w);
        double re = 0.0, im = 0.0, pr = 1.0, pi_ = 0.0;    // phasor e^{-jwn}
        for (int n = 0; n < len; ++n) {
            re += cur[n] * pr; im += cur[n] * pi_;
            const double npr = pr * cw + pi_ * sw, npi = pi_ * cw - pr * sw;
            pr = npr; pi_ = npi;
            if ((n & 63) == 63) { const double nn = pr * pr + pi_ * pi_; const double r = rsqrt(nn); pr *= r; pi_ *= r; }   // keep |phasor| = 1
        }
        mag = sqrt(re * re + im * im);
    }
    mag = wave_max(mag);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = mag;
    __syncthreads();
    if (tid == 0) { double m = red[0]; for (int w = 1; w < NC_THREADS / 64; ++w) m = red[w] > m ? red[w] : m; bc = pow(10.0, gain[f] / 20.0) / m; }
    __syncthreads();
    for (int n = tid; n < max_taps; n += NC_THREADS) coef[(size_t)f * max_taps + n] = n < len ? cur[n] * bc : 0.0;
    if (tid == 0) ntaps[f] = len;
}

// ---- on-device ISD (RawBoost.py:73-84): exactly-n random positions via a radix select on Philox keys ------------------------
__device__ __forceinline__ uint32_t isd_key(uint64_t seed, uint64_t sid, int b, int j, int word) {
    uint32_t r[4];
    philox4x32_10(((uint64_t)b << 32) | (uint32_t)j, sid, seed, r);
    return r[word];
}
// thr[b] = the n[b]-th smallest key (keys <= thr are selected; ties with thr add at most a handful of extra positions)
__global__ __launch_bounds__(1024) void isd_select_kernel(const int32_t* __restrict__ n, uint32_t* __restrict__ thr, int L, uint64_t seed, uint64_t sid) {
    __shared__ unsigned hist[2048];
    __shared__ unsigned prefix_s, want_s;
    const int b = blockIdx.x, tid = threadIdx.x;
    unsigned want = (unsigned)n[b];
    if (want == 0) { if (tid == 0) thr[b] = 0u; return; }
    if (want > (unsigned)L) want = L;
    unsigned prefix = 0;                      // bits already fixed (from the top)
    const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
        for (int i = tid; i < 2048; i += 1024) hist[i] = 0;
        __syncthreads();
        const unsigned hi_mask = pass == 0 ? 0u : (0xffffffffu << (shifts[pass - 1]));
        for (int j = tid; j < L; j += 1024) {
            const unsigned key = isd_key(seed, sid, b, j, 0);
            if ((key & hi_mask) == (prefix & hi_mask)) atomicAdd(&hist[(key >> shifts[pass]) & ((1u << bits[pass]) - 1)], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned acc = 0, bin = 0;
            const unsigned nb = 1u << bits[pass];
            for (bin = 0; bin < nb; ++bin) { if (acc + hist[bin] >= want) break; acc += hist[bin]; }
            if (bin >= nb) bin = nb - 1;
            prefix_s = prefix | (bin << shifts[pass]);
            want_s = want - acc;
        }
        __syncthreads();
        prefix = prefix_s; want = want_s;
        __syncthreads();
    }
    if (tid == 0) thr[b] = prefix;
}
__global__ void isd_apply_kernel(double* __restrict__ y, const int32_t* __restrict__ n, const uint32_t* __restrict__ thr, int L, double g_sd, uint64_t seed,
                                 uint64_t sid, int32_t* __restrict__ count) {
    const int b = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= L || n[b] <= 0) return;
    uint32_t r[4];
    philox4x32_10(((uint64_t)b << 32) | (uint32_t)j, sid, seed, r);
    if (r[0] > thr[b]) return;
    const double u1 = (double)r[1] * (1.0 / 4294967296.0), u2 = (double)r[2] * (1.0 / 4294967296.0);
    double* yp = y + (size_t)b * L + j;
    const double v = *yp;
    *yp = v + g_sd * v * ((2.0 * u1 - 1.0) * (2.0 * u2 - 1.0));
    if (count) atomicAdd(count + b, 1);
}

}  // namespace

extern "C" {

int occ_rawboost_fir_bank(const void* x, int x_dtype, double* y, const double* coef, const int32_t* ntaps,
                          int64_t B, int64_t L, int64_t n_filt, int64_t max_taps, int powers, void* stream) {
    OCC_CHECK_ARG(x && y && coef && ntaps, "occ_rawboost_fir_bank: null pointer");
    OCC_CHECK_ARG(B > 0 && L > 0 && n_filt > 0 && B < 65536, "occ_rawboost_fir_bank: bad shape B=%ld L=%ld n_filt=%ld", (long)B, (long)L, (long)n_filt);
    OCC_CHECK_ARG(max_taps >= 1 && max_taps <= FIR_MAXT, "occ_rawboost_fir_bank: max_taps %ld not in [1,%d]", (long)max_taps, FIR_MAXT);
    OCC_CHECK_ARG(x_dtype == OCC_F32 || x_dtype == OCC_F64, "occ_rawboost_fir_bank: x dtype must be f32 or f64");
    OCC_CHECK_ARG(L < (1ll << 30), "occ_rawboost_fir_bank: L too large");
    dim3 grid((unsigned)occ_cdiv(L, FIR_TILE), (unsigned)B);
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == OCC_F32)
        hipLaunchKernelGGL(fir_bank_kernel<float>, grid, dim3(FIR_THREADS), 0, s, (const float*)x, y, coef, ntaps, (int)L, (int)n_filt, (int)max_taps, powers);
    else
        hipLaunchKernelGGL(fir_bank_kernel<double>, grid, dim3(FIR_THREADS), 0, s, (const double*)x, y, coef, ntaps, (int)L, (int)n_filt, (int)max_taps, powers);
    OCC_LAUNCH_CHECK("occ_rawboost_fir_bank");
    return OCC_OK;
}

int occ_rawboost_center_norm(double* y, int64_t B, int64_t L, int subtract_mean, int norm_mode, double* partials, void* stream) {
    OCC_CHECK_ARG(y && partials, "occ_rawboost_center_norm: null pointer");
    OCC_CHECK_ARG(B > 0 && L > 0 && B < 65536 && L < (1ll << 30), "occ_rawboost_center_norm: bad shape");
    OCC_CHECK_ARG(norm_mode >= 0 && norm_mode <= 2, "occ_rawboost_center_norm: norm_mode must be 0,1,2");
    dim3 grid((unsigned)occ_cdiv(L, ST_TILE), (unsigned)B);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(stats_kernel, grid, dim3(ST_THREADS), 0, s, (const double*)y, (const double*)nullptr, partials, (int)L);
    hipLaunchKernelGGL(center_norm_apply_kernel, grid, dim3(ST_THREADS), 0, s, y, (const double*)partials, (int)L, subtract_mean, norm_mode);
    OCC_LAUNCH_CHECK("occ_rawboost_center_norm");
    return OCC_OK;
}

int occ_rawboost_isd_scatter(double* y, const int32_t* pos, const double* fr, const int32_t* n, int64_t B, int64_t L,
                             int64_t max_n, double g_sd, void* stream) {
    OCC_CHECK_ARG(y && pos && fr && n, "occ_rawboost_isd_scatter: null pointer");
    OCC_CHECK_ARG(B > 0 && L > 0 && max_n >= 0 && B < 65536 && L < (1ll << 30), "occ_rawboost_isd_scatter: bad shape");
    if (max_n == 0) return OCC_OK;
    dim3 grid((unsigned)occ_cdiv(max_n, 256), (unsigned)B);
    hipLaunchKernelGGL(isd_scatter_kernel, grid, dim3(256), 0, (hipStream_t)stream, y, pos, fr, n, (int)L, (int)max_n, g_sd);
    OCC_LAUNCH_CHECK("occ_rawboost_isd_scatter");
    return OCC_OK;
}

int occ_rawboost_ssi_mix(const double* x, const double* noise, const double* snr, double* out, int64_t B, int64_t L,
                         double* partials, void* stream) {
    OCC_CHECK_ARG(x && noise && snr && out && partials, "occ_rawboost_ssi_mix: null pointer");
    OCC_CHECK_ARG(B > 0 && L > 0 && B < 65536 && L < (1ll << 30), "occ_rawboost_ssi_mix: bad shape");
    dim3 grid((unsigned)occ_cdiv(L, ST_TILE), (unsigned)B);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(stats_kernel, grid, dim3(ST_THREADS), 0, s, noise, x, partials, (int)L);
    hipLaunchKernelGGL(ssi_mix_apply_kernel, grid, dim3(ST_THREADS), 0, s, x, noise, snr, out, (const double*)partials, (int)L);
    OCC_LAUNCH_CHECK("occ_rawboost_ssi_mix");
    return OCC_OK;
}

int occ_cast(const void* src, int sd, void* dst, int dd, int64_t n, void* stream) {
    OCC_CHECK_ARG(src && dst && n >= 0, "occ_cast: bad argument");
    if (n == 0) return OCC_OK;
    hipStream_t s = (hipStream_t)stream;
    const int blocks = (int)(occ_cdiv(n, 256) < 4096 ? occ_cdiv(n, 256) : 4096);
    if (sd == OCC_F64 && dd == OCC_F32) hipLaunchKernelGGL((cast_kernel<double, float>), dim3(blocks), dim3(256), 0, s, (const double*)src, (float*)dst, n);
    else if (sd == OCC_F32 && dd == OCC_F64) hipLaunchKernelGGL((cast_kernel<float, double>), dim3(blocks), dim3(256), 0, s, (const float*)src, (double*)dst, n);
    else if (sd == OCC_F32 && dd == OCC_BF16) hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(blocks), dim3(256), 0, s, (const float*)src, (unsigned short*)dst, n);
    else if (sd == OCC_BF16 && dd == OCC_F32) hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(blocks), dim3(256), 0, s, (const unsigned short*)src, (float*)dst, n);
    else { occ_set_error("occ_cast: unsupported dtype pair %d -> %d", sd, dd); return OCC_EUNSUPPORTED; }
    OCC_LAUNCH_CHECK("occ_cast");
    return OCC_OK;
}

int occ_add_f64(const double* a, const double* b, double* out, int64_t n, void* stream) {
    OCC_CHECK_ARG(a && b && out && n >= 0, "occ_add_f64: bad argument");
    if (n == 0) return OCC_OK;
    const int blocks = (int)(occ_cdiv(n, 256) < 4096 ? occ_cdiv(n, 256) : 4096);
    hipLaunchKernelGGL(add_f64_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
    OCC_LAUNCH_CHECK("occ_add_f64");
    return OCC_OK;
}

int occ_philox_fill(void* dst, int dtype, int64_t n, uint64_t seed, uint64_t stream_id, int normal, void* stream) {
    OCC_CHECK_ARG(dst && n >= 0, "occ_philox_fill: bad argument");
    if (n == 0) return OCC_OK;
    const int blocks = (int)(occ_cdiv(occ_cdiv(n, 4), 256) < 4096 ? occ_cdiv(occ_cdiv(n, 4), 256) : 4096);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == OCC_F32) hipLaunchKernelGGL(philox_fill_kernel<float>, dim3(blocks), dim3(256), 0, s, (float*)dst, n, seed, stream_id, normal);
    else if (dtype == OCC_F64) hipLaunchKernelGGL(philox_fill_kernel<double>, dim3(blocks), dim3(256), 0, s, (double*)dst, n, seed, stream_id, normal);
    else { occ_set_error("occ_philox_fill: dtype must be f32 or f64"); return OCC_EUNSUPPORTED; }
    OCC_LAUNCH_CHECK("occ_philox_fill");
    return OCC_OK;
}

int occ_notch_coeffs(const double* fc, const double* bw, const int32_t* c, const double* gain, int64_t n_filters, int64_t n_bands, double fs,
                     double* coef, int32_t* ntaps, int64_t max_taps, void* stream) {
    OCC_CHECK_ARG(fc && bw && c && gain && coef && ntaps && n_filters >= 1 && n_bands >= 1 && n_bands <= 8, "occ_notch_coeffs: bad argument");
    OCC_CHECK_ARG(max_taps >= 1 && max_taps <= NC_MAXT, "occ_notch_coeffs: max_taps must be in [1,%d]", NC_MAXT);
    hipLaunchKernelGGL(notch_coeffs_kernel, dim3((unsigned)n_filters), dim3(NC_THREADS), 0, (hipStream_t)stream, fc, bw, c, gain, (int)n_bands, fs, coef, ntaps,
                       (int)max_taps);
    OCC_LAUNCH_CHECK("occ_notch_coeffs");
    return OCC_OK;
}

int occ_rawboost_isd_device(double* y, const int32_t* n, uint32_t* thr_scratch, int32_t* count, int64_t B, int64_t L, double g_sd, uint64_t seed,
                            uint64_t stream_id, void* stream) {
    OCC_CHECK_ARG(y && n && thr_scratch && B >= 1 && B < 65536 && L >= 1 && L < (1ll << 30), "occ_rawboost_isd_device: bad argument");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(isd_select_kernel, dim3((unsigned)B), dim3(1024), 0, s, n, thr_scratch, (int)L, seed, stream_id);
    hipLaunchKernelGGL(isd_apply_kernel, dim3((unsigned)occ_cdiv(L, 256), (unsigned)B), dim3(256), 0, s, y, n, (const uint32_t*)thr_scratch, (int)L, g_sd, seed,
                       stream_id, count);
    OCC_LAUNCH_CHECK("occ_rawboost_isd_device");
    return OCC_OK;
}

}  // extern "C"
