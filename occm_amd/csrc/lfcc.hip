// LFCC front-end pieces: the elementwise / reduction stages around three f32 GEMMs (DFT as a [2052 x 480]
// matrix product, the 128 linear triangles, the DCT-II), which run through occ_gemm.
// Reference: utils.py:127-138 (extract_lfcc -> spafe.features.lfcc.lfcc; spafe is not vendored: oracle/lfcc_ref.py restates its published
// algorithm, parity unpinned).
#include "occ_common.h"

namespace {

constexpr int LT = 256;

// frames[(b*F + f)*ld + i] = window[i] * pe(b, f*hop + i),  pe(n) = x[n] - c*x[n-1] (pe(0) = x[0]), 0 beyond the signal (framing pads the
// pre-emphasised signal with zeros)
__global__ __launch_bounds__(LT) void lfcc_frames_kernel(const float* __restrict__ wav, const float* __restrict__ window, float* __restrict__ frames,
                                                        int L, int F, int flen, int hop, int ld, float coeff, long long total) {
    for (long long i = blockIdx.x * (long long)LT + threadIdx.x; i < total; i += (long long)gridDim.x * LT) {
        const int col = (int)(i % ld);
        const long long row = i / ld;
        const int f = (int)(row % F);
        const long long b = row / F;
        float v = 0.f;
        if (col < flen) {
            const int n = f * hop + col;
            if (n < L) {
                const float* x = wav + b * L;
                v = (n > 0 ? x[n] - coeff * x[n - 1] : x[0]) * window[col];
            }
        }
        frames[i] = v;
    }
}

// spec rows hold [re(0..nb-1) | im(0..nb-1)] (ld_spec floats); power[r][k] = scale * (re^2 + im^2), columns nb..ld_pow-1 zeroed
__global__ __launch_bounds__(LT) void lfcc_power_kernel(const float* __restrict__ spec, float* __restrict__ power, int nb, int ld_spec, int ld_pow, float scale,
                                                       long long total) {
    for (long long i = blockIdx.x * (long long)LT + threadIdx.x; i < total; i += (long long)gridDim.x * LT) {
        const int k = (int)(i % ld_pow);
        const long long r = i / ld_pow;
        float v = 0.f;
        if (k < nb) {
            const float re = spec[r * ld_spec + k], im = spec[r * ld_spec + nb + k];
            v = scale * (re * re + im * im);
        }
        power[i] = v;
    }
}

// x = log(x == 0 ? eps : x) in place
__global__ __launch_bounds__(LT) void log_eps_kernel(float* __restrict__ x, long long n, float eps) {
    for (long long i = blockIdx.x * (long long)LT + threadIdx.x; i < n; i += (long long)gridDim.x * LT) {
        const float v = x[i];
        x[i] = logf(v == 0.f ? eps : v);
    }
}

// per (utterance b, coefficient c): mean / population std over the F frames, out = (x - mean) / std.  One wave per (b, c); f64 sums.
__global__ __launch_bounds__(LT) void mvn_frames_kernel(const float* __restrict__ x, float* __restrict__ out, int F, int C, int ld_x, int ld_out, int B) {
    const int lane = threadIdx.x & 63;
    const long long job = blockIdx.x * (long long)(LT / 64) + (threadIdx.x >> 6);
    if (job >= (long long)B * C) return;
    const int c = (int)(job % C);
    const long long b = job / C;
    const float* xb = x + b * F * (long long)ld_x + c;
    double s = 0.0;
    for (int f = lane; f < F; f += 64) s += (double)xb[(long long)f * ld_x];
    const double mean = wave_sum(s) / (double)F;
    double q = 0.0;
    for (int f = lane; f < F; f += 64) { const double d = (double)xb[(long long)f * ld_x] - mean; q += d * d; }
    const double inv = 1.0 / sqrt(wave_sum(q) / (double)F);
    float* ob = out + b * F * (long long)ld_out + c;
    for (int f = lane; f < F; f += 64) ob[(long long)f * ld_out] = (float)(((double)xb[(long long)f * ld_x] - mean) * inv);
}

unsigned lgrid(long long n) { long long g = occ_cdiv(n, LT); return (unsigned)(g > 65535 * 8 ? 65535 * 8 : (g < 1 ? 1 : g)); }

}  // namespace

extern "C" int occ_lfcc_frames(const float* wav, const float* window, float* frames, int64_t B, int64_t L, int64_t F, int64_t frame_len, int64_t hop,
                               int64_t ld, float pre_emph, void* stream) {
    OCC_CHECK_ARG(wav && window && frames, "occ_lfcc_frames: null pointer");
    OCC_CHECK_ARG(B >= 1 && L >= 1 && F >= 1 && frame_len >= 1 && hop >= 1 && ld >= frame_len, "occ_lfcc_frames: bad shape");
    OCC_CHECK_ARG((F - 1) * hop + frame_len < L + hop + frame_len, "occ_lfcc_frames: F=%ld frames reach more than one hop past the %ld samples", (long)F, (long)L);
    const long long total = (long long)B * F * ld;
    hipLaunchKernelGGL(lfcc_frames_kernel, dim3(lgrid(total)), dim3(LT), 0, (hipStream_t)stream, wav, window, frames, (int)L, (int)F, (int)frame_len, (int)hop, (int)ld,
                       pre_emph, total);
    OCC_LAUNCH_CHECK("occ_lfcc_frames");
    return OCC_OK;
}

extern "C" int occ_lfcc_power(const float* spec, float* power, int64_t rows, int64_t nbins, int64_t ld_spec, int64_t ld_power, float scale, void* stream) {
    OCC_CHECK_ARG(spec && power, "occ_lfcc_power: null pointer");
    OCC_CHECK_ARG(rows >= 1 && nbins >= 1 && ld_spec >= 2 * nbins && ld_power >= nbins, "occ_lfcc_power: bad shape");
    const long long total = (long long)rows * ld_power;
    hipLaunchKernelGGL(lfcc_power_kernel, dim3(lgrid(total)), dim3(LT), 0, (hipStream_t)stream, spec, power, (int)nbins, (int)ld_spec, (int)ld_power, scale, total);
    OCC_LAUNCH_CHECK("occ_lfcc_power");
    return OCC_OK;
}

extern "C" int occ_log_eps(float* x, int64_t n, float eps, void* stream) {
    OCC_CHECK_ARG(x && n >= 1 && eps > 0.f, "occ_log_eps: bad argument");
    hipLaunchKernelGGL(log_eps_kernel, dim3(lgrid(n)), dim3(LT), 0, (hipStream_t)stream, x, (long long)n, eps);
    OCC_LAUNCH_CHECK("occ_log_eps");
    return OCC_OK;
}

extern "C" int occ_mvn_frames(const float* x, float* out, int64_t B, int64_t F, int64_t C, int64_t ld_x, int64_t ld_out, void* stream) {
    OCC_CHECK_ARG(x && out, "occ_mvn_frames: null pointer");
    OCC_CHECK_ARG(B >= 1 && F >= 1 && C >= 1 && ld_x >= C && ld_out >= C, "occ_mvn_frames: bad shape");
    hipLaunchKernelGGL(mvn_frames_kernel, dim3((unsigned)occ_cdiv(B * C, LT / 64)), dim3(LT), 0, (hipStream_t)stream, x, out, (int)F, (int)C, (int)ld_x, (int)ld_out,
                       (int)B);
    OCC_LAUNCH_CHECK("occ_mvn_frames");
    return OCC_OK;
}
