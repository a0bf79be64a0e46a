// Per-tensor fp8 quantisation (OCP e4m3fn / e5m2) with delayed scaling for the fp8 MFMA path of occ_gemm (the
// XLS-R-1B configuration): quantise with the scale derived from the PREVIOUS step's |max| while recording this step's |max|.
#include "occ_common.h"

namespace {

template <typename T> __device__ __forceinline__ void ld8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void ld8<float>(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void ld8<unsigned short>(const unsigned short* p, float (&v)[8]) {
    const uint4 a = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}

// One atomic per WORKGROUP, and only when the workgroup's maximum beats the value currently in memory: atomics on one address retire
// at ~12 ns each (MI355X_MICROARCH.md, fan-in), so the first version's 16 k per-wave atomics were 190 us of a 200 us launch.  The
// plain pre-read can be stale, which only costs an unnecessary (still correct) atomic.
__device__ __forceinline__ void amax_commit(float m, float* amax) {
    __shared__ float wm[4];
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        if (m > __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(reinterpret_cast<unsigned*>(amax), __float_as_uint(m));   // non-negative floats order as their bit patterns
    }
}

// E5M2: true -> bf8 (e5m2), false -> fp8 (e4m3fn).  Values are clamped to the largest finite magnitude first (saturating conversion).
template <typename T, bool E5M2, bool QUANT>
__global__ __launch_bounds__(256) void fp8_quantize_kernel(const T* __restrict__ src, unsigned char* __restrict__ dst, long long n, const float* __restrict__ scale,
                                                           float* __restrict__ amax) {
    const float sc = QUANT && scale ? *scale : 1.f;
    const float lim = E5M2 ? 57344.f : 448.f;
    float mx = 0.f;
    const long long n8 = n / 8;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        float v[8];
        ld8<T>(src + i * 8, v);
        unsigned w[2] = {0, 0};
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            mx = fmaxf(mx, fmaxf(fabsf(v[e]), fabsf(v[e + 1])));
            if (QUANT) {
                const float a = fminf(fmaxf(v[e] * sc, -lim), lim), b = fminf(fmaxf(v[e + 1] * sc, -lim), lim);
                int cur = (int)w[e >> 2];
                if (E5M2) cur = (e & 2) ? __builtin_amdgcn_cvt_pk_bf8_f32(a, b, cur, true) : __builtin_amdgcn_cvt_pk_bf8_f32(a, b, cur, false);
                else cur = (e & 2) ? __builtin_amdgcn_cvt_pk_fp8_f32(a, b, cur, true) : __builtin_amdgcn_cvt_pk_fp8_f32(a, b, cur, false);
                w[e >> 2] = (unsigned)cur;
            }
        }
        if (QUANT) *reinterpret_cast<uint2*>(dst + i * 8) = make_uint2(w[0], w[1]);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {              // tail (n % 8 elements)
        for (long long i = n8 * 8; i < n; ++i) {
            const float x = occ_load_f32(src + i);
            mx = fmaxf(mx, fabsf(x));
            if (QUANT) {
                const float a = fminf(fmaxf(x * sc, -lim), lim);
                const int pk = E5M2 ? __builtin_amdgcn_cvt_pk_bf8_f32(a, 0.f, 0, false) : __builtin_amdgcn_cvt_pk_fp8_f32(a, 0.f, 0, false);
                dst[i] = (unsigned char)(pk & 0xff);
            }
        }
    }
    if (amax) amax_commit(mx, amax);
}

// Many tensors in one launch (the weights of a model after an optimizer step: 2 x 96 launches of 10-30 us each on XLS-R-300M, twice that on
// 1B, were latency-bound).  jobs live in device memory, sorted by first_chunk; a workgroup handles one chunk of 8192 elements of one job.
// dst == nullptr: |max| only.  Sources are bf16, n % 8 == 0.
template <bool E5M2>
__global__ __launch_bounds__(256) void fp8_quantize_batch_kernel(const occ_fp8_job* __restrict__ jobs, int n_jobs) {
    int lo = 0, hi = n_jobs - 1;
    const long long b = blockIdx.x;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (jobs[mid].first_chunk <= b) lo = mid; else hi = mid - 1; }
    const occ_fp8_job j = jobs[lo];
    const long long e0 = (b - j.first_chunk) * 8192, e1 = e0 + 8192 < j.n ? e0 + 8192 : j.n;
    const unsigned short* src = reinterpret_cast<const unsigned short*>(j.src);
    unsigned char* dst = reinterpret_cast<unsigned char*>(j.dst);
    const float sc = j.scale ? *j.scale : 1.f;
    const float lim = E5M2 ? 57344.f : 448.f;
    float mx = 0.f;
    for (long long i = e0 + threadIdx.x * 8; i < e1; i += 256 * 8) {
        float v[8];
        ld8<unsigned short>(src + i, v);
        unsigned w[2] = {0, 0};
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            mx = fmaxf(mx, fmaxf(fabsf(v[e]), fabsf(v[e + 1])));
            if (dst) {
                const float a = fminf(fmaxf(v[e] * sc, -lim), lim), c = fminf(fmaxf(v[e + 1] * sc, -lim), lim);
                int cur = (int)w[e >> 2];
                if (E5M2) cur = (e & 2) ? __builtin_amdgcn_cvt_pk_bf8_f32(a, c, cur, true) : __builtin_amdgcn_cvt_pk_bf8_f32(a, c, cur, false);
                else cur = (e & 2) ? __builtin_amdgcn_cvt_pk_fp8_f32(a, c, cur, true) : __builtin_amdgcn_cvt_pk_fp8_f32(a, c, cur, false);
                w[e >> 2] = (unsigned)cur;
            }
        }
        if (dst) *reinterpret_cast<uint2*>(dst + i) = make_uint2(w[0], w[1]);
    }
    if (j.amax) amax_commit(mx, j.amax);
}

__global__ void fp8_update_scales_kernel(float* __restrict__ amax, float* __restrict__ scale, float* __restrict__ inv, int n, float fmax, float margin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = amax[i];
    if (a > 0.f) {                       // a site with no measurement since the last update (layer skipped by layerdrop, a second refresh
        const float sc = fmax / (a * margin);   // without a step in between) keeps its delayed scale instead of falling back to 1
        scale[i] = sc; inv[i] = 1.f / sc;
    }
    amax[i] = 0.f;
}

template <bool QUANT>
int launch_q(const void* src, int src_dtype, void* dst, int fmt, int64_t n, const float* scale, float* amax, hipStream_t s) {
    long long blocks = occ_cdiv(occ_cdiv(n, 8), 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
#define OCC_Q(T, E) hipLaunchKernelGGL((fp8_quantize_kernel<T, E, QUANT>), dim3((unsigned)blocks), dim3(256), 0, s, (const T*)src, (unsigned char*)dst, (long long)n, scale, amax)
    if (src_dtype == OCC_F32) { if (fmt == OCC_FP8_E5M2) OCC_Q(float, true); else OCC_Q(float, false); }
    else { if (fmt == OCC_FP8_E5M2) OCC_Q(unsigned short, true); else OCC_Q(unsigned short, false); }
#undef OCC_Q
    return OCC_OK;
}

}  // namespace

extern "C" {

int occ_fp8_quantize(const void* src, int src_dtype, void* dst, int fmt, int64_t n, const float* scale, float* amax, void* stream) {
    OCC_CHECK_ARG(src && dst && n >= 1, "occ_fp8_quantize: bad argument");
    OCC_CHECK_ARG((src_dtype == OCC_F32 || src_dtype == OCC_BF16) && (fmt == OCC_FP8_E4M3 || fmt == OCC_FP8_E5M2), "occ_fp8_quantize: src f32 / bf16, fmt e4m3 / e5m2");
    OCC_CHECK_ARG(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 7) == 0, "occ_fp8_quantize: alignment");
    launch_q<true>(src, src_dtype, dst, fmt, n, scale, amax, (hipStream_t)stream);
    OCC_LAUNCH_CHECK("occ_fp8_quantize");
    return OCC_OK;
}

int occ_fp8_quantize_batch(const occ_fp8_job* jobs_dev, int64_t n_jobs, int64_t total_chunks, int fmt, void* stream) {
    OCC_CHECK_ARG(jobs_dev && n_jobs >= 1 && n_jobs < (1 << 20) && total_chunks >= 1 && total_chunks < (1ll << 31) && (fmt == OCC_FP8_E4M3 || fmt == OCC_FP8_E5M2),
                  "occ_fp8_quantize_batch: bad argument");
    if (fmt == OCC_FP8_E5M2) hipLaunchKernelGGL(fp8_quantize_batch_kernel<true>, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)stream, jobs_dev, (int)n_jobs);
    else hipLaunchKernelGGL(fp8_quantize_batch_kernel<false>, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)stream, jobs_dev, (int)n_jobs);
    OCC_LAUNCH_CHECK("occ_fp8_quantize_batch");
    return OCC_OK;
}

int occ_fp8_amax(const void* src, int src_dtype, int64_t n, float* amax, void* stream) {
    OCC_CHECK_ARG(src && amax && n >= 1 && (src_dtype == OCC_F32 || src_dtype == OCC_BF16) && ((uintptr_t)src & 15) == 0, "occ_fp8_amax: bad argument");
    launch_q<false>(src, src_dtype, nullptr, OCC_FP8_E4M3, n, nullptr, amax, (hipStream_t)stream);
    OCC_LAUNCH_CHECK("occ_fp8_amax");
    return OCC_OK;
}

int occ_fp8_update_scales(float* amax, float* scale, float* inv_scale, int64_t n, float fmax, float margin, void* stream) {
    OCC_CHECK_ARG(amax && scale && inv_scale && n >= 1 && fmax > 0.f && margin > 0.f, "occ_fp8_update_scales: bad argument");
    hipLaunchKernelGGL(fp8_update_scales_kernel, dim3((unsigned)occ_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, amax, scale, inv_scale, (int)n, fmax, margin);
    OCC_LAUNCH_CHECK("occ_fp8_update_scales");
    return OCC_OK;
}

}  // extern "C"
