// Shared helpers for the libocc_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/occ_hip.h"

#define OCC_WAVE 64

void occ_set_error(const char* fmt, ...);

#define OCC_CHECK_ARG(cond, ...)                         \
    do {                                                 \
        if (!(cond)) {                                   \
            occ_set_error(__VA_ARGS__);                  \
            return OCC_EINVAL;                           \
        }                                                \
    } while (0)

#define OCC_LAUNCH_CHECK(name)                                                   \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            occ_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return OCC_ELAUNCH;                                                  \
        }                                                                        \
    } while (0)

typedef __hip_bfloat16 bf16_t;

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
    return __uint_as_float(((unsigned)b) << 16);
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    // plain cast keeps NaN a NaN (v_cvt_pk_bf16_f32 on gfx950)
    bf16_t h = __float2bfloat16(f);
    return *reinterpret_cast<unsigned short*>(&h);
}

// two floats -> one dword of bf16 (lo in bits 0-15): ONE v_cvt_pk_bf16_f32.  (Two scalar conversions or-ed together --
// bits(lo) | bits(hi) << 16 -- compile to four instructions a pair: cvt, cvt, shift, or.)
typedef __attribute__((ext_vector_type(2))) float occ_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 occ_bf16x2_t;
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const occ_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, occ_bf16x2_t));
}

template <typename T> __device__ __forceinline__ float occ_load_f32(const T* p);
template <> __device__ __forceinline__ float occ_load_f32<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float occ_load_f32<unsigned short>(const unsigned short* p) { return bf16_bits_to_f32(*p); }
template <typename T> __device__ __forceinline__ void occ_store_f32(T* p, float v);
template <> __device__ __forceinline__ void occ_store_f32<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void occ_store_f32<unsigned short>(unsigned short* p, float v) { *p = f32_to_bf16_bits(v); }

// ---- wave / block reductions (wave = 64 lanes) ------------------------------------------------
template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// f32: the four steps inside a 16-lane row go through DPP (quad_perm(1,0,3,2), quad_perm(2,3,0,1), row_half_mirror, row_mirror:
// VALU operand modifiers, no LDS crossbar round trip), only the two cross-row steps use ds_bpermute.  Every step pairs lanes
// symmetrically, so all 64 lanes still end with the bit-identical sum.
template <int CTRL> __device__ __forceinline__ float occ_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <> __device__ __forceinline__ float wave_sum<float>(float v) {
    v += occ_dpp<0xB1>(v); v += occ_dpp<0x4E>(v); v += occ_dpp<0x141>(v); v += occ_dpp<0x140>(v);
    v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    return v;
}
template <typename T> __device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { T u = __shfl_xor(v, o, 64); v = u > v ? u : v; }
    return v;
}
template <typename T> __device__ __forceinline__ T wave_min(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { T u = __shfl_xor(v, o, 64); v = u < v ? u : v; }
    return v;
}

// erf by Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7): one v_rcp + one v_exp + 7 FMAs, against the
// ~40-instruction libm erff -- GELU sits in GEMM epilogues where VALU time is not hidden.
__device__ __forceinline__ float fast_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.44269504088896340736f);
    const float r = fmaf(-p, e, 1.0f);
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752440f)); }
// d/dx [0.5 x (1 + erf(x/sqrt2))] = 0.5 (1 + erf(x/sqrt2)) + x * exp(-x^2/2) / sqrt(2 pi)
// (one exponential serves both terms: erf's exp(-(x/sqrt2)^2) is the density's exp(-x^2/2))
__device__ __forceinline__ float gelu_grad(float x) {
    const float ax = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.44269504088896340736f);
    const float cdf = 0.5f * (1.0f + copysignf(fmaf(-p, e, 1.0f), x));
    return fmaf(x, 0.39894228040143267794f * e, cdf);
}
// gelu(x) and gelu'(x) from one exponential and one reciprocal (the forward epilogue that also leaves gelu' for the backward pass)
__device__ __forceinline__ void gelu_fwd_grad(float x, float& y, float& dy) {
    const float ax = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.44269504088896340736f);
    const float cdf = 0.5f * (1.0f + copysignf(fmaf(-p, e, 1.0f), x));
    y = x * cdf;
    dy = fmaf(x, 0.39894228040143267794f * e, cdf);
}
__device__ __forceinline__ float selu_f(float x) {
    const float a = 1.6732632423543772848170429916717f, s = 1.0507009873554804934193349852946f;
    return x > 0.f ? s * x : s * a * (expf(x) - 1.0f);
}
__device__ __forceinline__ float selu_grad_from_in(float x) {
    const float a = 1.6732632423543772848170429916717f, s = 1.0507009873554804934193349852946f;
    return x > 0.f ? s : s * a * expf(x);
}

template <int ACT> __device__ __forceinline__ float occ_apply_act(float x) {
    if (ACT == OCC_ACT_GELU) return gelu_erf(x);
    if (ACT == OCC_ACT_SELU) return selu_f(x);
    if (ACT == OCC_ACT_RELU) return x > 0.f ? x : 0.f;
    if (ACT == OCC_ACT_TANH) return tanhf(x);
    return x;
}

static inline int64_t occ_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// device-side copy of occ_rowmap (element offsets)
// Row indices and rows_per_batch / rows_per_line fit 32 bits on every path.  The two divisions of a row index are done by
// multiplication with a host-computed magic number (floor(2^64 / d) + 1: exact for 32-bit dividends), and the only branch left
// is on rows_per_line, which is wave-uniform -- a kernel that maps four rows per thread per slab (gemm_tn) no longer runs
// divergent code paths or 32-bit division sequences in its inner loop.
struct RowMapI { long long rpb, bstride, rstride, rpl, lstride; unsigned long long mg_rpb, mg_rpl; };
static inline unsigned long long occ_div_magic(long long d) { return d <= 1 ? 0ull : (~0ull) / (unsigned long long)d + 1ull; }   // 0: divide by one
static inline RowMapI occ_make_rowmap(long long rpb, long long bstride, long long rstride, long long rpl, long long lstride) {
    return RowMapI{rpb, bstride, rstride, rpl, lstride, occ_div_magic(rpb), occ_div_magic(rpl)};
}
static inline RowMapI to_rowmap(const occ_rowmap& m) { return occ_make_rowmap(m.rows_per_batch, m.batch_stride, m.row_stride, m.rows_per_line, m.line_stride); }
__device__ __forceinline__ unsigned occ_fastdiv(unsigned n, unsigned long long magic) {
    return magic ? (unsigned)__umul64hi((unsigned long long)n, magic) : n;
}
__device__ __forceinline__ long long row_off(const RowMapI& m, long long row) {
    const unsigned rw = (unsigned)row;
    const unsigned b = occ_fastdiv(rw, m.mg_rpb), r = rw - b * (unsigned)m.rpb;
    if (m.rpl > 0) {
        const unsigned l = occ_fastdiv(r, m.mg_rpl);
        return b * m.bstride + l * m.lstride + (long long)(r - l * (unsigned)m.rpl) * m.rstride;
    }
    return b * m.bstride + (long long)r * m.rstride;
}
