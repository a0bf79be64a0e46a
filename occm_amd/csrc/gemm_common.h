// Shared pieces of the occ_gemm kernel family (gemm.hip: the 128x128 kernels and the occ_gemm entry point; gemm_p8.hip: the 256x256
// eight-phase kernel): argument block, epilogues, LDS-DMA / fragment-read helpers.
#pragma once
#include "occ_common.h"
#include <stdlib.h>

namespace occ_gemm_detail {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

constexpr int TM = 128, TN = 128, THREADS = 256;
constexpr int SLAB_BYTES = 128;                 // K bytes per row per slab
constexpr int CHUNKS = SLAB_BYTES / 16;         // 8 chunks of 16 B per row

struct GemmArgs {
    long long M, N, K;
    const char* X; RowMapI xmap; long long nseg, seg_len, seg_stride;
    const char* W; long long ldw;
    const float* bias;
    const char* R; RowMapI rmap; int r_dtype;
    char* C; RowMapI cmap; int c_dtype;
    int act; float alpha;
    const float* dq_a; const float* dq_w;   // fp8 operands: device scalars that undo the per-tensor quantisation scales (alpha *= *dq_a * *dq_w), or null
    unsigned short* aux;
    float* colsum_part;       // optional [2 * nbm][N] partial column sums of a bf16 result (row epilogue of the 256-row kernel)
    unsigned char* f8_out; const float* f8_scale; float* f8_amax; int f8_e5m2;   // optional fp8 copy of a bf16 result (row epilogue of the 256-row kernel)
    int nbm, nbn;
    int group_m;              // >0: walk GROUP_M m-tiles per n-tile before moving on (L2-sized working set), 0: n fastest
    long long a_gstride, w_gstride, c_gstride;
    int ngroups;             // persistent kernel: groups folded into the tile id
    int ksplit;              // > 1: the K range is cut into ksplit pieces handled by different workgroups, C += alpha*acc with f32 atomics
    int slabs_per_split;
    int tile_rows;           // default 128x128 kernels: rows per output tile (multiple of 16, <= 128; the LDS image stays 128 rows) -- see occ_gemm
    int q4_mode, q4_first, q4_delay;   // gemm_q4.hip: start stagger of the presumably second workgroup of a CU (speed only)
#ifdef P8_DIAG
    unsigned long long* diag;   // scripts/diag_p8.hip only: per workgroup {clock at entry, after the prologue, after the K loop, at exit, realtime entry, realtime exit}
#endif
};

__device__ __forceinline__ float act_rt(int act, float v) {
    switch (act) {
        case OCC_ACT_GELU: case OCC_ACT_GELU_KEEP_GRAD: return gelu_erf(v);
        case OCC_ACT_SELU: return selu_f(v);
        case OCC_ACT_RELU: return v > 0.f ? v : 0.f;
        case OCC_ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// Straight-line epilogue for the combinations launched thousands of times per step (no side
// tensor, activation none / GELU, residual none / f32): everything wave-uniform is a template parameter, the bias row is loaded
// once per 16-column block instead of once per 16x16 block.  The generic gemm_epilogue below handles every other combination
// with run-time switches; on the 128x128 tile that code executed ~1500 instructions per thread, which (with the workgroups of a
// round reaching it together) was a third of a K = 1024 GEMM's run time.
// AUXM: 0 no side tensor; 1 store the bf16 pre-activation (acc + bias) to aux (forward of fc1, kept for backward); 2 multiply by
// GELU'(aux) (the input gradient through fc1's activation); 3 store bf16 gelu'(acc + bias) to aux and apply GELU (GELU must be false: the
// pair comes from one exponential); 4 multiply by aux as it is.
template <int NJ, bool HASB, bool GELU, bool HASR, bool CBF, int AUXM = 0>
__device__ __forceinline__ void gemm_epilogue_fast(const GemmArgs& a, f32x4 (&acc)[4][NJ], long long mrow0, long long ncol0, int fr, int fq, long long cshift,
                                                   const f32x4* breg, long long mlim) {
    f32x4 bv[4];
    if (HASB) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (breg) { bv[i] = breg[i]; continue; }
            long long n = ncol0 + i * 16 + fq * 4; if (n > a.N - 4) n = a.N - 4;       // N % 4 == 0; out-of-range columns are never stored
            bv[i] = *reinterpret_cast<const f32x4*>(a.bias + cshift + n);
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const long long m = mrow0 + j * 16 + fr;
        if (m >= mlim) continue;
        const long long coff = row_off(a.cmap, m) + cshift;
        const long long roff = HASR ? row_off(a.rmap, m) + cshift : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long n = ncol0 + i * 16 + fq * 4;
            if (n >= a.N) continue;
            f32x4 v = acc[i][j];
            if (HASB) v += bv[i];
            if (AUXM == 1) {
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2*>(a.aux + coff + n) = o;
            }
            if (AUXM == 2) {
                const uint2 u = *reinterpret_cast<const uint2*>(a.aux + coff + n);
                v[0] *= gelu_grad(bf16_bits_to_f32((unsigned short)(u.x & 0xffff))); v[1] *= gelu_grad(bf16_bits_to_f32((unsigned short)(u.x >> 16)));
                v[2] *= gelu_grad(bf16_bits_to_f32((unsigned short)(u.y & 0xffff))); v[3] *= gelu_grad(bf16_bits_to_f32((unsigned short)(u.y >> 16)));
            }
            if (AUXM == 3) {
                float gq[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { float y; gelu_fwd_grad(v[e], y, gq[e]); v[e] = y; }
                uint2 o;
                o.x = pack_bf16x2(gq[0], gq[1]);
                o.y = pack_bf16x2(gq[2], gq[3]);
                *reinterpret_cast<uint2*>(a.aux + coff + n) = o;
            }
            if (AUXM == 4) {
                const uint2 u = *reinterpret_cast<const uint2*>(a.aux + coff + n);
                v[0] *= __uint_as_float(u.x << 16); v[1] *= __uint_as_float(u.x & 0xffff0000u); v[2] *= __uint_as_float(u.y << 16); v[3] *= __uint_as_float(u.y & 0xffff0000u);
            }
            if (GELU) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
            if (HASR) v += *reinterpret_cast<const f32x4*>(a.R + (roff + n) * 4);
            if (CBF) {
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2*>(a.C + (coff + n) * 2) = o;
            } else {
                *reinterpret_cast<f32x4*>(a.C + (coff + n) * 4) = v;
            }
        }
    }
}

// Split-K epilogue: C (f32) += alpha * acc with float atomics; used for weight-gradient GEMMs whose output has only a few dozen
// tiles while K is the whole batch (the caller passes R == C, i.e. "accumulate"; the pieces add onto what C holds).
template <int NJ>
__device__ __forceinline__ void gemm_epilogue_atomic(const GemmArgs& a, f32x4 (&acc)[4][NJ], long long mrow0, long long ncol0, int fr, int fq, long long cshift,
                                                     long long mlim_in = -1) {
    const long long mlim = mlim_in >= 0 ? mlim_in : a.M;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const long long m = mrow0 + j * 16 + fr;
        if (m >= mlim) continue;
        float* crow = reinterpret_cast<float*>(a.C) + row_off(a.cmap, m) + cshift;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long n = ncol0 + i * 16 + fq * 4;
            if (n >= a.N) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(crow + n + e, acc[i][j][e] * a.alpha);
        }
    }
}

// acc[i][j]: i = 16-column block of the wave's 64 output columns, j = 16-row block of its NJ*16 output rows;
// mrow0 / ncol0 = first row / column of the wave's sub-tile.  A lane owns C[m][n..n+3].
template <int NJ>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, f32x4 (&acc)[4][NJ], long long mrow0, long long ncol0, int fr, int fq, long long cshift,
                                              const f32x4* breg = nullptr, long long mlim_in = -1) {
    const long long mlim = mlim_in >= 0 ? mlim_in : a.M;        // rows this call may store: a short tile stops at its own end
    // the two side-tensor forms of the fine-tuning step (fc1 forward keeps its pre-activation; fc2's input gradient goes through GELU')
    if (a.alpha == 1.0f && a.aux && !a.R && a.c_dtype != OCC_F32) {
        if (a.act == OCC_ACT_GELU && a.bias) { gemm_epilogue_fast<NJ, true, true, false, true, 1>(a, acc, mrow0, ncol0, fr, fq, cshift, breg, mlim); return; }
        if (a.act == OCC_ACT_GELU_GRAD && !a.bias) { gemm_epilogue_fast<NJ, false, false, false, true, 2>(a, acc, mrow0, ncol0, fr, fq, cshift, breg, mlim); return; }
        if (a.act == OCC_ACT_GELU_KEEP_GRAD && a.bias) { gemm_epilogue_fast<NJ, true, false, false, true, 3>(a, acc, mrow0, ncol0, fr, fq, cshift, breg, mlim); return; }
        if (a.act == OCC_ACT_MUL_AUX && !a.bias) { gemm_epilogue_fast<NJ, false, false, false, true, 4>(a, acc, mrow0, ncol0, fr, fq, cshift, breg, mlim); return; }
    }
    if (a.alpha == 1.0f && !a.aux && (a.act == OCC_ACT_NONE || a.act == OCC_ACT_GELU) && (!a.R || a.r_dtype == OCC_F32)) {
        // wave-uniform flags -> one scalar branch chain into a straight-line instantiation
        const int key = (a.bias ? 8 : 0) | (a.act == OCC_ACT_GELU ? 4 : 0) | (a.R ? 2 : 0) | (a.c_dtype != OCC_F32 ? 1 : 0);
#define OCC_EPI(K, B, G, R, C) case K: gemm_epilogue_fast<NJ, B, G, R, C>(a, acc, mrow0, ncol0, fr, fq, cshift, breg, mlim); break;
        switch (key) {
            OCC_EPI(0, false, false, false, false) OCC_EPI(1, false, false, false, true) OCC_EPI(2, false, false, true, false) OCC_EPI(3, false, false, true, true)
            OCC_EPI(4, false, true, false, false) OCC_EPI(5, false, true, false, true) OCC_EPI(6, false, true, true, false) OCC_EPI(7, false, true, true, true)
            OCC_EPI(8, true, false, false, false) OCC_EPI(9, true, false, false, true) OCC_EPI(10, true, false, true, false) OCC_EPI(11, true, false, true, true)
            OCC_EPI(12, true, true, false, false) OCC_EPI(13, true, true, false, true) OCC_EPI(14, true, true, true, false) OCC_EPI(15, true, true, true, true)
        }
#undef OCC_EPI
        return;
    }
    const float al = a.alpha * (a.dq_a ? *a.dq_a : 1.f) * (a.dq_w ? *a.dq_w : 1.f);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const long long m = mrow0 + j * 16 + fr;
        if (m >= mlim) continue;
        const long long coff = row_off(a.cmap, m) + cshift;
        const long long roff = a.R ? row_off(a.rmap, m) + cshift : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long n = ncol0 + i * 16 + fq * 4;
            if (n >= a.N) continue;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] * al;
            if (breg) {                                                 // bias already in registers (staged through LDS by the caller)
                v[0] += breg[i][0]; v[1] += breg[i][1]; v[2] += breg[i][2]; v[3] += breg[i][3];
            } else if (a.bias) {
                const float4 bv = *reinterpret_cast<const float4*>(a.bias + cshift + n);
                v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
            }
            if (a.aux && a.act == OCC_ACT_GELU) {                       // keep the pre-activation for backward
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2*>(a.aux + coff + n) = o;
            }
            if (a.act == OCC_ACT_GELU_GRAD) {
                const uint2 u = *reinterpret_cast<const uint2*>(a.aux + coff + n);
                v[0] *= gelu_grad(bf16_bits_to_f32((unsigned short)(u.x & 0xffff))); v[1] *= gelu_grad(bf16_bits_to_f32((unsigned short)(u.x >> 16)));
                v[2] *= gelu_grad(bf16_bits_to_f32((unsigned short)(u.y & 0xffff))); v[3] *= gelu_grad(bf16_bits_to_f32((unsigned short)(u.y >> 16)));
            } else if (a.act == OCC_ACT_MUL_AUX) {
                const uint2 u = *reinterpret_cast<const uint2*>(a.aux + coff + n);
                v[0] *= __uint_as_float(u.x << 16); v[1] *= __uint_as_float(u.x & 0xffff0000u); v[2] *= __uint_as_float(u.y << 16); v[3] *= __uint_as_float(u.y & 0xffff0000u);
            } else if (a.act == OCC_ACT_GELU_KEEP_GRAD) {
                float gq[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { float y; gelu_fwd_grad(v[e], y, gq[e]); v[e] = y; }
                if (a.aux) {
                    uint2 o;
                    o.x = pack_bf16x2(gq[0], gq[1]);
                    o.y = pack_bf16x2(gq[2], gq[3]);
                    *reinterpret_cast<uint2*>(a.aux + coff + n) = o;
                }
            } else if (a.act != OCC_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act_rt(a.act, v[e]);
            }
            if (a.R) {
                if (a.r_dtype == OCC_F32) {
                    const float4 rv = *reinterpret_cast<const float4*>(a.R + (roff + n) * 4);
                    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                } else {
                    const uint2 rv = *reinterpret_cast<const uint2*>(a.R + (roff + n) * 2);
                    v[0] += bf16_bits_to_f32((unsigned short)(rv.x & 0xffff)); v[1] += bf16_bits_to_f32((unsigned short)(rv.x >> 16));
                    v[2] += bf16_bits_to_f32((unsigned short)(rv.y & 0xffff)); v[3] += bf16_bits_to_f32((unsigned short)(rv.y >> 16));
                }
            }
            if (a.c_dtype == OCC_F32) {
                *reinterpret_cast<float4*>(a.C + (coff + n) * 4) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2*>(a.C + (coff + n) * 2) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Row-layout epilogue.  In the MFMA result layout a lane owns 4 consecutive columns of ONE row per accumulator, so a store instruction
// of the epilogues above touches 16 rows with 32-64 bytes each; measured on the 256x256 kernel (scripts/diag_p8.hip) that store pattern
// made the epilogue 18 k (bf16 out) to 36 k cycles (f32 out + f32 residual) per tile against 36.7 k cycles for a whole K = 1024 loop.
// Here every wave sends its accumulators through a private LDS region in chunks of 32 rows x 64 columns (f32, 272-byte rows: the pad
// keeps the 16-byte writes of 8 rows on disjoint banks) and reads them back with a ROW per 16 lanes, so that bias, side tensor,
// residual and C are all accessed as whole contiguous row segments (64 columns = 256 B f32 / 128 B bf16, four rows per instruction).
// lds_wave: >= 8704 bytes private to the calling wave, 16-byte aligned, not in use by anything else.
template <int NJ, bool HASB, bool GELU, bool HASR, bool CBF, int AUXM>
__device__ __forceinline__ void gemm_epilogue_rows_t(const GemmArgs& a, f32x4 (&acc)[4][NJ], long long mrow0, long long ncol0, int lane, unsigned char* lds_wave, int rows,
                                                     int part_row) {
    static_assert(NJ % 2 == 0, "row epilogue works on pairs of 16-row blocks");
    constexpr int RS = 272;
    constexpr int NCH = NJ / 2;
    const int fr = lane & 15, fq = lane >> 4;
    const bool plain_c = a.cmap.rpl == 0 && a.cmap.rpb >= a.M, plain_r = !HASR || (a.rmap.rpl == 0 && a.rmap.rpb >= a.M);
    const float al = a.alpha * (a.dq_a ? *a.dq_a : 1.f) * (a.dq_w ? *a.dq_w : 1.f);       // (1.0 for the bf16 path: the product below is exact)
    // Side-tensor reads (the saved pre-activation of the GELU' form, the f32 residual) are issued for a whole 32-row chunk BEFORE the
    // chunk's LDS round trip and one chunk AHEAD of their use, from row / column indices clamped into the matrix so that no load sits
    // behind a bounds branch: behind `if (row valid)` every one of a lane's 16-32 loads was its own HBM round trip in program order
    // (out-proj at bs 64 with cold operands: 82 us, of which the K loop is 18).
    if constexpr (CBF && !HASR) {
        // bf16 results (and the bf16 side tensor): EIGHT columns per lane, a row per 8 lanes -- 16-byte stores, eight rows per instruction
        const int rr = lane >> 3, cc = lane & 7;
        const long long n = ncol0 + cc * 8;
        f32x4 b0 = (f32x4){0.f, 0.f, 0.f, 0.f}, b1 = b0;
        const bool whole = n + 8 <= a.N;
        if (HASB && n < a.N) { b0 = *reinterpret_cast<const f32x4*>(a.bias + n); if (whole) b1 = *reinterpret_cast<const f32x4*>(a.bias + n + 4); }
        const long long nld = n + 8 <= a.N ? n : (a.N >= 8 ? a.N - 8 : 0);        // a column group that lies inside the row (its values are unused when n is not whole)
        float f8max = 0.f;
        const float f8sc = a.f8_out && a.f8_scale ? *a.f8_scale : 1.f;
        float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        uint4 ux[4];                               // this chunk's four side-tensor vectors; slot k is refilled for the next chunk right after its use
        auto load_aux = [&](const int ch, const int k) {
            long long m = mrow0 + ch * 32 + k * 8 + rr; if (m > a.M - 1) m = a.M - 1;
            return *reinterpret_cast<const uint4*>(a.aux + (plain_c ? m * a.cmap.rstride : row_off(a.cmap, m)) + nld);
        };
        if ((AUXM == 2 || AUXM == 4) && a.N >= 8) {
#pragma unroll
            for (int k = 0; k < 4; ++k) ux[k] = load_aux(0, k);
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<f32x4*>(lds_wave + (jj * 16 + fr) * RS + i * 64 + fq * 16) = acc[i][ch * 2 + jj];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int rloc = k * 8 + rr;
                f32x4 v0 = *reinterpret_cast<const f32x4*>(lds_wave + rloc * RS + cc * 32), v1 = *reinterpret_cast<const f32x4*>(lds_wave + rloc * RS + cc * 32 + 16);
                const long long m = mrow0 + ch * 32 + rloc;
                uint4 u = make_uint4(0, 0, 0, 0);
                if ((AUXM == 2 || AUXM == 4) && a.N >= 8) { u = ux[k]; if (ch + 1 < NCH) ux[k] = load_aux(ch + 1, k); }
                if (m >= a.M || n >= a.N || ch * 32 + rloc >= rows) continue;
                const long long coff = (plain_c ? m * a.cmap.rstride : row_off(a.cmap, m)) + n;
                v0 = v0 * al; v1 = v1 * al;
                if (HASB) { v0 += b0; v1 += b1; }
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                auto pack = [&](uint4& o) {
                    o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
                    o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
                };
                if (AUXM == 1) {
                    uint4 o; pack(o);
                    if (whole) *reinterpret_cast<uint4*>(a.aux + coff) = o;
                    else { *reinterpret_cast<uint2*>(a.aux + coff) = make_uint2(o.x, o.y); }          // N % 4 == 0: a ragged last group is 4 columns
                }
                if (AUXM == 2) {
                    if (!whole) { const uint2 t = *reinterpret_cast<const uint2*>(a.aux + coff); u = make_uint4(t.x, t.y, 0, 0); }       // ragged last column group (rare): its own load
                    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[2 * e] *= gelu_grad(bf16_bits_to_f32((unsigned short)(w[e] & 0xffff))); v[2 * e + 1] *= gelu_grad(bf16_bits_to_f32((unsigned short)(w[e] >> 16))); }
                }
                if (AUXM == 3) {                    // gelu and gelu' from one exponential; the derivative (bf16) is what backward multiplies by
                    float gq[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) { float y; gelu_fwd_grad(v[e], y, gq[e]); v[e] = y; }
                    uint4 og;
                    og.x = pack_bf16x2(gq[0], gq[1]); og.y = pack_bf16x2(gq[2], gq[3]);
                    og.z = pack_bf16x2(gq[4], gq[5]); og.w = pack_bf16x2(gq[6], gq[7]);
                    if (whole) *reinterpret_cast<uint4*>(a.aux + coff) = og;
                    else *reinterpret_cast<uint2*>(a.aux + coff) = make_uint2(og.x, og.y);
                }
                if (AUXM == 4) {
                    if (!whole) { const uint2 t = *reinterpret_cast<const uint2*>(a.aux + coff); u = make_uint4(t.x, t.y, 0, 0); }
                    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[2 * e] *= __uint_as_float(w[e] << 16); v[2 * e + 1] *= __uint_as_float(w[e] & 0xffff0000u); }
                }
                if (GELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
                }
                uint4 o; pack(o);
                if (whole) *reinterpret_cast<uint4*>(a.C + coff * 2) = o;
                else *reinterpret_cast<uint2*>(a.C + coff * 2) = make_uint2(o.x, o.y);
                if (a.colsum_part) {                // column sums of the bf16-rounded result (the bias gradient a colsum pass over C would give)
                    const unsigned wc_[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) { csum[2 * e] += __uint_as_float(wc_[e] << 16); csum[2 * e + 1] += __uint_as_float(wc_[e] & 0xffff0000u); }
                }
                if (a.f8_out && whole) {            // fp8 of the bf16-rounded values (what a stand-alone quantisation pass would read), saturating
                    const unsigned w8[4] = {o.x, o.y, o.z, o.w};
                    float qv[8];
                    const float lim = a.f8_e5m2 ? 57344.f : 448.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lo = __uint_as_float(w8[e] << 16), hi = __uint_as_float(w8[e] & 0xffff0000u);
                        f8max = fmaxf(f8max, fmaxf(fabsf(lo), fabsf(hi)));
                        qv[2 * e] = fminf(fmaxf(lo * f8sc, -lim), lim); qv[2 * e + 1] = fminf(fmaxf(hi * f8sc, -lim), lim);
                    }
                    int q0, q1;
                    if (a.f8_e5m2) {
                        q0 = __builtin_amdgcn_cvt_pk_bf8_f32(qv[0], qv[1], 0, false); q1 = __builtin_amdgcn_cvt_pk_bf8_f32(qv[4], qv[5], 0, false);
                        q0 = __builtin_amdgcn_cvt_pk_bf8_f32(qv[2], qv[3], q0, true); q1 = __builtin_amdgcn_cvt_pk_bf8_f32(qv[6], qv[7], q1, true);
                    } else {
                        q0 = __builtin_amdgcn_cvt_pk_fp8_f32(qv[0], qv[1], 0, false); q1 = __builtin_amdgcn_cvt_pk_fp8_f32(qv[4], qv[5], 0, false);
                        q0 = __builtin_amdgcn_cvt_pk_fp8_f32(qv[2], qv[3], q0, true); q1 = __builtin_amdgcn_cvt_pk_fp8_f32(qv[6], qv[7], q1, true);
                    }
                    *reinterpret_cast<uint2*>(a.f8_out + m * a.N + n) = make_uint2((unsigned)q0, (unsigned)q1);
                }
            }
        }
        if (a.colsum_part) {                        // the 8 row groups of the wave (lane >> 3) hold the same 8 columns: add them up, lanes 0-7 store
#pragma unroll
            for (int e = 0; e < 8; ++e) { csum[e] += __shfl_xor(csum[e], 8, 64); csum[e] += __shfl_xor(csum[e], 16, 64); csum[e] += __shfl_xor(csum[e], 32, 64); }
            if (rr == 0 && whole) {
                float* pr = a.colsum_part + (long long)part_row * a.N + n;
                *reinterpret_cast<f32x4*>(pr) = (f32x4){csum[0], csum[1], csum[2], csum[3]};
                *reinterpret_cast<f32x4*>(pr + 4) = (f32x4){csum[4], csum[5], csum[6], csum[7]};
            }
        }
        if (a.f8_out && a.f8_amax) {                // one guarded atomic per wave (non-negative floats order as their bit patterns)
            f8max = wave_max(f8max);
            if (lane == 0 && f8max > __hip_atomic_load(a.f8_amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(reinterpret_cast<unsigned*>(a.f8_amax), __float_as_uint(f8max));
        }
        return;
    }
    const int rr = lane >> 4, cc = lane & 15;                  // read-back: row rr (+4k) of the chunk, columns 4*cc .. 4*cc+3
    const long long n = ncol0 + cc * 4;
    f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (HASB && n < a.N) bv = *reinterpret_cast<const f32x4*>(a.bias + n);
    const long long nld = n < a.N ? n : 0;                      // N % 4 == 0: a lane's four columns are inside the row or all outside it
    f32x4 rx[HASR ? 8 : 1];                        // this chunk's eight residual vectors; slot k is refilled for the next chunk right after its use
    auto load_res = [&](const int ch, const int k) {
        long long m = mrow0 + ch * 32 + k * 4 + rr; if (m > a.M - 1) m = a.M - 1;
        return *reinterpret_cast<const f32x4*>(a.R + ((plain_r ? m * a.rmap.rstride : row_off(a.rmap, m)) + nld) * 4);
    };
    if constexpr (HASR) {
#pragma unroll
        for (int k = 0; k < 8; ++k) rx[k] = load_res(0, k);
    }
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<f32x4*>(lds_wave + (jj * 16 + fr) * RS + i * 64 + fq * 16) = acc[i][ch * 2 + jj];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            f32x4 v = *reinterpret_cast<const f32x4*>(lds_wave + (k * 4 + rr) * RS + cc * 16);
            const long long m = mrow0 + ch * 32 + k * 4 + rr;
            f32x4 rv = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (HASR) { rv = rx[k]; if (ch + 1 < NCH) rx[k] = load_res(ch + 1, k); }
            if (m >= a.M || n >= a.N || ch * 32 + k * 4 + rr >= rows) continue;
            const long long coff = (plain_c ? m * a.cmap.rstride : row_off(a.cmap, m)) + n;
            v = v * al;
            if (HASB) v += bv;
            if (AUXM == 1) {
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2*>(a.aux + coff) = o;
            }
            if (AUXM == 2) {
                const uint2 u = *reinterpret_cast<const uint2*>(a.aux + coff);
                v[0] *= gelu_grad(bf16_bits_to_f32((unsigned short)(u.x & 0xffff))); v[1] *= gelu_grad(bf16_bits_to_f32((unsigned short)(u.x >> 16)));
                v[2] *= gelu_grad(bf16_bits_to_f32((unsigned short)(u.y & 0xffff))); v[3] *= gelu_grad(bf16_bits_to_f32((unsigned short)(u.y >> 16)));
            }
            if (AUXM == 3) {
                float gq[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { float y; gelu_fwd_grad(v[e], y, gq[e]); v[e] = y; }
                uint2 o;
                o.x = pack_bf16x2(gq[0], gq[1]);
                o.y = pack_bf16x2(gq[2], gq[3]);
                *reinterpret_cast<uint2*>(a.aux + coff) = o;
            }
            if (AUXM == 4) {
                const uint2 u = *reinterpret_cast<const uint2*>(a.aux + coff);
                v[0] *= __uint_as_float(u.x << 16); v[1] *= __uint_as_float(u.x & 0xffff0000u); v[2] *= __uint_as_float(u.y << 16); v[3] *= __uint_as_float(u.y & 0xffff0000u);
            }
            if (GELU) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
            if (HASR) v += rv;
            if (CBF) {
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2*>(a.C + coff * 2) = o;
            } else {
                *reinterpret_cast<f32x4*>(a.C + coff * 4) = v;
            }
        }
    }
}

// Dispatcher: the combinations the front-end launches (no side tensor or one of the two fine-tuning forms; activation none / GELU;
// residual none / f32) take the row-layout epilogue, anything else the generic one.  ncol0 / the wave's columns must lie inside one
// 64-column span (true for every kernel of this family: a wave owns 64 output columns).  rows: only the first `rows` rows of the wave's
// NJ * 16 are stored (the 224-row tile passes 112) -- honoured by the row-layout forms only, see rows_epilogue_applies.
__host__ __device__ inline bool rows_epilogue_applies(const GemmArgs& a) {
    if (a.R && a.r_dtype != OCC_F32) return false;
    if (a.aux) return !a.R && a.c_dtype != OCC_F32 && (((a.act == OCC_ACT_GELU || a.act == OCC_ACT_GELU_KEEP_GRAD) && a.bias) || ((a.act == OCC_ACT_GELU_GRAD || a.act == OCC_ACT_MUL_AUX) && !a.bias));
    return a.act == OCC_ACT_NONE || a.act == OCC_ACT_GELU;
}
template <int NJ>
__device__ __forceinline__ void gemm_epilogue_rows(const GemmArgs& a, f32x4 (&acc)[4][NJ], long long mrow0, long long ncol0, int lane, long long cshift,
                                                   unsigned char* lds_wave, int rows = NJ * 16, int part_row = 0) {
    const int fr = lane & 15, fq = lane >> 4;
    if (cshift == 0 && (!a.R || a.r_dtype == OCC_F32)) {
        if (a.aux && !a.R && a.c_dtype != OCC_F32) {
            if (a.act == OCC_ACT_GELU && a.bias) { gemm_epilogue_rows_t<NJ, true, true, false, true, 1>(a, acc, mrow0, ncol0, lane, lds_wave, rows, part_row); return; }
            if (a.act == OCC_ACT_GELU_GRAD && !a.bias) { gemm_epilogue_rows_t<NJ, false, false, false, true, 2>(a, acc, mrow0, ncol0, lane, lds_wave, rows, part_row); return; }
            if (a.act == OCC_ACT_GELU_KEEP_GRAD && a.bias) { gemm_epilogue_rows_t<NJ, true, false, false, true, 3>(a, acc, mrow0, ncol0, lane, lds_wave, rows, part_row); return; }
            if (a.act == OCC_ACT_MUL_AUX && !a.bias) { gemm_epilogue_rows_t<NJ, false, false, false, true, 4>(a, acc, mrow0, ncol0, lane, lds_wave, rows, part_row); return; }
        }
        if (!a.aux && (a.act == OCC_ACT_NONE || a.act == OCC_ACT_GELU)) {
            const int key = (a.bias ? 8 : 0) | (a.act == OCC_ACT_GELU ? 4 : 0) | (a.R ? 2 : 0) | (a.c_dtype != OCC_F32 ? 1 : 0);
#define OCC_EPR(K, B, G, R, C) case K: gemm_epilogue_rows_t<NJ, B, G, R, C, 0>(a, acc, mrow0, ncol0, lane, lds_wave, rows, part_row); break;
            switch (key) {
                OCC_EPR(0, false, false, false, false) OCC_EPR(1, false, false, false, true) OCC_EPR(2, false, false, true, false) OCC_EPR(3, false, false, true, true)
                OCC_EPR(4, false, true, false, false) OCC_EPR(5, false, true, false, true) OCC_EPR(6, false, true, true, false) OCC_EPR(7, false, true, true, true)
                OCC_EPR(8, true, false, false, false) OCC_EPR(9, true, false, false, true) OCC_EPR(10, true, false, true, false) OCC_EPR(11, true, false, true, true)
                OCC_EPR(12, true, true, false, false) OCC_EPR(13, true, true, false, true) OCC_EPR(14, true, true, true, false) OCC_EPR(15, true, true, true, true)
            }
#undef OCC_EPR
            return;
        }
    }
    gemm_epilogue<NJ>(a, acc, mrow0, ncol0, fr, fq, cshift);
}

// MODE 0: f32 operands, exact-f32 MFMA.  MODE 1: bf16 operands, bf16 MFMA.  MODE 2: f32 operands in memory, rounded to
// bf16 while they are staged into LDS, bf16 MFMA (f32 accumulate) -- the back-end's "bf16 compute" mode, which
// needs no bf16 copies of f32 activations / gradients.

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int OFF> __device__ __forceinline__ u32x4 ds_read128(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// N fragments of 16 rows each, 2 KiB apart (16 rows x 128 B)
template <int N> __device__ __forceinline__ void read_frags(u32x4 (&f)[N], unsigned addr) {
    f[0] = ds_read128<0>(addr); f[1] = ds_read128<2048>(addr); f[2] = ds_read128<4096>(addr); f[3] = ds_read128<6144>(addr);
    if constexpr (N == 8) {
        f[4] = ds_read128<8192>(addr); f[5] = ds_read128<10240>(addr); f[6] = ds_read128<12288>(addr); f[7] = ds_read128<14336>(addr);
    }
}
template <int N> __device__ __forceinline__ void wait_vm_then_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

inline int cu_count() {
    static int n = 0;
    if (!n) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}


// 256x256 eight-phase kernel (gemm_p8.hip); the caller has checked: bf16 operands, K % 64 == 0, one K segment, one group
void gemm_p8_launch(GemmArgs& a, hipStream_t s, int fmt = 0, int tile_rows = 256);      // tile_rows 224: only with rows_epilogue_applies(a)
// 256x128 four-wave kernel, two workgroups per CU (gemm_q4.hip); same preconditions, bf16 only
void gemm_q4_launch(GemmArgs& a, hipStream_t s, int tile_rows = 256);

}  // namespace occ_gemm_detail
