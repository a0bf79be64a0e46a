// First conv block of wav2vec2 on the matrix cores: Conv1d(1 -> 512, k = 10, stride 5) + bias -> LayerNorm(512) -> GELU, channels-last bf16 out
// (fairseq ConvFeatureExtractionModel block 0 in "layer_norm" mode, reached from sslassist.py:48), and its backward.
//
// The 512 x 10 contraction is a K = 10 GEMM; the VALU form (frontend.hip conv0_ln_gelu_kernel: one wave per frame, a lane owns 8 channels)
// spends 80 FMAs per frame and lane on it and runs four dependent wave reductions per frame.  Here a wave takes 16 frames at a time:
//
//   D[frame][channel] = A[frame][kappa] . B[kappa][channel]      one v_mfma_f32_16x16x32_bf16 per block of 16 channels, 32 per step
//
// with f32-grade precision from split operands: x = xh + xl, w = wh + wl (bf16 each, |lo| <= 2^-9 |hi|), and the 32 kappa slots hold
//   kappa  0.. 9   xh[tap] . wh[tap]
//   kappa 10..19   xl[tap] . wh[tap]
//   kappa 20..29   xh[tap] . wl[tap]          (xl . wl, 2^-16 relative, is dropped)
//   kappa 30, 31   1 . bias_hi, 1 . bias_lo   (the bias rides along)
// Block blk multiplies channels {c * 32 + blk : c = 0..15}: lane (c = lane & 15, g = lane >> 4) then holds, for each of its four frames
// 4g .. 4g+3, the 32 CONSECUTIVE channels c*32 .. c*32+31 -- 64 contiguous bytes of the bf16 output row, stored without a transposing pass.
// LayerNorm statistics are sums over a lane's 32 values and over the 16 lanes of a DPP row (4 row operations for 4 frames at once).
#include "occ_common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

constexpr int CM_K = 10;

__device__ __forceinline__ float row16_sum(float v) {      // sum over the 16 lanes of a DPP row; every lane of the row gets it
    v += occ_dpp<0xB1>(v); v += occ_dpp<0x4E>(v); v += occ_dpp<0x141>(v); v += occ_dpp<0x140>(v);
    return v;
}

// kappa slot -> (tap, kind): kind 0 = hi part, 1 = lo part, 2 = constant one
__device__ __forceinline__ void kappa_slot(int kappa, int& tap, int& kind) {
    if (kappa < 10) { tap = kappa; kind = 0; }
    else if (kappa < 20) { tap = kappa - 10; kind = 1; }
    else if (kappa < 30) { tap = kappa - 20; kind = 0; }
    else { tap = 0; kind = 2; }
}

__device__ __forceinline__ unsigned short hi_bits(float v) { return f32_to_bf16_bits(v); }
__device__ __forceinline__ unsigned short lo_bits(float v) { return f32_to_bf16_bits(v - bf16_bits_to_f32(f32_to_bf16_bits(v))); }

// LDS image of the weight operand: [blk 0..31][c 0..15][4 chunks of 8 kappa], chunk position (q ^ (c & 3)) so that the 16 lanes of a row
// (stride 64 B) spread over the banks.  Row (blk, c) = channel c*32 + blk.
__device__ __forceinline__ void build_w_image(unsigned short* wimg, const float* __restrict__ w, const float* __restrict__ bias, int tid) {
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {                      // two rows per thread: the row's ten taps are ten contiguous floats
        const int row = tid * 2 + rr;
        const int blk = row >> 4, c = row & 15, ch = c * 32 + blk;
        float t[CM_K];
#pragma unroll
        for (int i = 0; i < CM_K / 2; ++i) { const float2 v = *reinterpret_cast<const float2*>(w + ch * CM_K + 2 * i); t[2 * i] = v.x; t[2 * i + 1] = v.y; }
        const float bv = bias[ch];
        unsigned short v[32];
#pragma unroll
        for (int k = 0; k < CM_K; ++k) { v[k] = hi_bits(t[k]); v[10 + k] = v[k]; v[20 + k] = lo_bits(t[k]); }
        v[30] = hi_bits(bv); v[31] = lo_bits(bv);
        uint4* dst = reinterpret_cast<uint4*>(wimg + row * 32);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            dst[q ^ (c & 3)] = make_uint4((unsigned)v[q * 8] | ((unsigned)v[q * 8 + 1] << 16), (unsigned)v[q * 8 + 2] | ((unsigned)v[q * 8 + 3] << 16),
                                          (unsigned)v[q * 8 + 4] | ((unsigned)v[q * 8 + 5] << 16), (unsigned)v[q * 8 + 6] | ((unsigned)v[q * 8 + 7] << 16));
    }
}

// the A operand of one step: lane (fr = frame of the step, g = kappa chunk) builds its eight kappa values from the staged waveform
__device__ __forceinline__ bf16x8 build_x_frag(const float* smp, int frame_local, int stride, int g) {
    unsigned short v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        int tap, kind;
        kappa_slot(g * 8 + e, tap, kind);
        const float s = smp[frame_local * stride + tap];
        v[e] = kind == 0 ? hi_bits(s) : (kind == 1 ? lo_bits(s) : (unsigned short)0x3f80);
    }
    uint4 u = make_uint4((unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16), (unsigned)v[4] | ((unsigned)v[5] << 16),
                         (unsigned)v[6] | ((unsigned)v[7] << 16));
    return __builtin_bit_cast(bf16x8, u);
}

// CM_FRAMES: frames per workgroup (4 waves x CM_FRAMES / 64 steps x 16 frames); the weight image is built once per workgroup
template <int CM_FRAMES>
__global__ __launch_bounds__(256, 2) void conv0_mfma_kernel(const float* __restrict__ wav, const float* __restrict__ w, const float* __restrict__ bias,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, unsigned short* __restrict__ out,
                                                            int L, int Tout, int stride, float eps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
    unsigned short* wimg = reinterpret_cast<unsigned short*>(sm_raw);                       // 32 KiB
    float* smp = reinterpret_cast<float*>(sm_raw + 32768);                                  // the workgroup's waveform tile
    const int smp_bytes = (((CM_FRAMES - 1) * stride + CM_K) * 4 + 15) & ~15;
    const int b = blockIdx.y, f0 = blockIdx.x * CM_FRAMES;
    const int nsamp = (CM_FRAMES - 1) * stride + CM_K;
    const float* wb = wav + (size_t)b * L;
    for (int i = threadIdx.x; i < nsamp; i += 256) {
        const int gi = f0 * stride + i;
        smp[i] = gi < L ? wb[gi] : 0.f;
    }
    build_w_image(wimg, w, bias, threadIdx.x);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    // LayerNorm scale / shift in LDS as float4 [blk / 4][c]: the 16 lanes of a row read 16 consecutive float4 (the 64 values a lane needs
    // would cost 64 registers; the kernel then spills)
    float4* gl = reinterpret_cast<float4*>(sm_raw + 32768 + smp_bytes + 4 * 8192);
    float4* bl = gl + 128;
    if (threadIdx.x < 128) {
        const int q4 = threadIdx.x >> 4, cc = threadIdx.x & 15;
        gl[threadIdx.x] = *reinterpret_cast<const float4*>(gamma + cc * 32 + q4 * 4);
        bl[threadIdx.x] = *reinterpret_cast<const float4*>(beta + cc * 32 + q4 * 4);
    }
    __syncthreads();
    const uint4* wq = reinterpret_cast<const uint4*>(wimg) + c * 4 + (g ^ (c & 3));       // + blk * 64 (uint4 units): this lane's chunk of row (blk, c)
    for (int step = 0; step < CM_FRAMES / 64; ++step) {
        const int fl0 = wave * (CM_FRAMES / 4) + step * 16;                // first frame of the step, local to the workgroup
        if (f0 + fl0 >= Tout) break;                                       // wave-uniform
        const bf16x8 xa = build_x_frag(smp, fl0 + c, stride, g);
        f32x4 acc[32];
#pragma unroll
        for (int blk = 0; blk < 32; ++blk) {
            const uint4 wv = wq[blk * 64];
            acc[blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, __builtin_bit_cast(bf16x8, wv), (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
        // acc[blk][e] = pre-activation of frame fl0 + 4g + e, channel c*32 + blk
        float mean[4], rstd[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float s = 0.f;
#pragma unroll
            for (int blk = 0; blk < 32; ++blk) s += acc[blk][e];
            mean[e] = s;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) mean[e] = row16_sum(mean[e]) * (1.0f / 512.0f);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float q = 0.f;
#pragma unroll
            for (int blk = 0; blk < 32; ++blk) { const float d = acc[blk][e] - mean[e]; q = fmaf(d, d, q); }
            rstd[e] = q;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) rstd[e] = 1.0f / sqrtf(row16_sum(rstd[e]) * (1.0f / 512.0f) + eps);
        // Output: a lane's 64 bytes per frame sit 64 B apart across the row's 16 lanes -- stored directly, every instruction would touch 64
        // separate cache lines.  Eight frames at a time (e = 0, 1 of the four lane groups, then e = 2, 3) go through a wave-private 8 KiB
        // LDS tile [slot = 2g + (e & 1)][1 KiB row] and leave as whole rows, one KiB per instruction.  Within a lane's four 16-byte chunks
        // the position is permuted by (c >> 1) & 3 so that the eight lanes a ds_write_b128 serves together hit eight bank groups.
        uint4* tile = reinterpret_cast<uint4*>(sm_raw + 32768 + smp_bytes) + wave * 512;
        const int sw = (c >> 1) & 3;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int eh = 0; eh < 2; ++eh) {
                const int e = half * 2 + eh;
                unsigned pk[16];
#pragma unroll
                for (int q4 = 0; q4 < 8; ++q4) {
                    const float4 gv = gl[q4 * 16 + c], bv = bl[q4 * 16 + c];
                    const float gg[4] = {gv.x, gv.y, gv.z, gv.w}, bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                    for (int u = 0; u < 4; u += 2) {
                        const int blk = q4 * 4 + u;
                        const float y0 = gelu_erf((acc[blk][e] - mean[e]) * rstd[e] * gg[u] + bb[u]);
                        const float y1 = gelu_erf((acc[blk + 1][e] - mean[e]) * rstd[e] * gg[u + 1] + bb[u + 1]);
                        pk[blk >> 1] = pack_bf16x2(y0, y1);
                    }
                }
                uint4* row = tile + (2 * g + eh) * 64 + c * 4;
                row[0 ^ sw] = make_uint4(pk[0], pk[1], pk[2], pk[3]); row[1 ^ sw] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
                row[2 ^ sw] = make_uint4(pk[8], pk[9], pk[10], pk[11]); row[3 ^ sw] = make_uint4(pk[12], pk[13], pk[14], pk[15]);
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int pos = (lane & ~3) | ((lane & 3) ^ ((lane >> 3) & 3));          // where this lane's chunk of the row was put
#pragma unroll
            for (int slot = 0; slot < 8; ++slot) {
                const int f = f0 + fl0 + 4 * (slot >> 1) + half * 2 + (slot & 1);
                const uint4 v = tile[slot * 64 + pos];
                if (f < Tout) reinterpret_cast<uint4*>(out + ((size_t)b * Tout + f) * 512)[lane] = v;
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------- backward
// d(loss)/d(w, bias, gamma, beta) of the block above from dact = d(loss)/d(output); everything is recomputed from the waveform.
// The four waves of a workgroup share each step's 16 frames and split the channels: wave wv owns blocks 8 wv .. 8 wv + 7, i.e. for lane
// (c, g) the 8 consecutive channels c*32 + 8 wv .. + 7 of the frames 4g .. 4g+3 (one 16-byte piece of every dact row).  The four per-frame sums
// over all 512 channels (mean, variance, and the two of LayerNorm's backward) are row sums (DPP) + an exchange of the four waves' partial sums
// through LDS, three workgroup barriers per step.  The weight gradient is one more MFMA family:
//     dW^T[tap][channel] += A[tap][frame] . B[frame][channel]        v_mfma_f32_16x16x16_bf16, K = the step's 16 frames
// whose B operand -- lane (c, g) holds frames 4g .. 4g+3 of channel c -- is exactly the layout the gradient wrt the conv output is computed
// in (no transposition), and whose A operand is the waveform again: rows 0-9 the taps, row 10 ones (-> the bias gradient).  Split operands as
// in the forward pass: xh.dh + xl.dh + xh.dl.  A lane accumulates dgamma / dbeta of its 8 channels in registers.  A workgroup walks many
// frame tiles (persistent grid) and adds its sums to the global gradients once, with float atomics, as the VALU kernel did per 256 frames.
typedef __attribute__((ext_vector_type(4))) short bf16x4s;
template <typename T> struct LVec8;
template <> struct LVec8<float> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
};
template <> struct LVec8<unsigned short> {
    static __device__ __forceinline__ void load(const unsigned short* p, float (&v)[8]) {
        const uint4 a = *reinterpret_cast<const uint4*>(p);
        const unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    }
};
constexpr int CB_TILE = 128;            // frames per tile: 8 steps of 16

template <typename TD>
__global__ __launch_bounds__(256, 2) void conv0_bwd_mfma_kernel(const float* __restrict__ wav, const float* __restrict__ w, const float* __restrict__ bias,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta, const TD* __restrict__ dact,
                                                                float* __restrict__ dw, float* __restrict__ dbias, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, int L, int Tout, int stride, float eps, int tiles_per_utt, int total_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
    unsigned short* wimg = reinterpret_cast<unsigned short*>(sm_raw);                       // 32 KiB
    float* smp = reinterpret_cast<float*>(sm_raw + 32768);
    const int nsamp = (CB_TILE - 1) * stride + CM_K;
    float* xbuf = reinterpret_cast<float*>(sm_raw + 32768 + ((nsamp * 4 + 15) & ~15));     // [4 exchanges][4 waves][16 frames]
    build_w_image(wimg, w, bias, threadIdx.x);
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int ch0 = c * 32 + wv * 8;                       // this lane's 8 channels
    float dgam[8], dbet[8];
    // gamma / beta of the lane's 8 channels are read from LDS at each use (16 registers that decide whether the kernel spills)
    float* gbl = xbuf + 4 * 4 * 16;                          // [512 gamma][512 beta]
    for (int i = threadIdx.x; i < 512; i += 256) { gbl[i] = gamma[i]; gbl[512 + i] = beta[i]; }
    const float* gr = gbl + ch0; const float* br = gbl + 512 + ch0;
    f32x4 dwacc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { dwacc[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; dgam[j] = 0.f; dbet[j] = 0.f; }
    const uint4* wq = reinterpret_cast<const uint4*>(wimg) + (wv * 8) * 64 + c * 4 + (g ^ (c & 3));      // + j * 64: row (blk = 8 wv + j, c)

    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        const int b = tile / tiles_per_utt, f0 = (tile - b * tiles_per_utt) * CB_TILE;
        const float* wb = wav + (size_t)b * L;
        __syncthreads();                                   // the previous tile's readers of smp are done (and the weight image is complete)
        for (int i = threadIdx.x; i < nsamp; i += 256) {
            const int gi = f0 * stride + i;
            smp[i] = gi < L ? wb[gi] : 0.f;
        }
        __syncthreads();
        for (int step = 0; step < CB_TILE / 16; ++step) {
            const int fl0 = step * 16;
            if (f0 + fl0 >= Tout) break;                   // workgroup-uniform
            // dact of this lane's (frame, 8 channels) pieces, issued first: they are consumed after two exchanges
            float dv[4][8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int f = f0 + fl0 + 4 * g + e;
                if (f < Tout) LVec8<TD>::load(dact + ((size_t)b * Tout + f) * 512 + ch0, dv[e]);
                else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) dv[e][j] = 0.f;
                }
            }
            const bf16x8 xa = build_x_frag(smp, fl0 + c, stride, g);
            f32x4 acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, __builtin_bit_cast(bf16x8, wq[j * 64]), (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            // ---- mean
            float t4[4], mean[4], rstd[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) s += acc[j][e];
                t4[e] = row16_sum(s);
            }
            if (c == 0) *reinterpret_cast<float4*>(xbuf + (0 * 4 + wv) * 16 + 4 * g) = make_float4(t4[0], t4[1], t4[2], t4[3]);
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e)
                mean[e] = (xbuf[(0 * 4 + 0) * 16 + 4 * g + e] + xbuf[(0 * 4 + 1) * 16 + 4 * g + e] + xbuf[(0 * 4 + 2) * 16 + 4 * g + e] + xbuf[(0 * 4 + 3) * 16 + 4 * g + e]) * (1.0f / 512.0f);
            // ---- variance
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float q = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = acc[j][e] - mean[e]; q = fmaf(d, d, q); }
                t4[e] = row16_sum(q);
            }
            if (c == 0) *reinterpret_cast<float4*>(xbuf + (1 * 4 + wv) * 16 + 4 * g) = make_float4(t4[0], t4[1], t4[2], t4[3]);
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e)
                rstd[e] = 1.0f / sqrtf((xbuf[(1 * 4 + 0) * 16 + 4 * g + e] + xbuf[(1 * 4 + 1) * 16 + 4 * g + e] + xbuf[(1 * 4 + 2) * 16 + 4 * g + e] + xbuf[(1 * 4 + 3) * 16 + 4 * g + e]) * (1.0f / 512.0f) + eps);
            // ---- through GELU and the LayerNorm affine; the two sums of LayerNorm's backward
            float s1[4], s2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = (acc[j][e] - mean[e]) * rstd[e];
                    const float dz = dv[e][j] * gelu_grad(fmaf(xh, gr[j], br[j]));
                    dgam[j] = fmaf(dz, xh, dgam[j]); dbet[j] += dz;
                    const float dx = dz * gr[j];
                    a1 += dx; a2 = fmaf(dx, xh, a2);
                    acc[j][e] = xh; dv[e][j] = dx;
                }
                s1[e] = row16_sum(a1); s2[e] = row16_sum(a2);
                __builtin_amdgcn_sched_barrier(0);         // (keeps the four frames' chains from being interleaved into a register spill)
            }
            if (c == 0) {
                *reinterpret_cast<float4*>(xbuf + (2 * 4 + wv) * 16 + 4 * g) = make_float4(s1[0], s1[1], s1[2], s1[3]);
                *reinterpret_cast<float4*>(xbuf + (3 * 4 + wv) * 16 + 4 * g) = make_float4(s2[0], s2[1], s2[2], s2[3]);
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s1[e] = (xbuf[(2 * 4 + 0) * 16 + 4 * g + e] + xbuf[(2 * 4 + 1) * 16 + 4 * g + e] + xbuf[(2 * 4 + 2) * 16 + 4 * g + e] + xbuf[(2 * 4 + 3) * 16 + 4 * g + e]) * (1.0f / 512.0f);
                s2[e] = (xbuf[(3 * 4 + 0) * 16 + 4 * g + e] + xbuf[(3 * 4 + 1) * 16 + 4 * g + e] + xbuf[(3 * 4 + 2) * 16 + 4 * g + e] + xbuf[(3 * 4 + 3) * 16 + 4 * g + e]) * (1.0f / 512.0f);
            }
            // ---- the weight-gradient operands: A = the waveform (rows: taps 0-9, row 10 ones), lane (row c, frames 4g .. 4g+3)
            bf16x4s ah, al;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float sv = c < CM_K ? smp[(fl0 + 4 * g + e) * stride + c] : 0.f;
                ah[e] = (short)(c < CM_K ? hi_bits(sv) : (c == CM_K ? (unsigned short)0x3f80 : (unsigned short)0));
                al[e] = (short)(c < CM_K ? lo_bits(sv) : (unsigned short)0);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                bf16x4s dh, dl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dp = rstd[e] * (dv[e][j] - s1[e] - acc[j][e] * s2[e]);
                    dh[e] = (short)hi_bits(dp); dl[e] = (short)lo_bits(dp);
                }
                dwacc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, dl, dwacc[j], 0, 0, 0);
                dwacc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, dh, dwacc[j], 0, 0, 0);
                dwacc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, dh, dwacc[j], 0, 0, 0);
            }
        }
    }
    // ---- this workgroup's sums -> the global gradients.  dwacc[j][e] = dW^T[tap 4g+e][channel c*32 + 8 wv + j] (row 10: the bias gradient)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ch = ch0 + j;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int tap = 4 * g + e;
            if (tap < CM_K) atomicAdd(dw + (size_t)ch * CM_K + tap, dwacc[j][e]);
            else if (tap == CM_K) atomicAdd(dbias + ch, dwacc[j][e]);
        }
        float a = dgam[j], bb = dbet[j];
        a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
        bb += __shfl_xor(bb, 16, 64); bb += __shfl_xor(bb, 32, 64);
        if (g == 0) { atomicAdd(dgamma + ch, a); atomicAdd(dbeta + ch, bb); }
    }
}

}  // namespace

// called from occ_conv0_ln_gelu (frontend.hip) for bf16 output, k = 10, C = 512
int occ_conv0_mfma_launch(const float* wav, const float* w, const float* bias, const float* gamma, const float* beta, void* out, long long B, long long L,
                          long long Tout, long long stride, float eps, hipStream_t s) {
    static const int fr_env = getenv("OCC_C0_FRAMES") ? atoi(getenv("OCC_C0_FRAMES")) : 256;
    const int fr = fr_env >= 1024 ? 1024 : (fr_env >= 512 ? 512 : 256);
    const dim3 grid((unsigned)occ_cdiv(Tout, fr), (unsigned)B), block(256);
    const size_t shm = 32768 + (size_t)((((fr - 1) * stride + CM_K) * 4 + 15) & ~15) + 4 * 8192 + 4096;       // weight image, waveform tile, four 8 KiB store tiles, gamma / beta
    if (fr == 1024) hipLaunchKernelGGL(conv0_mfma_kernel<1024>, grid, block, shm, s, wav, w, bias, gamma, beta, (unsigned short*)out, (int)L, (int)Tout, (int)stride, eps);
    else if (fr == 512) hipLaunchKernelGGL(conv0_mfma_kernel<512>, grid, block, shm, s, wav, w, bias, gamma, beta, (unsigned short*)out, (int)L, (int)Tout, (int)stride, eps);
    else hipLaunchKernelGGL(conv0_mfma_kernel<256>, grid, block, shm, s, wav, w, bias, gamma, beta, (unsigned short*)out, (int)L, (int)Tout, (int)stride, eps);
    return 0;
}

// called from occ_conv0_ln_gelu_bwd (frontend_bwd.hip) for k = 10, C = 512
int occ_conv0_bwd_mfma_launch(const float* wav, const float* w, const float* bias, const float* gamma, const float* beta, const void* dact, int dact_bf16,
                              float* dw, float* dbias, float* dgamma, float* dbeta, long long B, long long L, long long Tout, long long stride, float eps,
                              hipStream_t s) {
    const long long tpu = occ_cdiv(Tout, CB_TILE), total = tpu * B;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v; }
    static const int per_cu = getenv("OCC_C0B_WGS_PER_CU") ? atoi(getenv("OCC_C0B_WGS_PER_CU")) : 2;
    long long wgs = (long long)cus * per_cu;
    if (wgs > total) wgs = total;
    const size_t shm = 32768 + (size_t)((((CB_TILE - 1) * stride + CM_K) * 4 + 15) & ~15) + (4 * 4 * 16 + 1024) * sizeof(float);
    if (dact_bf16)
        hipLaunchKernelGGL(conv0_bwd_mfma_kernel<unsigned short>, dim3((unsigned)wgs), dim3(256), shm, s, wav, w, bias, gamma, beta, (const unsigned short*)dact, dw, dbias,
                           dgamma, dbeta, (int)L, (int)Tout, (int)stride, eps, (int)tpu, (int)total);
    else
        hipLaunchKernelGGL(conv0_bwd_mfma_kernel<float>, dim3((unsigned)wgs), dim3(256), shm, s, wav, w, bias, gamma, beta, (const float*)dact, dw, dbias, dgamma, dbeta,
                           (int)L, (int)Tout, (int)stride, eps, (int)tpu, (int)total);
    return 0;
}
