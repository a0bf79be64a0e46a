// MFMA GEMM family for gfx950:  C = act(alpha * X.W^T + bias) + R
//   X: [M,K] activations addressed through an occ_rowmap (+ optional K-segments), K contiguous
//   W: [N,K] weights in torch Linear layout, K contiguous
// Every nn.Linear and Conv1d of the wav2vec2 front-end (reached from sslassist.py:48) and the Linear /
// 1x1-conv layers of the AASIST back-end (sslassist.py:448-471) map onto this kernel.
//
// Tiling (per workgroup of 4 waves): 128(m) x 128(n) output tile, 128-byte K slabs (64 bf16 / 32 f32),
// double-buffered in LDS (2 x 2 x 16 KiB), register-staged global->LDS copies issued one slab ahead,
// XOR-swizzled 16-byte chunks so the ds_read_b128 fragment reads spread over the banks.  Each wave
// owns a 64x64 sub-tile as 4x4 MFMA 16x16 blocks.  The MFMA is issued with W as the A operand and X as
// the B operand, so a lane ends up with 4 consecutive n of one output row m: bias/residual loads and
// the C store are 8/16-byte vector accesses.
// bf16 inputs use v_mfma_f32_16x16x32_bf16 (f32 accumulate); f32 inputs use the exact-f32
// v_mfma_f32_16x16x4_f32 with the k index permuted identically on both operands.
#include "occ_common.h"
#include <stdlib.h>

namespace {

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
    unsigned short* aux;
    int nbm, nbn;
    int group_m;              // >0: walk GROUP_M m-tiles per n-tile before moving on (L2-sized working set), 0: n fastest
    long long a_gstride, w_gstride, c_gstride;
    int ngroups;             // persistent kernel: groups folded into the tile id
    int ksplit;              // > 1: the K range is cut into ksplit pieces handled by different workgroups, C += alpha*acc with f32 atomics
    int slabs_per_split;
    int dbg;                 // ablation bits (timing experiments only, results wrong): 1 no loads in the K loop, 2 no MFMA, 4 no fragment reads
};

__device__ __forceinline__ float act_rt(int act, float v) {
    switch (act) {
        case OCC_ACT_GELU: return gelu_erf(v);
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
template <int NJ, bool HASB, bool GELU, bool HASR, bool CBF>
__device__ __forceinline__ void gemm_epilogue_fast(const GemmArgs& a, f32x4 (&acc)[4][NJ], long long mrow0, long long ncol0, int fr, int fq, long long cshift,
                                                   const f32x4* breg) {
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
        if (m >= a.M) continue;
        const long long coff = row_off(a.cmap, m) + cshift;
        const long long roff = HASR ? row_off(a.rmap, m) + cshift : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long n = ncol0 + i * 16 + fq * 4;
            if (n >= a.N) continue;
            f32x4 v = acc[i][j];
            if (HASB) v += bv[i];
            if (GELU) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
            if (HASR) v += *reinterpret_cast<const f32x4*>(a.R + (roff + n) * 4);
            if (CBF) {
                uint2 o;
                o.x = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
                o.y = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
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
__device__ __forceinline__ void gemm_epilogue_atomic(const GemmArgs& a, f32x4 (&acc)[4][NJ], long long mrow0, long long ncol0, int fr, int fq, long long cshift) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const long long m = mrow0 + j * 16 + fr;
        if (m >= a.M) continue;
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
                                              const f32x4* breg = nullptr) {
    if (a.alpha == 1.0f && !a.aux && (a.act == OCC_ACT_NONE || a.act == OCC_ACT_GELU) && (!a.R || a.r_dtype == OCC_F32)) {
        // wave-uniform flags -> one scalar branch chain into a straight-line instantiation
        const int key = (a.bias ? 8 : 0) | (a.act == OCC_ACT_GELU ? 4 : 0) | (a.R ? 2 : 0) | (a.c_dtype != OCC_F32 ? 1 : 0);
#define OCC_EPI(K, B, G, R, C) case K: gemm_epilogue_fast<NJ, B, G, R, C>(a, acc, mrow0, ncol0, fr, fq, cshift, breg); break;
        switch (key) {
            OCC_EPI(0, false, false, false, false) OCC_EPI(1, false, false, false, true) OCC_EPI(2, false, false, true, false) OCC_EPI(3, false, false, true, true)
            OCC_EPI(4, false, true, false, false) OCC_EPI(5, false, true, false, true) OCC_EPI(6, false, true, true, false) OCC_EPI(7, false, true, true, true)
            OCC_EPI(8, true, false, false, false) OCC_EPI(9, true, false, false, true) OCC_EPI(10, true, false, true, false) OCC_EPI(11, true, false, true, true)
            OCC_EPI(12, true, true, false, false) OCC_EPI(13, true, true, false, true) OCC_EPI(14, true, true, true, false) OCC_EPI(15, true, true, true, true)
        }
#undef OCC_EPI
        return;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const long long m = mrow0 + j * 16 + fr;
        if (m >= a.M) continue;
        const long long coff = row_off(a.cmap, m) + cshift;
        const long long roff = a.R ? row_off(a.rmap, m) + cshift : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long n = ncol0 + i * 16 + fq * 4;
            if (n >= a.N) continue;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] * a.alpha;
            if (breg) {                                                 // bias already in registers (staged through LDS by the caller)
                v[0] += breg[i][0]; v[1] += breg[i][1]; v[2] += breg[i][2]; v[3] += breg[i][3];
            } else if (a.bias) {
                const float4 bv = *reinterpret_cast<const float4*>(a.bias + cshift + n);
                v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
            }
            if (a.aux && a.act == OCC_ACT_GELU) {                       // keep the pre-activation for backward
                uint2 o;
                o.x = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
                o.y = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
                *reinterpret_cast<uint2*>(a.aux + coff + n) = o;
            }
            if (a.act == OCC_ACT_GELU_GRAD) {
                const uint2 u = *reinterpret_cast<const uint2*>(a.aux + coff + n);
                v[0] *= gelu_grad(bf16_bits_to_f32((unsigned short)(u.x & 0xffff))); v[1] *= gelu_grad(bf16_bits_to_f32((unsigned short)(u.x >> 16)));
                v[2] *= gelu_grad(bf16_bits_to_f32((unsigned short)(u.y & 0xffff))); v[3] *= gelu_grad(bf16_bits_to_f32((unsigned short)(u.y >> 16)));
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
                o.x = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
                o.y = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
                *reinterpret_cast<uint2*>(a.C + (coff + n) * 2) = o;
            }
        }
    }
}

// MODE 0: f32 operands, exact-f32 MFMA.  MODE 1: bf16 operands, bf16 MFMA.  MODE 2: f32 operands in memory, rounded to
// bf16 while they are staged into LDS, bf16 MFMA (f32 accumulate) -- the back-end's "bf16 compute" mode, which
// needs no bf16 copies of f32 activations / gradients.
__device__ __forceinline__ uint4 pack_bf16x8(const float4 lo, const float4 hi) {
    return make_uint4((unsigned)f32_to_bf16_bits(lo.x) | ((unsigned)f32_to_bf16_bits(lo.y) << 16),
                      (unsigned)f32_to_bf16_bits(lo.z) | ((unsigned)f32_to_bf16_bits(lo.w) << 16),
                      (unsigned)f32_to_bf16_bits(hi.x) | ((unsigned)f32_to_bf16_bits(hi.y) << 16),
                      (unsigned)f32_to_bf16_bits(hi.z) | ((unsigned)f32_to_bf16_bits(hi.w) << 16));
}

// TNT = 128: 2x2 waves, 64x64 per wave, 64 KiB LDS, 2 workgroups per CU.  TNT = 64 (narrow outputs: the back-ends' 32/64-channel
// convolutions): 4x1 waves, 32x64 per wave, 48 KiB LDS, 3 workgroups per CU, no MFMA work on columns that do not exist.
template <int MODE, int TNT = 128>
__global__ __launch_bounds__(THREADS, TNT == 128 ? 2 : 3) void gemm_kernel(const GemmArgs a) {
    constexpr int NJ = TNT == 128 ? 4 : 2;      // 16-row blocks per wave
    constexpr int WPASS = TNT / 32;             // staging passes over the W rows
    constexpr bool BF16 = MODE != 0;            // MFMA flavour
    constexpr int ES = MODE == 1 ? 2 : 4;       // element size of X in memory (MODE 3: X f32, W bf16)
    constexpr int WES = (MODE == 1 || MODE == 3) ? 2 : 4;
    constexpr int CE = BF16 ? 8 : 4;            // K elements per 16-B LDS chunk
    constexpr int SLAB_K = BF16 ? 64 : 32;      // K elements per slab
    __shared__ uint4 lds[2][(TM + TNT) * CHUNKS];   // [buffer][X rows, then W rows][row*8 + swizzled chunk]

    // ---- XCD-aware tile id: blocks b, b+8, ... share an XCD (and its L2); give each XCD a contiguous
    // range of tiles so neighbouring tiles re-use the same X / W panels out of that L2.
    const int total = a.nbm * a.nbn;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = total >> 3, r8 = total & 7;
    const int vid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const int tile_n = vid % a.nbn, tile_m = vid / a.nbn;
    const long long m0 = (long long)tile_m * TM, n0 = (long long)tile_n * TNT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = TNT == 128 ? (wave & 1) : wave, wn = TNT == 128 ? (wave >> 1) : 0;
    const long long grp = blockIdx.y;
    const char* Xg = a.X + grp * a.a_gstride * ES;
    const char* Wg = a.W + grp * a.w_gstride * WES;
    const long long cshift = grp * a.c_gstride;      // column shift of C / R / bias for this group

    // ---- staging assignment: this thread copies chunk `ch` of rows (tid>>3) + 32*i
    const int ch = tid & 7, srow = tid >> 3;
    long long xoff[4], woff[WPASS];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        long long m = m0 + srow + 32 * i; if (m > a.M - 1) m = a.M - 1;
        xoff[i] = row_off(a.xmap, m) * ES;
    }
#pragma unroll
    for (int i = 0; i < WPASS; ++i) {
        long long n = n0 + srow + 32 * i; if (n > a.N - 1) n = a.N - 1;
        woff[i] = n * a.ldw * WES;
    }
    const int nslab = (int)((a.K + SLAB_K - 1) / SLAB_K);

    uint4 px[4], pw[WPASS];
    auto issue_loads = [&](int slab) {
        const long long k0 = (long long)slab * SLAB_K + ch * CE;
        long long kx = k0;
        if (a.nseg > 1) { const long long sg = k0 / a.seg_len; kx = sg * a.seg_stride + (k0 - sg * a.seg_len); }
        const bool ok = k0 < a.K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (MODE == 2 || MODE == 3) {
                if (ok) {
                    const float4* xp = reinterpret_cast<const float4*>(Xg + xoff[i] + kx * ES);
                    px[i] = pack_bf16x8(xp[0], xp[1]);
                    if (i < WPASS) {
                        if constexpr (MODE == 2) {
                            const float4* wp = reinterpret_cast<const float4*>(Wg + woff[i % WPASS] + k0 * WES);
                            pw[i % WPASS] = pack_bf16x8(wp[0], wp[1]);
                        } else {
                            pw[i % WPASS] = *reinterpret_cast<const uint4*>(Wg + woff[i % WPASS] + k0 * WES);
                        }
                    }
                } else { px[i] = make_uint4(0, 0, 0, 0); if (i < WPASS) pw[i % WPASS] = px[i]; }
            } else {
                px[i] = ok ? *reinterpret_cast<const uint4*>(Xg + xoff[i] + kx * ES) : make_uint4(0, 0, 0, 0);
                if (i < WPASS) pw[i % WPASS] = ok ? *reinterpret_cast<const uint4*>(Wg + woff[i % WPASS] + k0 * ES) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto write_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = srow + 32 * i;
            const int idx = row * CHUNKS + (ch ^ (row & 7));
            lds[buf][idx] = px[i];
            if (i < WPASS) lds[buf][TM * CHUNKS + idx] = pw[i % WPASS];
        }
    };

    f32x4 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    issue_loads(0);
    write_lds(0);
    __syncthreads();

    const int fr = lane & 15, fq = lane >> 4;
    for (int slab = 0; slab < nslab; ++slab) {
        const int cur = slab & 1;
        if (slab + 1 < nslab) issue_loads(slab + 1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            uint4 wf[4], xf[NJ];
            const int chk = kb * 4 + fq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rw = wn * 64 + i * 16 + fr;
                wf[i] = lds[cur][TM * CHUNKS + rw * CHUNKS + (chk ^ (rw & 7))];
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int rx = wm * (NJ * 16) + j * 16 + fr;
                xf[j] = lds[cur][rx * CHUNKS + (chk ^ (rx & 7))];
            }
            if constexpr (BF16) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            *reinterpret_cast<bf16x8*>(&wf[i]), *reinterpret_cast<bf16x8*>(&xf[j]), acc[i][j], 0, 0, 0);
            } else {
                // lane group fq holds k = 4*fq + e of this 16-k block; step e multiplies k in {e, 4+e, 8+e, 12+e}
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                reinterpret_cast<const float*>(&wf[i])[e], reinterpret_cast<const float*>(&xf[j])[e], acc[i][j], 0, 0, 0);
            }
        }
        if (slab + 1 < nslab) write_lds(cur ^ 1);
        __syncthreads();
    }

    gemm_epilogue<NJ>(a, acc, m0 + wm * (NJ * 16), n0 + wn * 64, fr, fq, cshift);
}

// ------------------------------------------------------------------------------------------------
// bf16 fast path (K % 64 == 0): same 128x128 tile and fragment layout, but the slab goes global -> LDS by
// LDS-DMA (global_load_lds_dwordx4: no staging VGPRs, no ds_write), ONE 32 KiB LDS buffer and <= 128
// VGPRs, so 4 workgroups share a CU and hide each other's load latency (occupancy does the pipelining).
// The LDS image must be lane-linear per wave-instruction, so the XOR swizzle is applied to the per-lane
// SOURCE chunk instead: LDS[row][p] = G[row][p ^ (row & 7)]  <=>  chunk c of a row sits at p = c ^ (row & 7).
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

// TMT = 128: 4 waves (2x2), 32 KiB LDS, 4 workgroups per CU.  TMT = 256: 8 waves (4x2) on a 256x128 tile, 48 KiB LDS,
// 2 workgroups per CU: the same 16 waves per CU but 25 % fewer L2->LDS bytes per FLOP (the 128x128 tile moves one byte
// per 64 FLOP, which is about what a CU can pull from L2 at its MFMA rate).
template <int TMT, bool SPLITK = false>
__global__ __launch_bounds__(2 * TMT, 2 * TMT == 256 ? 4 : 2) void gemm_bf16_dma_kernel(const GemmArgs a) {
    constexpr int ES = 2, CE = 8, SLAB_K = 64;
    constexpr int NT = 2 * TMT;                 // threads
    constexpr int RPP = NT / 8;                 // rows staged per pass
    constexpr int XP = TMT / RPP, WP = TN / RPP;
    __shared__ uint4 lds[(TMT + TN) * CHUNKS];  // X rows then W rows: [row*8 + position]
    uint4* ldsX = lds; uint4* ldsW = lds + TMT * CHUNKS;
    const int total = a.nbm * a.nbn * (SPLITK ? a.ksplit : 1);
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = total >> 3, r8 = total & 7;
    const int vid_all = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const int kpart = SPLITK ? vid_all % a.ksplit : 0;                   // the pieces of one tile are neighbours (same XCD)
    const int vid = SPLITK ? vid_all / a.ksplit : vid_all;
    int tile_n = vid % a.nbn, tile_m = vid / a.nbn;
    if (a.group_m > 0) {                       // grouped order: GROUP_M m-tiles share each W panel while their X panels stay in L2
        const int per_group = a.group_m * a.nbn;
        const int gid = vid / per_group, first_m = gid * a.group_m;
        const int gsz = a.nbm - first_m < a.group_m ? a.nbm - first_m : a.group_m;
        const int loc = vid - gid * per_group;
        tile_m = first_m + loc % gsz;
        tile_n = loc / gsz;
    }
    const long long m0 = (long long)tile_m * TMT, n0 = (long long)tile_n * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = TMT == 128 ? (wave & 1) : (wave & 3), wn = TMT == 128 ? (wave >> 1) : (wave >> 2);
    const long long grp = blockIdx.y;
    const char* Xg = a.X + grp * a.a_gstride * ES;
    const char* Wg = a.W + grp * a.w_gstride * ES;
    const long long cshift = grp * a.c_gstride;

    // staging: one wave-instruction fills 8 rows (1 KiB, lane-linear); pass i covers rows i*RPP .. i*RPP + RPP-1
    const int pos = tid & 7, srow = tid >> 3;
    const char* xsrc[XP]; const char* wsrc[WP];
    int xc[XP], wc[WP];
#pragma unroll
    for (int i = 0; i < XP; ++i) {
        const int row = srow + RPP * i;
        long long m = m0 + row; if (m > a.M - 1) m = a.M - 1;
        xsrc[i] = Xg + row_off(a.xmap, m) * ES;
        xc[i] = pos ^ (row & 7);
    }
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int row = srow + RPP * i;
        long long n = n0 + row; if (n > a.N - 1) n = a.N - 1;
        wsrc[i] = Wg + n * a.ldw * ES;
        wc[i] = pos ^ (row & 7);
    }
    const int nslab_all = (int)(a.K / SLAB_K);
    const int slab0 = SPLITK ? kpart * a.slabs_per_split : 0;
    const int nslab = SPLITK ? (slab0 + a.slabs_per_split < nslab_all ? slab0 + a.slabs_per_split : nslab_all) : nslab_all;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    for (int slab = slab0; slab < nslab; ++slab) {
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const long long k0 = (long long)slab * SLAB_K + xc[i] * CE;
            long long kx = k0;
            if (a.nseg > 1) { const long long sg = k0 / a.seg_len; kx = sg * a.seg_stride + (k0 - sg * a.seg_len); }
            __builtin_amdgcn_global_load_lds((gbl_void*)(xsrc[i] + kx * ES), (lds_void*)&ldsX[(wave * 8 + RPP * i) * CHUNKS], 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WP; ++i) {
            const long long k0 = (long long)slab * SLAB_K + wc[i] * CE;
            __builtin_amdgcn_global_load_lds((gbl_void*)(wsrc[i] + k0 * ES), (lds_void*)&ldsW[(wave * 8 + RPP * i) * CHUNKS], 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            uint4 wf[4], xf[4];
            const int chk = kb * 4 + fq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rw = wn * 64 + i * 16 + fr;
                wf[i] = ldsW[rw * CHUNKS + (chk ^ (rw & 7))];
                const int rx = wm * 64 + i * 16 + fr;
                xf[i] = ldsX[rx * CHUNKS + (chk ^ (rx & 7))];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        *reinterpret_cast<bf16x8*>(&wf[i]), *reinterpret_cast<bf16x8*>(&xf[j]), acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    if constexpr (SPLITK) gemm_epilogue_atomic<4>(a, acc, m0 + wm * 64, n0 + wn * 64, fr, fq, cshift);
    else gemm_epilogue<4>(a, acc, m0 + wm * 64, n0 + wn * 64, fr, fq, cshift);
}

// ------------------------------------------------------------------------------------------------
// Multi-stage bf16 kernel: ONE workgroup per CU, block tile (WGM*NJ*16) x (WGN*64), NST LDS stages filled by LDS-DMA with
// NST-1 slabs in flight.  The loads stay in flight ACROSS the per-slab barrier: the wait is a counted `s_waitcnt vmcnt(n)`
// (n = DMA instructions of the later slabs, never 0 in the steady state) and the barrier is a raw `s_barrier` -- a
// `__syncthreads()` would drain the DMA queue (it fences with vmcnt(0)).  Why: at one or two resident workgroups per CU the
// L2->LDS path is latency-bound, not bandwidth-bound; sustaining the ~48 B/clk a 256x128 tile needs at ~2000 clk of latency
// takes ~96 KiB in flight per CU (Little), which is what 2 x 48 KiB stages in flight out of 3 provide.
//   256x128 (WGM=4, WGN=2, NJ=4): one L2 byte per 85 FLOP;  256x256 (WGM=2, WGN=4, NJ=8): one per 128 FLOP;
//   128x128 (WGM=2, WGN=2, NJ=4): one per 64 FLOP (the CU's ~64 B/clk L2 port is then the bound).
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
template <int WGM, int WGN, int NJ, int NST>
__global__ __launch_bounds__(WGM * WGN * 64, 1) void gemm_bf16_ms_kernel(const GemmArgs a) {
    constexpr int ES = 2, CE = 8, SLAB_K = 64;
    constexpr int NT = WGM * WGN * 64;
    constexpr int TMB = WGM * NJ * 16, TNB = WGN * 64;
    constexpr int RPP = NT / 8;                      // rows staged per pass
    constexpr int XP = TMB / RPP, WP = TNB / RPP;
    constexpr int PER = XP + WP;                     // DMA instructions per wave per slab
    constexpr int STAGE = (TMB + TNB) * CHUNKS;      // uint4 per stage
    static_assert((NST - 2) * PER <= 63, "vmcnt immediate");
    extern __shared__ uint4 plds[];                  // [NST][X rows | W rows][8 positions]
    const int total = a.nbm * a.nbn;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = total >> 3, r8 = total & 7;
    const int vid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    int tile_n = vid % a.nbn, tile_m = vid / a.nbn;
    if (a.group_m > 0) {
        const int per_group = a.group_m * a.nbn;
        const int gid = vid / per_group, first_m = gid * a.group_m;
        const int gsz = a.nbm - first_m < a.group_m ? a.nbm - first_m : a.group_m;
        const int loc = vid - gid * per_group;
        tile_m = first_m + loc % gsz;
        tile_n = loc / gsz;
    }
    const long long m0 = (long long)tile_m * TMB, n0 = (long long)tile_n * TNB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WGM, wn = wave / WGM;
    const long long grp = blockIdx.y;
    const char* Xg = a.X + grp * a.a_gstride * ES;
    const char* Wg = a.W + grp * a.w_gstride * ES;
    const long long cshift = grp * a.c_gstride;

    const int pos = tid & 7, srow = tid >> 3;
    const char* xsrc[XP]; const char* wsrc[WP];
#pragma unroll
    for (int i = 0; i < XP; ++i) {
        const int row = srow + RPP * i;
        long long m = m0 + row; if (m > a.M - 1) m = a.M - 1;
        xsrc[i] = Xg + row_off(a.xmap, m) * ES;
    }
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int row = srow + RPP * i;
        long long n = n0 + row; if (n > a.N - 1) n = a.N - 1;
        wsrc[i] = Wg + n * a.ldw * ES;
    }
    const int sc = (pos ^ (srow & 7)) * CE;          // RPP % 8 == 0: the source chunk is the same for every pass of this lane
    const int nslab = (int)(a.K / SLAB_K);
    auto stage = [&](int slab, int buf) {
        uint4* sx = plds + buf * STAGE;
        uint4* sw = sx + TMB * CHUNKS;
        const long long k0 = (long long)slab * SLAB_K + sc;
        long long kx = k0;
        if (a.nseg > 1) { const long long sg = k0 / a.seg_len; kx = sg * a.seg_stride + (k0 - sg * a.seg_len); }
#pragma unroll
        for (int i = 0; i < XP; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(xsrc[i] + kx * ES), (lds_void*)&sx[(wave * 8 + RPP * i) * CHUNKS], 16, 0, 0);
#pragma unroll
        for (int i = 0; i < WP; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(wsrc[i] + k0 * ES), (lds_void*)&sw[(wave * 8 + RPP * i) * CHUNKS], 16, 0, 0);
    };
    f32x4 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    // byte offsets of this lane's kb = 0 fragment rows inside a stage (kb = 1 is the same address ^ 64: chunk ^ 4)
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)plds;
    const unsigned xoff0 = (unsigned)((wm * (NJ * 16) + fr) * 128 + ((fq ^ (fr & 7)) << 4));
    const unsigned woff0 = (unsigned)(TMB * 128 + (wn * 64 + fr) * 128 + ((fq ^ (fr & 7)) << 4));
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nslab) stage(s, s);
    int buf = 0, nbuf = NST - 1;
    for (int slab = 0; slab < nslab; ++slab) {
        // slabs slab+1 .. slab+NST-2 may stay in flight; slab `slab` (this wave's pieces) must have landed
        const int later = nslab - 1 - slab;
        if (later >= NST - 2) wait_vm_then_barrier<(NST - 2) * PER>();
        else if (NST > 3 && later == 1) wait_vm_then_barrier<PER>();
        else wait_vm_then_barrier<0>();
        // every wave's pieces of `slab` have landed and nobody reads stage nbuf (= stage of slab-1) any more
        if (slab + NST - 1 < nslab && !(a.dbg & 1)) stage(slab + NST - 1, nbuf);
        // Fragment reads are inline asm on purpose: hipcc treats an LDS-DMA as a pending LDS store and would put
        // `s_waitcnt vmcnt(0)` in front of any ds_read it can see, draining the slabs that are meant to stay in flight.
        const unsigned sb = lds0 + (unsigned)buf * (STAGE * 16);
        u32x4 wf[2][4], xf[2][NJ];
        if (!(a.dbg & 4)) {
            read_frags<4>(wf[0], sb + woff0); read_frags<NJ>(xf[0], sb + xoff0);
            read_frags<4>(wf[1], sb + (woff0 ^ 64)); read_frags<NJ>(xf[1], sb + (xoff0 ^ 64));
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { wf[0][i] = (u32x4){sb, 1u, 2u, 3u}; wf[1][i] = wf[0][i]; }
#pragma unroll
            for (int j = 0; j < NJ; ++j) { xf[0][j] = (u32x4){sb, 1u, 2u, 3u}; xf[1][j] = xf[0][j]; }
        }
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(4 + NJ) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (!(a.dbg & 2)) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[0][i]), __builtin_bit_cast(bf16x8, xf[0][j]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (!(a.dbg & 2)) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[1][i]), __builtin_bit_cast(bf16x8, xf[1][j]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(wf[0][i]), "v"(wf[1][i]));
#pragma unroll
            for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(xf[0][j]), "v"(xf[1][j]));
        }
        buf = buf + 1 == NST ? 0 : buf + 1;
        nbuf = nbuf + 1 == NST ? 0 : nbuf + 1;
    }
    gemm_epilogue<NJ>(a, acc, m0 + wm * (NJ * 16), n0 + wn * 64, fr, fq, cshift);
}

// ------------------------------------------------------------------------------------------------
// Persistent form of the 128x128 LDS-DMA kernel (4 workgroups per CU, one 32 KiB slab buffer each; occupancy hides the load
// latency as before).  What it removes is the per-tile fixed cost that phase-aligned workgroups cannot hide from each other:
//  * slab 0 of the NEXT tile (and its bias row) is issued before the current tile's epilogue, so first-slab latency runs under
//    the epilogue arithmetic, and the epilogue's stores are YOUNGER than those loads: the next wait is `vmcnt(#stores)` and the
//    stores drain under the next tile's MFMAs instead of at wave exit (vmcnt retires loads, stores and LDS-DMA in issue order);
//  * the bias row comes through LDS (one 512-byte DMA per tile) instead of four dependent global loads per thread;
//  * workgroup launch, kernarg fetch and tile-independent setup happen once per workgroup.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
__global__ __launch_bounds__(256, 4) void gemm_bf16_p1_kernel(const GemmArgs a) {
    constexpr int ES = 2, CE = 8, SLAB_K = 64, TMT = 128;
    __shared__ uint4 lds[(TMT + TN) * CHUNKS + 2 * 32];          // X rows, W rows, then two bias rows (2 x 128 f32)
    uint4* ldsX = lds; uint4* ldsW = lds + TMT * CHUNKS;
    const int tiles_per_group = a.nbm * a.nbn;
    const int total = tiles_per_group * a.ngroups;
    const int G = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = total >> 3, r8 = total & 7;
    const int lo = xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;
    const int hi = lo + (xcd < r8 ? q + 1 : q);
    const int stride = (G - xcd + 7) >> 3;
    const int first = lo + (bid >> 3);
    const int nt_my = first < hi ? (hi - first + stride - 1) / stride : 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int pos = tid & 7, srow = tid >> 3;
    const int sc = (pos ^ (srow & 7)) * CE;
    const int nslab = (int)(a.K / SLAB_K);
    const int fr = lane & 15, fq = lane >> 4;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)lds;
    const unsigned bias_lds = lds0 + (TMT + TN) * CHUNKS * 16;

    auto decode = [&](int vid, int& grp, int& tile_m, int& tile_n) {
        grp = vid / tiles_per_group;
        const int v = vid - grp * tiles_per_group;
        tile_n = v % a.nbn; tile_m = v / a.nbn;
        if (a.group_m > 0) {
            const int per_group = a.group_m * a.nbn;
            const int gid = v / per_group, first_m = gid * a.group_m;
            const int gsz = a.nbm - first_m < a.group_m ? a.nbm - first_m : a.group_m;
            const int loc = v - gid * per_group;
            tile_m = first_m + loc % gsz;
            tile_n = loc / gsz;
        }
    };
    const char* xsrc[4]; const char* wsrc[4];
    auto set_tile = [&](int ord, long long& m0, long long& n0, long long& cshift) {
        int grp, tm, tn; decode(first + ord * stride, grp, tm, tn);
        const char* Xg = a.X + (long long)grp * a.a_gstride * ES;
        const char* Wg = a.W + (long long)grp * a.w_gstride * ES;
        m0 = (long long)tm * TMT; n0 = (long long)tn * TN; cshift = (long long)grp * a.c_gstride;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            long long m = m0 + srow + 32 * i; if (m > a.M - 1) m = a.M - 1;
            xsrc[i] = Xg + row_off(a.xmap, m) * ES;
            long long n = n0 + srow + 32 * i; if (n > a.N - 1) n = a.N - 1;
            wsrc[i] = Wg + n * a.ldw * ES;
        }
    };
    auto issue_slab = [&](int slab) {
        const long long k0 = (long long)slab * SLAB_K + sc;
        long long kx = k0;
        if (a.nseg > 1) { const long long sg = k0 / a.seg_len; kx = sg * a.seg_stride + (k0 - sg * a.seg_len); }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(xsrc[i] + kx * ES), (lds_void*)&ldsX[(wave * 8 + 32 * i) * CHUNKS], 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(wsrc[i] + k0 * ES), (lds_void*)&ldsW[(wave * 8 + 32 * i) * CHUNKS], 16, 0, 0);
    };
    // bias row of a tile: 128 f32, one 256-byte DMA from each of waves 0 and 1 (all waves issue one so the per-wave count is uniform)
    auto issue_bias = [&](long long n0, long long cshift, int par) {
        long long n = n0 + (wave & 1) * 64 + lane; if (n > a.N - 1) n = a.N - 1;
        const float* src = a.bias ? a.bias + cshift + n : reinterpret_cast<const float*>(a.W);
        float* dst = reinterpret_cast<float*>(lds + (TMT + TN) * CHUNKS) + par * 128 + (wave & 1) * 64;
        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)dst, 4, 0, 0);
    };
    if (nt_my == 0) return;
    long long m0, n0, cshift;
    set_tile(0, m0, n0, cshift);
    issue_slab(0);
    issue_bias(n0, cshift, 0);
    bool prev_full = false;                                   // did this wave issue its 16 epilogue stores after the loads in flight?
    for (int ord = 0; ord < nt_my; ++ord) {
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // ---- slab 0: its DMA was issued under the previous epilogue.  Fragment reads as inline asm here: hipcc would put
        // `s_waitcnt vmcnt(0)` in front of any ds_read it can see while an LDS-DMA may be pending, and that would also wait
        // for the previous tile's stores.
        if (prev_full) asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");   // slab 0 + bias landed; the 16 younger stores may still fly
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        {
            const unsigned xo = lds0 + (unsigned)((wm * 64 + fr) * 128 + ((fq ^ (fr & 7)) << 4));
            const unsigned wo = lds0 + (unsigned)(TMT * 128 + (wn * 64 + fr) * 128 + ((fq ^ (fr & 7)) << 4));
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                u32x4 wf[4], xf[4];
                read_frags<4>(wf, wo ^ (kb * 64)); read_frags<4>(xf, xo ^ (kb * 64));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[i]), __builtin_bit_cast(bf16x8, xf[j]), acc[i][j], 0, 0, 0);
            }
            asm volatile("s_barrier" ::: "memory");
        }
        // ---- slabs 1..: the plain loop of gemm_bf16_dma_kernel (compiler-scheduled fragment reads)
        for (int slab = 1; slab < nslab; ++slab) {
            issue_slab(slab);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                uint4 wf[4], xf[4];
                const int chk = kb * 4 + fq;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int rw = wn * 64 + i * 16 + fr;
                    wf[i] = ldsW[rw * CHUNKS + (chk ^ (rw & 7))];
                    const int rx = wm * 64 + i * 16 + fr;
                    xf[i] = ldsX[rx * CHUNKS + (chk ^ (rx & 7))];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            *reinterpret_cast<bf16x8*>(&wf[i]), *reinterpret_cast<bf16x8*>(&xf[j]), acc[i][j], 0, 0, 0);
            }
            __syncthreads();
        }
        // bias of this tile -> registers (its DMA was waited for at slab 0), then start the next tile's slab 0 under the epilogue
        f32x4 breg[4];
        {
            const unsigned ba = bias_lds + (unsigned)((ord & 1) * 512 + (wn * 64 + fq * 4) * 4);
            u32x4 t0 = ds_read128<0>(ba), t1 = ds_read128<64>(ba), t2 = ds_read128<128>(ba), t3 = ds_read128<192>(ba);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            breg[0] = __builtin_bit_cast(f32x4, t0); breg[1] = __builtin_bit_cast(f32x4, t1);
            breg[2] = __builtin_bit_cast(f32x4, t2); breg[3] = __builtin_bit_cast(f32x4, t3);
        }
        const long long em0 = m0 + wm * 64, en0 = n0 + wn * 64, ecs = cshift;
        const bool full = m0 + TMT <= a.M && n0 + TN <= a.N && !a.R && !a.aux;   // exactly 16 stores and no other memory operation per wave
        if (ord + 1 < nt_my) {
            set_tile(ord + 1, m0, n0, cshift);
            issue_slab(0);
            issue_bias(n0, cshift, (ord + 1) & 1);
        }
        gemm_epilogue<4>(a, acc, em0, en0, fr, fq, ecs, a.bias ? breg : nullptr);
        prev_full = full;
    }
}

// ------------------------------------------------------------------------------------------------
// Deep half-slab pipeline on a big tile (experiment, variant 15/16): block tile (WGM*NJ*16) x (WGN*64), K advanced in steps of 32
// through NST LDS stages of (TMB + TNB) x 64 B; NST-1 steps are in flight, the wait is a counted vmcnt, the barriers are raw.
// At 256x256 (8 waves, 128x64 per wave): 32 KiB per stage, 4 stages = 128 KiB, 96 KiB in flight per CU at 128 FLOP per L2 byte.
template <int WGM, int WGN, int NJ, int NST>
__global__ __launch_bounds__(WGM * WGN * 64, 1) void gemm_bf16_ms32_kernel(const GemmArgs a) {
    constexpr int ES = 2;
    constexpr int NT = WGM * WGN * 64;
    constexpr int TMB = WGM * NJ * 16, TNB = WGN * 64;
    constexpr int RPP = NT / 4;                       // rows staged per pass (one DMA = 16 rows x 64 B)
    constexpr int XP = TMB / RPP, WP = TNB / RPP;
    constexpr int PER = XP + WP;
    constexpr int STAGE = (TMB + TNB) * 4;            // uint4 per stage
    static_assert(TMB % RPP == 0 && TNB % RPP == 0 && (NST - 2) * PER <= 63, "tile / vmcnt");
    extern __shared__ uint4 plds[];
    const int total = a.nbm * a.nbn;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = total >> 3, r8 = total & 7;
    const int vid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    int tile_n = vid % a.nbn, tile_m = vid / a.nbn;
    if (a.group_m > 0) {
        const int per_group = a.group_m * a.nbn;
        const int gid = vid / per_group, first_m = gid * a.group_m;
        const int gsz = a.nbm - first_m < a.group_m ? a.nbm - first_m : a.group_m;
        const int loc = vid - gid * per_group;
        tile_m = first_m + loc % gsz;
        tile_n = loc / gsz;
    }
    const long long m0 = (long long)tile_m * TMB, n0 = (long long)tile_n * TNB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WGM, wn = wave / WGM;
    const long long grp = blockIdx.y;
    const char* Xg = a.X + grp * a.a_gstride * ES;
    const char* Wg = a.W + grp * a.w_gstride * ES;
    const long long cshift = grp * a.c_gstride;
    const int srow = tid >> 2, p4 = tid & 3;          // staging: thread -> (row inside a pass, 16-byte position)
    const int sck = (p4 ^ ((srow >> 1) & 3)) * 8;     // RPP % 8 == 0: same source chunk for every pass
    const char* xsrc[XP]; const char* wsrc[WP];
#pragma unroll
    for (int i = 0; i < XP; ++i) {
        long long m = m0 + srow + RPP * i; if (m > a.M - 1) m = a.M - 1;
        xsrc[i] = Xg + row_off(a.xmap, m) * ES;
    }
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        long long n = n0 + srow + RPP * i; if (n > a.N - 1) n = a.N - 1;
        wsrc[i] = Wg + n * a.ldw * ES;
    }
    const int nstep = (int)(a.K / 32);
    auto stage = [&](int h, int buf) {
        uint4* sx = plds + buf * STAGE;
        uint4* sw = sx + TMB * 4;
        const long long k0 = (long long)h * 32 + sck;
        long long kx = k0;
        if (a.nseg > 1) { const long long sg = k0 / a.seg_len; kx = sg * a.seg_stride + (k0 - sg * a.seg_len); }
#pragma unroll
        for (int i = 0; i < XP; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(xsrc[i] + kx * ES), (lds_void*)&sx[(wave * 16 + RPP * i) * 4], 16, 0, 0);
#pragma unroll
        for (int i = 0; i < WP; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(wsrc[i] + k0 * ES), (lds_void*)&sw[(wave * 16 + RPP * i) * 4], 16, 0, 0);
    };
    f32x4 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)plds;
    const unsigned sw16 = (unsigned)((fq ^ ((fr >> 1) & 3)) << 4);
    const unsigned xoff = (unsigned)((wm * (NJ * 16) + fr) * 64) + sw16;
    const unsigned woff = (unsigned)(TMB * 64 + (wn * 64 + fr) * 64) + sw16;
#pragma unroll
    for (int h = 0; h < NST - 1; ++h)
        if (h < nstep) stage(h, h);
    int buf = 0, nbuf = NST - 1;
    for (int h = 0; h < nstep; ++h) {
        const int later = nstep - 1 - h;
        if (later >= NST - 2) wait_vm_then_barrier<(NST - 2) * PER>();
        else if (NST > 3 && later == 2) wait_vm_then_barrier<2 * PER>();
        else if (NST > 3 && later == 1) wait_vm_then_barrier<PER>();
        else wait_vm_then_barrier<0>();
        if (h + NST - 1 < nstep) stage(h + NST - 1, nbuf);     // the stage of step h-1: every wave has passed the barrier after reading it
        const unsigned sb = lds0 + (unsigned)buf * (STAGE * 16);
        u32x4 wf[4], xf[NJ];
        wf[0] = ds_read128<0>(sb + woff); wf[1] = ds_read128<1024>(sb + woff); wf[2] = ds_read128<2048>(sb + woff); wf[3] = ds_read128<3072>(sb + woff);
        xf[0] = ds_read128<0>(sb + xoff); xf[1] = ds_read128<1024>(sb + xoff); xf[2] = ds_read128<2048>(sb + xoff); xf[3] = ds_read128<3072>(sb + xoff);
        if constexpr (NJ == 8) {
            xf[4] = ds_read128<4096>(sb + xoff); xf[5] = ds_read128<5120>(sb + xoff); xf[6] = ds_read128<6144>(sb + xoff); xf[7] = ds_read128<7168>(sb + xoff);
        }
        // x fragments are consumed column by column: wait only for what the next group of MFMAs needs
#define OCC_MS32_COL(J, CNT)                                                                                                 \
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(CNT) : "memory");                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                        \
            acc[i][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[i]), __builtin_bit_cast(bf16x8, xf[J]), acc[i][J], 0, 0, 0);
        __builtin_amdgcn_s_setprio(1);
        if constexpr (NJ == 8) { OCC_MS32_COL(0, 7) OCC_MS32_COL(1, 6) OCC_MS32_COL(2, 5) OCC_MS32_COL(3, 4) OCC_MS32_COL(4, 3) OCC_MS32_COL(5, 2) OCC_MS32_COL(6, 1) OCC_MS32_COL(7, 0) }
        else { OCC_MS32_COL(0, 3) OCC_MS32_COL(1, 2) OCC_MS32_COL(2, 1) OCC_MS32_COL(3, 0) }
        __builtin_amdgcn_s_setprio(0);
#undef OCC_MS32_COL
        buf = buf + 1 == NST ? 0 : buf + 1;
        nbuf = nbuf + 1 == NST ? 0 : nbuf + 1;
    }
    gemm_epilogue<NJ>(a, acc, m0 + wm * (NJ * 16), n0 + wn * 64, fr, fq, cshift);
}

template <int WGM, int WGN, int NJ, int NST>
int launch_ms32(GemmArgs& a, const occ_gemm_desc* d, long long ng, hipStream_t s) {
    constexpr int TMB = WGM * NJ * 16, TNB = WGN * 64;
    a.nbm = (int)occ_cdiv(d->M, TMB); a.nbn = (int)occ_cdiv(d->N, TNB);
    a.group_m = a.nbn >= 8 ? 4 : 0;
    const size_t shm = (size_t)NST * (TMB + TNB) * 4 * sizeof(uint4);
    static bool raised = false;
    if (!raised) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_ms32_kernel<WGM, WGN, NJ, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) { occ_set_error("occ_gemm: cannot raise the LDS limit: %s", hipGetErrorString(e)); return OCC_ELAUNCH; }
        raised = true;
    }
    hipLaunchKernelGGL((gemm_bf16_ms32_kernel<WGM, WGN, NJ, NST>), dim3((unsigned)((long long)a.nbm * a.nbn), (unsigned)ng), dim3(WGM * WGN * 64), shm, s, a);
    return OCC_OK;
}

// ------------------------------------------------------------------------------------------------
// Half-slab pipeline (experiment, variant 14): the 32 KiB of the default kernel cut into two 16 KiB halves of K = 32, so that one
// half is always in flight while the other is being multiplied -- the default kernel has nothing in flight while it computes.
// Same occupancy (4 workgroups per CU), twice the barriers.  64-byte LDS rows: chunk c of row r sits at c ^ ((r >> 1) & 3), which is
// conflict-free for the ds_read_b128 lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32).
__global__ __launch_bounds__(256, 4) void gemm_bf16_hs_kernel(const GemmArgs a) {
    constexpr int ES = 2, TMT = 128, HB = 1024;                 // uint4 per half: (128 + 128) rows x 4 chunks
    __shared__ uint4 lds[2 * HB];
    const int total = a.nbm * a.nbn;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = total >> 3, r8 = total & 7;
    const int vid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    int tile_n = vid % a.nbn, tile_m = vid / a.nbn;
    if (a.group_m > 0) {
        const int per_group = a.group_m * a.nbn;
        const int gid = vid / per_group, first_m = gid * a.group_m;
        const int gsz = a.nbm - first_m < a.group_m ? a.nbm - first_m : a.group_m;
        const int loc = vid - gid * per_group;
        tile_m = first_m + loc % gsz;
        tile_n = loc / gsz;
    }
    const long long m0 = (long long)tile_m * TMT, n0 = (long long)tile_n * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const long long grp = blockIdx.y;
    const char* Xg = a.X + grp * a.a_gstride * ES;
    const char* Wg = a.W + grp * a.w_gstride * ES;
    const long long cshift = grp * a.c_gstride;
    // staging: one DMA = 16 rows x 64 B; wave w issues row blocks w and w + 4 of X and of W per half
    const int rl = lane >> 2, p4 = lane & 3;
    const int sck = (p4 ^ ((rl >> 1) & 3)) * 8;                  // source k offset (elements) of this lane inside a half
    const char* xsrc[2]; const char* wsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        long long m = m0 + 16 * (wave + 4 * i) + rl; if (m > a.M - 1) m = a.M - 1;
        long long n = n0 + 16 * (wave + 4 * i) + rl; if (n > a.N - 1) n = a.N - 1;
        xsrc[i] = Xg + row_off(a.xmap, m) * ES;
        wsrc[i] = Wg + n * a.ldw * ES;
    }
    const int nhalf = (int)(a.K / 32);
    auto issue = [&](int h) {
        const long long k0 = (long long)h * 32 + sck;
        long long kx = k0;
        if (a.nseg > 1) { const long long sg = k0 / a.seg_len; kx = sg * a.seg_stride + (k0 - sg * a.seg_len); }
        uint4* base = lds + (h & 1) * HB;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((gbl_void*)(xsrc[i] + kx * ES), (lds_void*)&base[(wave + 4 * i) * 64], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void*)(wsrc[i] + k0 * ES), (lds_void*)&base[512 + (wave + 4 * i) * 64], 16, 0, 0);
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)lds;
    const unsigned sw = (unsigned)((fq ^ ((fr >> 1) & 3)) << 4);
    const unsigned xoff = (unsigned)((wm * 64 + fr) * 64) + sw;
    const unsigned woff = (unsigned)(8192 + (wn * 64 + fr) * 64) + sw;
    issue(0);
    if (nhalf > 1) issue(1);
    for (int h = 0; h < nhalf; ++h) {
        if (h + 1 < nhalf) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");     // half h landed, half h+1 stays in flight
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        const unsigned hb = lds0 + (unsigned)(h & 1) * (HB * 16);
        u32x4 xf[4], wf[4];
        xf[0] = ds_read128<0>(hb + xoff); xf[1] = ds_read128<1024>(hb + xoff); xf[2] = ds_read128<2048>(hb + xoff); xf[3] = ds_read128<3072>(hb + xoff);
        wf[0] = ds_read128<0>(hb + woff); wf[1] = ds_read128<1024>(hb + woff); wf[2] = ds_read128<2048>(hb + woff); wf[3] = ds_read128<3072>(hb + woff);
#define OCC_HS_ROW(I, CNT)                                                                                                       \
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(CNT) : "memory");                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                        \
            acc[I][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[I]), __builtin_bit_cast(bf16x8, xf[j]), acc[I][j], 0, 0, 0);
        OCC_HS_ROW(0, 3) OCC_HS_ROW(1, 2) OCC_HS_ROW(2, 1) OCC_HS_ROW(3, 0)
#undef OCC_HS_ROW
        if (h + 2 < nhalf) {
            asm volatile("s_barrier" ::: "memory");              // every wave has read this half: it can be refilled
            issue(h + 2);
        }
    }
    gemm_epilogue<4>(a, acc, m0 + wm * 64, n0 + wn * 64, fr, fq, cshift);
}

// ------------------------------------------------------------------------------------------------
// Persistent multi-stage kernel: one workgroup per CU walks a list of output tiles and keeps ONE slab pipeline running across
// tile boundaries -- while a tile's epilogue runs, the first NST-1 slabs of the workgroup's next tile are already in flight, so
// block start-up, first-slab latency and store drain are paid once per launch instead of once per tile.  (Measured on the
// non-persistent kernels: a 6368x4096 GEMM with K = 64 takes 30 us of which the K loop is < 4; at K = 1024 that fixed part is
// a third of the run time.)  Tile list: XCD x owns a contiguous range of tile ids; its 32 workgroups stride through it together,
// so the tiles in flight on one L2 at any time are neighbours (group_m x n panel).  Groups (grouped conv) are folded into the id.
template <int WGM, int WGN, int NJ, int NST>
__global__ __launch_bounds__(WGM * WGN * 64, 1) void gemm_bf16_persist_kernel(const GemmArgs a) {
    constexpr int ES = 2, CE = 8, SLAB_K = 64;
    constexpr int NT = WGM * WGN * 64;
    constexpr int TMB = WGM * NJ * 16, TNB = WGN * 64;
    constexpr int RPP = NT / 8;
    constexpr int XP = TMB / RPP, WP = TNB / RPP;
    constexpr int PER = XP + WP;
    constexpr int STAGE = (TMB + TNB) * CHUNKS;
    static_assert((NST - 2) * PER <= 63, "vmcnt immediate");
    extern __shared__ uint4 plds[];
    const int tiles_per_group = a.nbm * a.nbn;
    const int total = tiles_per_group * a.ngroups;
    const int G = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = total >> 3, r8 = total & 7;
    const int lo = xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;
    const int hi = lo + (xcd < r8 ? q + 1 : q);
    const int stride = (G - xcd + 7) >> 3;                 // workgroups on this XCD
    const int first = lo + (bid >> 3);
    const int nt_my = first < hi ? (hi - first + stride - 1) / stride : 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WGM, wn = wave / WGM;
    const int pos = tid & 7, srow = tid >> 3;
    const int sc = (pos ^ (srow & 7)) * CE;
    const int nslab = (int)(a.K / SLAB_K);
    const long long total_iters = (long long)nt_my * nslab;

    auto decode = [&](int vid, int& grp, int& tile_m, int& tile_n) {
        grp = vid / tiles_per_group;
        const int v = vid - grp * tiles_per_group;
        tile_n = v % a.nbn; tile_m = v / a.nbn;
        if (a.group_m > 0) {
            const int per_group = a.group_m * a.nbn;
            const int gid = v / per_group, first_m = gid * a.group_m;
            const int gsz = a.nbm - first_m < a.group_m ? a.nbm - first_m : a.group_m;
            const int loc = v - gid * per_group;
            tile_m = first_m + loc % gsz;
            tile_n = loc / gsz;
        }
    };
    // ---- stage side: the tile whose slabs are being issued (runs up to NST-1 slabs ahead of the compute side)
    const char* xsrc[XP]; const char* wsrc[WP];
    int ord_s = 0, slab_s = 0, sbuf = 0;
    auto set_stage_tile = [&](int ord) {
        int grp, tm, tn; decode(first + ord * stride, grp, tm, tn);
        const char* Xg = a.X + (long long)grp * a.a_gstride * ES;
        const char* Wg = a.W + (long long)grp * a.w_gstride * ES;
        const long long m0 = (long long)tm * TMB, n0 = (long long)tn * TNB;
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            long long m = m0 + srow + RPP * i; if (m > a.M - 1) m = a.M - 1;
            xsrc[i] = Xg + row_off(a.xmap, m) * ES;
        }
#pragma unroll
        for (int i = 0; i < WP; ++i) {
            long long n = n0 + srow + RPP * i; if (n > a.N - 1) n = a.N - 1;
            wsrc[i] = Wg + n * a.ldw * ES;
        }
    };
    auto stage_next = [&]() {
        if (ord_s >= nt_my) return;
        uint4* sx = plds + sbuf * STAGE;
        uint4* sw = sx + TMB * CHUNKS;
        const long long k0 = (long long)slab_s * SLAB_K + sc;
        long long kx = k0;
        if (a.nseg > 1) { const long long sg = k0 / a.seg_len; kx = sg * a.seg_stride + (k0 - sg * a.seg_len); }
#pragma unroll
        for (int i = 0; i < XP; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(xsrc[i] + kx * ES), (lds_void*)&sx[(wave * 8 + RPP * i) * CHUNKS], 16, 0, 0);
#pragma unroll
        for (int i = 0; i < WP; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(wsrc[i] + k0 * ES), (lds_void*)&sw[(wave * 8 + RPP * i) * CHUNKS], 16, 0, 0);
        sbuf = sbuf + 1 == NST ? 0 : sbuf + 1;
        if (++slab_s == nslab) { slab_s = 0; if (++ord_s < nt_my) set_stage_tile(ord_s); }
    };
    const int fr = lane & 15, fq = lane >> 4;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)plds;
    const unsigned xoff0 = (unsigned)((wm * (NJ * 16) + fr) * 128 + ((fq ^ (fr & 7)) << 4));
    const unsigned woff0 = (unsigned)(TMB * 128 + (wn * 64 + fr) * 128 + ((fq ^ (fr & 7)) << 4));
    if (nt_my > 0) set_stage_tile(0);
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) stage_next();
    int buf = 0;
    long long g = 0;
    for (int ord = 0; ord < nt_my; ++ord) {
        f32x4 acc[4][NJ];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int slab = 0; slab < nslab; ++slab, ++g) {
            // Everything younger than the slab needed now may stay in flight.  (NST-2)*PER is a lower bound of the number of
            // younger operations (vmcnt counts loads, stores and LDS-DMA together in issue order), so the wait never returns early;
            // right after an epilogue the stores make it wait a little longer than necessary.
            const long long later = total_iters - 1 - g;
            if (later >= NST - 2) wait_vm_then_barrier<(NST - 2) * PER>();
            else if (NST > 3 && later == 1) wait_vm_then_barrier<PER>();
            else wait_vm_then_barrier<0>();
            stage_next();
            const unsigned sb = lds0 + (unsigned)buf * (STAGE * 16);
            u32x4 wf[2][4], xf[2][NJ];
            read_frags<4>(wf[0], sb + woff0); read_frags<NJ>(xf[0], sb + xoff0);
            read_frags<4>(wf[1], sb + (woff0 ^ 64)); read_frags<NJ>(xf[1], sb + (xoff0 ^ 64));
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(4 + NJ) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[0][i]), __builtin_bit_cast(bf16x8, xf[0][j]), acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[1][i]), __builtin_bit_cast(bf16x8, xf[1][j]), acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            buf = buf + 1 == NST ? 0 : buf + 1;
        }
        int grp, tm, tn; decode(first + ord * stride, grp, tm, tn);
        gemm_epilogue<NJ>(a, acc, (long long)tm * TMB + wm * (NJ * 16), (long long)tn * TNB + wn * 64, fr, fq, (long long)grp * a.c_gstride);
    }
}

template <int WGM, int WGN, int NJ, int NST>
int launch_ms(GemmArgs& a, const occ_gemm_desc* d, long long ng, hipStream_t s) {
    constexpr int TMB = WGM * NJ * 16, TNB = WGN * 64;
    a.nbm = (int)occ_cdiv(d->M, TMB); a.nbn = (int)occ_cdiv(d->N, TNB);
    const size_t shm = (size_t)NST * (TMB + TNB) * CHUNKS * sizeof(uint4);
    static bool raised = false;                     // per instantiation
    if (!raised) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_ms_kernel<WGM, WGN, NJ, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) { occ_set_error("occ_gemm: cannot raise the LDS limit: %s", hipGetErrorString(e)); return OCC_ELAUNCH; }
        raised = true;
    }
    hipLaunchKernelGGL((gemm_bf16_ms_kernel<WGM, WGN, NJ, NST>), dim3((unsigned)((long long)a.nbm * a.nbn), (unsigned)ng), dim3(WGM * WGN * 64), shm, s, a);
    return OCC_OK;
}

int cu_count() {
    static int n = 0;
    if (!n) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}

template <int WGM, int WGN, int NJ, int NST>
int launch_persist(GemmArgs& a, const occ_gemm_desc* d, long long ng, hipStream_t s) {
    constexpr int TMB = WGM * NJ * 16, TNB = WGN * 64;
    a.nbm = (int)occ_cdiv(d->M, TMB); a.nbn = (int)occ_cdiv(d->N, TNB);
    a.ngroups = (int)ng;
    a.group_m = a.nbn >= 8 ? 4 : 0;
    const long long total = (long long)a.nbm * a.nbn * ng;
    const size_t shm = (size_t)NST * (TMB + TNB) * CHUNKS * sizeof(uint4);
    static bool raised = false;
    if (!raised) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_persist_kernel<WGM, WGN, NJ, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) { occ_set_error("occ_gemm: cannot raise the LDS limit: %s", hipGetErrorString(e)); return OCC_ELAUNCH; }
        raised = true;
    }
    const long long grid = total < cu_count() ? total : cu_count();
    hipLaunchKernelGGL((gemm_bf16_persist_kernel<WGM, WGN, NJ, NST>), dim3((unsigned)grid), dim3(WGM * WGN * 64), shm, s, a);
    return OCC_OK;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int g_dbg = 0;
int g_variant = getenv("OCC_GEMM_VARIANT") ? atoi(getenv("OCC_GEMM_VARIANT")) : 1;

}  // namespace

extern "C" int occ_gemm_variant(int v) {
    const int prev = g_variant;
    if (v >= 0) g_variant = v;
    return prev;
}

extern "C" int occ_gemm_debug(int bits) {
    const int prev = g_dbg;
    if (bits >= 0) g_dbg = bits;
    return prev;
}

extern "C" int occ_gemm(const occ_gemm_desc* d, void* stream) {
    OCC_CHECK_ARG(d, "occ_gemm: null descriptor");
    OCC_CHECK_ARG(d->A && d->W && d->C, "occ_gemm: null operand");
    OCC_CHECK_ARG(d->M >= 1 && d->N >= 1 && d->K >= 1, "occ_gemm: bad shape M=%ld N=%ld K=%ld", (long)d->M, (long)d->N, (long)d->K);
    OCC_CHECK_ARG(d->ab_dtype == OCC_BF16 || d->ab_dtype == OCC_F32 || d->ab_dtype == OCC_F32_AS_BF16 || d->ab_dtype == OCC_AF32_WBF16,
                  "occ_gemm: ab_dtype must be bf16, f32, f32-as-bf16 or af32-wbf16");
    OCC_CHECK_ARG(!(d->act == OCC_ACT_GELU_GRAD && !d->aux), "occ_gemm: OCC_ACT_GELU_GRAD needs aux");
    OCC_CHECK_ARG(d->c_dtype == OCC_BF16 || d->c_dtype == OCC_F32, "occ_gemm: c_dtype must be bf16 or f32");
    OCC_CHECK_ARG(!d->R || d->r_dtype == OCC_BF16 || d->r_dtype == OCC_F32, "occ_gemm: r_dtype must be bf16 or f32");
    const int ce = d->ab_dtype == OCC_F32 ? 4 : 8;            // K granularity: one 16-byte LDS chunk
    OCC_CHECK_ARG(d->K % ce == 0 && d->N % 4 == 0, "occ_gemm: needs K %% %d == 0 and N %% 4 == 0 (K=%ld N=%ld)", ce, (long)d->K, (long)d->N);
    const long long nseg = d->a_nseg > 1 ? d->a_nseg : 1;
    const long long seg_len = nseg > 1 ? d->a_seg_len : d->K;
    OCC_CHECK_ARG(nseg * seg_len == d->K, "occ_gemm: a_nseg*a_seg_len != K");
    OCC_CHECK_ARG(seg_len % ce == 0 && (nseg == 1 || d->a_seg_stride % ce == 0), "occ_gemm: K segments must be 16-byte granular");
    OCC_CHECK_ARG(d->a_map.rows_per_batch >= 1 && d->c_map.rows_per_batch >= 1, "occ_gemm: rows_per_batch must be >= 1");
    OCC_CHECK_ARG(d->a_map.line_stride % ce == 0 && d->c_map.line_stride % 4 == 0 && (!d->R || d->r_map.line_stride % 4 == 0),
                  "occ_gemm: line strides must keep rows 16-byte aligned");
    OCC_CHECK_ARG(d->a_map.row_stride % ce == 0 && d->a_map.batch_stride % ce == 0 && d->ldw % ce == 0 && d->ldw >= d->K,
                  "occ_gemm: A/W strides must keep rows 16-byte aligned");
    OCC_CHECK_ARG(d->c_map.row_stride % 4 == 0 && d->c_map.batch_stride % 4 == 0, "occ_gemm: C strides must be multiples of 4 elements");
    OCC_CHECK_ARG(aligned16(d->A) && aligned16(d->W) && aligned16(d->C) && (!d->bias || aligned16(d->bias)) && (!d->R || aligned16(d->R)),
                  "occ_gemm: operands must be 16-byte aligned");
    if (d->R) OCC_CHECK_ARG(d->r_map.rows_per_batch >= 1 && d->r_map.row_stride % 4 == 0 && d->r_map.batch_stride % 4 == 0, "occ_gemm: bad residual map");
    GemmArgs a;
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.X = (const char*)d->A; a.xmap = to_rowmap(d->a_map);
    a.nseg = nseg; a.seg_len = seg_len; a.seg_stride = d->a_seg_stride;
    a.W = (const char*)d->W; a.ldw = d->ldw;
    a.bias = (const float*)d->bias;
    a.R = (const char*)d->R; a.rmap = to_rowmap(d->r_map); a.r_dtype = d->r_dtype;
    a.C = (char*)d->C; a.cmap = to_rowmap(d->c_map); a.c_dtype = d->c_dtype;
    a.act = d->act; a.alpha = d->alpha; a.aux = (unsigned short*)d->aux;
    a.ksplit = 1; a.slabs_per_split = 0; a.ngroups = 1;
    a.dbg = g_dbg;
    a.nbm = (int)occ_cdiv(d->M, TM); a.nbn = (int)occ_cdiv(d->N, TN);
    // grouped tile order (8 m-tiles per W panel) measured +3 % on the N >= 3072 front-end GEMMs and +10 % at 4096^3, -2 % at N = 1024
    static const int group_m_env = getenv("OCC_GEMM_GROUP_M") ? atoi(getenv("OCC_GEMM_GROUP_M")) : -1;
    a.group_m = group_m_env >= 0 ? group_m_env : (a.nbn >= 16 ? 8 : 0);
    const long long ng = d->n_groups > 1 ? d->n_groups : 1;
    a.a_gstride = ng > 1 ? d->a_group_stride : 0; a.w_gstride = ng > 1 ? d->w_group_stride : 0; a.c_gstride = ng > 1 ? d->c_group_stride : 0;
    OCC_CHECK_ARG(ng < 65536 && a.a_gstride % ce == 0 && a.w_gstride % ce == 0 && a.c_gstride % 4 == 0, "occ_gemm: bad group strides");
    const long long total = (long long)a.nbm * a.nbn;
    OCC_CHECK_ARG(total < (1ll << 30), "occ_gemm: too many tiles");
    hipStream_t s = (hipStream_t)stream;
    const int variant = g_variant;
    const long long nbm256 = occ_cdiv(d->M, 256);
    // 256x128 tiles only pay on large square problems (4096^3: 999 vs 865 TFLOP/s); on the front-end shapes (M = 6368, or N = 512)
    // the 128x128 tile's finer granularity wins by 3-15 % (scripts/bench_gemm.py), so it stays the default there.
    const bool big = d->M >= 4096 && d->N >= 4096 && d->M % 256 == 0;
    if (d->ab_dtype == OCC_BF16 && d->K % 64 == 0 && (variant == 15 || variant == 16)) {
        const int rc = variant == 15 ? launch_ms32<2, 4, 8, 4>(a, d, ng, s) : launch_ms32<4, 2, 4, 5>(a, d, ng, s);
        if (rc != OCC_OK) return rc;
    } else if (d->ab_dtype == OCC_BF16 && d->K % 64 == 0 && variant == 14) {
        hipLaunchKernelGGL(gemm_bf16_hs_kernel, dim3((unsigned)total, (unsigned)ng), dim3(THREADS), 0, s, a);
    } else if (d->ab_dtype == OCC_BF16 && d->K % 64 == 0 && variant == 13) {
        a.nbm = (int)occ_cdiv(d->M, TM); a.nbn = (int)occ_cdiv(d->N, TN);
        a.ngroups = (int)ng;
        const long long tot = (long long)a.nbm * a.nbn * ng;
        const long long grid = tot < 4ll * cu_count() ? tot : 4ll * cu_count();
        hipLaunchKernelGGL(gemm_bf16_p1_kernel, dim3((unsigned)grid), dim3(256), 0, s, a);
    } else if (d->ab_dtype == OCC_BF16 && d->K % 64 == 0 && (variant == 11 || variant == 12)) {
        const int rc = variant == 11 ? launch_persist<4, 2, 4, 3>(a, d, ng, s) : launch_persist<2, 2, 4, 4>(a, d, ng, s);
        if (rc != OCC_OK) return rc;
    } else if (d->ab_dtype == OCC_BF16 && d->K % 64 == 0 && (variant == 5 || variant == 6 || variant == 8 || variant == 9)) {
        const int rc = variant == 5 ? launch_ms<4, 2, 4, 3>(a, d, ng, s) : variant == 6 ? launch_ms<2, 4, 8, 2>(a, d, ng, s)
                     : variant == 8 ? launch_ms<2, 2, 4, 4>(a, d, ng, s) : launch_ms<2, 2, 4, 3>(a, d, ng, s);
        if (rc != OCC_OK) return rc;
    } else if (d->ab_dtype == OCC_BF16 && d->K % 64 == 0 && (variant == 3 || (variant == 1 && big))) {
        a.nbm = (int)nbm256;
        hipLaunchKernelGGL(gemm_bf16_dma_kernel<256>, dim3((unsigned)(nbm256 * a.nbn), (unsigned)ng), dim3(512), 0, s, a);
    } else if (d->ab_dtype == OCC_BF16 && d->K % 64 == 0 && (variant == 1 || variant == 4)) {
        // Few output tiles but a very long K (weight gradients: K = batch x frames): split K over workgroups, f32 atomics into C.
        // Only for the accumulate form (R aliases C, f32, same row map), where adding the pieces onto C is the requested result.
        const bool accumulate = d->R == d->C && d->c_dtype == OCC_F32 && d->r_dtype == OCC_F32 && !d->bias && d->act == OCC_ACT_NONE && !d->aux &&
                                d->r_map.rows_per_batch == d->c_map.rows_per_batch && d->r_map.row_stride == d->c_map.row_stride &&
                                d->r_map.batch_stride == d->c_map.batch_stride && d->r_map.rows_per_line == d->c_map.rows_per_line &&
                                d->r_map.line_stride == d->c_map.line_stride;
        // measured (scripts/bench_wgrad.py): the atomics are expensive, so split only up to one workgroup per CU (two for the
        // conv-stack gradients whose K is hundreds of thousands): 1024x1024x12736 200 -> 104 us, 512x1536x409536 7817 -> 1164 us
        static const long long per_cu = getenv("OCC_GEMM_SPLIT_PER_CU") ? atoll(getenv("OCC_GEMM_SPLIT_PER_CU")) : -1;
        const long long nslab = d->K / 64, want = (per_cu >= 0 ? per_cu : (nslab >= 2048 ? 2 : 1)) * cu_count();
        long long split = 1;
        if (accumulate && total * ng <= want / 2 && nslab >= 32 && variant == 1) {
            split = occ_cdiv(want, total * ng);
            if (split > nslab / 8) split = nslab / 8;
            if (split < 1) split = 1;
        }
        // Under-filled launches with a long K (fc2 at M = 6368: 400 tiles for 1024 workgroup slots, 64 slabs each): the half-slab
        // pipeline keeps loads in flight under the MFMAs and wins 3-6 % there; with every slot busy the plain kernel is 15-20 % faster.
        if (split == 1 && variant == 1 && total * ng <= 2ll * cu_count() && nslab >= 32) {
            hipLaunchKernelGGL(gemm_bf16_hs_kernel, dim3((unsigned)total, (unsigned)ng), dim3(THREADS), 0, s, a);
        } else
        if (split > 1) {
            a.slabs_per_split = (int)occ_cdiv(nslab, split);
            a.ksplit = (int)occ_cdiv(nslab, a.slabs_per_split);
            a.R = nullptr;
            hipLaunchKernelGGL((gemm_bf16_dma_kernel<128, true>), dim3((unsigned)(total * a.ksplit), (unsigned)ng), dim3(THREADS), 0, s, a);
        } else
            hipLaunchKernelGGL(gemm_bf16_dma_kernel<128>, dim3((unsigned)total, (unsigned)ng), dim3(THREADS), 0, s, a);
    }
    else if (d->ab_dtype == OCC_BF16) hipLaunchKernelGGL(gemm_kernel<1>, dim3((unsigned)total, (unsigned)ng), dim3(THREADS), 0, s, a);
    else if (d->ab_dtype == OCC_F32_AS_BF16 && d->N <= 64 && g_variant != 14) {        // narrow outputs: 128x64 tile
        a.nbn = 1;
        hipLaunchKernelGGL((gemm_kernel<2, 64>), dim3((unsigned)a.nbm, (unsigned)ng), dim3(THREADS), 0, s, a);
    }
    else if (d->ab_dtype == OCC_F32_AS_BF16) hipLaunchKernelGGL(gemm_kernel<2>, dim3((unsigned)total, (unsigned)ng), dim3(THREADS), 0, s, a);
    else if (d->ab_dtype == OCC_AF32_WBF16) hipLaunchKernelGGL(gemm_kernel<3>, dim3((unsigned)total, (unsigned)ng), dim3(THREADS), 0, s, a);
    else hipLaunchKernelGGL(gemm_kernel<0>, dim3((unsigned)total, (unsigned)ng), dim3(THREADS), 0, s, a);
    OCC_LAUNCH_CHECK("occ_gemm");
    return OCC_OK;
}
