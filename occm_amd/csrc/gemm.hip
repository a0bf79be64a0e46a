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

// acc[i][j]: i = 16-column block of the wave's 64 output columns, j = 16-row block of its NJ*16 output rows;
// mrow0 / ncol0 = first row / column of the wave's sub-tile.  A lane owns C[m][n..n+3].
template <int NJ>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, f32x4 (&acc)[4][NJ], long long mrow0, long long ncol0, int fr, int fq, long long cshift) {
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
            if (a.bias) {
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

template <int MODE>
__global__ __launch_bounds__(THREADS, 2) void gemm_kernel(const GemmArgs a) {
    constexpr bool BF16 = MODE != 0;            // MFMA flavour
    constexpr int ES = MODE == 1 ? 2 : 4;       // element size of X in memory (MODE 3: X f32, W bf16)
    constexpr int WES = (MODE == 1 || MODE == 3) ? 2 : 4;
    constexpr int CE = BF16 ? 8 : 4;            // K elements per 16-B LDS chunk
    constexpr int SLAB_K = BF16 ? 64 : 32;      // K elements per slab
    __shared__ uint4 lds[2][2][TM * CHUNKS];    // [buffer][X|W][row*8 + swizzled chunk]

    // ---- XCD-aware tile id: blocks b, b+8, ... share an XCD (and its L2); give each XCD a contiguous
    // range of tiles so neighbouring tiles re-use the same X / W panels out of that L2.
    const int total = a.nbm * a.nbn;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = total >> 3, r8 = total & 7;
    const int vid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const int tile_n = vid % a.nbn, tile_m = vid / a.nbn;
    const long long m0 = (long long)tile_m * TM, n0 = (long long)tile_n * TN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const long long grp = blockIdx.y;
    const char* Xg = a.X + grp * a.a_gstride * ES;
    const char* Wg = a.W + grp * a.w_gstride * WES;
    const long long cshift = grp * a.c_gstride;      // column shift of C / R / bias for this group

    // ---- staging assignment: this thread copies chunk `ch` of rows (tid>>3) + 32*i
    const int ch = tid & 7, srow = tid >> 3;
    long long xoff[4], woff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        long long m = m0 + srow + 32 * i; if (m > a.M - 1) m = a.M - 1;
        long long n = n0 + srow + 32 * i; if (n > a.N - 1) n = a.N - 1;
        xoff[i] = row_off(a.xmap, m) * ES;
        woff[i] = n * a.ldw * WES;
    }
    const int nslab = (int)((a.K + SLAB_K - 1) / SLAB_K);

    uint4 px[4], pw[4];
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
                    if constexpr (MODE == 2) {
                        const float4* wp = reinterpret_cast<const float4*>(Wg + woff[i] + k0 * WES);
                        pw[i] = pack_bf16x8(wp[0], wp[1]);
                    } else {
                        pw[i] = *reinterpret_cast<const uint4*>(Wg + woff[i] + k0 * WES);
                    }
                } else { px[i] = make_uint4(0, 0, 0, 0); pw[i] = px[i]; }
            } else {
                px[i] = ok ? *reinterpret_cast<const uint4*>(Xg + xoff[i] + kx * ES) : make_uint4(0, 0, 0, 0);
                pw[i] = ok ? *reinterpret_cast<const uint4*>(Wg + woff[i] + k0 * ES) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto write_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = srow + 32 * i;
            const int idx = row * CHUNKS + (ch ^ (row & 7));
            lds[buf][0][idx] = px[i];
            lds[buf][1][idx] = pw[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    issue_loads(0);
    write_lds(0);
    __syncthreads();

    const int fr = lane & 15, fq = lane >> 4;
    for (int slab = 0; slab < nslab; ++slab) {
        const int cur = slab & 1;
        if (slab + 1 < nslab) issue_loads(slab + 1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            uint4 wf[4], xf[4];
            const int chk = kb * 4 + fq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rw = wn * 64 + i * 16 + fr;
                wf[i] = lds[cur][1][rw * CHUNKS + (chk ^ (rw & 7))];
                const int rx = wm * 64 + i * 16 + fr;
                xf[i] = lds[cur][0][rx * CHUNKS + (chk ^ (rx & 7))];
            }
            if constexpr (BF16) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            *reinterpret_cast<bf16x8*>(&wf[i]), *reinterpret_cast<bf16x8*>(&xf[j]), acc[i][j], 0, 0, 0);
            } else {
                // lane group fq holds k = 4*fq + e of this 16-k block; step e multiplies k in {e, 4+e, 8+e, 12+e}
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                reinterpret_cast<const float*>(&wf[i])[e], reinterpret_cast<const float*>(&xf[j])[e], acc[i][j], 0, 0, 0);
            }
        }
        if (slab + 1 < nslab) write_lds(cur ^ 1);
        __syncthreads();
    }

    gemm_epilogue<4>(a, acc, m0 + wm * 64, n0 + wn * 64, fr, fq, cshift);
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
template <int TMT>
__global__ __launch_bounds__(2 * TMT, 2 * TMT == 256 ? 4 : 2) void gemm_bf16_dma_kernel(const GemmArgs a) {
    constexpr int ES = 2, CE = 8, SLAB_K = 64;
    constexpr int NT = 2 * TMT;                 // threads
    constexpr int RPP = NT / 8;                 // rows staged per pass
    constexpr int XP = TMT / RPP, WP = TN / RPP;
    __shared__ uint4 lds[(TMT + TN) * CHUNKS];  // X rows then W rows: [row*8 + position]
    uint4* ldsX = lds; uint4* ldsW = lds + TMT * CHUNKS;
    const int total = a.nbm * a.nbn;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = total >> 3, r8 = total & 7;
    const int vid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
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
    const int nslab = (int)(a.K / SLAB_K);
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    for (int slab = 0; slab < nslab; ++slab) {
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
    gemm_epilogue<4>(a, acc, m0 + wm * 64, n0 + wn * 64, fr, fq, cshift);
}

__global__ __launch_bounds__(THREADS, 2) void gemm_bf16_dma2_kernel(const GemmArgs a) {
    constexpr int ES = 2, CE = 8, SLAB_K = 64;
    __shared__ uint4 lds[2][2][TM * CHUNKS];    // [buffer][X|W][row*8 + position]
    const int total = a.nbm * a.nbn;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = total >> 3, r8 = total & 7;
    const int vid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const int tile_n = vid % a.nbn, tile_m = vid / a.nbn;
    const long long m0 = (long long)tile_m * TM, n0 = (long long)tile_n * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const long long grp = blockIdx.y;
    const char* Xg = a.X + grp * a.a_gstride * ES;
    const char* Wg = a.W + grp * a.w_gstride * ES;
    const long long cshift = grp * a.c_gstride;

    // staging: wave-instruction i of wave `wave` fills rows 8*wave + 32*i .. +7 (1 KiB, lane-linear)
    const int pos = tid & 7, srow = tid >> 3;
    const char* xsrc[4]; const char* wsrc[4];
    int csrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = srow + 32 * i;
        long long m = m0 + row; if (m > a.M - 1) m = a.M - 1;
        long long n = n0 + row; if (n > a.N - 1) n = a.N - 1;
        xsrc[i] = Xg + row_off(a.xmap, m) * ES;
        wsrc[i] = Wg + n * a.ldw * ES;
        csrc[i] = pos ^ (row & 7);              // source chunk of this lane
    }
    const int nslab = (int)(a.K / SLAB_K);
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    auto stage = [&](int slab, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long k0 = (long long)slab * SLAB_K + csrc[i] * CE;
            long long kx = k0;
            if (a.nseg > 1) { const long long sg = k0 / a.seg_len; kx = sg * a.seg_stride + (k0 - sg * a.seg_len); }
            const int base = (wave * 8 + 32 * i) * CHUNKS;
            __builtin_amdgcn_global_load_lds((gbl_void*)(xsrc[i] + kx * ES), (lds_void*)&lds[buf][0][base], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void*)(wsrc[i] + k0 * ES), (lds_void*)&lds[buf][1][base], 16, 0, 0);
        }
    };
    stage(0, 0);
    for (int slab = 0; slab < nslab; ++slab) {
        const int cur = slab & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // slab `slab` has landed (this wave's pieces)
        __syncthreads();                                       // ... everyone's pieces; buffer cur^1 is free again
        if (slab + 1 < nslab) stage(slab + 1, cur ^ 1);        // next slab streams in under the MFMAs below
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            uint4 wf[4], xf[4];
            const int chk = kb * 4 + fq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rw = wn * 64 + i * 16 + fr;
                wf[i] = lds[cur][1][rw * CHUNKS + (chk ^ (rw & 7))];
                const int rx = wm * 64 + i * 16 + fr;
                xf[i] = lds[cur][0][rx * CHUNKS + (chk ^ (rx & 7))];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        *reinterpret_cast<bf16x8*>(&wf[i]), *reinterpret_cast<bf16x8*>(&xf[j]), acc[i][j], 0, 0, 0);
        }
    }
    gemm_epilogue<4>(a, acc, m0 + wm * 64, n0 + wn * 64, fr, fq, cshift);
}

// ------------------------------------------------------------------------------------------------
// Pipelined bf16 kernel for the big front-end GEMMs: ONE workgroup per CU (8 waves), block tile (WGM*NJ*16) x (WGN*64),
// two LDS stages filled by LDS-DMA; the DMA of slab s+1 is issued right after the barrier that publishes slab s and flies
// under the MFMAs of slab s, one barrier per slab.  256x128 (WGM=4, WGN=2, NJ=4) moves one L2 byte per 85 FLOP and
// 256x256 (WGM=2, WGN=4, NJ=8; 128x64 per wave) one per 128 FLOP -- the 128x128 tile's 64 FLOP/B is what caps it near
// 0.9 PFLOP/s (a CU pulls ~55-64 B/clk from L2, its MFMA rate needs 64 B/clk at that intensity).
template <int WGM, int WGN, int NJ>
__global__ __launch_bounds__(WGM * WGN * 64, 2) void gemm_bf16_pipe_kernel(const GemmArgs a) {
    constexpr int ES = 2, CE = 8, SLAB_K = 64;
    constexpr int NT = WGM * WGN * 64;
    constexpr int TMB = WGM * NJ * 16, TNB = WGN * 64;
    constexpr int RPP = NT / 8;                      // rows staged per pass (64 for 512 threads)
    constexpr int XP = TMB / RPP, WP = TNB / RPP;
    constexpr int STAGE = (TMB + TNB) * CHUNKS;      // uint4 per stage
    extern __shared__ uint4 plds[];                  // [2][X rows | W rows][8 positions]
    const int total = a.nbm * a.nbn;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = total >> 3, r8 = total & 7;
    const int vid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const int tile_n = vid % a.nbn, tile_m = vid / a.nbn;
    const long long m0 = (long long)tile_m * TMB, n0 = (long long)tile_n * TNB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WGM, wn = wave / WGM;
    const long long grp = blockIdx.y;
    const char* Xg = a.X + grp * a.a_gstride * ES;
    const char* Wg = a.W + grp * a.w_gstride * ES;
    const long long cshift = grp * a.c_gstride;

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
    const int nslab = (int)(a.K / SLAB_K);
    auto stage = [&](int slab, int buf) {
        uint4* sx = plds + buf * STAGE;
        uint4* sw = sx + TMB * CHUNKS;
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const long long k0 = (long long)slab * SLAB_K + xc[i] * CE;
            long long kx = k0;
            if (a.nseg > 1) { const long long sg = k0 / a.seg_len; kx = sg * a.seg_stride + (k0 - sg * a.seg_len); }
            __builtin_amdgcn_global_load_lds((gbl_void*)(xsrc[i] + kx * ES), (lds_void*)&sx[(wave * 8 + RPP * i) * CHUNKS], 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WP; ++i) {
            const long long k0 = (long long)slab * SLAB_K + wc[i] * CE;
            __builtin_amdgcn_global_load_lds((gbl_void*)(wsrc[i] + k0 * ES), (lds_void*)&sw[(wave * 8 + RPP * i) * CHUNKS], 16, 0, 0);
        }
    };
    f32x4 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    stage(0, 0);
    for (int slab = 0; slab < nslab; ++slab) {
        const int cur = slab & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of slab `slab` have landed
        __syncthreads();                                       // everyone's have; stage cur^1 is no longer being read
        if (slab + 1 < nslab) stage(slab + 1, cur ^ 1);
        const uint4* sx = plds + cur * STAGE;
        const uint4* sw = sx + TMB * CHUNKS;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            uint4 wf[4], xf[NJ];
            const int chk = kb * 4 + fq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rw = wn * 64 + i * 16 + fr;
                wf[i] = sw[rw * CHUNKS + (chk ^ (rw & 7))];
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int rx = wm * (NJ * 16) + j * 16 + fr;
                xf[j] = sx[rx * CHUNKS + (chk ^ (rx & 7))];
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        *reinterpret_cast<bf16x8*>(&wf[i]), *reinterpret_cast<bf16x8*>(&xf[j]), acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
    }
    gemm_epilogue<NJ>(a, acc, m0 + wm * (NJ * 16), n0 + wn * 64, fr, fq, cshift);
}

// ------------------------------------------------------------------------------------------------
// Producer / consumer variant: 256x128 block tile, 8 consumer waves (4x2, 64x64 each) that ONLY read LDS and issue MFMAs,
// plus 4 loader waves (one per SIMD) that ONLY issue the LDS-DMA of the next slab.  An LDS-DMA instruction costs its wave
// 60-180 issue cycles (MI355X_MICROARCH.md, cycle constants); taking the 12 per slab out of the MFMA waves' in-order streams
// is what the deep-pipelined templates achieve by hand interleaving.  Two LDS stages, one barrier per slab.
__global__ __launch_bounds__(768, 3) void gemm_bf16_pc_kernel(const GemmArgs a) {
    constexpr int ES = 2, CE = 8, SLAB_K = 64;
    constexpr int TMB = 256, TNB = 128;
    constexpr int STAGE = (TMB + TNB) * CHUNKS;
    extern __shared__ uint4 plds[];
    const int total = a.nbm * a.nbn;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = total >> 3, r8 = total & 7;
    const int vid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const int tile_n = vid % a.nbn, tile_m = vid / a.nbn;
    const long long m0 = (long long)tile_m * TMB, n0 = (long long)tile_n * TNB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long grp = blockIdx.y;
    const char* Xg = a.X + grp * a.a_gstride * ES;
    const char* Wg = a.W + grp * a.w_gstride * ES;
    const long long cshift = grp * a.c_gstride;
    const int nslab = (int)(a.K / SLAB_K);

    if (wave >= 8) {
        // ---------------- loader waves: 48 one-KiB pieces per slab, 12 per wave ----------------
        const int lw = wave - 8;
        const char* src[12]; int chunk[12]; int dstoff[12]; bool isx[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int blk = lw * 12 + i;                       // 0..31: X row blocks, 32..47: W row blocks (8 rows each)
            isx[i] = blk < 32;
            const int row = (isx[i] ? blk : blk - 32) * 8 + (lane >> 3);
            chunk[i] = (lane & 7) ^ (row & 7);
            if (isx[i]) {
                long long m = m0 + row; if (m > a.M - 1) m = a.M - 1;
                src[i] = Xg + row_off(a.xmap, m) * ES;
                dstoff[i] = (blk * 8) * CHUNKS;
            } else {
                long long n = n0 + row; if (n > a.N - 1) n = a.N - 1;
                src[i] = Wg + n * a.ldw * ES;
                dstoff[i] = (TMB + (blk - 32) * 8) * CHUNKS;
            }
        }
        auto stage = [&](int slab, int buf) {
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const long long k0 = (long long)slab * SLAB_K + chunk[i] * CE;
                long long kk = k0;
                if (isx[i] && a.nseg > 1) { const long long sg = k0 / a.seg_len; kk = sg * a.seg_stride + (k0 - sg * a.seg_len); }
                __builtin_amdgcn_global_load_lds((gbl_void*)(src[i] + kk * ES), (lds_void*)&plds[buf * STAGE + dstoff[i]], 16, 0, 0);
            }
        };
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int slab = 0; slab < nslab; ++slab) {
            if (slab + 1 < nslab) { stage(slab + 1, (slab + 1) & 1); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            __syncthreads();
        }
        return;
    }
    // ---------------- consumer waves ----------------
    const int wm = wave & 3, wn = wave >> 2;
    const int fr = lane & 15, fq = lane >> 4;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    for (int slab = 0; slab < nslab; ++slab) {
        const uint4* sx = plds + (slab & 1) * STAGE;
        const uint4* sw = sx + TMB * CHUNKS;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            uint4 wf[4], xf[4];
            const int chk = kb * 4 + fq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rw = wn * 64 + i * 16 + fr;
                wf[i] = sw[rw * CHUNKS + (chk ^ (rw & 7))];
                const int rx = wm * 64 + i * 16 + fr;
                xf[i] = sx[rx * CHUNKS + (chk ^ (rx & 7))];
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
    gemm_epilogue<4>(a, acc, m0 + wm * 64, n0 + wn * 64, fr, fq, cshift);
}

template <int WGM, int WGN, int NJ>
int launch_pipe(GemmArgs& a, const occ_gemm_desc* d, long long ng, hipStream_t s) {
    constexpr int TMB = WGM * NJ * 16, TNB = WGN * 64;
    a.nbm = (int)occ_cdiv(d->M, TMB); a.nbn = (int)occ_cdiv(d->N, TNB);
    const size_t shm = (size_t)2 * (TMB + TNB) * CHUNKS * sizeof(uint4);
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_pipe_kernel<WGM, WGN, NJ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) { occ_set_error("occ_gemm: cannot raise the LDS limit: %s", hipGetErrorString(e)); return OCC_ELAUNCH; }
    hipLaunchKernelGGL((gemm_bf16_pipe_kernel<WGM, WGN, NJ>), dim3((unsigned)((long long)a.nbm * a.nbn), (unsigned)ng), dim3(WGM * WGN * 64), shm, s, a);
    return OCC_OK;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

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
    static const int variant = getenv("OCC_GEMM_VARIANT") ? atoi(getenv("OCC_GEMM_VARIANT")) : 1;
    const long long nbm256 = occ_cdiv(d->M, 256);
    // 256x128 tiles only pay on large square problems (4096^3: 999 vs 865 TFLOP/s); on the front-end shapes (M = 6368, or N = 512)
    // the 128x128 tile's finer granularity wins by 3-15 % (scripts/bench_gemm.py), so it stays the default there.
    const bool big = d->M >= 4096 && d->N >= 4096 && d->M % 256 == 0;
    if (d->ab_dtype == OCC_BF16 && d->K % 64 == 0 && variant == 7) {
        a.nbm = (int)occ_cdiv(d->M, 256); a.nbn = (int)occ_cdiv(d->N, 128);
        const size_t shm = (size_t)2 * (256 + 128) * CHUNKS * sizeof(uint4);
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_pc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) { occ_set_error("occ_gemm: cannot raise the LDS limit: %s", hipGetErrorString(e)); return OCC_ELAUNCH; }
        hipLaunchKernelGGL(gemm_bf16_pc_kernel, dim3((unsigned)((long long)a.nbm * a.nbn), (unsigned)ng), dim3(768), shm, s, a);
    } else if (d->ab_dtype == OCC_BF16 && d->K % 64 == 0 && (variant == 5 || variant == 6)) {
        const int rc = variant == 5 ? launch_pipe<4, 2, 4>(a, d, ng, s) : launch_pipe<2, 4, 8>(a, d, ng, s);
        if (rc != OCC_OK) return rc;
    } else if (d->ab_dtype == OCC_BF16 && d->K % 64 == 0 && (variant == 3 || (variant == 1 && big))) {
        a.nbm = (int)nbm256;
        hipLaunchKernelGGL(gemm_bf16_dma_kernel<256>, dim3((unsigned)(nbm256 * a.nbn), (unsigned)ng), dim3(512), 0, s, a);
    } else if (d->ab_dtype == OCC_BF16 && d->K % 64 == 0 && (variant == 1 || variant == 4))
        hipLaunchKernelGGL(gemm_bf16_dma_kernel<128>, dim3((unsigned)total, (unsigned)ng), dim3(THREADS), 0, s, a);
    else if (d->ab_dtype == OCC_BF16 && d->K % 64 == 0 && variant == 2)
        hipLaunchKernelGGL(gemm_bf16_dma2_kernel, dim3((unsigned)total, (unsigned)ng), dim3(THREADS), 0, s, a);
    else if (d->ab_dtype == OCC_BF16) hipLaunchKernelGGL(gemm_kernel<1>, dim3((unsigned)total, (unsigned)ng), dim3(THREADS), 0, s, a);
    else if (d->ab_dtype == OCC_F32_AS_BF16) hipLaunchKernelGGL(gemm_kernel<2>, dim3((unsigned)total, (unsigned)ng), dim3(THREADS), 0, s, a);
    else if (d->ab_dtype == OCC_AF32_WBF16) hipLaunchKernelGGL(gemm_kernel<3>, dim3((unsigned)total, (unsigned)ng), dim3(THREADS), 0, s, a);
    else hipLaunchKernelGGL(gemm_kernel<0>, dim3((unsigned)total, (unsigned)ng), dim3(THREADS), 0, s, a);
    OCC_LAUNCH_CHECK("occ_gemm");
    return OCC_OK;
}
