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
#include "gemm_common.h"

namespace occ_gemm_detail {

__device__ __forceinline__ uint4 pack_bf16x8(const float4 lo, const float4 hi) {
    return make_uint4(pack_bf16x2(lo.x, lo.y),
                      pack_bf16x2(lo.z, lo.w),
                      pack_bf16x2(hi.x, hi.y),
                      pack_bf16x2(hi.z, hi.w));
}

// the lo parts of a hi / lo bf16 split of eight f32 values (hi = bf16(v) as pack_bf16x8 gives it, lo = bf16(v - hi))
__device__ __forceinline__ uint4 pack_bf16x8_lo(const float4 lo, const float4 hi) {
    auto r = [](float v) { return v - bf16_bits_to_f32(f32_to_bf16_bits(v)); };
    return pack_bf16x8(make_float4(r(lo.x), r(lo.y), r(lo.z), r(lo.w)), make_float4(r(hi.x), r(hi.y), r(hi.z), r(hi.w)));
}

// TNT = 128: 2x2 waves, 64x64 per wave, 64 KiB LDS, 2 workgroups per CU.  TNT = 64 (narrow outputs: the back-ends' 32/64-channel
// convolutions): 4x1 waves, 32x64 per wave, 48 KiB LDS, 3 workgroups per CU, no MFMA work on columns that do not exist.
template <int MODE, int TNT = 128>
__global__ __launch_bounds__(THREADS, TNT == 128 ? 2 : 3) void gemm_kernel(const GemmArgs a) {
    constexpr int NJ = TNT == 128 ? 4 : 2;      // 16-row blocks per wave
    constexpr int WPASS = TNT / 32;             // staging passes over the W rows
    constexpr bool BF16 = MODE != 0;            // MFMA flavour
    // MODE 4 ("f32 x3"): f32 operands in memory, each split while staged into bf16 hi = bf16(v) and lo = bf16(v - hi); the product is
    // three bf16 MFMAs per block, Wh.Xh + Wl.Xh + Wh.Xl (the dropped Wl.Xl term is 2^-16 relative): f32-grade results at 3/16 of
    // the exact-f32 MFMA's cycles.  A slab is 32 K elements: chunks 0-3 of a row hold the hi parts, chunks 4-7 the lo parts.
    constexpr bool SPLIT3 = MODE == 4;
    constexpr int ES = MODE == 1 ? 2 : 4;       // element size of X in memory (MODE 3: X f32, W bf16)
    constexpr int WES = (MODE == 1 || MODE == 3) ? 2 : 4;
    constexpr int CE = BF16 ? 8 : 4;            // K elements per 16-B LDS chunk
    constexpr int SLAB_K = (BF16 && !SPLIT3) ? 64 : 32;      // K elements per slab
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
        const long long k0 = (long long)slab * SLAB_K + (SPLIT3 ? (ch & 3) : ch) * CE;
        long long kx = k0;
        if (a.nseg > 1) { const long long sg = k0 / a.seg_len; kx = sg * a.seg_stride + (k0 - sg * a.seg_len); }
        const bool ok = k0 < a.K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (SPLIT3) {
                if (ok) {
                    const float4* xp = reinterpret_cast<const float4*>(Xg + xoff[i] + kx * ES);
                    px[i] = ch < 4 ? pack_bf16x8(xp[0], xp[1]) : pack_bf16x8_lo(xp[0], xp[1]);
                    if (i < WPASS) {
                        const float4* wp = reinterpret_cast<const float4*>(Wg + woff[i % WPASS] + k0 * WES);
                        pw[i % WPASS] = ch < 4 ? pack_bf16x8(wp[0], wp[1]) : pack_bf16x8_lo(wp[0], wp[1]);
                    }
                } else { px[i] = make_uint4(0, 0, 0, 0); if (i < WPASS) pw[i % WPASS] = px[i]; }
            } else
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
        if constexpr (SPLIT3) {
            uint4 wh[4], wl[4], xh[NJ], xl[NJ];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rw = wn * 64 + i * 16 + fr;
                wh[i] = lds[cur][TM * CHUNKS + rw * CHUNKS + (fq ^ (rw & 7))];
                wl[i] = lds[cur][TM * CHUNKS + rw * CHUNKS + ((4 + fq) ^ (rw & 7))];
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int rx = wm * (NJ * 16) + j * 16 + fr;
                xh[j] = lds[cur][rx * CHUNKS + (fq ^ (rx & 7))];
                xl[j] = lds[cur][rx * CHUNKS + ((4 + fq) ^ (rx & 7))];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    // smallest terms first
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wl[i]), *reinterpret_cast<bf16x8*>(&xh[j]), acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wh[i]), *reinterpret_cast<bf16x8*>(&xl[j]), acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&wh[i]), *reinterpret_cast<bf16x8*>(&xh[j]), acc[i][j], 0, 0, 0);
                }
        } else
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

// TMT = 128: 4 waves (2x2), 32 KiB LDS, 4 workgroups per CU.  TMT = 256: 8 waves (4x2) on a 256x128 tile, 48 KiB LDS,
// 2 workgroups per CU: the same 16 waves per CU but 25 % fewer L2->LDS bytes per FLOP (the 128x128 tile moves one byte
// per 64 FLOP, which is about what a CU can pull from L2 at its MFMA rate).
template <int TMT, bool SPLITK = false, int TR = TMT>      // TR: rows a tile owns; 112 (with TMT = 128) is the one short height compiled
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
    constexpr int trows = TR;                               // rows this tile owns (<= TMT; TR < TMT = short tile, the rows beyond belong to the next tile)
    const long long m0 = (long long)tile_m * trows, n0 = (long long)tile_n * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = TMT == 128 ? (wave & 1) : (wave & 3), wn = TMT == 128 ? (wave >> 1) : (wave >> 2);
    const int njv = TR == TMT || wm == 0 ? 4 : (TR - 64) / 16;   // 16-row blocks of this wave inside the tile (short tiles are 80 .. 112 rows)
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
            for (int j = 0; j < 4; ++j) {
                if (TR != TMT && j == 3 && njv < 4) continue;          // short tile (TR = 112): the upper waves own three row blocks (wave-uniform)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        *reinterpret_cast<bf16x8*>(&wf[i]), *reinterpret_cast<bf16x8*>(&xf[j]), acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    const long long mlim = TR == TMT ? -1 : (m0 + trows < a.M ? m0 + trows : a.M);     // short tile: stop storing at its own end
    if constexpr (SPLITK) gemm_epilogue_atomic<4>(a, acc, m0 + wm * 64, n0 + wn * 64, fr, fq, cshift, mlim);
    else gemm_epilogue<4>(a, acc, m0 + wm * 64, n0 + wn * 64, fr, fq, cshift, nullptr, mlim);
}

// ------------------------------------------------------------------------------------------------
// Narrow outputs (N <= 64 per group: the grouped positional conv has 16 groups of 64 output channels): a 128x64 tile, the four waves
// stacked along M (32 rows x 64 columns each, 8 accumulators).  In the 128x128 kernel half of every W image and half of the MFMAs
// of such a launch are padding; here the image is 16 + 8 KiB and a wave needs ~70 registers, so five workgroups share a CU.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void gemm_bf16_dma_n64_kernel(const GemmArgs a) {
    constexpr int ES = 2, CE = 8, SLAB_K = 64, TMT = 128, TNT = 64;
    constexpr int RPP = 32, XP = TMT / RPP, WP = TNT / RPP;
    __shared__ uint4 lds[(TMT + TNT) * CHUNKS];
    uint4* ldsX = lds; uint4* ldsW = lds + TMT * CHUNKS;
    const int total = a.nbm * a.nbn;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = total >> 3, r8 = total & 7;
    const int vid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const int tile_n = vid % a.nbn, tile_m = vid / a.nbn;
    const long long m0 = (long long)tile_m * TMT, n0 = (long long)tile_n * TNT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long grp = blockIdx.y;
    const char* Xg = a.X + grp * a.a_gstride * ES;
    const char* Wg = a.W + grp * a.w_gstride * ES;
    const long long cshift = grp * a.c_gstride;
    const int pos = tid & 7, srow = tid >> 3;
    const int ck = pos ^ (srow & 7);
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
    const int nslab = (int)(a.K / SLAB_K);
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const int fr = lane & 15, fq = lane >> 4;
    for (int slab = 0; slab < nslab; ++slab) {
        const long long k0 = (long long)slab * SLAB_K + ck * CE;
        long long kx = k0;
        if (a.nseg > 1) { const long long sg = k0 / a.seg_len; kx = sg * a.seg_stride + (k0 - sg * a.seg_len); }
#pragma unroll
        for (int i = 0; i < XP; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(xsrc[i] + kx * ES), (lds_void*)&ldsX[(wave * 8 + RPP * i) * CHUNKS], 16, 0, 0);
#pragma unroll
        for (int i = 0; i < WP; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(wsrc[i] + k0 * ES), (lds_void*)&ldsW[(wave * 8 + RPP * i) * CHUNKS], 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            uint4 wf[4], xf[2];
            const int chk = kb * 4 + fq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rw = i * 16 + fr;
                wf[i] = ldsW[rw * CHUNKS + (chk ^ (rw & 7))];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int rx = wave * 32 + j * 16 + fr;
                xf[j] = ldsX[rx * CHUNKS + (chk ^ (rx & 7))];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        *reinterpret_cast<bf16x8*>(&wf[i]), *reinterpret_cast<bf16x8*>(&xf[j]), acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    gemm_epilogue<2>(a, acc, m0 + wave * 32, n0, fr, fq, cshift);
}

// ------------------------------------------------------------------------------------------------
// Split-K inside the workgroup (under-filled long-K launches: fc2 at M = 6368 is 400 tiles x 64 slabs for 256 CUs): EIGHT waves per
// 128x128 tile, waves 0-3 walk the first half of K and waves 4-7 the second half, each group with its own 32 KiB LDS image and the
// default kernel's loop; at the end the upper group parks its accumulators in LDS (the two images are exactly the 64 KiB needed) and
// the lower group adds them and runs the epilogue.  Twice the loads in flight per tile and half the K steps, a CU that owns one or two
// tiles runs 8 / 16 waves instead of 4 / 8, and the sum order is fixed (no atomics: the frozen front-end stays bit-reproducible).
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_bf16_ks2_kernel(const GemmArgs a) {
    constexpr int ES = 2, CE = 8, SLAB_K = 64, TMT = 128;
    constexpr int RPP = 32, XP = 4, WP = 4;                     // each group of 256 threads stages its own image as the default kernel does
    __shared__ uint4 lds[2 * (TMT + TN) * CHUNKS];
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
    const int kh = threadIdx.x >> 8;                            // K half of this wave's group
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    uint4* ldsX = lds + kh * (TMT + TN) * CHUNKS; uint4* ldsW = ldsX + TMT * CHUNKS;
    const long long grp = blockIdx.y;
    const char* Xg = a.X + grp * a.a_gstride * ES;
    const char* Wg = a.W + grp * a.w_gstride * ES;
    const long long cshift = grp * a.c_gstride;
    const int pos = tid & 7, srow = tid >> 3;
    const char* xsrc[XP]; const char* wsrc[WP];
    const int ck = pos ^ (srow & 7);
#pragma unroll
    for (int i = 0; i < XP; ++i) {
        long long m = m0 + srow + RPP * i; if (m > a.M - 1) m = a.M - 1;
        xsrc[i] = Xg + row_off(a.xmap, m) * ES;
        long long n = n0 + srow + RPP * i; if (n > a.N - 1) n = a.N - 1;
        wsrc[i] = Wg + n * a.ldw * ES;
    }
    const int nhalf = (int)(a.K / SLAB_K) / 2;                  // slabs per K half (the launcher guarantees an even slab count)
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    for (int sl = 0; sl < nhalf; ++sl) {
        const long long k0 = (long long)(kh * nhalf + sl) * SLAB_K + ck * CE;
        long long kx = k0;
        if (a.nseg > 1) { const long long sg = k0 / a.seg_len; kx = sg * a.seg_stride + (k0 - sg * a.seg_len); }
#pragma unroll
        for (int i = 0; i < XP; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(xsrc[i] + kx * ES), (lds_void*)&ldsX[(wave * 8 + RPP * i) * CHUNKS], 16, 0, 0);
#pragma unroll
        for (int i = 0; i < WP; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(wsrc[i] + k0 * ES), (lds_void*)&ldsW[(wave * 8 + RPP * i) * CHUNKS], 16, 0, 0);
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
    // upper half -> LDS (lane-linear 16-byte slots: [wave][block][lane]), lower half adds in a fixed order and finishes the tile
    f32x4* red = reinterpret_cast<f32x4*>(lds);
    if (kh == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) red[(wave * 16 + i * 4 + j) * 64 + lane] = acc[i][j];
    }
    __syncthreads();
    if (kh == 1) return;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += red[(wave * 16 + i * 4 + j) * 64 + lane];
    gemm_epilogue<4>(a, acc, m0 + wm * 64, n0 + wn * 64, fr, fq, cshift);
}

// ------------------------------------------------------------------------------------------------
// Half-slab pipeline (occ_gemm_variant 14; the fallback for under-filled long-K launches whose slab count is odd): the 32 KiB of the default kernel cut into two 16 KiB halves of K = 32, so that one
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

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Rows [m1, M) of a row map as a map of their own: true (and the element offset of row m1) when moving the base pointer does it --
// no line structure, and m1 either inside the first batch of a single-batch map or on a batch boundary.
// out[n] += sum over the nrows per-tile partial rows the 256-row kernel's epilogue left (c_colsum), in row order: bit-reproducible
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ part, int nrows, long long N, float* __restrict__ out) {
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    int r = 0;
    for (; r + 8 <= nrows; r += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(long long)(r + u) * N + n];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; r < nrows; ++r) s += part[(long long)r * N + n];
    out[n] += s;
}
static void colsum_finalize(const GemmArgs& a, float* out, hipStream_t s) {
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((unsigned)occ_cdiv(a.N, 256)), dim3(256), 0, s, (const float*)a.colsum_part, 2 * a.nbm, a.N, out);
}

bool rows_rebase(const occ_rowmap& m, long long M, long long m1, long long* off) {
    if (m.rows_per_line > 0 || m.rows_per_batch < 1) return false;
    if (m.rows_per_batch >= M) { *off = m1 * m.row_stride; return true; }
    if (m1 % m.rows_per_batch == 0) { *off = (m1 / m.rows_per_batch) * m.batch_stride; return true; }
    return false;
}

int g_variant = getenv("OCC_GEMM_VARIANT") ? atoi(getenv("OCC_GEMM_VARIANT")) : 1;

}  // namespace occ_gemm_detail

using namespace occ_gemm_detail;

thread_local int g_last_kernel = -1;          // OCC_GEMM_KERNEL_* of the calling thread's last occ_gemm launch

extern "C" int occ_gemm_last_kernel(void) { return g_last_kernel; }

extern "C" int occ_gemm_variant(int v) {
    const int prev = g_variant;
    if (v >= 0) g_variant = v;
    return prev;
}

extern "C" int occ_gemm(const occ_gemm_desc* d, void* stream) {
    OCC_CHECK_ARG(d, "occ_gemm: null descriptor");
    OCC_CHECK_ARG(d->A && d->W && d->C, "occ_gemm: null operand");
    OCC_CHECK_ARG(d->M >= 1 && d->N >= 1 && d->K >= 1, "occ_gemm: bad shape M=%ld N=%ld K=%ld", (long)d->M, (long)d->N, (long)d->K);
    const bool fp8 = d->ab_dtype == OCC_FP8_E4M3 || d->ab_dtype == OCC_FP8_E5M2;
    OCC_CHECK_ARG(d->ab_dtype == OCC_BF16 || d->ab_dtype == OCC_F32 || d->ab_dtype == OCC_F32_AS_BF16 || d->ab_dtype == OCC_AF32_WBF16 || d->ab_dtype == OCC_F32X3 || fp8,
                  "occ_gemm: ab_dtype must be bf16, f32, f32-as-bf16, af32-wbf16 or fp8 (e4m3 / e5m2)");
    OCC_CHECK_ARG(!(d->act == OCC_ACT_GELU_GRAD && !d->aux), "occ_gemm: OCC_ACT_GELU_GRAD needs aux");
    if (d->act == OCC_ACT_GELU_KEEP_GRAD || d->act == OCC_ACT_MUL_AUX) {
        OCC_CHECK_ARG(d->aux && d->c_dtype == OCC_BF16 && !d->R && d->alpha == 1.0f, "occ_gemm: OCC_ACT_GELU_KEEP_GRAD / OCC_ACT_MUL_AUX need aux, a bf16 C, no residual, alpha = 1");
        if (!(d->ab_dtype == OCC_BF16 || d->ab_dtype == OCC_FP8_E4M3 || d->ab_dtype == OCC_FP8_E5M2)) {
            occ_set_error("occ_gemm: OCC_ACT_GELU_KEEP_GRAD / OCC_ACT_MUL_AUX exist in the bf16 / fp8 kernels' epilogues only (ab_dtype %d)", d->ab_dtype);
            return OCC_EUNSUPPORTED;
        }
    }
    OCC_CHECK_ARG(d->c_dtype == OCC_BF16 || d->c_dtype == OCC_F32, "occ_gemm: c_dtype must be bf16 or f32");
    OCC_CHECK_ARG(!d->R || d->r_dtype == OCC_BF16 || d->r_dtype == OCC_F32, "occ_gemm: r_dtype must be bf16 or f32");
    const int ce = d->ab_dtype == OCC_F32 ? 4 : (fp8 ? 16 : 8);            // K granularity: one 16-byte LDS chunk
    OCC_CHECK_ARG(!fp8 || (d->K % 128 == 0 && d->a_nseg <= 1 && d->n_groups <= 1), "occ_gemm: fp8 needs K %% 128 == 0, one K segment, one group (K=%ld)", (long)d->K);
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
    a.dq_a = fp8 ? d->a_dequant : nullptr; a.dq_w = fp8 ? d->w_dequant : nullptr;
    a.ksplit = 1; a.slabs_per_split = 0; a.ngroups = 1;
    a.colsum_part = d->c_colsum ? d->c_colsum_ws : nullptr;
    OCC_CHECK_ARG(!d->c_colsum || (d->c_colsum_ws && ((uintptr_t)d->c_colsum_ws & 15) == 0 && d->c_colsum_ws_floats >= 2 * occ_cdiv(d->M, 208) * d->N && d->c_dtype == OCC_BF16 && !d->R && d->N % 8 == 0),
                  "occ_gemm: c_colsum needs a bf16 result, no residual, N %% 8 == 0 and 2*ceil(M/208)*N floats of 16-byte aligned scratch");
    a.f8_out = (unsigned char*)d->c_f8; a.f8_scale = d->c_f8_scale; a.f8_amax = d->c_f8_amax; a.f8_e5m2 = d->c_f8_fmt == OCC_FP8_E5M2;
    OCC_CHECK_ARG(!d->c_f8 || (d->c_dtype == OCC_BF16 && !d->R && d->N % 8 == 0 && d->c_f8_scale && (d->c_f8_fmt == OCC_FP8_E4M3 || d->c_f8_fmt == OCC_FP8_E5M2)),
                  "occ_gemm: c_f8 needs a bf16 result, no residual, N %% 8 == 0, a scale and an fp8 format");
    OCC_CHECK_ARG(!d->c_colsum || (rows_epilogue_applies(a) && d->c_map.rows_per_line == 0 && d->c_map.rows_per_batch >= d->M), "occ_gemm: c_colsum needs one of the row-epilogue forms and a plain C row map");
    OCC_CHECK_ARG(!d->c_f8 || (rows_epilogue_applies(a) && d->c_map.rows_per_line == 0 && d->c_map.rows_per_batch >= d->M), "occ_gemm: c_f8 needs one of the row-epilogue forms and a plain C row map");
    a.nbm = (int)occ_cdiv(d->M, TM); a.nbn = (int)occ_cdiv(d->N, TN);
    a.tile_rows = TM;
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
    // The 256x256 eight-phase kernel (gemm_p8.hip) forms 32-bit byte offsets from the operand bases: largest one it can form = last row
    // of A through its row map / last row of W, plus one row of K.
    const long long es = fp8 ? 1 : 2;
    const long long last_a = ((d->M - 1) / d->a_map.rows_per_batch) * d->a_map.batch_stride +
                             (d->a_map.rows_per_line > 0 ? (d->a_map.rows_per_batch / d->a_map.rows_per_line + 1) * d->a_map.line_stride + d->a_map.rows_per_line * d->a_map.row_stride
                                                         : d->a_map.rows_per_batch * d->a_map.row_stride);
    const bool p8_fits = (last_a + d->K) * es < (1ll << 32) && d->N * d->ldw * es < (1ll << 32);
    if (fp8) {                                  // the only fp8 kernel (any size: edge tiles are clamped / masked as for bf16)
        OCC_CHECK_ARG(p8_fits, "occ_gemm: fp8 operands too large for 32-bit DMA offsets");
        g_last_kernel = OCC_GEMM_KERNEL_P8_FP8;
        // tile height by rounds x rows, as for bf16 below (until late round 4 fp8 always took 256-row tiles: 200 / 600 / 800 workgroups for
        // 256 CUs at M = 12736 -- one, three and four rounds at 78 % fill)
        static const int f8_rows_env = getenv("OCC_GEMM_FP8_ROWS") ? atoi(getenv("OCC_GEMM_FP8_ROWS")) : 0;     // 256 / 224 / 208 forces one; 0 = by cost
        int f8_rows = 256;
        if (rows_epilogue_applies(a)) {
            const long long nbn = occ_cdiv(d->N, 256), cus = cu_count();
            long long best = occ_cdiv(nbm256 * nbn, cus) * 256;
            for (const int r : {224, 208}) {
                const long long c = occ_cdiv(occ_cdiv(d->M, r) * nbn, cus) * r;
                if (c < best) { best = c; f8_rows = r; }
            }
            if (f8_rows_env == 256 || f8_rows_env == 224 || f8_rows_env == 208) f8_rows = f8_rows_env;
        }
        gemm_p8_launch(a, s, d->ab_dtype == OCC_FP8_E4M3 ? 1 : 2, f8_rows);
        if (d->c_colsum && !d->c_colsum_defer) colsum_finalize(a, d->c_colsum, s);
        OCC_LAUNCH_CHECK("occ_gemm");
        return OCC_OK;
    }
    // Default for well-filled bf16 launches.  Measured at M = 12736 (bs 64) against the 128x128 kernels below: fc2 1105 vs 838 TFLOP/s,
    // out-proj 760 vs 633, conv1 1084 vs 881, 4096^3 1248 vs 1058; a launch with fewer than ~0.7 tiles per CU (fc2 at M = 6368: 100
    // tiles) keeps the small-tile kernels.  OCC_GEMM_P8=0 switches it off; variant 30 forces it.
    static const int p8_env = getenv("OCC_GEMM_P8") ? atoi(getenv("OCC_GEMM_P8")) : 1;
    const bool p8_ok = d->ab_dtype == OCC_BF16 && d->K % 64 == 0 && nseg == 1 && ng == 1 && d->N >= 256 && d->M >= 256 && p8_fits;
    // The four-wave 256 x 128 kernel, two workgroups per CU (gemm_q4.hip).  Variant 40 / 41 force it (256- / 224-row tiles); by default
    // (q4_env) it takes the launches of the eight-phase kernel whose epilogue is long against the K loop -- see q4_pays below.
    static const int q4_env = getenv("OCC_GEMM_Q4") ? atoi(getenv("OCC_GEMM_Q4")) : 0;
    // q4_env bits: 1 = N >= 3072 with K <= 1536 (qkv / fc1 forward, fc2's input gradient: the two-tensor and GELU epilogues);
    // 2 = N < 3072, K <= 1536, f32 result (out-proj forward: f32 store + f32 residual read); 4 = every launch the eight-phase kernel would take
    const bool q4_pays = variant == 1 && q4_env && rows_epilogue_applies(a) && d->M >= 2048 &&
                         (((q4_env & 1) && d->N >= 3072 && d->K <= 1536) || ((q4_env & 2) && d->N < 3072 && d->K <= 1536 && d->c_dtype == OCC_F32) || (q4_env & 4));
    if (p8_ok && (variant == 40 || variant == 41 || q4_pays)) {
        OCC_CHECK_ARG(variant == 40 || rows_epilogue_applies(a), "occ_gemm: the 224-row tile has no epilogue for this combination");
        g_last_kernel = OCC_GEMM_KERNEL_Q4;
        static const int q4_rows = getenv("OCC_Q4_ROWS") ? atoi(getenv("OCC_Q4_ROWS")) : 256;
        gemm_q4_launch(a, s, variant == 41 ? 224 : (variant == 40 ? 256 : q4_rows));
        if (d->c_colsum && !d->c_colsum_defer) colsum_finalize(a, d->c_colsum, s);
        OCC_LAUNCH_CHECK("occ_gemm");
        return OCC_OK;
    }
    if (p8_ok && (variant == 30 || variant == 31 || variant == 32 || (variant == 1 && p8_env && nbm256 * occ_cdiv(d->N, 256) * 10 >= 7ll * cu_count()))) {
        // One workgroup per CU: a launch takes ceil(tiles / CUs) tile times, so 800 tiles of 256 rows on 256 CUs (fc1 at bs 64) pay four
        // rounds for 3.125 rounds of work and 200 tiles (N = 1024) leave 56 CUs idle.  Three forms, costed in tile-row units:
        //   whole   ceil(tiles256 / CUs) * 256
        //   tail    when the last round would be at most half full: the row tiles that fill whole rounds go to the eight-phase kernel,
        //           the remaining rows through this function again (they land on the 128x128 kernels, 4 workgroups per CU: about a
        //           third of a tile time plus a launch: costed 200, measured 943 vs 890 TFLOP/s whole vs 978 for 224-row tiles at 12736 x 4096 x 1024).  Needs rows that can be re-based by moving the pointers.
        //   224     57 instead of 50 row tiles at M = 12736: 228 / 684 / 912 workgroups, the same rounds of a tile that costs 7/8.
        static const int tail_env = getenv("OCC_GEMM_TAIL") ? atoi(getenv("OCC_GEMM_TAIL")) : 1;
        static const int r224_env = getenv("OCC_GEMM_224") ? atoi(getenv("OCC_GEMM_224")) : 1;
        const long long nbn256 = occ_cdiv(d->N, 256), tiles = nbm256 * nbn256, cus = cu_count(), rem = tiles % cus;
        long long oa = 0, oc = 0, orr = 0;
        const long long nbm1 = (tiles - rem) / nbn256, m1 = nbm1 * 256;
        const bool can_tail = !d->c_f8 && !d->c_colsum && variant == 1 && tail_env && tiles > cus && rem > 0 && rem * 2 <= cus && nbm1 >= 1 && nbm1 < nbm256 &&
                              rows_rebase(d->a_map, d->M, m1, &oa) && rows_rebase(d->c_map, d->M, m1, &oc) && (!d->R || rows_rebase(d->r_map, d->M, m1, &orr));
        const bool can_224 = rows_epilogue_applies(a) && (variant == 31 || (variant == 1 && r224_env));
        //   208     62 row tiles at M = 12736 (the lower wave row a block shorter): 248 / 744 / 992 workgroups = the same rounds, 13/14 of the loop
        static const int r208_env = getenv("OCC_GEMM_208") ? atoi(getenv("OCC_GEMM_208")) : 1;
        const bool can_208 = rows_epilogue_applies(a) && (variant == 32 || (variant == 1 && r208_env));
        const long long cost_whole = occ_cdiv(tiles, cus) * 256, cost_tail = can_tail ? (tiles / cus) * 256 + 200 : (1ll << 40),
                        cost_224 = can_224 ? occ_cdiv(occ_cdiv(d->M, 224) * nbn256, cus) * 224 : (1ll << 40),
                        cost_208 = can_208 ? occ_cdiv(occ_cdiv(d->M, 208) * nbn256, cus) * 208 : (1ll << 40);
        if (variant == 32 || (variant != 31 && cost_208 < cost_whole && cost_208 < cost_tail && cost_208 < cost_224)) {
            OCC_CHECK_ARG(can_208, "occ_gemm: the 208-row tile has no epilogue for this combination");
            g_last_kernel = OCC_GEMM_KERNEL_P8_224;                 // (reported with the 224-row form: the same kernel, a shorter lower wave row)
            gemm_p8_launch(a, s, 0, 208);
            if (d->c_colsum && !d->c_colsum_defer) colsum_finalize(a, d->c_colsum, s);
            OCC_LAUNCH_CHECK("occ_gemm");
            return OCC_OK;
        }
        if (variant == 31 || (cost_224 < cost_whole && cost_224 < cost_tail)) {
            OCC_CHECK_ARG(can_224, "occ_gemm: the 224-row tile has no epilogue for this combination");
            g_last_kernel = OCC_GEMM_KERNEL_P8_224;
            gemm_p8_launch(a, s, 0, 224);
            if (d->c_colsum && !d->c_colsum_defer) colsum_finalize(a, d->c_colsum, s);
            OCC_LAUNCH_CHECK("occ_gemm");
            return OCC_OK;
        }
        if (cost_tail < cost_whole) {
            occ_gemm_desc tail = *d;
            tail.M = d->M - m1;
            tail.A = (const char*)d->A + oa * 2;
            tail.C = (char*)d->C + oc * (d->c_dtype == OCC_BF16 ? 2 : 4);
            if (d->R) tail.R = (const char*)d->R + orr * (d->r_dtype == OCC_BF16 ? 2 : 4);
            if (d->aux) tail.aux = (char*)d->aux + oc * 2;
            a.M = m1;
            gemm_p8_launch(a, s);
            OCC_LAUNCH_CHECK("occ_gemm");
            const int rc = occ_gemm(&tail, stream);        // at most half a round of 256-row tiles: never splits again
            g_last_kernel = OCC_GEMM_KERNEL_P8_TAIL;
            return rc;
        }
        g_last_kernel = OCC_GEMM_KERNEL_P8;
        gemm_p8_launch(a, s);
        if (d->c_colsum && !d->c_colsum_defer) colsum_finalize(a, d->c_colsum, s);
        OCC_LAUNCH_CHECK("occ_gemm");
        return OCC_OK;
    }
    if (d->c_colsum) { occ_set_error("occ_gemm: c_colsum is produced by the 256-row kernel's epilogue only; this launch (M=%ld N=%ld K=%ld) does not take it", (long)d->M, (long)d->N, (long)d->K); return OCC_EUNSUPPORTED; }
    if (d->c_f8) { occ_set_error("occ_gemm: c_f8 is written by the 256-row kernel's epilogue only; this launch (M=%ld N=%ld K=%ld) does not take it", (long)d->M, (long)d->N, (long)d->K); return OCC_EUNSUPPORTED; }
    g_last_kernel = OCC_GEMM_KERNEL_OTHER;
    if (d->ab_dtype == OCC_BF16 && d->K % 128 == 0 && variant == 22) {
        hipLaunchKernelGGL(gemm_bf16_ks2_kernel, dim3((unsigned)total, (unsigned)ng), dim3(512), 0, s, a);
    } else if (d->ab_dtype == OCC_BF16 && d->K % 64 == 0 && variant == 14) {
        hipLaunchKernelGGL(gemm_bf16_hs_kernel, dim3((unsigned)total, (unsigned)ng), dim3(THREADS), 0, s, a);
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
        static const int n64_env = getenv("OCC_GEMM_N64") ? atoi(getenv("OCC_GEMM_N64")) : 1;
        // Under-filled launches (at most two tiles per CU: fc2, out-proj, conv 5/6, proj at M = 6368): split K inside the workgroup.
        // Measured against the kernels below: fc2 77 -> 65 us, out-proj 25.7 -> 23.7 us, conv5 27.0 -> 24.4 us; well-filled launches lose 10-15 %.
        static const int ks2_env = getenv("OCC_GEMM_KS2") ? atoi(getenv("OCC_GEMM_KS2")) : 1;
        if (split == 1 && variant == 1 && n64_env && d->N <= 64) {          // narrow (grouped) outputs: 128x64 tile, no padded half
            a.nbn = 1;
            hipLaunchKernelGGL(gemm_bf16_dma_n64_kernel, dim3((unsigned)a.nbm, (unsigned)ng), dim3(THREADS), 0, s, a);
        } else
        if (split == 1 && variant == 1 && ks2_env && total * ng <= 2ll * cu_count() && nslab >= 8 && nslab % 2 == 0) {
            hipLaunchKernelGGL(gemm_bf16_ks2_kernel, dim3((unsigned)total, (unsigned)ng), dim3(512), 0, s, a);
        } else
        if (split == 1 && variant == 1 && total * ng <= 2ll * cu_count() && nslab >= 32) {
            hipLaunchKernelGGL(gemm_bf16_hs_kernel, dim3((unsigned)total, (unsigned)ng), dim3(THREADS), 0, s, a);
        } else
        if (split > 1) {
            a.slabs_per_split = (int)occ_cdiv(nslab, split);
            a.ksplit = (int)occ_cdiv(nslab, a.slabs_per_split);
            a.R = nullptr;
            hipLaunchKernelGGL((gemm_bf16_dma_kernel<128, true>), dim3((unsigned)(total * a.ksplit), (unsigned)ng), dim3(THREADS), 0, s, a);
        } else
        {
            hipLaunchKernelGGL(gemm_bf16_dma_kernel<128>, dim3((unsigned)total, (unsigned)ng), dim3(THREADS), 0, s, a);
        }
    }
    else if (d->ab_dtype == OCC_BF16) hipLaunchKernelGGL(gemm_kernel<1>, dim3((unsigned)total, (unsigned)ng), dim3(THREADS), 0, s, a);
    else if (d->ab_dtype == OCC_F32_AS_BF16 && d->N <= 64 && g_variant != 14) {        // narrow outputs: 128x64 tile
        a.nbn = 1;
        hipLaunchKernelGGL((gemm_kernel<2, 64>), dim3((unsigned)a.nbm, (unsigned)ng), dim3(THREADS), 0, s, a);
    }
    else if (d->ab_dtype == OCC_F32_AS_BF16) hipLaunchKernelGGL(gemm_kernel<2>, dim3((unsigned)total, (unsigned)ng), dim3(THREADS), 0, s, a);
    else if (d->ab_dtype == OCC_AF32_WBF16) hipLaunchKernelGGL(gemm_kernel<3>, dim3((unsigned)total, (unsigned)ng), dim3(THREADS), 0, s, a);
    else if (d->ab_dtype == OCC_F32X3) hipLaunchKernelGGL(gemm_kernel<4>, dim3((unsigned)total, (unsigned)ng), dim3(THREADS), 0, s, a);
    else hipLaunchKernelGGL(gemm_kernel<0>, dim3((unsigned)total, (unsigned)ng), dim3(THREADS), 0, s, a);
    OCC_LAUNCH_CHECK("occ_gemm");
    return OCC_OK;
}

