// Experimental kernels of the bf16 GEMM family, selected with occ_gemm_variant / OCC_GEMM_VARIANT (5, 6, 8, 9: multi-stage LDS
// pipelines; 11, 12: persistent forms; 13: persistent 128x128 with the stores draining under the next tile; 15, 16: deep half-slab
// pipelines on 256-wide tiles).  All are parity-tested against the default (tests/test_gpu_frontend.py) and all measured slower
// than it on the front-end shapes; they stay as the starting points for the 8-phase schedule.
#include "gemm_common.h"

namespace occ_gemm_detail {

// ------------------------------------------------------------------------------------------------
// Multi-stage bf16 kernel: ONE workgroup per CU, block tile (WGM*NJ*16) x (WGN*64), NST LDS stages filled by LDS-DMA with
// NST-1 slabs in flight.  The loads stay in flight ACROSS the per-slab barrier: the wait is a counted `s_waitcnt vmcnt(n)`
// (n = DMA instructions of the later slabs, never 0 in the steady state) and the barrier is a raw `s_barrier` -- a
// `__syncthreads()` would drain the DMA queue (it fences with vmcnt(0)).  Why: at one or two resident workgroups per CU the
// L2->LDS path is latency-bound, not bandwidth-bound; sustaining the ~48 B/clk a 256x128 tile needs at ~2000 clk of latency
// takes ~96 KiB in flight per CU (Little), which is what 2 x 48 KiB stages in flight out of 3 provide.
//   256x128 (WGM=4, WGN=2, NJ=4): one L2 byte per 85 FLOP;  256x256 (WGM=2, WGN=4, NJ=8): one per 128 FLOP;
//   128x128 (WGM=2, WGN=2, NJ=4): one per 64 FLOP (the CU's ~64 B/clk L2 port is then the bound).
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
        if constexpr (WGM * WGN == 16) {
            // 16 waves = 4 per SIMD = 128 registers each: one fragment set, refilled for the second half-slab (the other three waves
            // of the SIMD cover the read latency)
            u32x4 wq[4], xq[NJ];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                read_frags<4>(wq, sb + (kb ? (woff0 ^ 64) : woff0)); read_frags<NJ>(xq, sb + (kb ? (xoff0 ^ 64) : xoff0));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wq[i]), __builtin_bit_cast(bf16x8, xq[j]), acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            buf = buf + 1 == NST ? 0 : buf + 1;
            nbuf = nbuf + 1 == NST ? 0 : nbuf + 1;
            continue;
        }
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


// ------------------------------------------------------------------------------------------------
// Big-register-tile pipeline (experiment, variant 17): 256x256 block tile, FOUR waves (2x2) each owning 128x128 of it (64 accumulators
// of 16x16 = 256 registers), one workgroup per CU -- the shape the vendor library picks for these problems (hipBLASLt: MT256x256x64,
// 256 threads, 133 KB LDS).  Idea: with LDS-DMA a CU pulls operands at (bytes in flight) / (load latency) and the LDS bounds the bytes in
// flight; a 256x256 tile does 128 FLOP per operand byte against the 128x128 tile's 64.  One wave per SIMD hides nothing by occupancy, so the
// loop is software-pipelined by hand: two LDS stages of one K = 64 slab each (128-byte rows, the default kernel's swizzle), stage s+2 is
// DMA'd while the second half of stage s is multiplied, the fragments of the next half-slab are read into a second register set under the
// MFMAs of the current one, one barrier per K = 64.
// Measured (4096^3, one tile per CU): 860-920 TFLOP/s against the default kernel's 1140-1250 and hipBLASLt's 1400.  Ablation by K-scaling:
// in-loop 1.9 us per slab (1.1 PFLOP/s) with ~29 us of per-tile fixed cost; fragment reads alone 1.2 us per slab, DMA + reads 1.65 us --
// each of the three streams (DMA, ds_read, MFMA) alone takes most of a slab's time and they overlap poorly with a single wave per SIMD.
// Kept as a correct, tested variant and as the starting point for a hand-scheduled (assembly-level) version; not selected by default.
__global__ __launch_bounds__(256, 1) void gemm_bf16_w4_kernel(const GemmArgs a) {
    constexpr int ES = 2, TMB = 256, TNB = 256;
    constexpr int STAGE = (TMB + TNB) * CHUNKS;       // uint4 per stage: 512 rows x 128 B = 64 KiB
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
    const int wm = wave & 1, wn = wave >> 1;
    const long long grp = blockIdx.y;
    const char* Xg = a.X + grp * a.a_gstride * ES;
    const char* Wg = a.W + grp * a.w_gstride * ES;
    const long long cshift = grp * a.c_gstride;
    // one DMA = 8 rows x 128 B; wave w stages row blocks w, w+4, ..., w+28 of X and of W (8 + 8 instructions per slab)
    const int rl = lane >> 3, pos = lane & 7;
    const int sck = (pos ^ rl) * 8;                   // source k offset (elements): LDS position p of row r holds chunk p ^ (r & 7)
    const char* xsrc[8]; const char* wsrc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        long long m = m0 + 8 * (wave + 4 * i) + rl; if (m > a.M - 1) m = a.M - 1;
        long long n = n0 + 8 * (wave + 4 * i) + rl; if (n > a.N - 1) n = a.N - 1;
        xsrc[i] = Xg + row_off(a.xmap, m) * ES;
        wsrc[i] = Wg + n * a.ldw * ES;
    }
    const int nslab = (int)(a.K / 64);
    // K segments (implicit-GEMM taps) are whole slabs here (seg_len % 64 == 0, checked by the launcher): one scalar mapping per slab
    const int seg_slabs = a.nseg > 1 ? (int)(a.seg_len / 64) : nslab + 1;
    auto xk = [&](int sl) -> long long {
        const int sg = sl / seg_slabs;
        return ((long long)sg * a.seg_stride + (long long)(sl - sg * seg_slabs) * 64 + sck) * ES;
    };
    // acc[ch][rg][i][j]: column half ch (64 columns), row group rg (32 rows), then the [4 column blocks][2 row blocks] chunk the epilogue takes
    f32x4 acc[2][4][4][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc[c][r][i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[c][r][i][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const int fr = lane & 15, fq = lane >> 4;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)plds;
    // fragment (row r, half kb): byte r*128 + (((kb*4 + fq) ^ (r & 7)) << 4); blocks of 16 rows are 2 KiB apart
    const unsigned xrow = (unsigned)((wm * 128 + fr) * 128), wrow = (unsigned)(TMB * 128 + (wn * 128 + fr) * 128);
    const unsigned sw0 = (unsigned)((fq ^ (fr & 7)) << 4), sw1 = (unsigned)(((4 + fq) ^ (fr & 7)) << 4);
    const unsigned x0 = xrow + sw0, x1 = xrow + sw1, w0 = wrow + sw0, w1 = wrow + sw1;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
        if (sl < nslab) {
            const long long kx = xk(sl), kw = ((long long)sl * 64 + sck) * ES;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_global_load_lds((gbl_void*)(xsrc[i] + kx), (lds_void*)&plds[sl * STAGE + (wave + 4 * i) * 64], 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gbl_void*)(wsrc[i] + kw), (lds_void*)&plds[sl * STAGE + TMB * CHUNKS + (wave + 4 * i) * 64], 16, 0, 0);
            }
        }
    u32x4 xa[8], wa[8], xb[8], wb[8];
    if (nslab > 1) wait_vm_then_barrier<16>(); else wait_vm_then_barrier<0>();     // slab 0 landed for everyone
    read_frags<8>(xa, lds0 + x0); read_frags<8>(wa, lds0 + w0);
    const bool domfma = !(a.dbg & 2), nodma = a.dbg & 1, noread = a.dbg & 4;       // ablation switches (timing experiments only)
    // 64 MFMAs in 16 groups of 4; before group G one fragment read (RD) and one DMA (FILL) are issued
#define OCC_W4_GROUP(G, XC, WC, XN, WN, RD, RB, RXO, RWO, FILL)                                                                        \
        if (RD) { if ((G) < 8) XN[(G) & 7] = ds_read128<((G) & 7) * 2048>((RB) + (RXO)); else WN[(G) & 7] = ds_read128<((G) & 7) * 2048>((RB) + (RWO)); } \
        if (FILL) {                                                                                                                    \
            if ((G) < 8) __builtin_amdgcn_global_load_lds((gbl_void*)(xsrc[(G) & 7] + kxf), (lds_void*)&plds[buf * STAGE + (wave + 4 * ((G) & 7)) * 64], 16, 0, 0); \
            else __builtin_amdgcn_global_load_lds((gbl_void*)(wsrc[(G) & 7] + kwf), (lds_void*)&plds[buf * STAGE + TMB * CHUNKS + (wave + 4 * ((G) & 7)) * 64], 16, 0, 0); \
        }                                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
        if (domfma) _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                                    \
            const int id = (G) * 4 + t, ii = id & 7, jj = id >> 3;                                                                     \
            acc[ii >> 2][jj >> 1][ii & 3][jj & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, WC[ii]), __builtin_bit_cast(bf16x8, XC[jj]), acc[ii >> 2][jj >> 1][ii & 3][jj & 1], 0, 0, 0); \
        }                                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);
#define OCC_W4_HALF(XC, WC, XN, WN, RD, RB, RXO, RWO, FILL)                                                                            \
        OCC_W4_GROUP(0, XC, WC, XN, WN, RD, RB, RXO, RWO, FILL) OCC_W4_GROUP(1, XC, WC, XN, WN, RD, RB, RXO, RWO, FILL)                \
        OCC_W4_GROUP(2, XC, WC, XN, WN, RD, RB, RXO, RWO, FILL) OCC_W4_GROUP(3, XC, WC, XN, WN, RD, RB, RXO, RWO, FILL)                \
        OCC_W4_GROUP(4, XC, WC, XN, WN, RD, RB, RXO, RWO, FILL) OCC_W4_GROUP(5, XC, WC, XN, WN, RD, RB, RXO, RWO, FILL)                \
        OCC_W4_GROUP(6, XC, WC, XN, WN, RD, RB, RXO, RWO, FILL) OCC_W4_GROUP(7, XC, WC, XN, WN, RD, RB, RXO, RWO, FILL)                \
        OCC_W4_GROUP(8, XC, WC, XN, WN, RD, RB, RXO, RWO, FILL) OCC_W4_GROUP(9, XC, WC, XN, WN, RD, RB, RXO, RWO, FILL)                \
        OCC_W4_GROUP(10, XC, WC, XN, WN, RD, RB, RXO, RWO, FILL) OCC_W4_GROUP(11, XC, WC, XN, WN, RD, RB, RXO, RWO, FILL)              \
        OCC_W4_GROUP(12, XC, WC, XN, WN, RD, RB, RXO, RWO, FILL) OCC_W4_GROUP(13, XC, WC, XN, WN, RD, RB, RXO, RWO, FILL)              \
        OCC_W4_GROUP(14, XC, WC, XN, WN, RD, RB, RXO, RWO, FILL) OCC_W4_GROUP(15, XC, WC, XN, WN, RD, RB, RXO, RWO, FILL)
    int buf = 0;
    for (int sl = 0; sl < nslab; ++sl) {
        const unsigned cb = lds0 + (unsigned)buf * (STAGE * 16), ob = lds0 + (unsigned)(buf ^ 1) * (STAGE * 16);
        long long kxf = 0, kwf = 0;
        // first half: (xa, wa) hold k 0..31 of this slab; read k 32..63 of the same stage into (xb, wb)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        {
            const bool rd = !noread;
            OCC_W4_HALF(xa, wa, xb, wb, rd, cb, x1, w1, false)
        }
        // second half: every wave has read this stage (their lgkmcnt(0) above precedes the barrier) and slab sl+1 has landed
        wait_vm_then_barrier<0>();
        {
            const bool rd = sl + 1 < nslab && !noread, fill = sl + 2 < nslab && !nodma;
            if (fill) { kxf = xk(sl + 2); kwf = ((long long)(sl + 2) * 64 + sck) * ES; }
            OCC_W4_HALF(xb, wb, xa, wa, rd, ob, x0, w0, fill)
        }
        buf ^= 1;
    }
#undef OCC_W4_HALF
#undef OCC_W4_GROUP
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // small chunks keep the epilogue's live registers low (256 accumulators leave little room; one 4x8-block call spilled)
#define OCC_W4_EPI(CH, RG) gemm_epilogue<2>(a, acc[CH][RG], m0 + wm * 128 + (RG) * 32, n0 + wn * 128 + (CH) * 64, fr, fq, cshift);
    OCC_W4_EPI(0, 0) OCC_W4_EPI(0, 1) OCC_W4_EPI(0, 2) OCC_W4_EPI(0, 3) OCC_W4_EPI(1, 0) OCC_W4_EPI(1, 1) OCC_W4_EPI(1, 2) OCC_W4_EPI(1, 3)
#undef OCC_W4_EPI
}

int launch_w4(GemmArgs& a, const occ_gemm_desc* d, long long ng, hipStream_t s) {
    if (a.nseg > 1 && a.seg_len % 64 != 0) return -100;          // K segments must be whole slabs: fall back to the default kernel
    a.nbm = (int)occ_cdiv(d->M, 256); a.nbn = (int)occ_cdiv(d->N, 256);
    a.group_m = a.nbn >= 8 ? 4 : 0;
    const size_t shm = (size_t)2 * 512 * CHUNKS * sizeof(uint4);
    static bool raised = false;
    if (!raised) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_w4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) { occ_set_error("occ_gemm: cannot raise the LDS limit: %s", hipGetErrorString(e)); return OCC_ELAUNCH; }
        raised = true;
    }
    hipLaunchKernelGGL(gemm_bf16_w4_kernel, dim3((unsigned)((long long)a.nbm * a.nbn), (unsigned)ng), dim3(256), shm, s, a);
    return OCC_OK;
}


int gemm_family_launch(int variant, GemmArgs& a, const occ_gemm_desc* d, long long ng, hipStream_t s) {
    if (variant == 17) return launch_w4(a, d, ng, s);
    if (variant == 20) return launch_ms<4, 4, 4, 2>(a, d, ng, s);          // 16 waves of 64x64 on a 256x256 tile, double-buffered
    if (variant == 15 || variant == 16) return variant == 15 ? launch_ms32<2, 4, 8, 4>(a, d, ng, s) : launch_ms32<4, 2, 4, 5>(a, d, ng, s);
    if (variant == 13) {
        a.nbm = (int)occ_cdiv(d->M, TM); a.nbn = (int)occ_cdiv(d->N, TN);
        a.ngroups = (int)ng;
        const long long tot = (long long)a.nbm * a.nbn * ng;
        const long long grid = tot < 4ll * cu_count() ? tot : 4ll * cu_count();
        hipLaunchKernelGGL(gemm_bf16_p1_kernel, dim3((unsigned)grid), dim3(256), 0, s, a);
        return OCC_OK;
    }
    if (variant == 11 || variant == 12) return variant == 11 ? launch_persist<4, 2, 4, 3>(a, d, ng, s) : launch_persist<2, 2, 4, 4>(a, d, ng, s);
    if (variant == 5 || variant == 6 || variant == 8 || variant == 9)
        return variant == 5 ? launch_ms<4, 2, 4, 3>(a, d, ng, s) : variant == 6 ? launch_ms<2, 4, 8, 2>(a, d, ng, s)
             : variant == 8 ? launch_ms<2, 2, 4, 4>(a, d, ng, s) : launch_ms<2, 2, 4, 3>(a, d, ng, s);
    return -100;                       // not a family variant
}

}  // namespace occ_gemm_detail
