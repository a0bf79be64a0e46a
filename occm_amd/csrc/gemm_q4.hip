// 256 x 128 x 64 bf16 GEMM, FOUR waves, TWO workgroups per CU -- the form of the large Linear launches of the wav2vec2 front-end
// (fairseq layers reached from sslassist.py:48) whose epilogue is long against its K loop (K = 1024 with a GELU / side-tensor / f32
// residual epilogue): gemm_p8.hip runs one 128-KiB workgroup per CU, so while its eight waves store their tile (HBM-write-bound,
// ~10 B/clk per CU) the CU's matrix pipe idles, and while they multiply, its store path does.  Here a CU holds two independent
// workgroups of half the tile; one multiplies while the other stores.
//
//   C[256 m x 128 n] per workgroup, 4 waves as 2 (m) x 2 (n), 128 x 64 outputs per wave (32 accumulators of 16x16 -- the same wave tile and
//   the same row epilogue as gemm_p8.hip), one wave per SIMD; 256 threads, <= 256 VGPRs, 80 KiB of LDS: two workgroups per CU.
//   LDS: a ring of FIVE 16 KiB pieces (128 rows x 128 B).  A K-tile (64 deep) is three pieces, consumed in the order
//       W   rows wc*64 + 0..63 of both wave columns            (piece 3t)
//       X0  rows wr*128 + 0..63 of both wave rows              (piece 3t + 1)
//       X1  rows wr*128 + 64..127 of both wave rows            (piece 3t + 2)
//   piece p lives in slot p % 5 (3 * 5 = 15 pieces = 5 K-tiles bring the ring round: the slot numbers are compile-time constants of
//   t % 5).  A K-tile is two phases, each ONE workgroup barrier:
//       PA(t)  s_waitcnt vmcnt (W_t, X0_t landed) . barrier . DMA X0_{t+1} -> the slot PB(t-1) read . read W_t, X0_t . 32 MFMA (m-half 0)
//       PB(t)  s_waitcnt vmcnt (X1_t landed)      . barrier . DMA X1_{t+1}, W_{t+2} -> the slots PA(t) read . read X1_t . 32 MFMA (m-half 1)
//   Hazards by construction: a piece is read after the issuing waves' counted vmcnt AND the barrier behind it; a slot is re-staged
//   after the barrier that follows the phase whose reads (retired by lgkmcnt(0) ahead of that phase's MFMAs) were its last.  Every piece is
//   issued two phases (X0, X1) or three (W) ahead of its use.
//   The LDS image is lane-linear per DMA instruction (8 rows x 128 B); the bank swizzle chunk ^ (row & 7) is applied to the per-lane SOURCE
//   chunk and again on the fragment read (as in gemm_p8.hip).
//   Which two workgroups share a CU is the dispatcher's business; for SPEED only, workgroups that are presumably second on their CU
//   (q4_mode 1: blockIdx >= CUs; 2: odd blockIdx) sleep q4_delay x 64 cycles before their prologue so that the pair starts out of phase.
#include "gemm_common.h"
#include <type_traits>

namespace occ_gemm_detail {

#define Q4_MFMA(ACC, WF, XF) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, WF), __builtin_bit_cast(bf16x8, XF), ACC, 0, 0, 0)

// NF fragments of 16 rows each, 2 KiB apart, k-step ks of the row's 128 bytes
template <int OFF, int NF> __device__ __forceinline__ void q4_read(u32x4 (&d)[4][2], const int ks, const unsigned addr) {
    d[0][ks] = ds_read128<OFF>(addr); d[1][ks] = ds_read128<OFF + 2048>(addr); d[2][ks] = ds_read128<OFF + 4096>(addr);
    if constexpr (NF == 4) d[3][ks] = ds_read128<OFF + 6144>(addr);
}

template <int MB = 8>
__global__ __launch_bounds__(256, 2) void gemm_q4_kernel(const GemmArgs a) {
    static_assert(MB == 8 || MB == 7, "8 or 7 blocks of 16 rows per wave");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[5 * 16384];
    const int total = a.nbm * a.nbn;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q8 = total >> 3, r8 = total & 7;
    const int vid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    int tile_n = vid % a.nbn, tile_m = vid / a.nbn;
    if (a.group_m > 0) {
        const int per_group = a.group_m * a.nbn;
        const int gid = vid / per_group, first_m = gid * a.group_m;
        const int gsz = a.nbm - first_m < a.group_m ? a.nbm - first_m : a.group_m;
        const int loc = vid - gid * per_group;
        tile_m = first_m + loc % gsz;
        tile_n = loc / gsz;
    }
    const long long m0 = (long long)tile_m * (MB * 32), n0 = (long long)tile_n * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    if ((a.q4_mode == 1 && bid >= a.q4_first) || (a.q4_mode == 2 && (bid & 1))) {
        for (int i = 0; i < a.q4_delay; ++i) __builtin_amdgcn_s_sleep(1);            // 64 cycles each; wave-uniform
    }

    // ---- LDS-DMA sources: instruction (q, wave) of a piece fills its rows (q*4 + wave)*8 .. +7; LDS position p of a row holds source
    // chunk p ^ (row & 7)
    const int srow = lane >> 3;
    const int sch = (lane & 7) ^ srow;
    unsigned sx[2][4], sw[4];                  // 32-bit byte offsets from the (wave-uniform) operand bases
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int R = (q * 4 + wave) * 8 + srow;                     // piece row 0..127
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            long long m = m0 + (R >> 6) * (MB * 16) + h * 64 + (R & 63); if (m > a.M - 1) m = a.M - 1;
            sx[h][q] = (unsigned)(row_off(a.xmap, m) * 2 + sch * 16);
        }
        long long n = n0 + R; if (n > a.N - 1) n = a.N - 1;
        sw[q] = (unsigned)(n * a.ldw * 2 + sch * 16);
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)lds;
    const unsigned dma_dst = lds0 + (unsigned)wave * 1024u;
    auto dma16 = [&](const char* sbase, unsigned voff, unsigned ldst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(ldst) : "memory");
    };
    // the W piece / the X piece of m-half H of K-tile KT into ring slot SLOT
#define Q4_STAGE_W(SLOT, KT)                                                                                                           \
    {                                                                                                                                  \
        const char* base__ = a.W + (long long)(KT) * 128;                                                                              \
        _Pragma("unroll") for (int q__ = 0; q__ < 4; ++q__) dma16(base__, sw[q__], dma_dst + (SLOT) * 16384 + q__ * 4096);             \
    }
#define Q4_STAGE_X(H, SLOT, KT)                                                                                                        \
    {                                                                                                                                  \
        const char* base__ = a.X + (long long)(KT) * 128;                                                                              \
        _Pragma("unroll") for (int q__ = 0; q__ < 4; ++q__) dma16(base__, sx[H][q__], dma_dst + (SLOT) * 16384 + q__ * 4096);          \
    }

    // ---- fragment read addresses: row fr of a 16-row block, chunk (ks*4 + fq) ^ (fr & 7); slots 0-1 from the low base, 2-4 from base + 32 KiB
    // (the ds_read offset field is 16 bits)
    const int fr = lane & 15, fq = lane >> 4;
    const unsigned cb = (unsigned)((fq ^ (fr & 7)) << 4);
    unsigned xa[2][2], wa[2][2];               // [low / high base][k-step]
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        xa[b][0] = lds0 + b * 32768 + (wr * 64 + fr) * 128 + cb;  xa[b][1] = xa[b][0] ^ 64u;
        wa[b][0] = lds0 + b * 32768 + (wc * 64 + fr) * 128 + cb;  wa[b][1] = wa[b][0] ^ 64u;
    }
#define Q4_RD(DST, NF, ARR, SLOT, KS) q4_read<((SLOT) < 2 ? (SLOT) : (SLOT) - 2) * 16384, NF>(DST, KS, ARR[(SLOT) < 2 ? 0 : 1][KS])

    f32x4 acc[4][8];                           // [16-column block of the wave's 64][16-row block of its 128]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nt = (int)(a.K / 64);
    // ---- prologue: pieces 0..3 = W_0, X0_0, X1_0, W_1
    Q4_STAGE_W(0, 0) Q4_STAGE_X(0, 1, 0) Q4_STAGE_X(1, 2, 0)
    if (nt > 1) Q4_STAGE_W(3, 1)

    u32x4 w[4][2], x[4][2];                    // [16-row block][k-step]
    auto tile = [&](auto jc, const int t) {
        constexpr int J = decltype(jc)::value;                      // t % 5
        constexpr int SW = (3 * J) % 5, SX0 = (3 * J + 1) % 5, SX1 = (3 * J + 2) % 5;
        // -------- PA
        if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { Q4_RD(w, 4, wa, SW, ks); Q4_RD(x, 4, xa, SX0, ks); }
        if (t + 1 < nt) Q4_STAGE_X(0, (3 * J + 4) % 5, t + 1)
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) Q4_MFMA(acc[nf][mf], w[nf][0], x[mf][0]);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) Q4_MFMA(acc[nf][mf], w[nf][1], x[mf][1]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        // -------- PB
        if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) Q4_RD(x, MB - 4, xa, SX1, ks);
        if (t + 1 < nt) Q4_STAGE_X(1, (3 * J + 5) % 5, t + 1)
        if (t + 2 < nt) Q4_STAGE_W((3 * J + 6) % 5, t + 2)
        asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(MB - 4) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mf = 0; mf < MB - 4; ++mf)
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) Q4_MFMA(acc[nf][4 + mf], w[nf][0], x[mf][0]);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mf = 0; mf < MB - 4; ++mf)
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) Q4_MFMA(acc[nf][4 + mf], w[nf][1], x[mf][1]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    int t = 0;
    for (; t + 5 <= nt; t += 5) {
        tile(std::integral_constant<int, 0>{}, t);
        tile(std::integral_constant<int, 1>{}, t + 1);
        tile(std::integral_constant<int, 2>{}, t + 2);
        tile(std::integral_constant<int, 3>{}, t + 3);
        tile(std::integral_constant<int, 4>{}, t + 4);
    }
    if (t < nt) { tile(std::integral_constant<int, 0>{}, t); ++t; }
    if (t < nt) { tile(std::integral_constant<int, 1>{}, t); ++t; }
    if (t < nt) { tile(std::integral_constant<int, 2>{}, t); ++t; }
    if (t < nt) { tile(std::integral_constant<int, 3>{}, t); ++t; }
    // every wave has retired its fragment reads (lgkmcnt(0) ahead of its last MFMAs) once it arrives here: behind this barrier the ring is
    // free, no DMA is outstanding (the last phase waited vmcnt(0)), and each wave takes one 16 KiB slot as its epilogue's staging rows
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    gemm_epilogue_rows<8>(a, acc, m0 + wr * (MB * 16), n0 + wc * 64, lane, 0, lds + wave * 16384, MB * 16, tile_m * 2 + wr);
}

// Launch helper used by occ_gemm.  Preconditions (checked by the caller): bf16 operands, K % 64 == 0, one K segment, one group, 32-bit
// operand offsets; tile_rows 224 only with rows_epilogue_applies(a).
void gemm_q4_launch(GemmArgs& a, hipStream_t s, int tile_rows) {
    a.nbm = (int)occ_cdiv(a.M, tile_rows);
    a.nbn = (int)occ_cdiv(a.N, 128);
    static const int gm_env = getenv("OCC_Q4_GROUP_M") ? atoi(getenv("OCC_Q4_GROUP_M")) : 8;
    a.group_m = a.nbm >= gm_env && gm_env > 0 ? gm_env : 0;
    static const int mode_env = getenv("OCC_Q4_STAGGER") ? atoi(getenv("OCC_Q4_STAGGER")) : 0;
    static const int delay_env = getenv("OCC_Q4_DELAY") ? atoi(getenv("OCC_Q4_DELAY")) : -1;
    a.q4_mode = mode_env;
    a.q4_first = cu_count();
    // half of a lone tile's K loop, in 64-cycle sleeps: a K-tile of one workgroup alone takes ~2.3 k cycles
    a.q4_delay = delay_env >= 0 ? delay_env : (int)((a.K / 64) * 2300 / 2 / 64);
    const dim3 grid((unsigned)((long long)a.nbm * a.nbn));
    if (tile_rows == 224) hipLaunchKernelGGL((gemm_q4_kernel<7>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((gemm_q4_kernel<8>), grid, dim3(256), 0, s, a);
}

}  // namespace occ_gemm_detail
