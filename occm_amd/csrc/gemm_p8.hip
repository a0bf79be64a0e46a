// 256 x 256 x 64 bf16 GEMM, eight waves, software-pipelined LDS-DMA ("eight-phase" loop) -- the kernel behind the large
// Linear / Conv1d launches of the wav2vec2 front-end (fairseq layers reached from sslassist.py:48) and their input gradients.
//
//   C[256 m x 256 n] per workgroup, 8 waves as 2 (m) x 4 (n), 128 x 64 outputs per wave (32 accumulators of 16x16), 2 waves per SIMD.
//   LDS: two K-tile buffers of 64 KiB; a buffer = four 16 KiB half-tiles {X0, X1, W0, W1} of 128 rows x 128 B.  Half h of X holds,
//   for both wave rows, the rows of that wave's m-half h (tile rows wr*128 + h*64 + 0..63); half h of W the rows of every wave
//   column's n-half h (tile rows wc*64 + h*32 + 0..31): one phase of one wave reads exactly one half-tile per operand.
//   A K-tile (64 deep) is four phases, each: fragment reads of one operand half (ds_read_b128), ONE half-tile of LDS-DMA for a later
//   K-tile (2 x global_load_lds_dwordx4 per thread), barrier, 16 MFMA 16x16x32 (one 64 x 32 quadrant of the wave's outputs), barrier:
//       P1  read W0, X0   stage X1 of tile t+1      Q(m0, n0)
//       P2  read W1       stage W0 of tile t+2      Q(m0, n1)
//       P3  read X1       stage X0 of tile t+2      Q(m1, n1)
//       P4  --            stage W1 of tile t+2      Q(m1, n0)        s_waitcnt vmcnt(6): tile t+1 has landed, three half-tiles stay in flight
//   The two wave rows run one barrier apart (waves 4-7 take an extra barrier before the loop, waves 0-3 one after it), so on every SIMD
//   one wave multiplies while its partner reads fragments and issues DMA.
//   Hazards by construction: a half-tile is read one phase (or more) after the counted vmcnt + barrier that retires its DMA, and is
//   re-staged one phase (or more) after the phase whose reads -- retired by lgkmcnt(0) ahead of that phase's first barrier -- were its
//   last; both hold for the leading and the lagging wave row.
//   The LDS image is lane-linear per DMA instruction (8 rows x 128 B); the bank swizzle chunk ^ (row & 7) is applied to the per-lane
//   SOURCE chunk and again on the fragment read.
//
// FMT: 0 = bf16 operands (v_mfma_f32_16x16x32_bf16, two 32-deep k-steps per 64-element K-tile).  1 / 2 = OCP fp8: W e4m3, X e4m3 (forward)
// or e5m2 (the gradient operand of an input-gradient GEMM); a K-tile is then 128 elements in the same 128-byte rows, a lane's fragment is
// 32 consecutive bytes (two ds_read_b128) and the product is ONE v_mfma_scale_f32_16x16x128_f8f6f4 per 16x16 block with unit block
// scales (E8M0 127) -- the per-tensor scales of both operands are undone in the epilogue (alpha * dq_a * dq_w).  That instruction takes
// twice the cycles of the bf16 one for four times the depth: the same loop, twice the FLOPs per K-tile, half the operand bytes per FLOP.
#include "gemm_common.h"
#include <type_traits>

namespace occ_gemm_detail {

typedef __attribute__((ext_vector_type(8))) int i32x8;

#define P8_MFMA(ACC, WF, XF) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, WF), __builtin_bit_cast(bf16x8, XF), ACC, 0, 0, 0)

// 16 MFMAs: outputs [n-half NH][m-half MH] of the wave, both 32-deep k-steps; an accumulator is revisited 8 MFMAs later
#define P8_QUAD(NH, MH, WQ)                                                                        \
    __builtin_amdgcn_s_setprio(1);                                                                 \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                               \
        _Pragma("unroll") for (int mf = 0; mf < ((MH) ? MB - 4 : 4); ++mf) {                       \
            if (ASYM && (MH) && mf == MB - 5 && wr == 1) continue;   /* the lower wave row's last block is not part of a 208-row tile */ \
            _Pragma("unroll") for (int nf = 0; nf < 2; ++nf)                                       \
                P8_MFMA(acc[(NH) * 2 + nf][(MH) * 4 + mf], WQ[nf][ks], x[mf][ks]);                 \
        }                                                                                          \
    __builtin_amdgcn_s_setprio(0);                                                                 \
    __builtin_amdgcn_sched_barrier(0);

// fp8: 8 MFMAs of depth 128 on the 32-byte fragments.  Inline asm with the accumulator tied in place: through the builtin hipcc gives
// every result a fresh register quad (the accumulators then do not fit and spill, 420 bytes per lane).  Consecutive MFMAs of one
// accumulator need no wait states; the epilogue's first VALU read of an accumulator sits behind P8_MFMA_DRAIN.
#define P8_MFMA_F8(ACC, WF, XF, XT)                                                                                                       \
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] blgp:" #XT : "+v"(ACC) : "v"(WF), "v"(XF), "v"(unit_scales));
#define P8_QUAD_F8(NH, MH, WQ, XT)                                                                 \
    __builtin_amdgcn_s_setprio(1);                                                                 \
    _Pragma("unroll") for (int mf = 0; mf < ((MH) ? MB - 4 : 4); ++mf) {                           \
        if (ASYM && (MH) && mf == MB - 5 && wr == 1) continue;       /* (208-row tiles: see P8_QUAD) */ \
        _Pragma("unroll") for (int nf = 0; nf < 2; ++nf)                                           \
            P8_MFMA_F8(acc[(NH) * 2 + nf][(MH) * 4 + mf], WQ[nf], xq[mf], XT)                      \
    }                                                                                              \
    __builtin_amdgcn_s_setprio(0);                                                                 \
    __builtin_amdgcn_sched_barrier(0);
#define P8_Q(NH, MH, WQ)                                                                           \
    if constexpr (FMT == 0) { P8_QUAD(NH, MH, WQ) } else if constexpr (FMT == 1) { P8_QUAD_F8(NH, MH, WQ##q, 0) } else { P8_QUAD_F8(NH, MH, WQ##q, 1) }

#define P8_SYNC_READS()                                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                             \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    __builtin_amdgcn_s_barrier();                                                                  \
    __builtin_amdgcn_sched_barrier(0);

// (the empty asm with a memory clobber pins compiler-visible LDS loads -- the fp8 fragments -- below the barrier: a raw s_barrier
// builtin carries no fence, and a fragment load hoisted above it reads a half-tile before the lagging wave row has retired its DMA)
#define P8_END_PHASE()                                                                             \
    __builtin_amdgcn_s_barrier();                                                                  \
    asm volatile("" ::: "memory");                                                                 \
    __builtin_amdgcn_sched_barrier(0);

// MB: 16-row blocks per wave, 8 (256-row tiles) or 7 (224-row tiles: the LDS image keeps its 128-row slots per wave row, the last block
// of the second m-half is neither multiplied nor stored).  M = 12736 (bs 64) is 49.75 tiles of 256 rows: with N = 1024 that is 200
// workgroups for 256 CUs, and 600 / 800 for N = 3072 / 4096 -- 57 tiles of 224 rows make it 228 / 684 / 912, the same number of rounds
// of a tile that costs 7/8.
// ASYM (MB = 7): 208-row tiles -- the upper wave row keeps its 7 blocks (112 rows), the lower one multiplies and stores 6 (96 rows;
// its tile rows start at 112 as before).  The two waves of a SIMD share its matrix core, so a K-tile costs 13/14 of the 224-row form's
// MFMAs: 62 row tiles at M = 12736 = 248 / 744 / 992 workgroups -- one, three and four rounds again, of a shorter loop.
template <int FMT, int MB = 8, bool ASYM = false>
__global__ __launch_bounds__(512, 2) void gemm_p8_kernel(const GemmArgs a) {
    static_assert(MB == 8 || MB == 7, "8 or 7 blocks of 16 rows per wave");
    static_assert(!ASYM || MB == 7, "the 208-row form is the 224-row kernel with a shorter lower wave row");
    constexpr int ES = FMT == 0 ? 2 : 1;           // bytes per operand element; a K-tile is 128 bytes of every row
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * 65536];
    const int total = a.nbm * a.nbn;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q8 = total >> 3, r8 = total & 7;
    const int vid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    int tile_n = vid % a.nbn, tile_m = vid / a.nbn;
    if (a.group_m > 0) {                       // GROUP_M m-tiles share each W panel: the 32 tiles an XCD runs together form a compact block
        const int per_group = a.group_m * a.nbn;
        const int gid = vid / per_group, first_m = gid * a.group_m;
        const int gsz = a.nbm - first_m < a.group_m ? a.nbm - first_m : a.group_m;
        const int loc = vid - gid * per_group;
        tile_m = first_m + loc % gsz;
        tile_n = loc / gsz;
    }
    const long long m0 = (long long)tile_m * (ASYM ? 208 : MB * 32), n0 = (long long)tile_n * 256;
#ifdef P8_DIAG
    const unsigned long long dg_t0 = __builtin_amdgcn_s_memtime(), dg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // ---- LDS-DMA sources: one instruction = 8 rows x 128 B; LDS position p of a row holds source chunk p ^ (row & 7)
    // (fp8: 32-byte blocks are swizzled whole -- chunk ^ (((row >> 1) & 3) << 1) -- so that a lane's 32-byte fragment stays contiguous
    // and in k order; the bf16 form swizzles 16-byte chunks)
    const int srow = lane >> 3;
    const int sch = FMT == 0 ? (lane & 7) ^ srow : (lane & 7) ^ (((srow >> 1) & 3) << 1);
    // 32-bit byte offsets from the (wave-uniform) operand base: the DMA takes its address as SGPR base + VGPR offset, the K advance is scalar
    unsigned soff[4][2];                       // [X0, X1, W0, W1][pass]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            long long m = m0 + q * (MB * 16) + h * 64 + wave * 8 + srow; if (m > a.M - 1) m = a.M - 1;
            soff[h][q] = (unsigned)(row_off(a.xmap, m) * ES + sch * 16);
            long long n = n0 + (q * 2 + (wave >> 2)) * 64 + h * 32 + (wave & 3) * 8 + srow; if (n > a.N - 1) n = a.N - 1;
            soff[2 + h][q] = (unsigned)(n * a.ldw * ES + sch * 16);
        }
    // One LDS-DMA instruction in the SGPR-base + 32-bit-VGPR-offset form (hipcc's builtin keeps a 64-bit address pair per source).  It is
    // invisible to the compiler's s_waitcnt bookkeeping, which is what this loop wants: every wait on it below is hand-counted.
    // M0 (the LDS destination of the wave's 1 KiB piece) is saved and restored inside the statement.
    const unsigned dma_dst = (unsigned)(uintptr_t)(lds_void*)lds + (unsigned)wave * 1024u;
    auto dma16 = [&](const char* sbase, unsigned voff, unsigned ldst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(ldst) : "memory");
    };
#define P8_STAGE(HID, BUF, KT)                                                                                                         \
    {                                                                                                                                  \
        const char* base__ = ((HID) < 2 ? a.X : a.W) + (long long)(KT) * 128;                                                          \
        dma16(base__, soff[HID][0], dma_dst + (BUF) * 65536 + (HID) * 16384);                                                          \
        dma16(base__, soff[HID][1], dma_dst + (BUF) * 65536 + (HID) * 16384 + 8192);                                                   \
    }

    // ---- fragment read addresses (byte addresses in LDS): row fr of a 16-row block; bf16: chunk (ks*4 + fq) ^ (fr & 7) for k-step ks,
    // fp8: the two chunks (2*fq + c) ^ (fr & 7) of the lane's 32 consecutive bytes
    const int fr = lane & 15, fq = lane >> 4;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)lds;
    const unsigned cb = (unsigned)((fq ^ (fr & 7)) << 4);
    constexpr unsigned SECOND = 64u;
    unsigned xa[2][2], wa[2][2];               // [buffer][k-step (bf16) / 16-byte half (fp8)]
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        xa[b][0] = lds0 + b * 65536 + (wr * 64 + fr) * 128 + cb;          xa[b][1] = xa[b][0] ^ SECOND;
        wa[b][0] = lds0 + b * 65536 + 32768 + (wc * 32 + fr) * 128 + cb;  wa[b][1] = wa[b][0] ^ SECOND;
    }

    f32x4 acc[4][8];                           // [16-column block of the wave's 64][16-row block of its 128]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nt = (int)(a.K / (128 / ES));
    // ---- prologue: all of K-tile 0, three half-tiles of K-tile 1
    P8_STAGE(2, 0, 0) P8_STAGE(0, 0, 0) P8_STAGE(3, 0, 0) P8_STAGE(1, 0, 0)
    if (nt > 1) {
        P8_STAGE(2, 1, 1) P8_STAGE(0, 1, 1) P8_STAGE(3, 1, 1)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    if (wr == 1) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }       // the lagging wave row

#ifdef P8_DIAG
    const unsigned long long dg_t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    u32x4 w0[2][2], w1[2][2], x[4][2];         // bf16: [16-row block][k-step]
    i32x8 w0q[2], w1q[2], xq[4];               // fp8: [16-row block], 32 consecutive bytes of the row per lane
    // E8M0 block scales of 1.0 for both operands, set once through an asm statement: the compiler then keeps ONE register for it and
    // cannot re-materialise a v_mov right in front of an asm MFMA (a VALU write -> MFMA operand hazard its recogniser does not see).
    int unit_scales;
    asm volatile("v_mov_b32 %0, 0x7f7f7f7f" : "=v"(unit_scales));
    // fp8 fragments are plain 32-byte LDS loads (the DMA is inline asm, so the compiler sees no pending LDS writes to wait for): byte
    // offset of this lane's block in row fr of a 16-row group
    const unsigned f8x = (unsigned)((wr * 64 + fr) * 128 + ((fq ^ ((fr >> 1) & 3)) << 5));
    const unsigned f8w = (unsigned)(32768 + (wc * 32 + fr) * 128 + ((fq ^ ((fr >> 1) & 3)) << 5));
#define P8_LD8(OFF) (*reinterpret_cast<const i32x8*>(lds + (OFF)))
    auto tile = [&](auto bufc, const int t) {
        constexpr int B = decltype(bufc)::value;
        // -------- P1
        if constexpr (FMT == 0) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) { w0[0][ks] = ds_read128<0>(wa[B][ks]); w0[1][ks] = ds_read128<2048>(wa[B][ks]); }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                x[0][ks] = ds_read128<0>(xa[B][ks]); x[1][ks] = ds_read128<2048>(xa[B][ks]);
                x[2][ks] = ds_read128<4096>(xa[B][ks]); x[3][ks] = ds_read128<6144>(xa[B][ks]);
            }
        } else {
            w0q[0] = P8_LD8(B * 65536 + f8w); w0q[1] = P8_LD8(B * 65536 + f8w + 2048);
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) xq[mf] = P8_LD8(B * 65536 + f8x + mf * 2048);
        }
        if (t + 1 < nt) P8_STAGE(1, B ^ 1, t + 1)
        P8_SYNC_READS()
        P8_Q(0, 0, w0)
        P8_END_PHASE()
        // -------- P2
        if constexpr (FMT == 0) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) { w1[0][ks] = ds_read128<16384>(wa[B][ks]); w1[1][ks] = ds_read128<16384 + 2048>(wa[B][ks]); }
        } else {
            w1q[0] = P8_LD8(B * 65536 + f8w + 16384); w1q[1] = P8_LD8(B * 65536 + f8w + 16384 + 2048);
        }
        if (t + 2 < nt) P8_STAGE(2, B, t + 2)
        P8_SYNC_READS()
        P8_Q(1, 0, w1)
        P8_END_PHASE()
        // -------- P3
        if constexpr (FMT == 0) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                x[0][ks] = ds_read128<16384>(xa[B][ks]); x[1][ks] = ds_read128<16384 + 2048>(xa[B][ks]);
                x[2][ks] = ds_read128<16384 + 4096>(xa[B][ks]);
                if constexpr (MB == 8) x[3][ks] = ds_read128<16384 + 6144>(xa[B][ks]);
            }
        } else {
#pragma unroll
            for (int mf = 0; mf < MB - 4; ++mf) xq[mf] = P8_LD8(B * 65536 + f8x + 16384 + mf * 2048);
        }
        if (t + 2 < nt) P8_STAGE(0, B, t + 2)
        P8_SYNC_READS()
        P8_Q(1, 1, w1)
        P8_END_PHASE()
        // -------- P4
        if (t + 2 < nt) {
            P8_STAGE(3, B, t + 2)
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        P8_Q(0, 1, w0)
        P8_END_PHASE()
    };
    int t = 0;
    for (; t + 1 < nt; t += 2) {
        tile(std::integral_constant<int, 0>{}, t);
        tile(std::integral_constant<int, 1>{}, t + 1);
    }
    if (t < nt) tile(std::integral_constant<int, 0>{}, t);
    if (wr == 0) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }        // pairs with the lagging row's last barrier

    if constexpr (FMT != 0) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last asm MFMAs' results -> the epilogue's VALU reads
#ifdef P8_DIAG
    const unsigned long long dg_t2 = __builtin_amdgcn_s_memtime();
#endif
    // all LDS is free here: no DMA is outstanding (the last phase waited vmcnt(0)) and every wave has finished its fragment reads
    gemm_epilogue_rows<8>(a, acc, m0 + wr * (MB * 16), n0 + wc * 64, lane, 0, lds + wave * 16384, ASYM && wr == 1 ? 96 : MB * 16, tile_m * 2 + wr);
#ifdef P8_DIAG
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) {
        unsigned long long* o = a.diag + (long long)blockIdx.x * 6;
        o[0] = dg_t0; o[1] = dg_t1; o[2] = dg_t2; o[3] = __builtin_amdgcn_s_memtime(); o[4] = dg_r0; o[5] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// Launch helper used by occ_gemm.  Preconditions (checked by the caller): bf16 operands and K % 64 == 0, or fp8 operands (fmt 1: A e4m3,
// fmt 2: A e5m2; W e4m3) and K % 128 == 0; one K segment, one group.
void gemm_p8_launch(GemmArgs& a, hipStream_t s, int fmt, int tile_rows) {
    a.nbm = (int)occ_cdiv(a.M, tile_rows);
    a.nbn = (int)occ_cdiv(a.N, 256);
    a.group_m = a.nbm >= 8 ? 8 : 0;
    const dim3 grid((unsigned)((long long)a.nbm * a.nbn));
    if (tile_rows == 208) {
        if (fmt == 0) hipLaunchKernelGGL((gemm_p8_kernel<0, 7, true>), grid, dim3(512), 0, s, a);
        else if (fmt == 1) hipLaunchKernelGGL((gemm_p8_kernel<1, 7, true>), grid, dim3(512), 0, s, a);
        else hipLaunchKernelGGL((gemm_p8_kernel<2, 7, true>), grid, dim3(512), 0, s, a);
        return;
    }
    if (tile_rows == 224) {
        if (fmt == 0) hipLaunchKernelGGL((gemm_p8_kernel<0, 7>), grid, dim3(512), 0, s, a);
        else if (fmt == 1) hipLaunchKernelGGL((gemm_p8_kernel<1, 7>), grid, dim3(512), 0, s, a);
        else hipLaunchKernelGGL((gemm_p8_kernel<2, 7>), grid, dim3(512), 0, s, a);
        return;
    }
    if (fmt == 0) hipLaunchKernelGGL(gemm_p8_kernel<0>, grid, dim3(512), 0, s, a);
    else if (fmt == 1) hipLaunchKernelGGL(gemm_p8_kernel<1>, grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL(gemm_p8_kernel<2>, grid, dim3(512), 0, s, a);
}

}  // namespace occ_gemm_detail
