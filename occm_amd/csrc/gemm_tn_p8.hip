// Weight gradients of the large Linear / Conv1d layers of the wav2vec2 front-end when it is fine-tuned (the reference's optimizer holds
// every XLS-R parameter, oc_training.py:324):   C[n1, n2] += alpha * sum_m dY[m, n1] * X[m, n2],   dY / X bf16 row-major, C f32.
//
// Same eight-phase loop as gemm_p8.hip (256 x 256 output tile, 8 waves as 2 x 4, 128 x 64 outputs per wave, K-tiles of 64 reduction
// rows in two 64 KiB LDS buffers of four half-tiles, one half-tile of LDS-DMA per phase, counted vmcnt, the two wave rows one barrier
// apart), with the operands turned around: the reduction index m is the ROW of both operands in memory, so a half-tile is 64 rows x
// 256 B (the 128 columns one phase needs: for dY the 64-column halves of both wave rows, for X the 32-column halves of the four wave
// columns), staged as it lies in memory (one 1 KiB DMA = 4 rows), and the fragments -- 8 consecutive m per lane -- are read with the
// transposing ds_read_b64_tr_b16.  k-slot (g, u, q) of a 32-row k-step <-> LDS row 16u + 4g + q on both operands.  The 32-byte column
// block b of row r sits at block b ^ (r & 7) (applied on the DMA's source column), so the 8 rows a 32-lane half of a transposing read
// touches fall on disjoint bank octets.
//
// These products have few output tiles (fc1: 64, out-proj: 16) and a long reduction (12736 rows at bs 64), so the reduction is cut
// into S pieces over workgroups.  Float atomics into C would run at the chip's ~1.3 TB/s atomic rate (S x 16 MB for fc1); instead
// every piece stores its f32 tile to a lane-linear slab of the caller's workspace and a second, bandwidth-bound launch
// (tn_p8_reduce_kernel) adds the slabs of a tile in piece order onto C: no inter-workgroup protocol, and a fixed summation order
// (bit-reproducible gradients).  Measured against "the piece with the last ticket sums in-launch": 856 vs 801 TFLOP/s on fc1 (4 pieces),
// 522 vs 286 on out-proj (16 pieces: one workgroup reading 15 slabs is serial).
#include "occ_common.h"
#include <stdlib.h>
#include <type_traits>

namespace occ_tn_p8 {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((address_space(3))) void lds_void;

struct Args {
    long long N1, N2;
    const char* A; RowMapI amap;       // dY [M, N1]
    const char* B; RowMapI bmap;       // X  [M, N2]
    float* C; long long ldc; float alpha;
    int t1, t2, S;                     // tiles along N1 / N2, pieces of the reduction
    int piece_major;                   // block -> (piece, tile) order, see tn_map
    int nt;                            // K-tiles (64 rows) in all; piece s takes [s*nt/S, (s+1)*nt/S)
    int plain;                         // both row maps are m * row_stride: the K advance is a scalar add
    float* slabs;                      // [tile][piece][8 waves][32 accumulators][64 lanes] f32x4, pieces > 1 only
    int store;                         // C is known to hold zeros: store alpha * product instead of read-add-write
};

template <int OFF> __device__ __forceinline__ u32x2 tr_read(unsigned addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// one 16-column x 32-row operand fragment: rows 4g+q (+16 for the second read) of the k-step, at byte offset OFF of the lane's block address
template <int OFF> __device__ __forceinline__ u32x4 frag(unsigned addr) {
    const u32x2 lo = tr_read<OFF>(addr), hi = tr_read<OFF + 4096>(addr);
    return (u32x4){lo[0], lo[1], hi[0], hi[1]};
}

#define TP_MFMA(ACC, WF, XF) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, WF), __builtin_bit_cast(bf16x8, XF), ACC, 0, 0, 0)
#define TP_QUAD(NH, MH, WQ)                                                                        \
    __builtin_amdgcn_s_setprio(1);                                                                 \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                               \
        _Pragma("unroll") for (int mf = 0; mf < 4; ++mf)                                           \
            _Pragma("unroll") for (int nf = 0; nf < 2; ++nf)                                       \
                TP_MFMA(acc[(NH) * 2 + nf][(MH) * 4 + mf], WQ[nf][ks], x[mf][ks]);                 \
    __builtin_amdgcn_s_setprio(0);                                                                 \
    __builtin_amdgcn_sched_barrier(0);
#define TP_SYNC_READS()                                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                             \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    __builtin_amdgcn_s_barrier();                                                                  \
    __builtin_amdgcn_sched_barrier(0);
#define TP_END_PHASE()                                                                             \
    __builtin_amdgcn_s_barrier();                                                                  \
    __builtin_amdgcn_sched_barrier(0);

// XCD-contiguous virtual block id: blocks b, b + 8, ... share an XCD and get a contiguous range of work
__device__ __forceinline__ int tn_vid() {
    const int total = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q8 = total >> 3, r8 = total & 7;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
}

template <bool PLAIN>
__device__ __forceinline__ void tn_p8_body(const Args& a, const int piece, const int tile) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * 65536];
    const int tx = tile % a.t1, ty = tile / a.t1;
    const long long n1_0 = (long long)tx * 256, n2_0 = (long long)ty * 256;
    const int kt0 = (int)((long long)piece * a.nt / a.S), kt1 = (int)((long long)(piece + 1) * a.nt / a.S);
    const int nt = kt1 - kt0;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;           // wave row: 128 columns of dY (n1); wave column: 64 columns of X (n2)

    // ---- LDS-DMA: one instruction = 4 rows x 256 B; lane -> row rl of the 4, 16-byte position p16; pass q covers rows q*32 + wave*4 + rl
    const int rl = lane >> 4, p16 = lane & 15;
    const int sb = (p16 >> 1) ^ ((4 * (wave & 1) + rl) & 7);        // source 32-byte block of this lane's position (row & 7 is the same in both passes)
    // dY half h: block sb = (wave row j, 16-column block f) -> column j*128 + h*64 + f*16; X half h: (wave column j, f) -> j*64 + h*32 + f*16
    const unsigned colA = (unsigned)((n1_0 + (sb >> 2) * 128 + (sb & 3) * 16 + (p16 & 1) * 8) * 2);
    const unsigned colB = (unsigned)((n2_0 + (sb >> 1) * 64 + (sb & 1) * 16 + (p16 & 1) * 8) * 2);
    const int rloc = wave * 4 + rl;
    unsigned voA[2], voB[2];                           // byte offsets of this lane's two rows (+ column) from the scalar base
    const char* baseA = a.A; const char* baseB = a.B;
    long long advA = 0, advB = 0;
    if (PLAIN) {
        voA[0] = (unsigned)(rloc * a.amap.rstride * 2) + colA; voA[1] = (unsigned)((rloc + 32) * a.amap.rstride * 2) + colA;
        voB[0] = (unsigned)(rloc * a.bmap.rstride * 2) + colB; voB[1] = (unsigned)((rloc + 32) * a.bmap.rstride * 2) + colB;
        advA = 64 * a.amap.rstride * 2; advB = 64 * a.bmap.rstride * 2;
        baseA += (long long)kt0 * advA; baseB += (long long)kt0 * advB;
    }
    auto rowsA = [&](int t) {                          // general row maps (conv windows, padded buffers): offsets of K-tile kt0 + t
        const long long m = ((long long)(kt0 + t)) * 64 + rloc;
        voA[0] = (unsigned)(row_off(a.amap, m) * 2) + colA; voA[1] = (unsigned)(row_off(a.amap, m + 32) * 2) + colA;
    };
    auto rowsB = [&](int t) {
        const long long m = ((long long)(kt0 + t)) * 64 + rloc;
        voB[0] = (unsigned)(row_off(a.bmap, m) * 2) + colB; voB[1] = (unsigned)(row_off(a.bmap, m + 32) * 2) + colB;
    };
    const unsigned dma_dst = (unsigned)(uintptr_t)(lds_void*)lds + (unsigned)wave * 1024u;
    // (no instruction offset on the DMA: an immediate would move the LDS destination as well as the source; the half's column shift goes
    // into the scalar base instead)
#define TP_DMA(SBASE, VOFF, LDST)                                                                                                     \
    {                                                                                                                                  \
        unsigned keep__;                                                                                                               \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"            \
                     : "=&s"(keep__) : "v"(VOFF), "s"(SBASE), "s"(LDST) : "memory");                                                   \
    }
    // half-tile ids: 0 = dY half 0, 1 = dY half 1, 2 = X half 0, 3 = X half 1 (64 rows x 256 B each); pass 1 lands 32 rows = 8 KiB further
#define TP_STAGE_A(H, BUF, T)                                                                                                          \
    {                                                                                                                                  \
        const char* sb__ = baseA + (PLAIN ? (long long)(T) * advA : 0ll) + (H) * 128;                                                  \
        TP_DMA(sb__, voA[0], dma_dst + (BUF) * 65536 + (H) * 16384)                                                                    \
        TP_DMA(sb__, voA[1], dma_dst + (BUF) * 65536 + (H) * 16384 + 8192)                                                             \
    }
#define TP_STAGE_B(H, BUF, T)                                                                                                          \
    {                                                                                                                                  \
        const char* sb__ = baseB + (PLAIN ? (long long)(T) * advB : 0ll) + (H) * 64;                                                   \
        TP_DMA(sb__, voB[0], dma_dst + (BUF) * 65536 + 32768 + (H) * 16384)                                                            \
        TP_DMA(sb__, voB[1], dma_dst + (BUF) * 65536 + 32768 + (H) * 16384 + 8192)                                                     \
    }

    // ---- transposing fragment reads: lane (g, q, pp) supplies row 4g + q of the block, 8 bytes at pp*8 of the 32-byte column block
    const int g = lane >> 4, fr = lane & 15, q = fr >> 2, pp = lane & 3;
    const unsigned swz = (unsigned)((4 * (g & 1) + q) & 7);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)lds;
    const unsigned lane_off = (unsigned)((4 * g + q) * 256 + pp * 8);
    // block b of the wave sits at (b ^ swz) << 5: bits 5-7 of the address, which nothing else of it touches, so the other blocks of
    // the wave are XORs of the first one's address
    unsigned xa0[2], wa0[2];                           // [buffer]: dY block wr*4, X block wc*2
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        xa0[b] = lds0 + b * 65536 + lane_off + ((((unsigned)(wr * 4)) ^ swz) << 5);
        wa0[b] = lds0 + b * 65536 + 32768 + lane_off + ((((unsigned)(wc * 2)) ^ swz) << 5);
    }
#define XA(B, MF) (xa0[B] ^ ((MF) << 5))
#define WA(B, NF) (wa0[B] ^ ((NF) << 5))

    f32x4 acc[4][8];                                   // [16-column block of the wave's 64 n2][16-column block of its 128 n1]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- prologue: K-tile 0 whole, three half-tiles of K-tile 1 (order per tile: X0, dY0, X1, dY1)
    if (!PLAIN) { rowsA(0); rowsB(0); }
    TP_STAGE_B(0, 0, 0) TP_STAGE_A(0, 0, 0) TP_STAGE_B(1, 0, 0) TP_STAGE_A(1, 0, 0)
    if (nt > 1) {
        if (!PLAIN) { rowsA(1); rowsB(1); }
        TP_STAGE_B(0, 1, 1) TP_STAGE_A(0, 1, 1) TP_STAGE_B(1, 1, 1)
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 1) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }

    u32x4 w0[2][2], w1[2][2], x[4][2];                 // [16-column block][k-step]
    auto tile_body = [&](auto bufc, const int t) {
        constexpr int B = decltype(bufc)::value;
        // -------- P1: X half 0, dY half 0; stage dY half 1 of tile t+1 (its row offsets are still in voA from tile t+1's dY half 0)
        w0[0][0] = frag<0>(WA(B, 0)); w0[1][0] = frag<0>(WA(B, 1)); w0[0][1] = frag<8192>(WA(B, 0)); w0[1][1] = frag<8192>(WA(B, 1));
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) { x[mf][0] = frag<0>(XA(B, mf)); x[mf][1] = frag<8192>(XA(B, mf)); }
        if (t + 1 < nt) TP_STAGE_A(1, B ^ 1, t + 1)
        TP_SYNC_READS()
        TP_QUAD(0, 0, w0)
        TP_END_PHASE()
        // -------- P2: X half 1; stage X half 0 of tile t+2
        w1[0][0] = frag<16384>(WA(B, 0)); w1[1][0] = frag<16384>(WA(B, 1)); w1[0][1] = frag<16384 + 8192>(WA(B, 0)); w1[1][1] = frag<16384 + 8192>(WA(B, 1));
        if (t + 2 < nt) {
            if (!PLAIN) rowsB(t + 2);
            TP_STAGE_B(0, B, t + 2)
        }
        TP_SYNC_READS()
        TP_QUAD(1, 0, w1)
        TP_END_PHASE()
        // -------- P3: dY half 1; stage dY half 0 of tile t+2
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) { x[mf][0] = frag<16384>(XA(B, mf)); x[mf][1] = frag<16384 + 8192>(XA(B, mf)); }
        if (t + 2 < nt) {
            if (!PLAIN) rowsA(t + 2);
            TP_STAGE_A(0, B, t + 2)
        }
        TP_SYNC_READS()
        TP_QUAD(1, 1, w1)
        TP_END_PHASE()
        // -------- P4: stage X half 1 of tile t+2; tile t+1 has landed
        if (t + 2 < nt) {
            TP_STAGE_B(1, B, t + 2)
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        TP_QUAD(0, 1, w0)
        TP_END_PHASE()
    };
    int t = 0;
    for (; t + 1 < nt; t += 2) {
        tile_body(std::integral_constant<int, 0>{}, t);
        tile_body(std::integral_constant<int, 1>{}, t + 1);
    }
    if (t < nt) tile_body(std::integral_constant<int, 0>{}, t);
    if (wr == 0) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }

    // ---- pieces > 1: store the slab
    if (a.S > 1) {
        f32x4* mine = reinterpret_cast<f32x4*>(a.slabs) + ((long long)(tile * a.S + piece) * 8 + wave) * 32 * 64 + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) mine[(i * 8 + j) * 64] = acc[i][j];
        return;                                    // tn_p8_reduce_kernel sums the slabs of every tile in piece order
    }
    // ---- C[n1][n2 .. n2+3] += alpha * acc: D rows (4g + e) are X columns (n2), D column fr is a dY column (n1)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const long long n1 = n1_0 + wr * 128 + j * 16 + fr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long n2 = n2_0 + wc * 64 + i * 16 + g * 4;
            float* cp = a.C + n1 * a.ldc + n2;
            f32x4 c = acc[i][j] * a.alpha;
            if (!a.store) c += *reinterpret_cast<f32x4*>(cp);
            *reinterpret_cast<f32x4*>(cp) = c;
        }
    }
}

// block -> (piece, tile).  Every workgroup of a launch is resident at once (tiles x pieces <= CUs), so what an XCD's L2 can share is decided by
// which 32 workgroups land on it.  piece_major (round 4): an XCD takes ONE piece of the reduction (a contiguous range of operand rows) for a
// run of consecutive tiles, so the row range of every operand panel those tiles touch is fetched once per XCD and reused by all of them --
// the old order (a tile's pieces side by side) gave an XCD eight tiles x all pieces, i.e. eight full-length panels of dY that nothing shared
// (measured 615 MB per paired launch against 170 / 290 MB algorithmic).
__device__ __forceinline__ void tn_map(const int vid, const int tiles, const int S, const int piece_major, int& piece, int& tile) {
    if (piece_major) { piece = vid / tiles; tile = vid - piece * tiles; }
    else { piece = vid % S; tile = vid / S; }
}

template <bool PLAIN>
__global__ __launch_bounds__(512, 2) void gemm_tn_p8_kernel(const Args a) {
    int piece, tile;
    tn_map(tn_vid(), a.t1 * a.t2, a.S, a.piece_major, piece, tile);
    tn_p8_body<PLAIN>(a, piece, tile);
}

// Two products of one layer in one launch (out-proj with qkv, fc2 with fc1: same reduction length, different operands).  Each launch has
// one workgroup per CU and leaves 64 MB of f32 slabs however few tiles it has, so two launches of 16 + 48 tiles cut their reductions 16
// and 5 ways where one launch of 64 tiles cuts them 4 ways: half the slab traffic, K loops four times as long for the small product.
struct ArgsPair { Args v[2]; int wgs0; };
template <bool PLAIN>
__global__ __launch_bounds__(512, 2) void gemm_tn_p8_pair_kernel(const ArgsPair ap) {
    const int vid = tn_vid();
    const int tiles0 = ap.v[0].t1 * ap.v[0].t2;
    if (ap.v[0].piece_major) {                          // (piece, then the tiles of both products in a row)
        int piece, t;
        tn_map(vid, tiles0 + ap.v[1].t1 * ap.v[1].t2, ap.v[0].S, 1, piece, t);
        const int second = t >= tiles0 ? 1 : 0;         // wave-uniform: the argument block is read from the kernel-argument segment
        tn_p8_body<PLAIN>(ap.v[second], piece, t - (second ? tiles0 : 0));
        return;
    }
    const int second = vid >= ap.wgs0 ? 1 : 0;
    const int v2 = vid - (second ? ap.wgs0 : 0);
    tn_p8_body<PLAIN>(ap.v[second], v2 % ap.v[second].S, v2 / ap.v[second].S);
}

// C += alpha * sum over pieces of the slabs (reduce_kernel mode): thread = one f32x4 of one (tile, wave, accumulator, lane) slot,
// i.e. the slab layout itself (coalesced 1 KiB reads per wave and piece), C addressed as the main kernel's epilogue does.
__global__ __launch_bounds__(256) void tn_p8_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ C, long long ldc, float alpha, int S, int t1, int store) {
    const int tile = blockIdx.y;
    const int slot = blockIdx.x * 256 + threadIdx.x;          // [wave 8][acc 32][lane 64]
    const int lane = slot & 63, k = (slot >> 6) & 31, wave = slot >> 11;
    const f32x4* p = reinterpret_cast<const f32x4*>(slabs) + (long long)tile * S * 16384 + slot;
    f32x4 v = p[0];
    for (int s = 1; s < S; ++s) v += p[(long long)s * 16384];
    const int i = k >> 3, j = k & 7, wr = wave >> 2, wc = wave & 3, fr = lane & 15, g = lane >> 4;
    const long long n1 = (long long)(tile % t1) * 256 + wr * 128 + j * 16 + fr, n2 = (long long)(tile / t1) * 256 + wc * 64 + i * 16 + g * 4;
    f32x4* cp = reinterpret_cast<f32x4*>(C + n1 * ldc + n2);
    v = v * alpha;
    if (!store) v += *cp;                              // (store: C is known to hold zeros -- no read)
    *cp = v;
}

// the same for a pair of products sharing S: blockIdx.y < tiles0 -> first product
__global__ __launch_bounds__(256) void tn_p8_reduce_pair_kernel(const float* __restrict__ slabs0, float* __restrict__ C0, long long ldc0, int t1_0, int tiles0,
                                                               const float* __restrict__ slabs1, float* __restrict__ C1, long long ldc1, int t1_1, float alpha0,
                                                               float alpha1, int S, int store0, int store1) {
    const bool second = (int)blockIdx.y >= tiles0;
    const int tile = second ? blockIdx.y - tiles0 : blockIdx.y;
    const float* slabs = second ? slabs1 : slabs0;
    float* C = second ? C1 : C0;
    const long long ldc = second ? ldc1 : ldc0;
    const int t1 = second ? t1_1 : t1_0;
    const float alpha = second ? alpha1 : alpha0;
    const int store = second ? store1 : store0;
    const int slot = blockIdx.x * 256 + threadIdx.x;
    const int lane = slot & 63, k = (slot >> 6) & 31, wave = slot >> 11;
    const f32x4* p = reinterpret_cast<const f32x4*>(slabs) + (long long)tile * S * 16384 + slot;
    f32x4 v = p[0];
    for (int s = 1; s < S; ++s) v += p[(long long)s * 16384];
    const int i = k >> 3, j = k & 7, wr = wave >> 2, wc = wave & 3, fr = lane & 15, g = lane >> 4;
    const long long n1 = (long long)(tile % t1) * 256 + wr * 128 + j * 16 + fr, n2 = (long long)(tile / t1) * 256 + wc * 64 + i * 16 + g * 4;
    f32x4* cp = reinterpret_cast<f32x4*>(C + n1 * ldc + n2);
    v = v * alpha;
    if (!store) v += *cp;
    *cp = v;
}

}  // namespace occ_tn_p8

// Two products in one launch.  Returns 1 when launched, 0 when the pair is not eligible (the caller runs them one by one).
// Requirements on top of occ_tn_p8_try's: the same M (a multiple of 64), plain row maps on all four operands, a workspace for the slabs.
int occ_tn_p8_pair_try(long long M, const long long* N1, const long long* N2, const void* const* A, const RowMapI* amap, const void* const* B, const RowMapI* bmap,
                       float* const* C, const long long* ldc, const float* alpha, void* workspace, long long workspace_bytes, const long long* max_a_off,
                       const long long* max_b_off, const int* store, hipStream_t s) {
    using namespace occ_tn_p8;
    static const int en = getenv("OCC_TN_P8") ? atoi(getenv("OCC_TN_P8")) : 1;
    static const int pair_en = getenv("OCC_TN_PAIR") ? atoi(getenv("OCC_TN_PAIR")) : 1;
    if (!en || !pair_en || M < 1024 || M % 64 || !workspace || ((uintptr_t)workspace & 15)) return 0;
    ArgsPair ap;
    long long tiles[2];
    for (int p = 0; p < 2; ++p) {
        if (N1[p] % 256 || N2[p] % 256 || ldc[p] % 4 || ((uintptr_t)C[p] & 15)) return 0;
        if (max_a_off[p] * 2 >= (1ll << 32) || max_b_off[p] * 2 >= (1ll << 32)) return 0;
        if (!(amap[p].rpl == 0 && bmap[p].rpl == 0 && amap[p].rpb >= M && bmap[p].rpb >= M)) return 0;
        Args& a = ap.v[p];
        a.N1 = N1[p]; a.N2 = N2[p]; a.A = (const char*)A[p]; a.amap = amap[p]; a.B = (const char*)B[p]; a.bmap = bmap[p]; a.C = C[p]; a.ldc = ldc[p]; a.alpha = alpha[p];
        a.t1 = (int)(N1[p] / 256); a.t2 = (int)(N2[p] / 256);
        a.nt = (int)(M / 64);
        a.plain = 1; a.store = store[p];
        static const int pm_env = getenv("OCC_TN_PIECE_MAJOR") ? atoi(getenv("OCC_TN_PIECE_MAJOR")) : 1;
        a.piece_major = pm_env;
        tiles[p] = (long long)a.t1 * a.t2;
    }
    static int cus = 0;
    if (!cus) { int dev = 0, v = 0; cus = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256; }
    const long long T = tiles[0] + tiles[1];
    long long S = cus / T;
    if (S > ap.v[0].nt / 8) S = ap.v[0].nt / 8;
    if (S < 2 || T * S * 262144 > workspace_bytes) return 0;          // (a single piece has no slabs: nothing to gain over two launches)
    for (int p = 0; p < 2; ++p) ap.v[p].S = (int)S;
    ap.v[0].slabs = (float*)workspace;
    ap.v[1].slabs = (float*)workspace + tiles[0] * S * 65536;
    ap.wgs0 = (int)(tiles[0] * S);
    hipLaunchKernelGGL(gemm_tn_p8_pair_kernel<true>, dim3((unsigned)(T * S)), dim3(512), 0, s, ap);
    hipLaunchKernelGGL(tn_p8_reduce_pair_kernel, dim3(64, (unsigned)T), dim3(256), 0, s, ap.v[0].slabs, ap.v[0].C, ap.v[0].ldc, ap.v[0].t1, (int)tiles[0], ap.v[1].slabs,
                       ap.v[1].C, ap.v[1].ldc, ap.v[1].t1, ap.v[0].alpha, ap.v[1].alpha, (int)S, store[0], store[1]);
    return 1;
}

// Host side.  Returns 1 when the launch was made, 0 when the problem is not this kernel's (the caller falls back), < 0 on error.
// Covers rows [0, rows64) with rows64 = M - M % 64; the caller adds the last M % 64 rows with the small-tile kernel.
int occ_tn_p8_try(long long M, long long N1, long long N2, const void* A, const RowMapI& amap, const void* B, const RowMapI& bmap, float* C, long long ldc,
                  float alpha, void* workspace, long long workspace_bytes, long long max_a_off, long long max_b_off, int store, hipStream_t s) {
    using namespace occ_tn_p8;
    static const int en = getenv("OCC_TN_P8") ? atoi(getenv("OCC_TN_P8")) : 1;
    if (!en || N1 % 256 || N2 % 256 || M < 1024 || ldc % 4 || ((uintptr_t)C & 15)) return 0;
    if (max_a_off * 2 >= (1ll << 32) || max_b_off * 2 >= (1ll << 32)) return 0;       // DMA offsets are 32-bit byte offsets from the operand base
    Args a;
    a.N1 = N1; a.N2 = N2; a.A = (const char*)A; a.amap = amap; a.B = (const char*)B; a.bmap = bmap; a.C = C; a.ldc = ldc; a.alpha = alpha;
    a.t1 = (int)(N1 / 256); a.t2 = (int)(N2 / 256);
    a.nt = (int)(M / 64); a.store = store;
    static const int pm_env = getenv("OCC_TN_PIECE_MAJOR") ? atoi(getenv("OCC_TN_PIECE_MAJOR")) : 1;
    a.piece_major = pm_env;
    a.plain = amap.rpl == 0 && bmap.rpl == 0 && amap.rpb >= M && bmap.rpb >= M;
    static int cus = 0;
    if (!cus) { int dev = 0, v = 0; cus = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256; }
    const long long tiles = (long long)a.t1 * a.t2;
    long long S = cus / tiles;                         // one workgroup per CU at most; a piece keeps at least 8 K-tiles
    if (S > a.nt / 8) S = a.nt / 8;
    if (S < 1) S = 1;
    // a tile's slabs: S x 256 KiB
    while (S > 1 && tiles * S * 262144 > workspace_bytes) --S;
    if (S > 1 && (!workspace || ((uintptr_t)workspace & 15))) S = 1;
    a.S = (int)S;
    a.slabs = S > 1 ? (float*)workspace : nullptr;
    if (a.plain) hipLaunchKernelGGL(gemm_tn_p8_kernel<true>, dim3((unsigned)(tiles * S)), dim3(512), 0, s, a);
    else hipLaunchKernelGGL(gemm_tn_p8_kernel<false>, dim3((unsigned)(tiles * S)), dim3(512), 0, s, a);
    if (S > 1)
        hipLaunchKernelGGL(tn_p8_reduce_kernel, dim3(64, (unsigned)tiles), dim3(256), 0, s, a.slabs, a.C, a.ldc, a.alpha, a.S, a.t1, a.store);
    return 1;
}
