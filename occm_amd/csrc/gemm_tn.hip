// Weight-gradient GEMM for gfx950 (f32, exact-f32 MFMA):   C[n1, n2] += sum_m A[m, n1] * B[m, n2]
//   A: [M, N1] rows through an occ_rowmap (dY of a Linear / conv layer, N1 contiguous)
//   B: [M, N2] rows through an occ_rowmap + K-segments (the layer input, or its conv windows)
//   C: f32 [N1, ldc], accumulated with float atomics (the M range is split over workgroups)
// and the matching bias gradient  out[n] += sum_m A[m, n].
// Backward of every nn.Linear / Conv2d weight of the AASIST back-end (sslassist.py:58-504).
//
// Workgroup = 4 waves, output tile 64(n1) x 64(n2); 32-row slabs of A and B are staged in LDS
// (row stride 80 floats: the 4 k-groups of a 16x16x4 MFMA land on disjoint bank sets) and every wave
// owns a 32x32 quadrant = 2x2 MFMA tiles.  v_mfma_f32_16x16x4_f32 takes ONE f32 per lane for each
// operand, so no transposed copy of either operand is ever materialised.
#include "occ_common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int TT = 64, SLAB = 32, LDS_STRIDE = 80, THREADS = 256;

struct TnArgs {
    long long M, N1, N2;
    const void* A; RowMapI amap;
    const void* B; RowMapI bmap; long long nseg, seg_len, seg_stride;
    float* C; long long ldc;
    long long rows_per_split;
    float alpha;
    int atomic;               // gemm_tn_dma_kernel: several row splits add into C -> atomics; one split: plain read-modify-write
    int t1, t2;               // tiles along N1 / N2 (grid is 1-D: XCD-aware order, see tn_block)
    float* colsum;            // optional: colsum[n1] += alpha * sum_m A[m, n1] (bias gradient), done by the n2-tile-0 workgroups
    long long m_first;        // gemm_tn_dma_kernel: first reduction row of this launch (the rows before it were done by gemm_tn_p8)
    int ngroups;              // gemm_tn_dma_kernel: independent products (grouped conv: one per channel group) folded into the grid
    long long a_gs, b_gs, c_gs;   // element strides between the groups' A / B / C
};

__device__ __forceinline__ float4 ld4_bf16(const unsigned short* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}

// 1-D grid -> (n1 tile, n2 tile, row split).  Workgroup b runs on XCD b % 8; the blocks of one XCD get a contiguous range of
// logical ids ordered split-major, so every tile that reads one row range sits on the same XCD and that range crosses the
// Infinity-Cache -> L2 boundary once instead of once per XCD (measured on the 88704 x 64 x 384 conv weight gradient: the
// un-mapped grid moved ~8x the operand bytes into the L2s and ran at 1 TB/s of unique operand bytes).
__device__ __forceinline__ int tn_lid() {
    const int total = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = total >> 3, r8 = total & 7;
    return (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
}
__device__ __forceinline__ void tn_block(const TnArgs& a, int& bx, int& by, int& bz) {
    const int l = tn_lid();
    bx = l % a.t1; by = (l / a.t1) % a.t2; bz = l / (a.t1 * a.t2);
}

template <bool ABF, bool BBF>      // operands f32 or bf16 in memory; the MFMA is the exact-f32 one either way
__global__ __launch_bounds__(THREADS) void gemm_tn_kernel(const TnArgs a) {
    __shared__ __attribute__((aligned(16))) float As[SLAB * LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) float Bs[SLAB * LDS_STRIDE];
    int bx, by, bz; tn_block(a, bx, by, bz);
    const long long n1_0 = (long long)bx * TT, n2_0 = (long long)by * TT;
    const long long m_begin = (long long)bz * a.rows_per_split;
    const long long m_end = m_begin + a.rows_per_split < a.M ? m_begin + a.rows_per_split : a.M;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1, fr = lane & 15, g = lane >> 4;

    // staging: thread copies float4 #c4 of rows srow, srow+16
    const int c4 = tid & 15, srow = tid >> 4;
    const long long n1c = n1_0 + c4 * 4, n2c = n2_0 + c4 * 4;
    const bool a_ok = n1c < a.N1, b_ok = n2c < a.N2;      // N1, N2 are multiples of 4
    long long bseg_off = 0;
    if (b_ok) {
        if (a.nseg > 1) { const long long sg = n2c / a.seg_len; bseg_off = sg * a.seg_stride + (n2c - sg * a.seg_len); }
        else bseg_off = n2c;
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long m0 = m_begin; m0 < m_end; m0 += SLAB) {
        float4 va[2], vb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long long m = m0 + srow + 16 * i;
            va[i] = make_float4(0.f, 0.f, 0.f, 0.f); vb[i] = va[i];
            if (m < m_end) {
                if (a_ok) va[i] = ABF ? ld4_bf16((const unsigned short*)a.A + row_off(a.amap, m) + n1c)
                                      : *reinterpret_cast<const float4*>((const float*)a.A + row_off(a.amap, m) + n1c);
                if (b_ok) vb[i] = BBF ? ld4_bf16((const unsigned short*)a.B + row_off(a.bmap, m) + bseg_off)
                                      : *reinterpret_cast<const float4*>((const float*)a.B + row_off(a.bmap, m) + bseg_off);
            }
        }
        csum.x += va[0].x + va[1].x; csum.y += va[0].y + va[1].y; csum.z += va[0].z + va[1].z; csum.w += va[0].w + va[1].w;
        __syncthreads();                                   // previous slab fully consumed
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<float4*>(&As[(srow + 16 * i) * LDS_STRIDE + c4 * 4]) = va[i];
            *reinterpret_cast<float4*>(&Bs[(srow + 16 * i) * LDS_STRIDE + c4 * 4]) = vb[i];
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < SLAB / 4; ++ks) {
            const int kr = (ks * 4 + g) * LDS_STRIDE;
            float af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { af[i] = As[kr + wi * 32 + i * 16 + fr]; bf[i] = Bs[kr + wj * 32 + i * 16 + fr]; }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    if (a.colsum && by == 0) {                     // bias gradient: reduce the 16 row-groups through LDS
        __syncthreads();
        *reinterpret_cast<float4*>(&As[srow * LDS_STRIDE + c4 * 4]) = csum;
        __syncthreads();
        if (tid < TT && n1_0 + tid < a.N1) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += As[r * LDS_STRIDE + tid];
            atomicAdd(a.colsum + n1_0 + tid, s * a.alpha);
        }
    }
    // D[row = n1 (A index)][col = n2]: lane holds col fr, rows 4g + r
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long long n2 = n2_0 + wj * 32 + j * 16 + fr;
            if (n2 >= a.N2) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long n1 = n1_0 + wi * 32 + i * 16 + g * 4 + r;
                if (n1 < a.N1) atomicAdd(a.C + n1 * a.ldc + n2, acc[i][j][r] * a.alpha);
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16-MFMA form (occ_gemm_tn_desc.compute = OCC_BF16): operands are rounded to bf16 while they are staged, products are
// accumulated in f32 by v_mfma_f32_16x16x32_bf16 -- 16x the MFMA rate of the exact-f32 form, which turns the kernel from
// MFMA-bound into load-bound.  The reduction index m is the LDS row, so a fragment (8 consecutive k per lane) is a COLUMN
// piece of the staged slab: read with ds_read_b64_tr_b16 (gfx950 transposing read; 4 rows x 16 columns per 16-lane group).
// k-slot (g, h, q) of a 32-row block maps to LDS row 16h + 4g + q on BOTH operands (any bijection works for a dot product);
// with a 160-byte row stride the 8 consecutive rows a 32-lane half touches fall on 8 disjoint bank octets.
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
constexpr int SLABB = 64, ROWB = 160;            // rows per slab, bytes per LDS row (64 bf16 + pad)

template <bool ABF, bool BBF>
__global__ __launch_bounds__(THREADS) void gemm_tn_bf16_kernel(const TnArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char As[SLABB * ROWB];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[SLABB * ROWB];
    int bx, by, bz; tn_block(a, bx, by, bz);
    const long long n1_0 = (long long)bx * TT, n2_0 = (long long)by * TT;
    const long long m_begin = (long long)bz * a.rows_per_split;
    const long long m_end = m_begin + a.rows_per_split < a.M ? m_begin + a.rows_per_split : a.M;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1, fr = lane & 15, g = lane >> 4;
    const int c4 = tid & 15, srow = tid >> 4;
    const long long n1c = n1_0 + c4 * 4, n2c = n2_0 + c4 * 4;
    const bool a_ok = n1c < a.N1, b_ok = n2c < a.N2;
    long long bseg_off = 0;
    if (b_ok) {
        if (a.nseg > 1) { const long long sg = n2c / a.seg_len; bseg_off = sg * a.seg_stride + (n2c - sg * a.seg_len); }
        else bseg_off = n2c;
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 va[4], vb[4];
    // branch-free: out-of-range rows / columns read a valid element (last row, column 0) and are zeroed by a select, so the
    // eight loads of a slab are all in flight together instead of sitting in eight divergent regions
    const long long acol = a_ok ? n1c : 0, bcol = b_ok ? bseg_off : 0;
    auto fetch = [&](long long m0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long m = m0 + srow + 16 * i;
            const bool ok = m < m_end;
            const long long mm = ok ? m : m_end - 1;
            va[i] = ABF ? ld4_bf16((const unsigned short*)a.A + row_off(a.amap, mm) + acol)
                        : *reinterpret_cast<const float4*>((const float*)a.A + row_off(a.amap, mm) + acol);
            vb[i] = BBF ? ld4_bf16((const unsigned short*)a.B + row_off(a.bmap, mm) + bcol)
                        : *reinterpret_cast<const float4*>((const float*)a.B + row_off(a.bmap, mm) + bcol);
            if (!(ok && a_ok)) va[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!(ok && b_ok)) vb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto pack4 = [](const float4 v) {
        return make_uint2(pack_bf16x2(v.x, v.y),
                          pack_bf16x2(v.z, v.w));
    };
    // per-lane byte offsets of the transposing reads: row 16h + 4g + q, columns 4p .. 4p+3 of the fragment's 16 columns
    const int q = (lane & 15) >> 2, p = lane & 3;
    const int roff = (4 * g + q) * ROWB + p * 8;
    fetch(m_begin);
    for (long long m0 = m_begin; m0 < m_end; m0 += SLABB) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { csum.x += va[i].x; csum.y += va[i].y; csum.z += va[i].z; csum.w += va[i].w; }
        __syncthreads();                                   // previous slab fully consumed
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<uint2*>(&As[(srow + 16 * i) * ROWB + c4 * 8]) = pack4(va[i]);
            *reinterpret_cast<uint2*>(&Bs[(srow + 16 * i) * ROWB + c4 * 8]) = pack4(vb[i]);
        }
        __syncthreads();
        if (m0 + SLABB < m_end) fetch(m0 + SLABB);         // next slab's loads fly under the MFMAs below
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            bf16x8_t af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int base = kb * 32 * ROWB + roff;
                const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(As + base + (wi * 32 + i * 16) * 2));
                const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(As + base + 16 * ROWB + (wi * 32 + i * 16) * 2));
                const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(Bs + base + (wj * 32 + i * 16) * 2));
                const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(Bs + base + 16 * ROWB + (wj * 32 + i * 16) * 2));
                af[i] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
                bf[i] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    if (a.colsum && by == 0) {                     // bias gradient (sums of the un-rounded values)
        __syncthreads();
        float* red = reinterpret_cast<float*>(As);         // 16 x 64 f32 = 4 KiB of the 10 KiB buffer
        *reinterpret_cast<float4*>(&red[srow * 64 + c4 * 4]) = csum;
        __syncthreads();
        if (tid < TT && n1_0 + tid < a.N1) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += red[r * 64 + tid];
            atomicAdd(a.colsum + n1_0 + tid, s * a.alpha);
        }
    }
    // D[row = n1][col = n2]: lane holds col fr, rows 4g + r
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long long n2 = n2_0 + wj * 32 + j * 16 + fr;
            if (n2 >= a.N2) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long n1 = n1_0 + wi * 32 + i * 16 + g * 4 + r;
                if (n1 < a.N1) atomicAdd(a.C + n1 * a.ldc + n2, acc[i][j][r] * a.alpha);
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// Large weight gradients (front-end fine-tuning: C[N1,N2] += dY[M,N1]^T . X[M,N2] with N1, N2 in the thousands): the 128x128
// LDS-DMA tiling of occ_gemm turned around.  Both operands are bf16 row-major with the reduction index m as the ROW, so they are
// staged as they lie in memory (64 rows x 256 B per operand and slab, one 1 KiB DMA = 4 rows) and the fragments -- 8 consecutive
// m per lane -- are read with the transposing ds_read_b64_tr_b16.  No transposed copies of dY / X are ever written, which is
// what the transpose + occ_gemm formulation spent ~10 % of a fine-tuning step on.
//   * swizzle: the 32-byte column block cb of row r sits at block cb ^ (r & 7); the 8 consecutive rows a 32-lane half of a
//     transposing read touches then fall on 8 disjoint bank octets (applied on the DMA's source address, LDS stays lane-linear);
//   * k-slot (g, h, q) of a 32-row block <-> LDS row 16h + 4g + q on both operands;
//   * MFMA(a = X fragment, b = dY fragment): a lane ends with C[n1][n2 .. n2+3], i.e. 16-byte accesses to C;
//   * the bias gradient (column sums of dY) is a separate colsum launch: fusing it (one MFMA per dY fragment against a fragment of
//     ones) pushed the kernel over 128 VGPRs and cost 15-45 % of its run time.
static inline long long cu_count_tn() {
    static int n = 0;
    if (!n) { int dev = 0, v = 0; n = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256; }
    return n;
}
typedef __attribute__((address_space(3))) void tn_lds_void;
typedef __attribute__((address_space(1))) const void tn_gbl_void;
//   * KG > 1: split of the reduction INSIDE the workgroup.  These products have few output tiles (fc1: 256, out-proj: 64) and a
//     reduction of 12736 rows; with one 4-wave workgroup per tile a CU has one wave per SIMD and nothing hides the load latency, and
//     splitting over workgroups costs f32 atomics (measured: every extra split adds ~20 us on a 1024x1024 output).  Here KG groups of
//     four waves each take a contiguous part of the workgroup's rows with their own 32 KiB LDS image and the same loop; at the end the
//     upper groups park their accumulators in LDS (pairwise tree: 64 KiB per parked group) and group 0 stores.  Fixed summation order.
constexpr int TD = 128, SLD = 64;
template <int KG>
__global__ __launch_bounds__(256 * KG) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_tn_dma_kernel(const TnArgs a) {
    __shared__ uint4 lds_all[KG * 2 * SLD * 16];          // per group: A slab (64 rows x 16 chunks), then B slab
    int bx, by, bz;
    long long grp = 0;                                    // grouped products: the group is the slowest index of the (XCD-contiguous) block id
    if (a.ngroups > 1) {
        const int l = tn_lid(), per = (int)(gridDim.x / a.ngroups), l2 = l % per;
        grp = l / per;
        bx = l2 % a.t1; by = (l2 / a.t1) % a.t2; bz = l2 / (a.t1 * a.t2);
    } else tn_block(a, bx, by, bz);
    const long long n1_0 = (long long)bx * TD, n2_0 = (long long)by * TD;
    const long long wg_begin = a.m_first + (long long)bz * a.rows_per_split;
    const long long wg_end = wg_begin + a.rows_per_split < a.M ? wg_begin + a.rows_per_split : a.M;
    const int kg = KG > 1 ? threadIdx.x >> 8 : 0;
    // every group runs the same number of slabs (the barriers are workgroup-wide); a group whose rows are used up stages zeros
    const long long rows_per_group = KG > 1 ? ((wg_end - wg_begin + KG - 1) / KG + SLD - 1) / SLD * SLD : wg_end - wg_begin;
    const long long m_begin = wg_begin + kg * rows_per_group;
    const long long m_end = m_begin + rows_per_group < wg_end ? m_begin + rows_per_group : wg_end;     // may be <= m_begin
    const int nsl = (int)((rows_per_group + SLD - 1) / SLD);
    uint4* lds = lds_all + kg * 2 * SLD * 16;
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1;
    // staging: DMA ii = pass * 4 + wave covers rows 4*ii .. 4*ii+3; lane -> (row_local = lane >> 4, 16-byte position p = lane & 15)
    const int rl = lane >> 4, p16 = lane & 15;
    // rows 4*ii + rl of this lane have (row & 7) = (4*wave + rl) & 7 for every pass: one swizzled column offset per operand
    const int cb = (p16 >> 1) ^ ((4 * wave + rl) & 7);
    long long c1 = n1_0 + cb * 16 + (p16 & 1) * 8; if (c1 > a.N1 - 8) c1 = a.N1 - 8;           // columns past N1 / N2 feed outputs that are never stored
    long long c2 = n2_0 + cb * 16 + (p16 & 1) * 8; if (c2 > a.N2 - 8) c2 = a.N2 - 8;
    // K-segmented B (conv windows): the lane's eight columns lie inside one segment (seg_len % 8 == 0); a tile may span several segments
    if (a.nseg > 1) { const long long sg = c2 / a.seg_len; c2 = sg * a.seg_stride + (c2 - sg * a.seg_len); }
    const unsigned short* Ap = (const unsigned short*)a.A + grp * a.a_gs + c1;
    const unsigned short* Bp = (const unsigned short*)a.B + grp * a.b_gs + c2;
    const bool plain = a.amap.rpl == 0 && a.bmap.rpl == 0 && a.amap.rpb >= a.M && a.bmap.rpb >= a.M;     // plain matrices: offset = m * row_stride
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int fr = lane & 15, g = lane >> 4, q = fr >> 2, pp = lane & 3;
    unsigned char* ldsA = reinterpret_cast<unsigned char*>(lds);
    unsigned char* ldsB = ldsA + SLD * 256;
    typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
    for (int sl = 0; sl < nsl; ++sl) {
        const long long m0 = m_begin + (long long)sl * SLD;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ii = i * 4 + wave;
            long long m = m0 + 4 * ii + rl; if (m > wg_end - 1) m = wg_end - 1;
            const long long oa = plain ? m * a.amap.rstride : row_off(a.amap, m), ob = plain ? m * a.bmap.rstride : row_off(a.bmap, m);
            __builtin_amdgcn_global_load_lds((tn_gbl_void*)(Ap + oa), (tn_lds_void*)&lds[ii * 64], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((tn_gbl_void*)(Bp + ob), (tn_lds_void*)&lds[SLD * 16 + ii * 64], 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (m0 + SLD > m_end) {                             // partial (or, for a used-up group, empty) slab: rows past the end must not contribute
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ii = i * 4 + wave;
                if (m0 + 4 * ii + rl >= m_end) { lds[ii * 64 + lane] = make_uint4(0, 0, 0, 0); lds[SLD * 16 + ii * 64 + lane] = make_uint4(0, 0, 0, 0); }
            }
        }
        __syncthreads();
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            bf8 af[4], bf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                s16x4 ar[2], br[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = kb * 32 + 16 * h + 4 * g + q;
                    ar[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ldsA + row * 256 + (((wi * 4 + t) ^ (row & 7)) << 5) + pp * 8));
                    br[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ldsB + row * 256 + (((wj * 4 + t) ^ (row & 7)) << 5) + pp * 8));
                }
                af[t] = __builtin_bit_cast(bf8, __builtin_shufflevector(ar[0], ar[1], 0, 1, 2, 3, 4, 5, 6, 7));
                bf[t] = __builtin_bit_cast(bf8, __builtin_shufflevector(br[0], br[1], 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int ia = 0; ia < 4; ++ia)
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) acc[ia][ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ib], af[ia], acc[ia][ib], 0, 0, 0);
        }
        __syncthreads();
    }
    if constexpr (KG > 1) {
        // pairwise tree over the groups: the upper half parks its accumulators (64 KiB per group = two groups' slab images), the lower
        // half adds them; the whole LDS is free after the loop's last barrier
        f32x4* red = reinterpret_cast<f32x4*>(lds_all);
#pragma unroll
        for (int half = KG / 2; half >= 1; half >>= 1) {
            if (kg >= half && kg < 2 * half) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) red[(((kg - half) * 4 + wave) * 16 + i * 4 + j) * 64 + lane] = acc[i][j];
            }
            __syncthreads();
            if (kg < half) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] += red[((kg * 4 + wave) * 16 + i * 4 + j) * 64 + lane];
            }
            __syncthreads();
        }
        if (kg != 0) return;
    }
    // D[row = n2 (4g + r)][col = n1 (fr)]: this lane owns C[n1][n2 .. n2+3]
#pragma unroll
    for (int ia = 0; ia < 4; ++ia) {
        const long long n1 = n1_0 + wi * 64 + ia * 16 + fr;
        if (n1 >= a.N1) continue;
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            const long long n2 = n2_0 + wj * 64 + ib * 16 + g * 4;
            if (n2 >= a.N2) continue;                       // N2 % 4 == 0
            float* cp = a.C + grp * a.c_gs + n1 * a.ldc + n2;
            if (a.atomic) {
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(cp + e, acc[ia][ib][e] * a.alpha);
            } else {
                f32x4 c = *reinterpret_cast<f32x4*>(cp);
                c += acc[ia][ib] * a.alpha;
                *reinterpret_cast<f32x4*>(cp) = c;
            }
        }
    }
}

// out[n] += alpha * sum_m A[m, n]
template <typename T>
__global__ __launch_bounds__(THREADS) void colsum_kernel(const T* __restrict__ A, RowMapI amap, long long M, long long N,
                                                         long long rows_per_split, float* __restrict__ out, float alpha) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long n = (long long)blockIdx.x * 64 + lane;
    const long long m_begin = (long long)blockIdx.y * rows_per_split;
    const long long m_end = m_begin + rows_per_split < M ? m_begin + rows_per_split : M;
    float s = 0.f;
    if (n < N)
        for (long long m = m_begin + wave; m < m_end; m += 4) s += occ_load_f32(A + row_off(amap, m) + n);
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && n < N) atomicAdd(out + n, (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) * alpha);
}

// bf16 column sums with 16-byte loads: a workgroup owns 64 columns x one row range; thread = (8-column group, row lane)
__global__ __launch_bounds__(THREADS) void colsum_bf16_vec_kernel(const unsigned short* __restrict__ A, RowMapI amap, long long M, long long N,
                                                                 long long rows_per_split, float* __restrict__ out, float alpha) {
    __shared__ float red[32][64];
    const int cg = threadIdx.x & 7, rlane = threadIdx.x >> 3;
    const long long c0 = (long long)blockIdx.x * 64 + cg * 8;
    const long long m_begin = (long long)blockIdx.y * rows_per_split;
    const long long m_end = m_begin + rows_per_split < M ? m_begin + rows_per_split : M;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c0 < N)                                           // N % 8 == 0
        for (long long m = m_begin + rlane; m < m_end; m += 32) {
            const uint4 u = *reinterpret_cast<const uint4*>(A + row_off(amap, m) + c0);
            const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[2 * e] += __uint_as_float(w[e] << 16); s[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u); }
        }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rlane][cg * 8 + e] = s[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) t += red[r][threadIdx.x];
        const long long n = (long long)blockIdx.x * 64 + threadIdx.x;
        if (n < N) atomicAdd(out + n, t * alpha);
    }
}

static void launch_colsum_bf16_vec(const unsigned short* A, const RowMapI& amap, long long M, long long N, float* out, float alpha, hipStream_t s) {
    // ~8 workgroups per CU: with 1024-row pieces a [12736 x 1024] sum ran on 208 workgroups (15 us, 1.7 TB/s)
    long long split = occ_cdiv(M, 192);
    const long long cap = occ_cdiv(2048, occ_cdiv(N, 64));
    if (split > cap) split = cap;
    if (split < 1) split = 1;
    const long long rps = occ_cdiv(M, split);
    split = occ_cdiv(M, rps);
    hipLaunchKernelGGL(colsum_bf16_vec_kernel, dim3((unsigned)occ_cdiv(N, 64), (unsigned)split), dim3(THREADS), 0, s, A, amap, M, N, rps, out, alpha);
}

}  // namespace

// gemm_tn_p8.hip
int occ_tn_p8_try(long long M, long long N1, long long N2, const void* A, const RowMapI& amap, const void* B, const RowMapI& bmap, float* C, long long ldc,
                  float alpha, void* workspace, long long workspace_bytes, long long max_a_off, long long max_b_off, int store, hipStream_t s);
int occ_tn_p8_pair_try(long long M, const long long* N1, const long long* N2, const void* const* A, const RowMapI* amap, const void* const* B, const RowMapI* bmap,
                       float* const* C, const long long* ldc, const float* alpha, void* workspace, long long workspace_bytes, const long long* max_a_off,
                       const long long* max_b_off, const int* store, hipStream_t s);

static long long max_row_off(const occ_rowmap& m, long long M) {       // upper bound of the element offset of any row < M
    const long long nb = (M - 1) / m.rows_per_batch;
    long long in_batch = m.rows_per_line > 0 ? (m.rows_per_batch / m.rows_per_line + 1) * m.line_stride + m.rows_per_line * m.row_stride : m.rows_per_batch * m.row_stride;
    if (in_batch < 0) in_batch = 0;
    return nb * (m.batch_stride > 0 ? m.batch_stride : 0) + in_batch;
}

extern "C" {

int occ_gemm_tn(const occ_gemm_tn_desc* d, void* stream);

// Two weight gradients of one layer (same reduction rows) in one launch of the eight-phase kernel where both qualify; otherwise one by one.
int occ_gemm_tn_pair(const occ_gemm_tn_desc* d0, const occ_gemm_tn_desc* d1, void* stream) {
    OCC_CHECK_ARG(d0 && d1, "occ_gemm_tn_pair: null descriptor");
    const occ_gemm_tn_desc* d[2] = {d0, d1};
    bool ok = d0->M == d1->M;
    // reduction rows beyond the last whole 64-row K-tile (variable-length groups: M = 12 * T is rarely a multiple of 64) go through the
    // small-tile kernel per product afterwards, as occ_gemm_tn does; the paired launch covers rows [0, rows64)
    const long long rows64 = d0->M - d0->M % 64;
    for (int p = 0; p < 2 && ok; ++p) {
        const occ_gemm_tn_desc* q = d[p];
        ok = q->A && q->B && q->C && q->compute == OCC_BF16 && q->a_dtype == OCC_BF16 && q->b_dtype == OCC_BF16 && q->b_nseg <= 1 && q->ldc >= q->N2 &&
             q->a_map.rows_per_batch >= 1 && q->b_map.rows_per_batch >= 1 && q->a_map.row_stride % 8 == 0 && q->b_map.row_stride % 8 == 0 &&
             ((uintptr_t)q->A & 15) == 0 && ((uintptr_t)q->B & 15) == 0;
    }
    if (ok) {
        long long N1[2], N2[2], ldc[2], ma[2], mb[2];
        const void* A[2]; const void* B[2]; float* C[2]; float alpha[2]; RowMapI am[2], bm[2];
        for (int p = 0; p < 2; ++p) {
            N1[p] = d[p]->N1; N2[p] = d[p]->N2; ldc[p] = d[p]->ldc; A[p] = d[p]->A; B[p] = d[p]->B; C[p] = (float*)d[p]->C; alpha[p] = d[p]->alpha;
            am[p] = to_rowmap(d[p]->a_map); bm[p] = to_rowmap(d[p]->b_map);
            ma[p] = max_row_off(d[p]->a_map, d[p]->M) + d[p]->N1; mb[p] = max_row_off(d[p]->b_map, d[p]->M) + d[p]->N2;
        }
        const bool tail_ok = rows64 == d0->M || (N1[0] % 8 == 0 && N2[0] % 8 == 0 && N1[1] % 8 == 0 && N2[1] % 8 == 0);
        const int store[2] = {d0->c_is_zero != 0, d1->c_is_zero != 0};
        if (tail_ok && occ_tn_p8_pair_try(rows64, N1, N2, A, am, B, bm, C, ldc, alpha, d0->workspace, d0->workspace_bytes, ma, mb, store, (hipStream_t)stream) == 1) {
            for (int p = 0; p < 2; ++p) {
                if (d[p]->colsum) launch_colsum_bf16_vec((const unsigned short*)d[p]->A, am[p], d[p]->M, d[p]->N1, (float*)d[p]->colsum, d[p]->alpha, (hipStream_t)stream);
                if (rows64 < d[p]->M) {
                    TnArgs t;
                    t.M = d[p]->M; t.N1 = N1[p]; t.N2 = N2[p]; t.A = A[p]; t.amap = am[p]; t.B = B[p]; t.bmap = bm[p]; t.nseg = 1; t.seg_len = N2[p]; t.seg_stride = 0;
                    t.C = C[p]; t.ldc = ldc[p]; t.alpha = alpha[p]; t.colsum = nullptr; t.atomic = 0; t.m_first = rows64; t.rows_per_split = SLD;
                    t.ngroups = 1; t.a_gs = t.b_gs = t.c_gs = 0;
                    t.t1 = (int)occ_cdiv(N1[p], TD); t.t2 = (int)occ_cdiv(N2[p], TD);
                    hipLaunchKernelGGL(gemm_tn_dma_kernel<1>, dim3((unsigned)(t.t1 * t.t2)), dim3(THREADS), 0, (hipStream_t)stream, t);
                }
            }
            OCC_LAUNCH_CHECK("occ_gemm_tn_pair");
            return OCC_OK;
        }
    }
    const int rc = occ_gemm_tn(d0, stream);
    return rc != OCC_OK ? rc : occ_gemm_tn(d1, stream);
}

int occ_gemm_tn(const occ_gemm_tn_desc* d, void* stream) {
    OCC_CHECK_ARG(d && d->A && d->B && d->C, "occ_gemm_tn: null operand");
    OCC_CHECK_ARG(d->M >= 1 && d->N1 >= 4 && d->N2 >= 4 && d->N1 % 4 == 0 && d->N2 % 4 == 0, "occ_gemm_tn: N1, N2 must be multiples of 4 (M=%ld N1=%ld N2=%ld)",
                  (long)d->M, (long)d->N1, (long)d->N2);
    const long long nseg = d->b_nseg > 1 ? d->b_nseg : 1;
    const long long seg_len = nseg > 1 ? d->b_seg_len : d->N2;
    OCC_CHECK_ARG(nseg * seg_len == d->N2 && seg_len % 4 == 0 && (nseg == 1 || d->b_seg_stride % 4 == 0), "occ_gemm_tn: bad B segments");
    OCC_CHECK_ARG(d->a_map.rows_per_batch >= 1 && d->b_map.rows_per_batch >= 1, "occ_gemm_tn: rows_per_batch must be >= 1");
    OCC_CHECK_ARG(d->a_map.row_stride % 4 == 0 && d->a_map.batch_stride % 4 == 0 && d->a_map.line_stride % 4 == 0 &&
                  d->b_map.row_stride % 4 == 0 && d->b_map.batch_stride % 4 == 0 && d->b_map.line_stride % 4 == 0,
                  "occ_gemm_tn: strides must keep rows 16-byte aligned");
    OCC_CHECK_ARG(((uintptr_t)d->A & 15) == 0 && ((uintptr_t)d->B & 15) == 0 && d->ldc >= d->N2, "occ_gemm_tn: alignment / ldc");
    TnArgs a;
    a.M = d->M; a.N1 = d->N1; a.N2 = d->N2;
    a.A = d->A; a.amap = to_rowmap(d->a_map);
    a.B = d->B; a.bmap = to_rowmap(d->b_map); a.nseg = nseg; a.seg_len = seg_len; a.seg_stride = d->b_seg_stride;
    a.C = (float*)d->C; a.ldc = d->ldc; a.alpha = d->alpha; a.colsum = (float*)d->colsum; a.atomic = 1; a.m_first = 0;
    const long long ngr = d->n_groups > 1 ? d->n_groups : 1;
    a.ngroups = (int)ngr; a.a_gs = ngr > 1 ? d->a_group_stride : 0; a.b_gs = ngr > 1 ? d->b_group_stride : 0; a.c_gs = ngr > 1 ? d->c_group_stride : 0;
    const bool abf0 = d->a_dtype == OCC_BF16, bbf0 = d->b_dtype == OCC_BF16;
    static const int dma_env = getenv("OCC_TN_DMA") ? atoi(getenv("OCC_TN_DMA")) : 1;
    if (dma_env && d->compute == OCC_BF16 && abf0 && bbf0 && d->N1 % 8 == 0 && d->N2 % 8 == 0 && d->N1 >= 64 && d->N2 >= 128 && d->M >= 256 &&
        (nseg == 1 || seg_len % 8 == 0) && (ngr == 1 || (d->a_group_stride % 8 == 0 && d->b_group_stride % 8 == 0 && d->c_group_stride % 4 == 0 && !d->colsum)) && d->ldc % 4 == 0 && ((uintptr_t)d->C & 15) == 0 &&
        d->a_map.row_stride % 8 == 0 && d->a_map.batch_stride % 8 == 0 && d->a_map.line_stride % 8 == 0 &&
        d->b_map.row_stride % 8 == 0 && d->b_map.batch_stride % 8 == 0 && d->b_map.line_stride % 8 == 0 && (nseg == 1 || d->b_seg_stride % 8 == 0)) {
        // Large outputs (multiples of 256 both ways, one K segment): the 256x256 eight-phase kernel on all whole 64-row K-tiles, reduction
        // pieces joined through workspace slabs instead of atomics; what is left (M % 64 rows) goes through the kernel below.
        if (nseg == 1 && ngr == 1 && d->N1 >= 128) {
            const long long rows64 = d->M - d->M % 64;
            const int r = occ_tn_p8_try(rows64, d->N1, d->N2, d->A, a.amap, d->B, a.bmap, a.C, a.ldc, a.alpha, d->workspace, d->workspace_bytes,
                                        max_row_off(d->a_map, d->M) + d->N1, max_row_off(d->b_map, d->M) + d->N2, d->c_is_zero != 0, (hipStream_t)stream);
            if (r < 0) { occ_set_error("occ_gemm_tn: workspace memset failed"); return OCC_ELAUNCH; }
            if (r == 1) {
                if (a.colsum) launch_colsum_bf16_vec((const unsigned short*)a.A, a.amap, a.M, a.N1, a.colsum, a.alpha, (hipStream_t)stream);
                if (rows64 < d->M) {
                    a.m_first = rows64; a.rows_per_split = SLD; a.atomic = 0;
                    a.t1 = (int)occ_cdiv(d->N1, TD); a.t2 = (int)occ_cdiv(d->N2, TD);
                    hipLaunchKernelGGL(gemm_tn_dma_kernel<1>, dim3((unsigned)(a.t1 * a.t2)), dim3(THREADS), 0, (hipStream_t)stream, a);
                }
                OCC_LAUNCH_CHECK("occ_gemm_tn");
                return OCC_OK;
            }
        }
        const long long u1 = occ_cdiv(d->N1, TD), u2 = occ_cdiv(d->N2, TD);
        // KG groups of four waves split the rows inside a workgroup (no atomics between them); workgroups beyond one per tile split the
        // rows further and do need atomics, so there are only as many as it takes to have ~one 16-wave workgroup per CU
        // Measured (scripts/bench_tn_kg.py, 48-tile outputs): KG = 4 wins up to ~50 k reduction rows (12736 rows: 113 -> 61 us; the
        // transformer gradients at 12736 rows: 251 -> 150 us), beyond ~100 k rows many independent 4-wave workgroups win (409536 rows: 1.0 vs 1.9 ms)
        static const int kg_env = getenv("OCC_TN_KG") ? atoi(getenv("OCC_TN_KG")) : 0;
        static const long long dma_target = getenv("OCC_TN_DMA_TARGET") ? atoll(getenv("OCC_TN_DMA_TARGET")) : 0;
        const int KGv = kg_env == 1 || kg_env == 2 || kg_env == 4 ? kg_env : (d->M <= 65536 ? 4 : 1);
        long long sp = occ_cdiv(dma_target > 0 ? dma_target : (KGv == 1 ? 512 : 2 * cu_count_tn() / KGv * 1), u1 * u2 * ngr);
        const long long msp = d->M / (256 * KGv);
        if (sp > msp) sp = msp;
        if (sp < 1) sp = 1;
        a.rows_per_split = occ_cdiv(occ_cdiv(d->M, sp), SLD) * SLD;
        sp = occ_cdiv(d->M, a.rows_per_split);
        a.t1 = (int)u1; a.t2 = (int)u2; a.atomic = sp > 1;
        OCC_CHECK_ARG(u1 * u2 * sp * ngr < (1ll << 30), "occ_gemm_tn: output too large");
        const dim3 grid((unsigned)(u1 * u2 * sp * ngr));
        if (KGv == 4) hipLaunchKernelGGL(gemm_tn_dma_kernel<4>, grid, dim3(1024), 0, (hipStream_t)stream, a);
        else if (KGv == 2) hipLaunchKernelGGL(gemm_tn_dma_kernel<2>, grid, dim3(512), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL(gemm_tn_dma_kernel<1>, grid, dim3(THREADS), 0, (hipStream_t)stream, a);
        if (a.colsum) launch_colsum_bf16_vec((const unsigned short*)a.A, a.amap, a.M, a.N1, a.colsum, a.alpha, (hipStream_t)stream);
        OCC_LAUNCH_CHECK("occ_gemm_tn");
        return OCC_OK;
    }
    if (ngr > 1) {                               // grouped, but not the LDS-DMA kernel's case (short reductions, f32 operands ...): one product per group
        OCC_CHECK_ARG(!d->colsum, "occ_gemm_tn: grouped products take no colsum (sum the whole A once with occ_colsum)");
        const long long ea = d->a_dtype == OCC_BF16 ? 2 : 4, eb = d->b_dtype == OCC_BF16 ? 2 : 4;
        for (long long g = 0; g < ngr; ++g) {
            occ_gemm_tn_desc one = *d;
            one.n_groups = 1;
            one.A = (const char*)d->A + g * d->a_group_stride * ea;
            one.B = (const char*)d->B + g * d->b_group_stride * eb;
            one.C = (float*)d->C + g * d->c_group_stride;
            const int rc = occ_gemm_tn(&one, stream);
            if (rc != OCC_OK) return rc;
        }
        return OCC_OK;
    }
    const long long t1 = occ_cdiv(d->N1, TT), t2 = occ_cdiv(d->N2, TT);
    static const long long target = getenv("OCC_TN_BLOCKS") ? atoll(getenv("OCC_TN_BLOCKS")) : 1024;
    long long split = occ_cdiv(target, t1 * t2);                     // aim at ~1024 workgroups
    const long long max_split = occ_cdiv(d->M, 4 * SLAB);            // at least 128 rows each
    if (split > max_split) split = max_split;
    if (split < 1) split = 1;
    if (split > 65535) split = 65535;
    a.rows_per_split = occ_cdiv(occ_cdiv(d->M, split), SLABB) * SLABB;          // multiple of both kernels' slab heights
    split = occ_cdiv(d->M, a.rows_per_split);
    OCC_CHECK_ARG(t1 * t2 * split < (1ll << 30), "occ_gemm_tn: output too large");
    a.t1 = (int)t1; a.t2 = (int)t2;
    const dim3 grid((unsigned)(t1 * t2 * split));
    const bool abf = d->a_dtype == OCC_BF16, bbf = d->b_dtype == OCC_BF16;
    OCC_CHECK_ARG((abf || d->a_dtype == OCC_F32) && (bbf || d->b_dtype == OCC_F32), "occ_gemm_tn: operand dtypes must be f32 or bf16");
    hipStream_t s = (hipStream_t)stream;
    OCC_CHECK_ARG(d->compute == OCC_F32 || d->compute == OCC_BF16, "occ_gemm_tn: compute must be OCC_F32 or OCC_BF16");
    if (d->compute == OCC_BF16) {
        if (abf && bbf) hipLaunchKernelGGL((gemm_tn_bf16_kernel<true, true>), grid, dim3(THREADS), 0, s, a);
        else if (abf) hipLaunchKernelGGL((gemm_tn_bf16_kernel<true, false>), grid, dim3(THREADS), 0, s, a);
        else if (bbf) hipLaunchKernelGGL((gemm_tn_bf16_kernel<false, true>), grid, dim3(THREADS), 0, s, a);
        else hipLaunchKernelGGL((gemm_tn_bf16_kernel<false, false>), grid, dim3(THREADS), 0, s, a);
        OCC_LAUNCH_CHECK("occ_gemm_tn");
        return OCC_OK;
    }
    if (abf && bbf) hipLaunchKernelGGL((gemm_tn_kernel<true, true>), grid, dim3(THREADS), 0, s, a);
    else if (abf) hipLaunchKernelGGL((gemm_tn_kernel<true, false>), grid, dim3(THREADS), 0, s, a);
    else if (bbf) hipLaunchKernelGGL((gemm_tn_kernel<false, true>), grid, dim3(THREADS), 0, s, a);
    else hipLaunchKernelGGL((gemm_tn_kernel<false, false>), grid, dim3(THREADS), 0, s, a);
    OCC_LAUNCH_CHECK("occ_gemm_tn");
    return OCC_OK;
}

int occ_colsum(const void* A, int a_dtype, const occ_rowmap* a_map, int64_t M, int64_t N, float* out, float alpha, void* stream) {
    OCC_CHECK_ARG(A && a_map && out && M >= 1 && N >= 1 && a_map->rows_per_batch >= 1, "occ_colsum: bad argument");
    OCC_CHECK_ARG(a_dtype == OCC_F32 || a_dtype == OCC_BF16, "occ_colsum: A must be f32 or bf16");
    if (a_dtype == OCC_BF16 && N % 8 == 0 && ((uintptr_t)A & 15) == 0 && a_map->row_stride % 8 == 0 && a_map->batch_stride % 8 == 0 && a_map->line_stride % 8 == 0) {
        launch_colsum_bf16_vec((const unsigned short*)A, to_rowmap(*a_map), M, N, out, alpha, (hipStream_t)stream);      // 16-byte loads
        OCC_LAUNCH_CHECK("occ_colsum");
        return OCC_OK;
    }
    long long split = occ_cdiv(M, 512);
    if (split > 256) split = 256;
    const long long rps = occ_cdiv(M, split);
    split = occ_cdiv(M, rps);
    if (a_dtype == OCC_F32)
        hipLaunchKernelGGL(colsum_kernel<float>, dim3((unsigned)occ_cdiv(N, 64), (unsigned)split), dim3(THREADS), 0, (hipStream_t)stream, (const float*)A,
                           to_rowmap(*a_map), (long long)M, (long long)N, rps, out, alpha);
    else
        hipLaunchKernelGGL(colsum_kernel<unsigned short>, dim3((unsigned)occ_cdiv(N, 64), (unsigned)split), dim3(THREADS), 0, (hipStream_t)stream,
                           (const unsigned short*)A, to_rowmap(*a_map), (long long)M, (long long)N, rps, out, alpha);
    OCC_LAUNCH_CHECK("occ_colsum");
    return OCC_OK;
}

}  // extern "C"
