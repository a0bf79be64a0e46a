// Backward kernels of the XLS-R transformer encoder (fine-tuning path): transposing cast for the weight-gradient
// GEMMs, LayerNorm backward, and flash-style attention backward on MFMA.  Reference arithmetic: autograd of fairseq's
// TransformerSentenceEncoderLayer (pre-LN) reached from models/sslassist.py:48 when the optimizer holds the SSL
// parameters (oc_training.py:324).
#include "occ_common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// ------------------------------------------------------------------------------------------------
// dst[c][r] = bf16(src[r][c]); dst rows have ld_dst >= rows elements (pad columns are left untouched: callers keep them 0).
template <typename TS>
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const TS* __restrict__ src, unsigned short* __restrict__ dst, long long rows, long long cols,
                                                            RowMapI smap, long long ld_dst, float* __restrict__ colsum) {
    __shared__ unsigned short tile[64][66];
    __shared__ float csum[4][64];
    float cs = 0.f;
    const long long r0 = (long long)blockIdx.y * 64, c0 = (long long)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const long long r = r0 + ty + 4 * i, c = c0 + tx;
        unsigned short v = 0;
        if (r < rows && c < cols) {
            const long long so = row_off(smap, r) + c;
            if (sizeof(TS) == 4) { const float f = reinterpret_cast<const float*>(src)[so]; cs += f; v = f32_to_bf16_bits(f); }
            else { v = reinterpret_cast<const unsigned short*>(src)[so]; cs += bf16_bits_to_f32(v); }
        }
        tile[ty + 4 * i][tx] = v;
    }
    if (colsum) csum[ty][tx] = cs;
    __syncthreads();
    if (colsum && ty == 0 && c0 + tx < cols) atomicAdd(colsum + c0 + tx, csum[0][tx] + csum[1][tx] + csum[2][tx] + csum[3][tx]);
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const long long c = c0 + ty + 4 * i, r = r0 + tx;
        if (c < cols && r < rows) dst[c * ld_dst + r] = tile[tx][ty + 4 * i];
    }
}

// Many transposes in one launch (the W^T operands of a training step's input-gradient GEMMs: 112 small launches of 9-14 us each were
// latency-bound at 1.5 TB/s).  jobs live in device memory; workgroup b finds its job by first_tile (ascending) and does one 64x64 tile.
__global__ __launch_bounds__(256) void transpose_bf16_batch_kernel(const occ_transpose_job* __restrict__ jobs, int n_jobs) {
    __shared__ unsigned short tile[64][66];
    int lo = 0, hi = n_jobs - 1;
    const long long b = blockIdx.x;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (jobs[mid].first_tile <= b) lo = mid; else hi = mid - 1; }
    const occ_transpose_job j = jobs[lo];
    const long long t = b - j.first_tile, tiles_x = (j.cols + 63) / 64;
    const long long r0 = (t / tiles_x) * 64, c0 = (t % tiles_x) * 64;
    unsigned short* dst = reinterpret_cast<unsigned short*>(j.dst);
    if (j.src_dtype == OCC_BF16 && r0 + 64 <= j.rows && c0 + 64 <= j.cols && ((j.ld_src | j.ld_dst) & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(j.src) | reinterpret_cast<uintptr_t>(j.dst)) & 7) == 0) {
        // whole bf16 tile: 8-byte global accesses both ways (a 2-byte access per lane moves 128 B per wave instruction)
        const unsigned short* src = reinterpret_cast<const unsigned short*>(j.src);
        const int q = threadIdx.x & 15, rr = threadIdx.x >> 4;                 // 16 lanes x 4 elements = one 64-element row segment; 16 rows per pass
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = rr + 16 * i;
            const uint2 v = *reinterpret_cast<const uint2*>(src + (r0 + r) * j.ld_src + c0 + q * 4);
            tile[r][q * 4 + 0] = (unsigned short)(v.x & 0xffff); tile[r][q * 4 + 1] = (unsigned short)(v.x >> 16);
            tile[r][q * 4 + 2] = (unsigned short)(v.y & 0xffff); tile[r][q * 4 + 3] = (unsigned short)(v.y >> 16);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = rr + 16 * i;
            uint2 v;
            v.x = (unsigned)tile[q * 4 + 0][c] | ((unsigned)tile[q * 4 + 1][c] << 16);
            v.y = (unsigned)tile[q * 4 + 2][c] | ((unsigned)tile[q * 4 + 3][c] << 16);
            *reinterpret_cast<uint2*>(dst + (c0 + c) * j.ld_dst + r0 + q * 4) = v;
        }
        return;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const long long r = r0 + ty + 4 * i, c = c0 + tx;
        unsigned short v = 0;
        if (r < j.rows && c < j.cols) {
            const long long so = r * j.ld_src + c;
            v = j.src_dtype == OCC_F32 ? f32_to_bf16_bits(reinterpret_cast<const float*>(j.src)[so]) : reinterpret_cast<const unsigned short*>(j.src)[so];
        }
        tile[ty + 4 * i][tx] = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const long long c = c0 + ty + 4 * i, r = r0 + tx;
        if (c < j.cols && r < j.rows) dst[c * j.ld_dst + r] = tile[tx][ty + 4 * i];
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward over rows of width C (C % 8 == 0, C <= 2048), one wave per row, persistent waves:
//   dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat)) (+ dres);  dgamma += sum dy*xhat;  dbeta += sum dy.
template <typename T> struct LVec8;
template <> struct LVec8<float> {
    struct raw_t { float4 a, b; };
    static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    // raw load now, convert at use: lets the next row's loads fly under the current row's reductions
    static __device__ __forceinline__ raw_t load_raw(const float* p) { return raw_t{*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4)}; }
    static __device__ __forceinline__ raw_t zero_raw() { return raw_t{make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)}; }
    static __device__ __forceinline__ void convert(const raw_t& r, float (&v)[8]) {
        v[0] = r.a.x; v[1] = r.a.y; v[2] = r.a.z; v[3] = r.a.w; v[4] = r.b.x; v[5] = r.b.y; v[6] = r.b.z; v[7] = r.b.w;
    }
};
template <> struct LVec8<unsigned short> {
    typedef uint4 raw_t;
    static __device__ __forceinline__ void load(const unsigned short* p, float (&v)[8]) { convert(*reinterpret_cast<const uint4*>(p), v); }
    static __device__ __forceinline__ raw_t load_raw(const unsigned short* p) { return *reinterpret_cast<const uint4*>(p); }
    static __device__ __forceinline__ raw_t zero_raw() { return make_uint4(0, 0, 0, 0); }
    static __device__ __forceinline__ void convert(const raw_t& a, float (&v)[8]) {
        const unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    }
};

// GELU: dy is the gradient wrt gelu(LN(x)) (the conv blocks of the feature extractor); the kernel rebuilds z = LN(x) and
// multiplies dy by gelu'(z) first.  dx (f32, contiguous) and dx_bf16 (through a row map) are both optional outputs.
template <typename TDY, typename TX, int NIT, bool GELU>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const TDY* __restrict__ dy, const TX* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ dres, float* __restrict__ dx,
                                                           unsigned short* __restrict__ dx_bf16, RowMapI bmap, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, long long rows, int C, float eps) {
    __shared__ float red[3][2][NIT * 512];           // waves 1..3 park their partial dgamma / dbeta here
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long wave0 = (long long)blockIdx.x * 4 + wave, nwaves = (long long)gridDim.x * 4;
    float dg[NIT][8], db[NIT][8], g[NIT][8], bt[NIT][8];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = (it * 64 + lane) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { dg[it][e] = 0.f; db[it][e] = 0.f; g[it][e] = 0.f; bt[it][e] = 0.f; }
        if (c < C) { LVec8<float>::load(gamma + c, g[it]); if (GELU) LVec8<float>::load(beta + c, bt[it]); }
    }
    // The grid is one 4-wave workgroup per CU (more workgroups means more dgamma / dbeta atomics on the same 2C addresses, measured slower),
    // so a wave has no neighbours on its SIMD to hide latency: the NEXT row's x / dy and THIS row's residual gradient are loaded (raw)
    // at the top of an iteration and converted only when used, i.e. they fly under the four dependent wave reductions of the row.
    typename LVec8<TX>::raw_t nx[NIT]; typename LVec8<TDY>::raw_t nd[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = (it * 64 + lane) * 8;
        nx[it] = LVec8<TX>::zero_raw(); nd[it] = LVec8<TDY>::zero_raw();
        if (c < C && wave0 < rows) { nx[it] = LVec8<TX>::load_raw(x + wave0 * C + c); nd[it] = LVec8<TDY>::load_raw(dy + wave0 * C + c); }
    }
    for (long long row = wave0; row < rows; row += nwaves) {
        float xv[NIT][8], dv[NIT][8];
        LVec8<float>::raw_t rr[NIT];
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            LVec8<TX>::convert(nx[it], xv[it]);
            LVec8<TDY>::convert(nd[it], dv[it]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += xv[it][e];
        }
        {
            const long long nrow = row + nwaves;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c = (it * 64 + lane) * 8;
                rr[it] = LVec8<float>::zero_raw();
                if (c < C) {
                    if (dres) rr[it] = LVec8<float>::load_raw(dres + row * C + c);
                    if (nrow < rows) { nx[it] = LVec8<TX>::load_raw(x + nrow * C + c); nd[it] = LVec8<TDY>::load_raw(dy + nrow * C + c); }
                }
            }
        }
        const float mean = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = (it * 64 + lane) * 8;
            if (c < C) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = xv[it][e] - mean; q += d * d; }
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = (xv[it][e] - mean) * rstd;
                if (GELU) dv[it][e] *= gelu_grad(xh * g[it][e] + bt[it][e]);
                const float dxh = dv[it][e] * g[it][e];
                xv[it][e] = xh;                    // keep xhat (zero-gamma padding lanes contribute nothing)
                s1 += dxh; s2 += dxh * xh;
                dg[it][e] += dv[it][e] * xh; db[it][e] += dv[it][e];
            }
        s1 = wave_sum(s1) / (float)C; s2 = wave_sum(s2) / (float)C;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = (it * 64 + lane) * 8;
            if (c < C) {
                float v[8], r[8];
                LVec8<float>::convert(rr[it], r);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[e] = rstd * (dv[it][e] * g[it][e] - s1 - xv[it][e] * s2);
                    if (dres) v[e] += r[e];
                }
                if (dx) {
                    *reinterpret_cast<float4*>(dx + row * C + c) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(dx + row * C + c + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
                if (dx_bf16) {
                    unsigned w[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
                    *reinterpret_cast<uint4*>(dx_bf16 + row_off(bmap, row) + c) = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
        }
    }
    // one set of atomics per workgroup: waves 1..3 hand their partials to wave 0 through LDS
    if (wave > 0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int e = 0; e < 8; ++e) { red[wave - 1][0][(it * 64 + lane) * 8 + e] = dg[it][e]; red[wave - 1][1][(it * 64 + lane) * 8 + e] = db[it][e]; }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = (it * 64 + lane) * 8;
            if (c < C) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float a = dg[it][e], bb = db[it][e];
#pragma unroll
                    for (int w2 = 0; w2 < 3; ++w2) { a += red[w2][0][c + e]; bb += red[w2][1][c + e]; }
                    atomicAdd(dgamma + c + e, a); atomicAdd(dbeta + c + e, bb);
                }
            }
        }
    }
}

// Sixteen-wave form of the kernel above (4 waves per SIMD on one workgroup per CU).  The row loop is a chain of four dependent wave
// reductions, so a wave alone on its SIMD runs at that latency (measured: 94 us for 12736 x 1024 = 2.2 TB/s); more waves per SIMD hide
// it, but more WORKGROUPS would multiply the dgamma / dbeta atomics on the same 2C addresses.  Here the per-column sums live in LDS
// (ds_add_f32 from all 16 waves, lane-contiguous layout: no bank conflicts) instead of 32 registers per lane, which is what lets
// the wave fit 128 registers; the workgroup still issues one set of global atomics at the end.
// WAVES: 16 (128 registers per lane) or 12 (168: room for the EXTRA state without spilling).  EXTRA: (1) the column sums of the
// OUTPUT dx as a third partial set -- the bias gradient of the Linear whose output gradient this is (out-proj / fc2 of a transformer
// layer: x = residual + Linear(.)), which was a separate pass over the bf16 copy; (2) an fp8 (e5m2) copy of the bf16-rounded dx
// with the site's delayed scale, recording |max| for the next step's scale (the fp8 input-gradient GEMM's operand, else its own pass).
struct LnbExtra { unsigned char* f8; const float* f8_scale; float* f8_amax; int want_bias; };
template <typename TDY, typename TX, int NIT, bool GELU, int WAVES = 16, bool EXTRA = false>
__global__ __launch_bounds__(64 * WAVES) void layernorm_bwd16_kernel(const TDY* __restrict__ dy, const TX* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ dres, float* __restrict__ dx,
                                                              unsigned short* __restrict__ dx_bf16, RowMapI bmap, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, long long rows, int C, float eps, float* __restrict__ partials,
                                                              const LnbExtra ex) {
    constexpr int NT = 64 * WAVES;
    __shared__ float accg[8 * NIT * 64], accb[8 * NIT * 64];     // [element e of the lane's 8][it*64 + lane]
    __shared__ float accs[EXTRA ? 8 * NIT * 64 : 4];
    __shared__ float amx[EXTRA ? WAVES : 1];
    __shared__ __attribute__((aligned(16))) float gs[NIT * 512], bs[GELU ? NIT * 512 : 4];   // gamma / beta: read from LDS at each use (register budget)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8 * NIT * 64; i += NT) {
        accg[i] = 0.f; accb[i] = 0.f;
        if (EXTRA) accs[i] = 0.f;
        gs[i] = i < C ? gamma[i] : 0.f;
        if (GELU) bs[i] = i < C ? beta[i] : 0.f;
    }
    __syncthreads();
    const long long wave0 = (long long)blockIdx.x * WAVES + wave, nwaves = (long long)gridDim.x * WAVES;
    float dg[NIT][8], db[NIT][8];                  // per-lane column sums over this wave's rows (LDS float atomics per row measured 2.5x slower)
    float dsum[EXTRA ? NIT : 1][8];
    float f8max = 0.f;
    const float f8sc = EXTRA && ex.f8 && ex.f8_scale ? *ex.f8_scale : 1.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int e = 0; e < 8; ++e) { dg[it][e] = 0.f; db[it][e] = 0.f; if (EXTRA) dsum[it][e] = 0.f; }
    // no next-row prefetch here: with four waves per SIMD the other waves cover the load latency, and the 24 registers decide whether
    // the kernel fits the 128-register budget of a 16-wave workgroup
    for (long long row = wave0; row < rows; row += nwaves) {
        float xv[NIT][8], dv[NIT][8];
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = (it * 64 + lane) * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) { xv[it][e] = 0.f; dv[it][e] = 0.f; }
            if (c < C) { LVec8<TX>::load(x + row * C + c, xv[it]); LVec8<TDY>::load(dy + row * C + c, dv[it]); }
#pragma unroll
            for (int e = 0; e < 8; ++e) s += xv[it][e];
        }
        const float mean = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = (it * 64 + lane) * 8;
            if (c < C) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = xv[it][e] - mean; q += d * d; }
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = (xv[it][e] - mean) * rstd;
                const float ge = gs[(it * 64 + lane) * 8 + e];
                if (GELU) dv[it][e] *= gelu_grad(xh * ge + bs[GELU ? (it * 64 + lane) * 8 + e : 0]);
                const float dxh = dv[it][e] * ge;
                xv[it][e] = xh;
                s1 += dxh; s2 += dxh * xh;
                dg[it][e] += dv[it][e] * xh; db[it][e] += dv[it][e];
            }
        s1 = wave_sum(s1) / (float)C; s2 = wave_sum(s2) / (float)C;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = (it * 64 + lane) * 8;
            if (c < C) {
                float v[8], r[8];
                if (dres) LVec8<float>::load(dres + row * C + c, r);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[e] = rstd * (dv[it][e] * gs[c + e] - s1 - xv[it][e] * s2);
                    if (dres) v[e] += r[e];
                }
                if (dx) {
                    *reinterpret_cast<float4*>(dx + row * C + c) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(dx + row * C + c + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
                if (EXTRA && ex.want_bias) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) dsum[it][e] += v[e];
                }
                if (dx_bf16) {
                    unsigned w[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
                    *reinterpret_cast<uint4*>(dx_bf16 + row_off(bmap, row) + c) = make_uint4(w[0], w[1], w[2], w[3]);
                    if (EXTRA && ex.f8) {           // e5m2 of the bf16-rounded values (what the stand-alone quantisation pass reads), saturating
                        float qv[8];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float a = __uint_as_float(w[i] << 16), b = __uint_as_float(w[i] & 0xffff0000u);
                            f8max = fmaxf(f8max, fmaxf(fabsf(a), fabsf(b)));
                            qv[2 * i] = fminf(fmaxf(a * f8sc, -57344.f), 57344.f); qv[2 * i + 1] = fminf(fmaxf(b * f8sc, -57344.f), 57344.f);
                        }
                        int q0 = __builtin_amdgcn_cvt_pk_bf8_f32(qv[0], qv[1], 0, false), q1 = __builtin_amdgcn_cvt_pk_bf8_f32(qv[4], qv[5], 0, false);
                        q0 = __builtin_amdgcn_cvt_pk_bf8_f32(qv[2], qv[3], q0, true); q1 = __builtin_amdgcn_cvt_pk_bf8_f32(qv[6], qv[7], q1, true);
                        *reinterpret_cast<uint2*>(ex.f8 + row * C + c) = make_uint2((unsigned)q0, (unsigned)q1);
                    }
                }
            }
        }
    }
    // the waves add their sums into the LDS arrays one after the other (fixed order), then one set of global atomics / partial stores
    for (int w = 0; w < WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int it = 0; it < NIT; ++it)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    accg[e * (NIT * 64) + it * 64 + lane] += dg[it][e]; accb[e * (NIT * 64) + it * 64 + lane] += db[it][e];
                    if (EXTRA) accs[e * (NIT * 64) + it * 64 + lane] += dsum[it][e];
                }
        }
        __syncthreads();
    }
    if (EXTRA && ex.f8 && ex.f8_amax) {            // one guarded atomic per workgroup (non-negative floats order as their bit patterns)
        f8max = wave_max(f8max);
        if (lane == 0) amx[wave] = f8max;
        __syncthreads();
        if (threadIdx.x == 0) {
            float m = amx[0];
            for (int w = 1; w < WAVES; ++w) m = fmaxf(m, amx[w]);
            if (m > __hip_atomic_load(ex.f8_amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(reinterpret_cast<unsigned*>(ex.f8_amax), __float_as_uint(m));
        }
    }
    constexpr int NS = EXTRA ? 3 : 2;              // partial sets per block: dgamma, dbeta (, bias)
    for (int i = threadIdx.x; i < 8 * NIT * 64; i += NT) {
        const int e = i / (NIT * 64), l = i - e * (NIT * 64), c = l * 8 + e;
        if (c < C) {
            // 256 workgroups adding to the same 2C addresses cost ~25 us of a 77 us launch: with caller scratch the sums go out as
            // plain stores [block][sets][C] and layernorm_bwd_finalize_kernel adds them up in block order
            if (partials) {
                partials[((long long)blockIdx.x * NS) * C + c] = accg[i]; partials[((long long)blockIdx.x * NS + 1) * C + c] = accb[i];
                if (EXTRA) partials[((long long)blockIdx.x * NS + 2) * C + c] = accs[i];
            }
            else { atomicAdd(dgamma + c, accg[i]); atomicAdd(dbeta + c, accb[i]); }
        }
    }
}

// 64 columns per workgroup; the four waves take every fourth block (independent loads, eight in flight), wave 0 adds the four sums
__global__ __launch_bounds__(256) void layernorm_bwd_finalize_kernel(const float* __restrict__ partials, int nblocks, int C, float* __restrict__ dgamma,
                                                                    float* __restrict__ dbeta, float* __restrict__ dbias, int nsets) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;                  // [nsets][C]
    const int tot = nsets * C;
    float s = 0.f;
    if (i < tot) {
        int b = wave;
        for (; b + 28 < nblocks; b += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partials[(long long)(b + 4 * u) * tot + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < nblocks; b += 4) s += partials[(long long)b * tot + i];
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && i < tot) {
        const float t = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
        if (i < C) dgamma[i] += t; else if (i < 2 * C) dbeta[i - C] += t; else if (dbias) dbias[i - 2 * C] += t;
    }
}

// y = residual + scale * x * keep / (1 - p): fairseq's train-mode dropouts inside wav2vec2 (dropout_input, the encoder's and the
// layers' dropout / activation_dropout) and their backward (the same call with the stored mask on the gradient).  mask NULL = no
// dropout (plain scale / add); gen != 0 draws the keep-mask (Philox 4x32-10, one counter per 4 elements, as occ_dropout) and stores it.
__device__ __forceinline__ void fb_philox_r(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
template <typename TX, typename TY>
__global__ void dropout_ex_kernel(const TX* __restrict__ x, TY* __restrict__ y, unsigned char* __restrict__ mask, const float* __restrict__ residual,
                                  long long n, float p, float scale, uint64_t seed, uint64_t sid, int gen) {
    const float sc = scale / (1.0f - p);
    const long long nq = (n + 3) / 4;
    for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < nq; q += (long long)gridDim.x * blockDim.x) {
        uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)sid, (uint32_t)(sid >> 32)};
        if (gen && mask) {
            uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
            for (int r = 0; r < 10; ++r) { fb_philox_r(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const long long i = q * 4 + h;
            if (i < n) {
                unsigned char m = 1;
                if (mask) { if (gen) { m = ((float)c[h] * (1.0f / 4294967296.0f)) >= p ? 1 : 0; mask[i] = m; } else m = mask[i]; }
                float v = m ? occ_load_f32(x + i) * sc : 0.f;
                if (residual) v += residual[i];
                occ_store_f32(y + i, v);
            }
        }
    }
}

// The keep-mask alone, bit-identical to what dropout_ex_kernel(gen = 1) draws for the same (n, p, seed, sid): the attention-dropout
// mask [B*H, T, Tp] is consumed inside the attention kernels, there is no tensor to apply it to here.
__global__ void dropout_mask_kernel(unsigned char* __restrict__ mask, long long n, float p, uint64_t seed, uint64_t sid) {
    const long long nq = (n + 3) / 4;
    for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < nq; q += (long long)gridDim.x * blockDim.x) {
        uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)sid, (uint32_t)(sid >> 32)};
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) { fb_philox_r(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
        unsigned w = 0;
#pragma unroll
        for (int h = 0; h < 4; ++h) w |= (((float)c[h] * (1.0f / 4294967296.0f)) >= p ? 1u : 0u) << (8 * h);
        if (q * 4 + 3 < n && ((uintptr_t)mask & 3) == 0) *reinterpret_cast<unsigned*>(mask + q * 4) = w;
        else for (int h = 0; h < 4; ++h) if (q * 4 + h < n) mask[q * 4 + h] = (unsigned char)((w >> (8 * h)) & 1u);
    }
}

// out[omap(r)][c] = bf16(dy[r][c] * gelu'(u[r][c]))   (gradient through the positional conv's GELU, written into the padded buffer)
__global__ void gelu_bwd_rows_kernel(const float* __restrict__ dy, const unsigned short* __restrict__ u, unsigned short* __restrict__ out, RowMapI omap,
                                     long long rows, int C) {
    const long long n = rows * (C / 4);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / (C / 4); const int c = (int)(i - r * (C / 4)) * 4;
        const float4 d = *reinterpret_cast<const float4*>(dy + r * C + c);
        const uint2 uu = *reinterpret_cast<const uint2*>(u + r * C + c);
        const float v0 = d.x * gelu_grad(bf16_bits_to_f32((unsigned short)(uu.x & 0xffff))), v1 = d.y * gelu_grad(bf16_bits_to_f32((unsigned short)(uu.x >> 16)));
        const float v2 = d.z * gelu_grad(bf16_bits_to_f32((unsigned short)(uu.y & 0xffff))), v3 = d.w * gelu_grad(bf16_bits_to_f32((unsigned short)(uu.y >> 16)));
        uint2 o;
        o.x = pack_bf16x2(v0, v1);
        o.y = pack_bf16x2(v2, v3);
        *reinterpret_cast<uint2*>(out + row_off(omap, r) + c) = o;
    }
}

// Backward of the fused first conv block (Conv1d(1->512,k,stride) + LayerNorm + GELU): everything is recomputed from the
// waveform, nothing was saved.  One wave per frame, lane = 8 channels; weight / bias / LayerNorm gradients are reduced in
// registers over the wave's frames, then over the 4 waves through LDS, then one set of atomics per workgroup.
constexpr int C0B_FRAMES = 256, C0B_MAXK = 16;   // C0B_MAXK: largest supported tap count
// KT = compile-time tap count held in registers (weights + their gradients: 16 * KT VGPRs per lane); KT = 10 is XLS-R's first conv.
template <typename TD, int KT>
__global__ __launch_bounds__(256, KT <= 10 ? 2 : 1) void conv0_bwd_kernel(const float* __restrict__ wav, const float* __restrict__ w, const float* __restrict__ bias,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, const TD* __restrict__ dact,
                                                       float* __restrict__ dw, float* __restrict__ dbias, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                       int L, int Tout, int k, int stride, float eps) {
    extern __shared__ __attribute__((aligned(16))) float smp[];          // samples | reduction scratch [512*(k+3)]
    const int b = blockIdx.y, f0 = blockIdx.x * C0B_FRAMES;
    const int nsamp = (C0B_FRAMES - 1) * stride + k;
    float* red = smp + ((nsamp + 3) & ~3);
    const float* wb = wav + (size_t)b * L;
    for (int i = threadIdx.x; i < nsamp; i += 256) { const int g = f0 * stride + i; smp[i] = g < L ? wb[g] : 0.f; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = lane * 8;
    // conv bias / LayerNorm gamma / beta of the lane's channels are read from LDS at each use: 24 registers that decide whether two waves
    // fit a SIMD without spilling (wr + gw alone are 160)
    __shared__ float cst[3][512];
    for (int i = threadIdx.x; i < 512; i += 256) { cst[0][i] = bias[i]; cst[1][i] = gamma[i]; cst[2][i] = beta[i]; }
    const float* br = cst[0] + c0; const float* gr = cst[1] + c0; const float* ber = cst[2] + c0;
    float wr[8][KT];
    float gw[8][KT], gb[8], gg[8], gbe[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        gb[e] = 0.f; gg[e] = 0.f; gbe[e] = 0.f;
#pragma unroll
        for (int t = 0; t < KT; ++t) { wr[e][t] = t < k ? w[(size_t)(c0 + e) * k + t] : 0.f; gw[e][t] = 0.f; }
    }
    __syncthreads();
    for (int fi = 0; fi < C0B_FRAMES / 4; ++fi) {
        const int fl = wave * (C0B_FRAMES / 4) + fi, f = f0 + fl;
        if (f >= Tout) break;
        float acc[8], xs[KT];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = br[e];
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            xs[t] = t < k ? smp[fl * stride + t] : 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(wr[e][t], xs[t], acc[e]);
        }
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += acc[e];
        const float mean = wave_sum(s) * (1.0f / 512.0f);
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = acc[e] - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / 512.0f) + eps);
        float dv[8];
        LVec8<TD>::load(dact + ((size_t)b * Tout + f) * 512 + c0, dv);
        float xh[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            xh[e] = (acc[e] - mean) * rstd;
            const float dz = dv[e] * gelu_grad(xh[e] * gr[e] + ber[e]);
            gg[e] += dz * xh[e]; gbe[e] += dz;
            dv[e] = dz * gr[e];
            s1 += dv[e]; s2 += dv[e] * xh[e];
        }
        s1 = wave_sum(s1) * (1.0f / 512.0f); s2 = wave_sum(s2) * (1.0f / 512.0f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float dp = rstd * (dv[e] - s1 - xh[e] * s2);
            gb[e] += dp;
#pragma unroll
            for (int t = 0; t < KT; ++t) gw[e][t] = fmaf(dp, xs[t], gw[e][t]);
        }
    }
    // reduce over the 4 waves through ONE [512][k+3] = (dw[k], dbias, dgamma, dbeta) scratch, wave after wave (three separate images made
    // the workgroup 85 KB of LDS: one workgroup = one wave per SIMD on a kernel that is a chain of dependent wave reductions)
    const int rs = k + 3;
    for (int w2 = 1; w2 < 4; ++w2) {
        __syncthreads();
        if (wave == w2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float* r = red + (c0 + e) * rs;
#pragma unroll
                for (int t = 0; t < KT; ++t) if (t < k) r[t] = gw[e][t];      // static index: a run-time index would push gw to scratch
                r[k] = gb[e]; r[k + 1] = gg[e]; r[k + 2] = gbe[e];
            }
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float* r = red + (c0 + e) * rs;
#pragma unroll
                for (int t = 0; t < KT; ++t) if (t < k) gw[e][t] += r[t];
                gb[e] += r[k]; gg[e] += r[k + 1]; gbe[e] += r[k + 2];
            }
        }
    }
    if (wave == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + e;
#pragma unroll
            for (int t = 0; t < KT; ++t) if (t < k) atomicAdd(dw + (size_t)c * k + t, gw[e][t]);
            atomicAdd(dbias + c, gb[e]); atomicAdd(dgamma + c, gg[e]); atomicAdd(dbeta + c, gbe[e]);
        }
    }
}

// weight_norm(dim=2) of the positional conv: w[o,i,k] = g[k] * v[o,i,k] / ||v[:,:,k]||.  One workgroup per tap k.
// pack: writes the forward GEMM operand  wf[G][n][k][c]  and the input-gradient operand  wb[G][c][j][n] = w[(G,n), c, K-1-j]  (bf16).
__global__ __launch_bounds__(256) void weight_norm_pack_kernel(const float* __restrict__ v, const float* __restrict__ g, unsigned short* __restrict__ wf,
                                                              unsigned short* __restrict__ wb, float* __restrict__ norms, int O, int I, int K, int G) {
    __shared__ float red[4];
    const int k = blockIdx.x, cgn = O / G;
    float s = 0.f;
    for (int idx = threadIdx.x; idx < O * I; idx += 256) { const float x = v[(size_t)idx * K + k]; s += x * x; }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]);
    if (threadIdx.x == 0) norms[k] = nrm;
    const float sc = g[k] / nrm;
    for (int idx = threadIdx.x; idx < O * I; idx += 256) {
        const int o = idx / I, i = idx - o * I, gi = o / cgn, n = o - gi * cgn;
        const unsigned short val = f32_to_bf16_bits(v[(size_t)idx * K + k] * sc);
        wf[(((size_t)gi * cgn + n) * K + k) * I + i] = val;
        if (wb) wb[(((size_t)gi * I + i) * K + (K - 1 - k)) * cgn + n] = val;
    }
}
// bwd: dWp[G][n][k][c] (f32, GEMM layout) -> dg[k] += sum dW*v/||v_k|| ; dv[o,i,k] += g/||v|| * (dW - v * sum(dW*v)/||v||^2)
__global__ __launch_bounds__(256) void weight_norm_bwd_kernel(const float* __restrict__ v, const float* __restrict__ g, const float* __restrict__ norms,
                                                             const float* __restrict__ dwp, float* __restrict__ dv, float* __restrict__ dg, int O, int I,
                                                             int K, int G) {
    __shared__ float red[4];
    const int k = blockIdx.x, cgn = O / G;
    float s = 0.f;
    for (int idx = threadIdx.x; idx < O * I; idx += 256) {
        const int o = idx / I, i = idx - o * I, gi = o / cgn, n = o - gi * cgn;
        s += dwp[(((size_t)gi * cgn + n) * K + k) * I + i] * v[(size_t)idx * K + k];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float dot = red[0] + red[1] + red[2] + red[3];
    const float nrm = norms[k], gk = g[k];
    if (threadIdx.x == 0) dg[k] += dot / nrm;
    for (int idx = threadIdx.x; idx < O * I; idx += 256) {
        const int o = idx / I, i = idx - o * I, gi = o / cgn, n = o - gi * cgn;
        const float d = dwp[(((size_t)gi * cgn + n) * K + k) * I + i];
        dv[(size_t)idx * K + k] += gk / nrm * (d - v[(size_t)idx * K + k] * dot / (nrm * nrm));
    }
}


// ---- coalesced forms of the two kernels above (taken when the caller passes scratch and 256 % K == 0) ---------------------------------
// v [O*I rows][K] has the taps innermost, the GEMM layouts have them outermost: one workgroup per tap reads every element with a
// stride of K floats (4 of every 512 bytes fetched: 331 us / 547 us for the 8.4 M weights of XLS-R's positional conv).  Here every
// global access is a contiguous row segment and the (i,k) <-> (k,i) / (n,k) <-> (k,n) transposes happen in LDS (row pitch + 1).
__global__ __launch_bounds__(256) void wn_sumsq_kernel(const float* __restrict__ v, float* __restrict__ partial, long long rows, int K) {
    __shared__ float red[256];
    const int k = threadIdx.x % K, lr = threadIdx.x / K, rpp = 256 / K;
    const long long per = (rows + gridDim.x - 1) / gridDim.x, r0 = blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
    float s = 0.f;
    for (long long r = r0 + lr; r < r1; r += rpp) { const float x = v[r * K + k]; s += x * x; }
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < K) {
        float t = 0.f;
        for (int j = 0; j < rpp; ++j) t += red[j * K + threadIdx.x];
        partial[(long long)blockIdx.x * K + threadIdx.x] = t;
    }
}
// one workgroup per output channel o: wf[o][k][i] = bf16(v[o][i][k] * g[k] / ||v_k||)
// norms[k] = sqrt(sum of the S partial rows), in row order
__global__ __launch_bounds__(256) void wn_norms_kernel(const float* __restrict__ partial, int S, float* __restrict__ norms, int K) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    float t = 0.f;
    int j = 0;
    for (; j + 32 <= S; j += 32) {                  // 32 independent loads in flight, added in row order (one latency per 32 rows, not per row)
        float v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = partial[(long long)(j + u) * K + k];
#pragma unroll
        for (int u = 0; u < 32; ++u) t += v[u];
    }
    for (; j < S; ++j) t += partial[(long long)j * K + k];
    norms[k] = sqrtf(t);
}
__global__ __launch_bounds__(256) void wn_pack_fwd_kernel(const float* __restrict__ v, const float* __restrict__ g, const float* __restrict__ norms,
                                                         unsigned short* __restrict__ wf, int I, int K) {
    extern __shared__ float wn_lds[];
    float* sc = wn_lds;                 // [K]
    float* tile = wn_lds + K;           // [I][K + 1]
    const long long o = blockIdx.x;
    for (int k = threadIdx.x; k < K; k += 256) sc[k] = g[k] / norms[k];
    for (int idx = threadIdx.x; idx < I * K; idx += 256) { const int i = idx / K, k = idx - i * K; tile[i * (K + 1) + k] = v[(o * I + i) * K + k]; }
    __syncthreads();
    for (int idx = threadIdx.x; idx < I * K; idx += 256) {
        const int k = idx / I, i = idx - k * I;
        wf[(o * K + k) * I + i] = f32_to_bf16_bits(tile[i * (K + 1) + k] * sc[k]);
    }
}
// one workgroup per (group gi, input channel i): wb[gi][i][K-1-k][n] = bf16(w[(gi,n)][i][k])
__global__ __launch_bounds__(256) void wn_pack_bwd_kernel(const float* __restrict__ v, const float* __restrict__ g, const float* __restrict__ norms,
                                                         unsigned short* __restrict__ wb, int I, int K, int cgn) {
    extern __shared__ float wn_lds[];
    float* sc = wn_lds;                 // [K]
    float* tile = wn_lds + K;           // [cgn][K + 1]
    const long long gi = blockIdx.x / I, i = blockIdx.x - gi * I;
    for (int k = threadIdx.x; k < K; k += 256) sc[k] = g[k] / norms[k];
    __syncthreads();
    for (int idx = threadIdx.x; idx < cgn * K; idx += 256) {
        const int n = idx / K, k = idx - n * K;
        tile[n * (K + 1) + k] = v[((gi * cgn + n) * I + i) * K + k] * sc[k];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < K * cgn; idx += 256) {
        const int j = idx / cgn, n = idx - j * cgn;
        wb[((gi * I + i) * K + j) * cgn + n] = f32_to_bf16_bits(tile[n * (K + 1) + (K - 1 - j)]);
    }
}
// backward: per output channel o, partial[o][k] = sum_i dWp[o][k][i] * v[o][i][k]
__global__ __launch_bounds__(256) void wn_bwd_dot_kernel(const float* __restrict__ v, const float* __restrict__ dwp, float* __restrict__ partial, int I, int K) {
    extern __shared__ float wn_lds[];
    float* tile = wn_lds;               // [K][I + 1]
    float* red = wn_lds + K * (I + 1);  // [256]
    const long long o = blockIdx.x;
    for (int idx = threadIdx.x; idx < K * I; idx += 256) { const int k = idx / I, i = idx - k * I; tile[k * (I + 1) + i] = dwp[(o * K + k) * I + i]; }
    __syncthreads();
    const int k = threadIdx.x % K, lr = threadIdx.x / K, rpp = 256 / K;
    float s = 0.f;
    for (int i = lr; i < I; i += rpp) s += tile[k * (I + 1) + i] * v[(o * I + i) * K + k];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < K) {
        float t = 0.f;
        for (int j = 0; j < rpp; ++j) t += red[j * K + threadIdx.x];
        partial[o * K + threadIdx.x] = t;
    }
}
// dot[k] = sum_o partial[o][k] (fixed order); dg[k] += dot / ||v_k||
__global__ __launch_bounds__(256) void wn_bwd_finalize_kernel(const float* __restrict__ partial, const float* __restrict__ norms, float* __restrict__ dot,
                                                             float* __restrict__ dg, int O, int K) {
    __shared__ float red[4];
    const int k = blockIdx.x;
    float s = 0.f;
    for (int o = threadIdx.x; o < O; o += 256) s += partial[(long long)o * K + k];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { const float d = red[0] + red[1] + red[2] + red[3]; dot[k] = d; dg[k] += d / norms[k]; }
}
// dv[o][i][k] += g/||v|| * (dWp[o][k][i] - v * dot / ||v||^2)
__global__ __launch_bounds__(256) void wn_bwd_apply_kernel(const float* __restrict__ v, const float* __restrict__ g, const float* __restrict__ norms,
                                                          const float* __restrict__ dwp, const float* __restrict__ dot, float* __restrict__ dv, int I, int K) {
    extern __shared__ float wn_lds[];
    float* tile = wn_lds;               // [K][I + 1]
    float* a = wn_lds + K * (I + 1);    // [K] g / nrm
    float* b = a + K;                   // [K] dot / nrm^2
    const long long o = blockIdx.x;
    for (int k = threadIdx.x; k < K; k += 256) { const float nrm = norms[k]; a[k] = g[k] / nrm; b[k] = dot[k] / (nrm * nrm); }
    for (int idx = threadIdx.x; idx < K * I; idx += 256) { const int k = idx / I, i = idx - k * I; tile[k * (I + 1) + i] = dwp[(o * K + k) * I + i]; }
    __syncthreads();
    for (int idx = threadIdx.x; idx < I * K; idx += 256) {
        const int i = idx / K, k = idx - i * K;
        const long long at = (o * I + i) * K + k;
        dv[at] += a[k] * (tile[k * (I + 1) + i] - v[at] * b[k]);
    }
}

}  // namespace

int occ_conv0_bwd_mfma_launch(const float* wav, const float* w, const float* bias, const float* gamma, const float* beta, const void* dact, int dact_bf16,
                              float* dw, float* dbias, float* dgamma, float* dbeta, long long B, long long L, long long Tout, long long stride, float eps,
                              hipStream_t s);     // conv0_mfma.hip

extern "C" {

int occ_transpose_bf16_rows(const void* src, int src_dtype, const occ_rowmap* src_map, void* dst, int64_t rows, int64_t cols, int64_t ld_dst,
                            float* colsum, void* stream) {
    OCC_CHECK_ARG(src && dst && src_map && rows >= 1 && cols >= 1 && ld_dst >= rows && src_map->rows_per_batch >= 1, "occ_transpose_bf16: bad argument");
    OCC_CHECK_ARG(src_dtype == OCC_F32 || src_dtype == OCC_BF16, "occ_transpose_bf16: source must be f32 or bf16");
    const dim3 grid((unsigned)occ_cdiv(cols, 64), (unsigned)occ_cdiv(rows, 64)), block(256);
    OCC_CHECK_ARG(grid.y < 65536, "occ_transpose_bf16: too many rows");
    hipStream_t s = (hipStream_t)stream;
    const RowMapI sm = to_rowmap(*src_map);
    if (src_dtype == OCC_F32) hipLaunchKernelGGL(transpose_bf16_kernel<float>, grid, block, 0, s, (const float*)src, (unsigned short*)dst, (long long)rows, (long long)cols, sm, (long long)ld_dst, colsum);
    else hipLaunchKernelGGL(transpose_bf16_kernel<unsigned short>, grid, block, 0, s, (const unsigned short*)src, (unsigned short*)dst, (long long)rows, (long long)cols, sm, (long long)ld_dst, colsum);
    OCC_LAUNCH_CHECK("occ_transpose_bf16");
    return OCC_OK;
}

int occ_transpose_bf16_batch(const occ_transpose_job* jobs_dev, int64_t n_jobs, int64_t total_tiles, void* stream) {
    OCC_CHECK_ARG(jobs_dev && n_jobs >= 1 && n_jobs < (1 << 20) && total_tiles >= 1 && total_tiles < (1ll << 31), "occ_transpose_bf16_batch: bad argument");
    hipLaunchKernelGGL(transpose_bf16_batch_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, jobs_dev, (int)n_jobs);
    OCC_LAUNCH_CHECK("occ_transpose_bf16_batch");
    return OCC_OK;
}

int occ_transpose_bf16(const void* src, int src_dtype, void* dst, int64_t rows, int64_t cols, int64_t ld_src, int64_t ld_dst, float* colsum,
                       void* stream) {
    OCC_CHECK_ARG(ld_src >= cols, "occ_transpose_bf16: ld_src < cols");
    const occ_rowmap m{rows, 0, ld_src, 0, 0};
    return occ_transpose_bf16_rows(src, src_dtype, &m, dst, rows, cols, ld_dst, colsum, stream);
}

static int lnb_impl(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* gamma, const float* beta, const float* dres, float* dx,
                    void* dx_bf16, const occ_rowmap* dx_bf16_map, float* dgamma, float* dbeta, int64_t rows, int64_t C, float eps, int gelu,
                    float* scratch, int64_t scratch_floats, void* stream, float* dbias, void* dx_f8, const float* f8_scale, float* f8_amax);

int occ_layernorm_bwd_ex(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* gamma, const float* beta, const float* dres, float* dx,
                         void* dx_bf16, const occ_rowmap* dx_bf16_map, float* dgamma, float* dbeta, int64_t rows, int64_t C, float eps, int gelu,
                         float* scratch, int64_t scratch_floats, void* stream) {
    return lnb_impl(dy, dy_dtype, x, x_dtype, gamma, beta, dres, dx, dx_bf16, dx_bf16_map, dgamma, dbeta, rows, C, eps, gelu, scratch, scratch_floats, stream,
                    nullptr, nullptr, nullptr, nullptr);
}

static thread_local int g_lnb_defer = 0;        // occ_layernorm_bwd_fused(defer = 1): the partial sums stay in the caller's scratch (occ_finalize_batch adds them)

int occ_layernorm_bwd_fused(const void* dy, int dy_dtype, const float* x, const float* gamma, const float* dres, float* dx, void* dx_bf16, float* dgamma,
                            float* dbeta, float* dbias, void* dx_f8, const float* f8_scale, float* f8_amax, int64_t rows, int64_t C, float eps,
                            float* scratch, int64_t scratch_floats, int defer, void* stream) {
    g_lnb_defer = defer;
    OCC_CHECK_ARG(dbias || dx_f8, "occ_layernorm_bwd_fused: nothing to fuse (use occ_layernorm_bwd)");
    OCC_CHECK_ARG(!dx_f8 || (dx_bf16 && f8_scale), "occ_layernorm_bwd_fused: the fp8 copy is made of the bf16 output and needs its scale");
    OCC_CHECK_ARG(rows >= 2048 && C <= 1536 && scratch && scratch_floats >= 256 * 3 * C && ((uintptr_t)scratch & 15) == 0,
                  "occ_layernorm_bwd_fused: needs rows >= 2048, C <= 1536 and 768*C floats of 16-byte aligned scratch");
    const int rc = lnb_impl(dy, dy_dtype, x, OCC_F32, gamma, nullptr, dres, dx, dx_bf16, nullptr, dgamma, dbeta, rows, C, eps, 0, scratch, scratch_floats, stream,
                            dbias, dx_f8, f8_scale, f8_amax);
    g_lnb_defer = 0;
    return rc;
}

static int lnb_impl(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* gamma, const float* beta, const float* dres, float* dx,
                    void* dx_bf16, const occ_rowmap* dx_bf16_map, float* dgamma, float* dbeta, int64_t rows, int64_t C, float eps, int gelu,
                    float* scratch, int64_t scratch_floats, void* stream, float* dbias, void* dx_f8, const float* f8_scale, float* f8_amax) {
    OCC_CHECK_ARG(dy && x && gamma && (dx || dx_bf16) && dgamma && dbeta, "occ_layernorm_bwd: null pointer");
    OCC_CHECK_ARG(!gelu || beta, "occ_layernorm_bwd: the fused GELU needs beta");
    OCC_CHECK_ARG(rows >= 1 && C >= 8 && C % 8 == 0 && C <= 2048, "occ_layernorm_bwd: C must be a multiple of 8 in [8,2048]");
    OCC_CHECK_ARG((dy_dtype == OCC_F32 || dy_dtype == OCC_BF16) && (x_dtype == OCC_F32 || x_dtype == OCC_BF16), "occ_layernorm_bwd: dy / x must be f32 or bf16");
    long long blocks = occ_cdiv(rows, 4 * 8);           // >= 8 rows per wave so the per-workgroup atomics amortise
    static const long long lnb_blocks = getenv("OCC_LNB_BLOCKS") ? atoll(getenv("OCC_LNB_BLOCKS")) : 256;
    if (blocks > lnb_blocks) blocks = lnb_blocks;
    if (blocks < 1) blocks = 1;
    const int nit = (int)((C + 511) / 512);
    OCC_CHECK_ARG(nit <= 2 || !gelu, "occ_layernorm_bwd: fused GELU supports C <= 1024");
    hipStream_t s = (hipStream_t)stream;
    RowMapI bm = occ_make_rowmap(rows, 0, C, 0, 0);
    if (dx_bf16_map) { OCC_CHECK_ARG(dx_bf16_map->rows_per_batch >= 1 && dx_bf16_map->row_stride % 8 == 0 && dx_bf16_map->batch_stride % 8 == 0, "occ_layernorm_bwd: bad bf16 row map"); bm = to_rowmap(*dx_bf16_map); }
#define OCC_LNB(TD, TXX, N, G) hipLaunchKernelGGL((layernorm_bwd_kernel<TD, TXX, N, G>), dim3((unsigned)blocks), dim3(256), 0, s, (const TD*)dy, (const TXX*)x, gamma, beta, dres, dx, (unsigned short*)dx_bf16, bm, dgamma, dbeta, (long long)rows, (int)C, eps)
#define OCC_LNB16(TD, TXX, N, G) hipLaunchKernelGGL((layernorm_bwd16_kernel<TD, TXX, N, G>), dim3((unsigned)blocks16), dim3(1024), 0, s, (const TD*)dy, (const TXX*)x, gamma, beta, dres, dx, (unsigned short*)dx_bf16, bm, dgamma, dbeta, (long long)rows, (int)C, eps, part, LnbExtra{nullptr, nullptr, nullptr, 0})
    if (dbias || dx_f8) {                           // the fused form (transformer layers): 12 waves per workgroup, three partial sets
        const int nitx = (int)((C + 511) / 512);
        long long bl = occ_cdiv(rows, 12 * 4);
        if (bl > 256) bl = 256;
        const LnbExtra ex{(unsigned char*)dx_f8, f8_scale, f8_amax, dbias ? 1 : 0};
        // C <= 1024: 12 waves (158 registers, no spill); 1024 < C <= 1536 (XLS-R-1B: 1280): 8 waves, whose 256-register budget holds the 120 accumulators
#define OCC_LNBX(TD, N, W) hipLaunchKernelGGL((layernorm_bwd16_kernel<TD, float, N, false, W, true>), dim3((unsigned)bl), dim3(64 * W), 0, s, (const TD*)dy, (const float*)x, gamma, beta, dres, dx, (unsigned short*)dx_bf16, bm, dgamma, dbeta, (long long)rows, (int)C, eps, scratch, ex)
        if (dy_dtype == OCC_F32) { if (nitx == 1) OCC_LNBX(float, 1, 12); else if (nitx == 2) OCC_LNBX(float, 2, 12); else OCC_LNBX(float, 3, 8); }
        else { if (nitx == 1) OCC_LNBX(unsigned short, 1, 12); else if (nitx == 2) OCC_LNBX(unsigned short, 2, 12); else OCC_LNBX(unsigned short, 3, 8); }
#undef OCC_LNBX
        if (!g_lnb_defer)
            hipLaunchKernelGGL(layernorm_bwd_finalize_kernel, dim3((unsigned)occ_cdiv(3 * C, 64)), dim3(256), 0, s, scratch, (int)bl, (int)C, dgamma, dbeta, dbias, 3);
        OCC_LAUNCH_CHECK("occ_layernorm_bwd_fused");
        return OCC_OK;
    }
    static const int lnb16 = getenv("OCC_LNB16") ? atoi(getenv("OCC_LNB16")) : 1;
    // C <= 512 (the conv stack): 16 waves of 128 registers; 512 < C <= 1024 (a transformer LayerNorm on the separate-pass route taken with
    // residual dropout / layerdrop): 12 waves -- the 16-wave form of that width spilled 16 bytes per lane
    const int wv = nit == 2 ? 12 : 16;
    long long blocks16 = occ_cdiv(rows, wv * 4);        // >= 4 rows per wave
    if (blocks16 > lnb_blocks) blocks16 = lnb_blocks;
    const bool wide = lnb16 && rows >= 2048 && !(gelu && nit == 2) && nit <= 2;      // (the GELU form at C > 512 does not fit 128 registers)
    float* part = wide && scratch && scratch_floats >= blocks16 * 2 * C && ((uintptr_t)scratch & 15) == 0 ? scratch : nullptr;
#define OCC_LNB16W(TD, TXX, N, G, W) hipLaunchKernelGGL((layernorm_bwd16_kernel<TD, TXX, N, G, W>), dim3((unsigned)blocks16), dim3(64 * W), 0, s, (const TD*)dy, (const TXX*)x, gamma, beta, dres, dx, (unsigned short*)dx_bf16, bm, dgamma, dbeta, (long long)rows, (int)C, eps, part, LnbExtra{nullptr, nullptr, nullptr, 0})
    // (the GELU form exists at C <= 512 only: no wide instantiation of it at NIT = 2)
#define OCC_LNB_NG(TD, TXX) do { if (nit == 1) { if (wide) OCC_LNB16W(TD, TXX, 1, true, 16); else OCC_LNB(TD, TXX, 1, true); } else if (nit == 2) OCC_LNB(TD, TXX, 2, true); \
                                 else if (nit == 3) OCC_LNB(TD, TXX, 3, false); else OCC_LNB(TD, TXX, 4, false); } while (0)
#define OCC_LNB_NN(TD, TXX) do { if (nit == 1) { if (wide) OCC_LNB16W(TD, TXX, 1, false, 16); else OCC_LNB(TD, TXX, 1, false); } \
                                 else if (nit == 2) { if (wide) OCC_LNB16W(TD, TXX, 2, false, 12); else OCC_LNB(TD, TXX, 2, false); } \
                                 else if (nit == 3) OCC_LNB(TD, TXX, 3, false); else OCC_LNB(TD, TXX, 4, false); } while (0)
    const bool df = dy_dtype == OCC_F32, xf = x_dtype == OCC_F32;
    if (gelu) {
        if (df && xf) OCC_LNB_NG(float, float); else if (df) OCC_LNB_NG(float, unsigned short);
        else if (xf) OCC_LNB_NG(unsigned short, float); else OCC_LNB_NG(unsigned short, unsigned short);
    } else {
        if (df && xf) OCC_LNB_NN(float, float); else if (df) OCC_LNB_NN(float, unsigned short);
        else if (xf) OCC_LNB_NN(unsigned short, float); else OCC_LNB_NN(unsigned short, unsigned short);
    }
#undef OCC_LNB_NG
#undef OCC_LNB_NN
#undef OCC_LNB16W
#undef OCC_LNB16
#undef OCC_LNB
    if (part) hipLaunchKernelGGL(layernorm_bwd_finalize_kernel, dim3((unsigned)occ_cdiv(2 * C, 64)), dim3(256), 0, s, part, (int)blocks16, (int)C, dgamma, dbeta, (float*)nullptr, 2);
    OCC_LAUNCH_CHECK("occ_layernorm_bwd");
    return OCC_OK;
}

int occ_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* gamma, const float* dres, float* dx, void* dx_bf16, float* dgamma,
                      float* dbeta, int64_t rows, int64_t C, float eps, float* scratch, int64_t scratch_floats, void* stream) {
    return occ_layernorm_bwd_ex(dy, dy_dtype, x, OCC_F32, gamma, nullptr, dres, dx, dx_bf16, nullptr, dgamma, dbeta, rows, C, eps, 0, scratch, scratch_floats, stream);
}

int occ_dropout_ex(const void* x, int x_dtype, void* y, int y_dtype, uint8_t* mask, const float* residual, int64_t n, float p, float scale, uint64_t seed,
                   uint64_t stream_id, int generate, void* stream) {
    OCC_CHECK_ARG(x && y && n >= 0 && p >= 0.f && p < 1.f, "occ_dropout_ex: bad argument");
    OCC_CHECK_ARG((x_dtype == OCC_F32 || x_dtype == OCC_BF16) && (y_dtype == OCC_F32 || y_dtype == OCC_BF16), "occ_dropout_ex: dtypes must be f32 or bf16");
    if (n == 0) return OCC_OK;
    long long blocks = occ_cdiv(occ_cdiv(n, 4), 256);
    if (blocks > 8192) blocks = 8192;
    hipStream_t s = (hipStream_t)stream;
#define OCC_DX(TX, TY) hipLaunchKernelGGL((dropout_ex_kernel<TX, TY>), dim3((unsigned)blocks), dim3(256), 0, s, (const TX*)x, (TY*)y, mask, residual, (long long)n, p, scale, seed, stream_id, generate)
    if (x_dtype == OCC_F32 && y_dtype == OCC_F32) OCC_DX(float, float);
    else if (x_dtype == OCC_F32) OCC_DX(float, unsigned short);
    else if (y_dtype == OCC_F32) OCC_DX(unsigned short, float);
    else OCC_DX(unsigned short, unsigned short);
#undef OCC_DX
    OCC_LAUNCH_CHECK("occ_dropout_ex");
    return OCC_OK;
}

int occ_dropout_mask(uint8_t* mask, int64_t n, float p, uint64_t seed, uint64_t stream_id, void* stream) {
    OCC_CHECK_ARG(mask && n >= 0 && p >= 0.f && p < 1.f, "occ_dropout_mask: bad argument");
    if (n == 0) return OCC_OK;
    long long blocks = occ_cdiv(occ_cdiv(n, 4), 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, mask, (long long)n, p, seed, stream_id);
    OCC_LAUNCH_CHECK("occ_dropout_mask");
    return OCC_OK;
}

int occ_gelu_bwd_rows(const float* dy, const void* u, void* out, const occ_rowmap* out_map, int64_t rows, int64_t C, void* stream) {
    OCC_CHECK_ARG(dy && u && out && out_map && rows >= 1 && C >= 4 && C % 4 == 0 && out_map->rows_per_batch >= 1, "occ_gelu_bwd_rows: bad argument");
    long long blocks = occ_cdiv(rows * (C / 4), 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(gelu_bwd_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, (const unsigned short*)u, (unsigned short*)out,
                       to_rowmap(*out_map), (long long)rows, (int)C);
    OCC_LAUNCH_CHECK("occ_gelu_bwd_rows");
    return OCC_OK;
}

int occ_conv0_ln_gelu_bwd(const float* wav, const float* w, const float* bias, const float* gamma, const float* beta, const void* dact, int dact_dtype,
                          float* dw, float* dbias, float* dgamma, float* dbeta, int64_t B, int64_t L, int64_t Tout, int64_t C, int64_t k, int64_t stride,
                          float eps, void* stream) {
    OCC_CHECK_ARG(wav && w && bias && gamma && beta && dact && dw && dbias && dgamma && dbeta, "occ_conv0_ln_gelu_bwd: null pointer");
    OCC_CHECK_ARG(C == 512 && k >= 1 && k <= C0B_MAXK && stride >= 1 && stride <= 16 && B >= 1 && B < 65536 && Tout == (L - k) / stride + 1,
                  "occ_conv0_ln_gelu_bwd: bad shape");
    OCC_CHECK_ARG(dact_dtype == OCC_F32 || dact_dtype == OCC_BF16, "occ_conv0_ln_gelu_bwd: dact must be f32 or bf16");
    // XLS-R geometry: the matrix-core form (conv0_mfma.hip).  OCC_CONV0_BWD_MFMA=0 switches back to the VALU kernel below (A/B).
    static const int c0b_mfma = getenv("OCC_CONV0_BWD_MFMA") ? atoi(getenv("OCC_CONV0_BWD_MFMA")) : 1;
    if (c0b_mfma && k == 10 && (reinterpret_cast<uintptr_t>(dact) & 15) == 0 && (reinterpret_cast<uintptr_t>(gamma) & 15) == 0 && (reinterpret_cast<uintptr_t>(beta) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(w) & 7) == 0) {
        occ_conv0_bwd_mfma_launch(wav, w, bias, gamma, beta, dact, dact_dtype == OCC_BF16, dw, dbias, dgamma, dbeta, B, L, Tout, stride, eps, (hipStream_t)stream);
        OCC_LAUNCH_CHECK("occ_conv0_ln_gelu_bwd(mfma)");
        return OCC_OK;
    }
    const int nsamp = (int)((C0B_FRAMES - 1) * stride + k);
    const size_t shm = ((size_t)((nsamp + 3) & ~3) + (size_t)512 * (k + 3)) * sizeof(float);
    const dim3 grid((unsigned)occ_cdiv(Tout, C0B_FRAMES), (unsigned)B), block(256);
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
#define OCC_C0B(TD, KT, CAST)                                                                                                              \
    e = hipFuncSetAttribute((const void*)conv0_bwd_kernel<TD, KT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                       \
    if (e != hipSuccess) { occ_set_error("occ_conv0_ln_gelu_bwd: %s", hipGetErrorString(e)); return OCC_ELAUNCH; }                              \
    hipLaunchKernelGGL((conv0_bwd_kernel<TD, KT>), grid, block, shm, s, wav, w, bias, gamma, beta, (const TD*)dact, dw, dbias, dgamma, dbeta,   \
                       (int)L, (int)Tout, (int)k, (int)stride, eps)
    if (dact_dtype == OCC_F32) { if (k <= 10) { OCC_C0B(float, 10, 0); } else { OCC_C0B(float, C0B_MAXK, 0); } }
    else { if (k <= 10) { OCC_C0B(unsigned short, 10, 0); } else { OCC_C0B(unsigned short, C0B_MAXK, 0); } }
#undef OCC_C0B
    OCC_LAUNCH_CHECK("occ_conv0_ln_gelu_bwd");
    return OCC_OK;
}

int occ_weight_norm_pack(const float* v, const float* g, void* w_fwd, void* w_bwd, float* norms, int64_t O, int64_t I, int64_t K, int64_t G, float* scratch,
                         int64_t scratch_floats, void* stream) {
    OCC_CHECK_ARG(v && g && w_fwd && norms && O >= 1 && I >= 1 && K >= 1 && G >= 1 && O % G == 0, "occ_weight_norm_pack: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int64_t cgn = O / G, S = 512;
    const size_t lds_f = (size_t)(K + I * (K + 1)) * 4, lds_b = (size_t)(K + cgn * (K + 1)) * 4;
    if (scratch && scratch_floats >= S * K && K <= 256 && 256 % K == 0 && lds_f <= 64 * 1024 && lds_b <= 64 * 1024) {
        hipLaunchKernelGGL(wn_sumsq_kernel, dim3((unsigned)S), dim3(256), 0, s, v, scratch, (long long)(O * I), (int)K);
        hipLaunchKernelGGL(wn_norms_kernel, dim3((unsigned)occ_cdiv(K, 256)), dim3(256), 0, s, (const float*)scratch, (int)S, norms, (int)K);
        hipLaunchKernelGGL(wn_pack_fwd_kernel, dim3((unsigned)O), dim3(256), lds_f, s, v, g, (const float*)norms, (unsigned short*)w_fwd, (int)I, (int)K);
        if (w_bwd) hipLaunchKernelGGL(wn_pack_bwd_kernel, dim3((unsigned)(G * I)), dim3(256), lds_b, s, v, g, (const float*)norms, (unsigned short*)w_bwd, (int)I, (int)K, (int)cgn);
    } else {
        hipLaunchKernelGGL(weight_norm_pack_kernel, dim3((unsigned)K), dim3(256), 0, s, v, g, (unsigned short*)w_fwd, (unsigned short*)w_bwd, norms, (int)O, (int)I, (int)K, (int)G);
    }
    OCC_LAUNCH_CHECK("occ_weight_norm_pack");
    return OCC_OK;
}

int occ_weight_norm_bwd(const float* v, const float* g, const float* norms, const float* dw_packed, float* dv, float* dg, int64_t O, int64_t I, int64_t K,
                        int64_t G, float* scratch, int64_t scratch_floats, void* stream) {
    OCC_CHECK_ARG(v && g && norms && dw_packed && dv && dg && O >= 1 && I >= 1 && K >= 1 && G >= 1 && O % G == 0, "occ_weight_norm_bwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const size_t lds_d = (size_t)(K * (I + 1) + 256) * 4, lds_a = (size_t)(K * (I + 1) + 2 * K) * 4;
    if (scratch && scratch_floats >= O * K + K && K <= 256 && 256 % K == 0 && lds_d <= 64 * 1024 && lds_a <= 64 * 1024) {
        float* dot = scratch + O * K;
        hipLaunchKernelGGL(wn_bwd_dot_kernel, dim3((unsigned)O), dim3(256), lds_d, s, v, dw_packed, scratch, (int)I, (int)K);
        hipLaunchKernelGGL(wn_bwd_finalize_kernel, dim3((unsigned)K), dim3(256), 0, s, (const float*)scratch, norms, dot, dg, (int)O, (int)K);
        hipLaunchKernelGGL(wn_bwd_apply_kernel, dim3((unsigned)O), dim3(256), lds_a, s, v, g, norms, dw_packed, (const float*)dot, dv, (int)I, (int)K);
    } else {
        hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3((unsigned)K), dim3(256), 0, s, v, g, norms, dw_packed, dv, dg, (int)O, (int)I, (int)K, (int)G);
    }
    OCC_LAUNCH_CHECK("occ_weight_norm_bwd");
    return OCC_OK;
}

}  // extern "C"
