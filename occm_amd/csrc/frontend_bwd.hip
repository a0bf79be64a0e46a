// Backward kernels of the XLS-R transformer encoder (fine-tuning path): transposing cast for the weight-gradient
// GEMMs, LayerNorm backward, and flash-style attention backward on MFMA.  Reference arithmetic: autograd of fairseq's
// TransformerSentenceEncoderLayer (pre-LN) reached from models/sslassist.py:48 when the optimizer holds the SSL
// parameters (oc_training.py:324).
#include "occ_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// ------------------------------------------------------------------------------------------------
// dst[c][r] = bf16(src[r][c]); dst rows have ld_dst >= rows elements (pad columns are left untouched: callers keep them 0).
template <typename TS>
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const TS* __restrict__ src, unsigned short* __restrict__ dst, long long rows, long long cols,
                                                            long long ld_src, long long ld_dst, float* __restrict__ colsum) {
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
            if (sizeof(TS) == 4) { const float f = reinterpret_cast<const float*>(src)[r * ld_src + c]; cs += f; v = f32_to_bf16_bits(f); }
            else { v = reinterpret_cast<const unsigned short*>(src)[r * ld_src + c]; cs += bf16_bits_to_f32(v); }
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

// ------------------------------------------------------------------------------------------------
// LayerNorm backward over rows of width C (C % 8 == 0, C <= 2048), one wave per row, persistent waves:
//   dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat)) (+ dres);  dgamma += sum dy*xhat;  dbeta += sum dy.
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

template <typename TDY, int NIT>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const TDY* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ dres, float* __restrict__ dx, unsigned short* __restrict__ dx_bf16,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, long long rows, int C, float eps) {
    __shared__ float red[3][2][NIT * 512];           // waves 1..3 park their partial dgamma / dbeta here
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long wave0 = (long long)blockIdx.x * 4 + wave, nwaves = (long long)gridDim.x * 4;
    float dg[NIT][8], db[NIT][8], g[NIT][8];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = (it * 64 + lane) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { dg[it][e] = 0.f; db[it][e] = 0.f; g[it][e] = 0.f; }
        if (c < C) LVec8<float>::load(gamma + c, g[it]);
    }
    for (long long row = wave0; row < rows; row += nwaves) {
        float xv[NIT][8], dv[NIT][8];
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = (it * 64 + lane) * 8;
            if (c < C) {
                LVec8<float>::load(x + row * C + c, xv[it]);
                LVec8<TDY>::load(dy + row * C + c, dv[it]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { xv[it][e] = 0.f; dv[it][e] = 0.f; }
            }
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
                if (dres) LVec8<float>::load(dres + row * C + c, r);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[e] = rstd * (dv[it][e] * g[it][e] - s1 - xv[it][e] * s2);
                    if (dres) v[e] += r[e];
                }
                *reinterpret_cast<float4*>(dx + row * C + c) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(dx + row * C + c + 4) = make_float4(v[4], v[5], v[6], v[7]);
                if (dx_bf16) {
                    unsigned w[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) w[i] = (unsigned)f32_to_bf16_bits(v[2 * i]) | ((unsigned)f32_to_bf16_bits(v[2 * i + 1]) << 16);
                    *reinterpret_cast<uint4*>(dx_bf16 + row * C + c) = make_uint4(w[0], w[1], w[2], w[3]);
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

// ------------------------------------------------------------------------------------------------
// Attention backward, head_dim 64, T <= 256, one workgroup (4 waves) per (batch, head).
// Keys are partitioned over the waves (64 each) and live on the MFMA lane, so S[q][key] and dP[q][key] come out of the
// MFMA already shaped as the B operands of dV^T += dO^T.P and dK^T += Q^T.dS (each wave keeps dK^T/dV^T of its keys in
// registers over the whole query sweep).  Probabilities are rebuilt from the forward's log-sum-exp (no row reductions), the
// row term delta = rowsum(dO * O) is computed in the prologue, and only dS crosses LDS, once per 32-query step, for dQ.
constexpr int AB_TP = 256;             // padded sequence length
constexpr int AB_TS = AB_TP + 8;       // row stride (elements) of the transposed / dS LDS images: 16-byte aligned rows

__global__ __launch_bounds__(256, 1) void attention_bwd_kernel(const unsigned short* __restrict__ qkv, const unsigned short* __restrict__ o,
                                                              const unsigned short* __restrict__ dout, const float* __restrict__ lse,
                                                              unsigned short* __restrict__ dqkv, int Tn, int H, long long ld_qkv, long long ld_o,
                                                              float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned short sm[];
    unsigned short* Qt = sm;                         // [64][AB_TS]  Q^T
    unsigned short* dOt = Qt + 64 * AB_TS;           // [64][AB_TS]  dO^T
    unsigned short* Kt = dOt + 64 * AB_TS;           // [64][AB_TS]  K^T
    unsigned short* dSs = Kt + 64 * AB_TS;           // [32][AB_TS]  dS of the current 32-query step
    float* lse_s = reinterpret_cast<float*>(dSs + 32 * AB_TS);   // [AB_TP]
    float* dlt_s = lse_s + AB_TP;                                // [AB_TP]
    const int D = H * 64;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const unsigned short* base = qkv + (size_t)b * Tn * ld_qkv + (size_t)h * 64;
    const unsigned short* obase = o + (size_t)b * Tn * ld_o + (size_t)h * 64;
    const unsigned short* dobase = dout + (size_t)b * Tn * ld_o + (size_t)h * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, g = lane >> 4;

    // ---- prologue: transposed images (zero beyond Tn), lse and delta ----
    for (int idx = tid; idx < AB_TP * 8; idx += 256) {
        const int t = idx >> 3, ch = idx & 7;
        uint4 qv = make_uint4(0, 0, 0, 0), dv = qv, kv = qv;
        if (t < Tn) {
            qv = *reinterpret_cast<const uint4*>(base + (size_t)t * ld_qkv + ch * 8);
            kv = *reinterpret_cast<const uint4*>(base + (size_t)t * ld_qkv + D + ch * 8);
            dv = *reinterpret_cast<const uint4*>(dobase + (size_t)t * ld_o + ch * 8);
        }
        const unsigned qw[4] = {qv.x, qv.y, qv.z, qv.w}, dw[4] = {dv.x, dv.y, dv.z, dv.w}, kw[4] = {kv.x, kv.y, kv.z, kv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            Qt[(ch * 8 + 2 * e) * AB_TS + t] = (unsigned short)(qw[e] & 0xffff); Qt[(ch * 8 + 2 * e + 1) * AB_TS + t] = (unsigned short)(qw[e] >> 16);
            dOt[(ch * 8 + 2 * e) * AB_TS + t] = (unsigned short)(dw[e] & 0xffff); dOt[(ch * 8 + 2 * e + 1) * AB_TS + t] = (unsigned short)(dw[e] >> 16);
            Kt[(ch * 8 + 2 * e) * AB_TS + t] = (unsigned short)(kw[e] & 0xffff); Kt[(ch * 8 + 2 * e + 1) * AB_TS + t] = (unsigned short)(kw[e] >> 16);
        }
    }
    for (int t = tid; t < AB_TP; t += 256) {
        float dl = 0.f, ls = 1.0e30f;                    // rows beyond Tn: P = exp2(. - 1e30) = 0
        if (t < Tn) {
            ls = lse[(size_t)bh * Tn + t];
            for (int d = 0; d < 64; ++d) dl += bf16_bits_to_f32(dobase[(size_t)t * ld_o + d]) * bf16_bits_to_f32(obase[(size_t)t * ld_o + d]);
        }
        lse_s[t] = ls; dlt_s[t] = dl;
    }
    // ---- this wave's keys: K and V row fragments stay in registers ----
    uint4 kf[4][2], vf[4][2];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        const int key = wave * 64 + kt * 16 + fr;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            kf[kt][s] = make_uint4(0, 0, 0, 0); vf[kt][s] = kf[kt][s];
            if (key < Tn) {
                kf[kt][s] = *reinterpret_cast<const uint4*>(base + (size_t)key * ld_qkv + D + s * 32 + g * 8);
                vf[kt][s] = *reinterpret_cast<const uint4*>(base + (size_t)key * ld_qkv + 2 * D + s * 32 + g * 8);
            }
        }
    }
    f32x4 dvacc[4][4], dkacc[4][4];                   // [d-tile][key-tile]: rows d = 16*dt + 4g + r, col key = 16*kt + fr
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { dvacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; dkacc[i][j] = dvacc[i][j]; }
    const float c2 = scale * 1.44269504088896340736f;
    __syncthreads();

    const int nstep = (Tn + 31) / 32;
    for (int st = 0; st < nstep; ++st) {
        const int q0 = st * 32;
        // Q / dO row fragments of the two 16-query tiles (A operands), straight from global (L2-resident)
        uint4 qf[2][2], df[2][2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int q = q0 + qt * 16 + fr;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                qf[qt][s] = make_uint4(0, 0, 0, 0); df[qt][s] = qf[qt][s];
                if (q < Tn) {
                    qf[qt][s] = *reinterpret_cast<const uint4*>(base + (size_t)q * ld_qkv + s * 32 + g * 8);
                    df[qt][s] = *reinterpret_cast<const uint4*>(dobase + (size_t)q * ld_o + s * 32 + g * 8);
                }
            }
        }
        unsigned pp[4][4], ds[4][4];                   // [key-tile][packed bf16 pairs]: slots j<4 from q-tile 0, j>=4 from q-tile 1
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int key = wave * 64 + kt * 16 + fr;
            float pv[2][4], dsv[2][4];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                f32x4 sacc = (f32x4){0.f, 0.f, 0.f, 0.f}, dpacc = sacc;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&qf[qt][s]), *reinterpret_cast<bf16x8*>(&kf[kt][s]), sacc, 0, 0, 0);
                    dpacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&df[qt][s]), *reinterpret_cast<bf16x8*>(&vf[kt][s]), dpacc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = q0 + qt * 16 + g * 4 + r;
                    const float p = key < Tn ? __builtin_amdgcn_exp2f(sacc[r] * c2 - lse_s[q]) : 0.f;
                    pv[qt][r] = p;
                    dsv[qt][r] = p * (dpacc[r] - dlt_s[q]);
                    dSs[(qt * 16 + g * 4 + r) * AB_TS + key] = f32_to_bf16_bits(dsv[qt][r]);
                }
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                pp[kt][e] = (unsigned)f32_to_bf16_bits(pv[0][2 * e]) | ((unsigned)f32_to_bf16_bits(pv[0][2 * e + 1]) << 16);
                pp[kt][2 + e] = (unsigned)f32_to_bf16_bits(pv[1][2 * e]) | ((unsigned)f32_to_bf16_bits(pv[1][2 * e + 1]) << 16);
                ds[kt][e] = (unsigned)f32_to_bf16_bits(dsv[0][2 * e]) | ((unsigned)f32_to_bf16_bits(dsv[0][2 * e + 1]) << 16);
                ds[kt][2 + e] = (unsigned)f32_to_bf16_bits(dsv[1][2 * e]) | ((unsigned)f32_to_bf16_bits(dsv[1][2 * e + 1]) << 16);
            }
        }
        // dV^T += dO^T . P ; dK^T += Q^T . dS   (k = the 32 queries of this step, slot j <-> q0 + (j<4 ? 4g+j : 16+4g+j-4))
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const unsigned short* dr = dOt + (dt * 16 + fr) * AB_TS + q0 + g * 4;
            const unsigned short* qr = Qt + (dt * 16 + fr) * AB_TS + q0 + g * 4;
            const uint2 d0 = *reinterpret_cast<const uint2*>(dr), d1 = *reinterpret_cast<const uint2*>(dr + 16);
            const uint2 q0v = *reinterpret_cast<const uint2*>(qr), q1v = *reinterpret_cast<const uint2*>(qr + 16);
            uint4 da = make_uint4(d0.x, d0.y, d1.x, d1.y), qa = make_uint4(q0v.x, q0v.y, q1v.x, q1v.y);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                uint4 pb = make_uint4(pp[kt][0], pp[kt][1], pp[kt][2], pp[kt][3]);
                uint4 sb = make_uint4(ds[kt][0], ds[kt][1], ds[kt][2], ds[kt][3]);
                dvacc[dt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&da), *reinterpret_cast<bf16x8*>(&pb), dvacc[dt][kt], 0, 0, 0);
                dkacc[dt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&qa), *reinterpret_cast<bf16x8*>(&sb), dkacc[dt][kt], 0, 0, 0);
            }
        }
        __syncthreads();                               // dS of all 256 keys is in LDS
        // dQ[q][d] = scale * sum_key dS[q][key] K[key][d]: wave w owns d-tile w, all keys
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            f32x4 qacc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < AB_TP / 32; ++ks) {
                uint4 af = *reinterpret_cast<const uint4*>(dSs + (qt * 16 + fr) * AB_TS + ks * 32 + g * 8);
                uint4 bfv = *reinterpret_cast<const uint4*>(Kt + (wave * 16 + fr) * AB_TS + ks * 32 + g * 8);
                qacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&af), *reinterpret_cast<bf16x8*>(&bfv), qacc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = q0 + qt * 16 + g * 4 + r;
                if (q < Tn) dqkv[((size_t)b * Tn + q) * ld_qkv + (size_t)h * 64 + wave * 16 + fr] = f32_to_bf16_bits(qacc[r] * scale);
            }
        }
        __syncthreads();                               // dSs is rewritten by the next step
    }
    // ---- dK, dV of this wave's keys: lane holds 4 consecutive d of one key ----
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        const int key = wave * 64 + kt * 16 + fr;
        if (key >= Tn) continue;
        unsigned short* dkrow = dqkv + ((size_t)b * Tn + key) * ld_qkv + D + (size_t)h * 64;
        unsigned short* dvrow = dqkv + ((size_t)b * Tn + key) * ld_qkv + 2 * D + (size_t)h * 64;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            uint2 kk, vv;
            kk.x = (unsigned)f32_to_bf16_bits(dkacc[dt][kt][0] * scale) | ((unsigned)f32_to_bf16_bits(dkacc[dt][kt][1] * scale) << 16);
            kk.y = (unsigned)f32_to_bf16_bits(dkacc[dt][kt][2] * scale) | ((unsigned)f32_to_bf16_bits(dkacc[dt][kt][3] * scale) << 16);
            vv.x = (unsigned)f32_to_bf16_bits(dvacc[dt][kt][0]) | ((unsigned)f32_to_bf16_bits(dvacc[dt][kt][1]) << 16);
            vv.y = (unsigned)f32_to_bf16_bits(dvacc[dt][kt][2]) | ((unsigned)f32_to_bf16_bits(dvacc[dt][kt][3]) << 16);
            *reinterpret_cast<uint2*>(dkrow + dt * 16 + g * 4) = kk;
            *reinterpret_cast<uint2*>(dvrow + dt * 16 + g * 4) = vv;
        }
    }
}

}  // namespace

extern "C" {

int occ_transpose_bf16(const void* src, int src_dtype, void* dst, int64_t rows, int64_t cols, int64_t ld_src, int64_t ld_dst, float* colsum,
                       void* stream) {
    OCC_CHECK_ARG(src && dst && rows >= 1 && cols >= 1 && ld_src >= cols && ld_dst >= rows, "occ_transpose_bf16: bad argument");
    OCC_CHECK_ARG(src_dtype == OCC_F32 || src_dtype == OCC_BF16, "occ_transpose_bf16: source must be f32 or bf16");
    const dim3 grid((unsigned)occ_cdiv(cols, 64), (unsigned)occ_cdiv(rows, 64)), block(256);
    OCC_CHECK_ARG(grid.y < 65536, "occ_transpose_bf16: too many rows");
    hipStream_t s = (hipStream_t)stream;
    if (src_dtype == OCC_F32) hipLaunchKernelGGL(transpose_bf16_kernel<float>, grid, block, 0, s, (const float*)src, (unsigned short*)dst, (long long)rows, (long long)cols, (long long)ld_src, (long long)ld_dst, colsum);
    else hipLaunchKernelGGL(transpose_bf16_kernel<unsigned short>, grid, block, 0, s, (const unsigned short*)src, (unsigned short*)dst, (long long)rows, (long long)cols, (long long)ld_src, (long long)ld_dst, colsum);
    OCC_LAUNCH_CHECK("occ_transpose_bf16");
    return OCC_OK;
}

int occ_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* gamma, const float* dres, float* dx, void* dx_bf16, float* dgamma,
                      float* dbeta, int64_t rows, int64_t C, float eps, void* stream) {
    OCC_CHECK_ARG(dy && x && gamma && dx && dgamma && dbeta, "occ_layernorm_bwd: null pointer");
    OCC_CHECK_ARG(rows >= 1 && C >= 8 && C % 8 == 0 && C <= 2048, "occ_layernorm_bwd: C must be a multiple of 8 in [8,2048]");
    OCC_CHECK_ARG(dy_dtype == OCC_F32 || dy_dtype == OCC_BF16, "occ_layernorm_bwd: dy must be f32 or bf16");
    long long blocks = occ_cdiv(rows, 4 * 8);           // >= 8 rows per wave so the per-workgroup atomics amortise
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    const int nit = (int)((C + 511) / 512);
    hipStream_t s = (hipStream_t)stream;
#define OCC_LNB(T, N) hipLaunchKernelGGL((layernorm_bwd_kernel<T, N>), dim3((unsigned)blocks), dim3(256), 0, s, (const T*)dy, x, gamma, dres, dx, (unsigned short*)dx_bf16, dgamma, dbeta, (long long)rows, (int)C, eps)
    if (dy_dtype == OCC_F32) { if (nit == 1) OCC_LNB(float, 1); else if (nit == 2) OCC_LNB(float, 2); else if (nit == 3) OCC_LNB(float, 3); else OCC_LNB(float, 4); }
    else { if (nit == 1) OCC_LNB(unsigned short, 1); else if (nit == 2) OCC_LNB(unsigned short, 2); else if (nit == 3) OCC_LNB(unsigned short, 3); else OCC_LNB(unsigned short, 4); }
#undef OCC_LNB
    OCC_LAUNCH_CHECK("occ_layernorm_bwd");
    return OCC_OK;
}

int occ_attention_bwd(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, int64_t B, int64_t T, int64_t H, int64_t hd,
                      int64_t ld_qkv, int64_t ld_o, float scale, void* stream) {
    OCC_CHECK_ARG(qkv && o && dout && lse && dqkv, "occ_attention_bwd: null pointer");
    OCC_CHECK_ARG(hd == 64 && T >= 1 && T <= AB_TP && B >= 1 && H >= 1, "occ_attention_bwd: needs head_dim 64 and T <= %d (T=%ld hd=%ld)", AB_TP, (long)T, (long)hd);
    OCC_CHECK_ARG(ld_qkv % 8 == 0 && ld_o % 8 == 0 && ld_qkv >= 3 * H * hd && ld_o >= H * hd, "occ_attention_bwd: leading dimensions");
    const size_t shm = (size_t)(3 * 64 + 32) * AB_TS * 2 + 2 * AB_TP * 4;
    hipError_t e = hipFuncSetAttribute((const void*)attention_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) { occ_set_error("occ_attention_bwd: cannot raise LDS limit: %s", hipGetErrorString(e)); return OCC_ELAUNCH; }
    hipLaunchKernelGGL(attention_bwd_kernel, dim3((unsigned)(B * H)), dim3(256), shm, (hipStream_t)stream, (const unsigned short*)qkv, (const unsigned short*)o,
                       (const unsigned short*)dout, lse, (unsigned short*)dqkv, (int)T, (int)H, (long long)ld_qkv, (long long)ld_o, scale);
    OCC_LAUNCH_CHECK("occ_attention_bwd");
    return OCC_OK;
}

}  // extern "C"
