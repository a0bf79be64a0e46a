// Row kernels of the wav2vec2 / XLS-R front-end (fairseq Wav2Vec2Model reached from
// models/sslassist.py:48, models/xlsr.py:46): LayerNorm(+GELU), the first conv block fused with its
// LayerNorm and GELU, and the self-attention core.
#include "occ_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// LayerNorm over rows of width C (C % 8 == 0, C <= 2048): one wave per row, the row lives in
// registers, mean and variance are two exact passes over those registers, 16-byte accesses.
template <typename T> struct Vec8;
template <> struct Vec8<float> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
};
template <> struct Vec8<unsigned short> {
    static __device__ __forceinline__ void load(const unsigned short* p, float (&v)[8]) {
        const uint4 a = *reinterpret_cast<const uint4*>(p);
        const unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    }
    static __device__ __forceinline__ void store(unsigned short* p, const float (&v)[8]) {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};

// F8: besides y (bf16) the kernel writes an e4m3 copy of the bf16-rounded outputs times *f8_scale (saturating) and records their |max|
// in *f8_amax -- operand and next-step statistics of the fp8 GEMM that consumes this LayerNorm (delayed per-tensor scaling), which
// otherwise cost a separate pass over y.
template <typename TI, typename TO, int NIT, bool F8 = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const TI* __restrict__ x, TO* __restrict__ y, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, long long rows, int C, float eps, int gelu,
                                                       unsigned char* __restrict__ yq = nullptr, const float* __restrict__ f8_scale = nullptr,
                                                       float* __restrict__ f8_amax = nullptr) {
    __shared__ float wmx[F8 ? 4 : 1];
    float f8max = 0.f;
    // (F8: a grid-stride loop over the row groups, so that a launch ends in ~1000 guarded |max| updates of ONE address instead of one per
    // four rows -- 3184 of them serialised in the L2 for 22 of the kernel's 54 us at [12736, 1024])
    for (long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long long)gridDim.x * 4) {
    const int lane = threadIdx.x & 63;
    const TI* xr = x + row * C;
    float v[NIT][8];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = (it * 64 + lane) * 8;
        if (c < C) {
            Vec8<TI>::load(xr + c, v[it]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[it][e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[it][e] = 0.f;
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = (it * 64 + lane) * 8;
        if (c < C) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[it][e] - mean; q += d * d; }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    TO* yr = y + row * C;
    const float f8sc = F8 ? *f8_scale : 1.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = (it * 64 + lane) * 8;
        if (c < C) {
            float g[8], b[8], o[8];
            Vec8<float>::load(gamma + c, g);
            Vec8<float>::load(beta + c, b);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = (v[it][e] - mean) * rstd * g[e] + b[e];
                o[e] = gelu ? gelu_erf(t) : t;
            }
            Vec8<TO>::store(yr + c, o);
            if constexpr (F8) {
                float qv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float r = bf16_bits_to_f32(f32_to_bf16_bits(o[e]));       // the value the bf16 tensor holds
                    f8max = fmaxf(f8max, fabsf(r));
                    qv[e] = fminf(fmaxf(r * f8sc, -448.f), 448.f);
                }
                int q0 = __builtin_amdgcn_cvt_pk_fp8_f32(qv[0], qv[1], 0, false), q1 = __builtin_amdgcn_cvt_pk_fp8_f32(qv[4], qv[5], 0, false);
                q0 = __builtin_amdgcn_cvt_pk_fp8_f32(qv[2], qv[3], q0, true); q1 = __builtin_amdgcn_cvt_pk_fp8_f32(qv[6], qv[7], q1, true);
                *reinterpret_cast<uint2*>(yq + row * C + c) = make_uint2((unsigned)q0, (unsigned)q1);
            }
        }
    }
    }
    if constexpr (F8) {                             // one guarded atomic per workgroup (non-negative floats order as their bit patterns)
        f8max = wave_max(f8max);
        if ((threadIdx.x & 63) == 0) wmx[threadIdx.x >> 6] = f8max;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float m = fmaxf(fmaxf(wmx[0], wmx[1]), fmaxf(wmx[2], wmx[3]));
            if (m > __hip_atomic_load(f8_amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(reinterpret_cast<unsigned*>(f8_amax), __float_as_uint(m));
        }
    }
}

template <typename TI, typename TO>
int launch_layernorm(const void* x, void* y, const float* gamma, const float* beta, long long rows, int C, float eps, int gelu, hipStream_t s) {
    const dim3 grid((unsigned)occ_cdiv(rows, 4)), block(256);
    const int nit = (C + 511) / 512;
    switch (nit) {
        case 1: hipLaunchKernelGGL((layernorm_kernel<TI, TO, 1>), grid, block, 0, s, (const TI*)x, (TO*)y, gamma, beta, rows, C, eps, gelu); break;
        case 2: hipLaunchKernelGGL((layernorm_kernel<TI, TO, 2>), grid, block, 0, s, (const TI*)x, (TO*)y, gamma, beta, rows, C, eps, gelu); break;
        case 3: hipLaunchKernelGGL((layernorm_kernel<TI, TO, 3>), grid, block, 0, s, (const TI*)x, (TO*)y, gamma, beta, rows, C, eps, gelu); break;
        case 4: hipLaunchKernelGGL((layernorm_kernel<TI, TO, 4>), grid, block, 0, s, (const TI*)x, (TO*)y, gamma, beta, rows, C, eps, gelu); break;
        default: return -1;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// First conv block: Conv1d(1 -> 512, k, stride) + bias -> LayerNorm(512) -> GELU, channels-last out.
// One wave per output frame; lane owns 8 channels whose k taps stay in registers; the waveform tile
// (256 frames * stride + k samples) is staged once per workgroup in LDS and read back as broadcasts.
constexpr int C0_FRAMES = 256;     // frames per workgroup (64 per wave)
constexpr int C0_MAXK = 16;

// sums of NF independent values over the wave, the NF chains interleaved so their latencies overlap
template <int NF> __device__ __forceinline__ void wave_sum_multi(float (&v)[NF]) {
#pragma unroll
    for (int u = 0; u < NF; ++u) v[u] += occ_dpp<0xB1>(v[u]);
#pragma unroll
    for (int u = 0; u < NF; ++u) v[u] += occ_dpp<0x4E>(v[u]);
#pragma unroll
    for (int u = 0; u < NF; ++u) v[u] += occ_dpp<0x141>(v[u]);
#pragma unroll
    for (int u = 0; u < NF; ++u) v[u] += occ_dpp<0x140>(v[u]);
#pragma unroll
    for (int u = 0; u < NF; ++u) v[u] += __shfl_xor(v[u], 16, 64);
#pragma unroll
    for (int u = 0; u < NF; ++u) v[u] += __shfl_xor(v[u], 32, 64);
}

// KT = compile-time tap count (10 for wav2vec2, 16 = generic with zero-padded taps): no per-tap branches, the frame's samples are
// fetched as one batch of LDS broadcasts; C0_NF frames are in flight per wave so the two cross-lane reductions of a frame overlap
// with those of its neighbours (the loop is VALU/latency bound: ~330 instructions per frame before, 2 serial reductions each).
constexpr int C0_NF = 4;
template <typename TO, int KT>
__global__ __launch_bounds__(256) void conv0_ln_gelu_kernel(const float* __restrict__ wav, const float* __restrict__ w, const float* __restrict__ bias,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, TO* __restrict__ out,
                                                           int L, int Tout, int k, int stride, float eps) {
    extern __shared__ __attribute__((aligned(16))) float smp[];
    const int b = blockIdx.y, f0 = blockIdx.x * C0_FRAMES;
    const int nsamp = (C0_FRAMES - 1) * stride + KT;
    const float* wb = wav + (size_t)b * L;
    for (int i = threadIdx.x; i < nsamp; i += 256) {
        const int g = f0 * stride + i;
        smp[i] = g < L ? wb[g] : 0.f;
    }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c0 = lane * 8;
    float wr[8][KT], br[8], gr[8], ber[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        br[e] = bias[c0 + e]; gr[e] = gamma[c0 + e]; ber[e] = beta[c0 + e];
#pragma unroll
        for (int t = 0; t < KT; ++t) wr[e][t] = t < k ? w[(size_t)(c0 + e) * k + t] : 0.f;
    }
    __syncthreads();
    constexpr int FPW = C0_FRAMES / 4;
    for (int fi = 0; fi < FPW; fi += C0_NF) {
        const int fl = wave * FPW + fi;
        const int f = f0 + fl;
        if (f >= Tout) break;
        float acc[C0_NF][8], sm[C0_NF], q[C0_NF];
#pragma unroll
        for (int u = 0; u < C0_NF; ++u) {
            float xv[KT];
#pragma unroll
            for (int t = 0; t < KT; ++t) xv[t] = smp[(fl + u) * stride + t];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[u][e] = br[e];
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[u][e] = fmaf(wr[e][t], xv[t], acc[u][e]);
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s += acc[u][e];
            sm[u] = s;
        }
        wave_sum_multi<C0_NF>(sm);
#pragma unroll
        for (int u = 0; u < C0_NF; ++u) {
            sm[u] *= (1.0f / 512.0f);
            float qq = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = acc[u][e] - sm[u]; qq += d * d; }
            q[u] = qq;
        }
        wave_sum_multi<C0_NF>(q);
#pragma unroll
        for (int u = 0; u < C0_NF; ++u) {
            if (f + u >= Tout) break;
            const float rstd = 1.0f / sqrtf(q[u] * (1.0f / 512.0f) + eps);
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = gelu_erf((acc[u][e] - sm[u]) * rstd * gr[e] + ber[e]);
            Vec8<TO>::store(out + ((size_t)b * Tout + f + u) * 512 + c0, o);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Self-attention core, f32 arithmetic, storage type T (f32 or bf16): softmax(scale * Q.K^T).V per
// (batch, head).  K and V of one head are staged in LDS as f32 (rows padded to hd+1 floats so the
// key-per-lane reads are conflict-free); each wave walks query rows: lanes = keys for the scores,
// lanes = head dims for the weighted sum.  Short sequences only (T*(hd+1)*8 + ... <= 160 KiB).
template <typename T>
__global__ __launch_bounds__(256) void attention_kernel(const T* __restrict__ qkv, T* __restrict__ out, int Tn, int H, int hd,
                                                       long long ld_qkv, long long ld_out, float scale, int qsplit, const int* __restrict__ kv_len) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int D = H * hd;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    // kv_len: keys (= valid frames) of utterance b in a zero-padded batch; rows past it are neither attended to nor meaningful as queries
    const int Tk = kv_len ? min(max(kv_len[b], 1), Tn) : Tn;
    const int hp = hd + 1;
    float* Ks = sm;                       // [Tn][hd+1]
    float* Vs = Ks + (size_t)Tn * hp;     // [Tn][hd+1]
    float* Ps = Vs + (size_t)Tn * hp;     // [4 waves][Tn]
    float* Qs = Ps + 4 * Tn;              // [4 waves][hd]
    const T* base = qkv + (size_t)b * Tn * ld_qkv + (size_t)h * hd;
    for (int i = threadIdx.x; i < Tn * hd; i += 256) {
        const int t = i / hd, d = i - t * hd;
        Ks[t * hp + d] = occ_load_f32(base + (size_t)t * ld_qkv + D + d);
        Vs[t * hp + d] = occ_load_f32(base + (size_t)t * ld_qkv + 2 * D + d);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* pw = Ps + wave * Tn;
    float* qw = Qs + wave * hd;
    const int rows_per = (Tn + qsplit - 1) / qsplit;
    const int q_begin = blockIdx.y * rows_per, q_end = min(Tn, q_begin + rows_per);
    const int nkey = (Tk + 63) / 64;
    for (int qi = q_begin + wave; qi < q_end; qi += 4) {
        for (int d = lane; d < hd; d += 64) qw[d] = occ_load_f32(base + (size_t)qi * ld_qkv + d) * scale;
        __builtin_amdgcn_wave_barrier();
        float mx = -3.0e38f;
#pragma unroll 4
        for (int kb = 0; kb < nkey; ++kb) {
            const int key = kb * 64 + lane;
            float s = -3.0e38f;
            if (key < Tk) {
                s = 0.f;
                const float* kr = Ks + key * hp;
                for (int d = 0; d < hd; ++d) s = fmaf(qw[d], kr[d], s);
            }
            mx = fmaxf(mx, s);
            if (key < Tk) pw[key] = s;
        }
        mx = wave_max(mx);
        float sum = 0.f;
        for (int kb = 0; kb < nkey; ++kb) {
            const int key = kb * 64 + lane;
            if (key < Tk) { const float p = expf(pw[key] - mx); pw[key] = p; sum += p; }
        }
        sum = wave_sum(sum);
        __builtin_amdgcn_wave_barrier();
        const float inv = 1.0f / sum;
        for (int d = lane; d < hd; d += 64) {
            float o = 0.f;
            for (int key = 0; key < Tk; ++key) o = fmaf(pw[key], Vs[key * hp + d], o);
            occ_store_f32(out + ((size_t)b * Tn + qi) * ld_out + (size_t)h * hd + d, o * inv);
        }
        __builtin_amdgcn_wave_barrier();
    }
}


// Any-length form of the kernel above (the f32 scoring path on utterances longer than ~6 s, whose K and V of a head no longer fit in
// LDS): one workgroup per (batch, head, 32 queries), keys streamed through LDS in blocks of 64 with the online-softmax recurrence; a wave
// owns 8 queries, lanes = keys for the scores and = head dims for the weighted sum.  f32 arithmetic throughout (expf, not exp2).
template <typename T>
__global__ __launch_bounds__(256) void attention_stream_kernel(const T* __restrict__ qkv, T* __restrict__ out, int Tn, int H, int hd,
                                                              long long ld_qkv, long long ld_out, float scale, const int* __restrict__ kv_len) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int KB = 64, QB = 32, QW = QB / 4;
    const int D = H * hd, hp = hd + 1;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int Tk = kv_len ? min(max(kv_len[b], 1), Tn) : Tn;
    float* Ks = sm;                       // [KB][hd+1]
    float* Vs = Ks + KB * hp;             // [KB][hd+1]
    float* Qs = Vs + KB * hp;             // [QB][hd], pre-scaled
    float* Ps = Qs + QB * hd;             // [4 waves][KB]
    const T* base = qkv + (size_t)b * Tn * ld_qkv + (size_t)h * hd;
    const int q0 = blockIdx.y * QB;
    for (int i = threadIdx.x; i < QB * hd; i += 256) {
        const int q = i / hd, d = i - q * hd;
        Qs[i] = q0 + q < Tn ? occ_load_f32(base + (size_t)(q0 + q) * ld_qkv + d) * scale : 0.f;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* pw = Ps + wave * KB;
    float m[QW], l[QW], o[QW][2];
#pragma unroll
    for (int j = 0; j < QW; ++j) { m[j] = -3.0e38f; l[j] = 0.f; o[j][0] = 0.f; o[j][1] = 0.f; }
    for (int k0 = 0; k0 < Tk; k0 += KB) {
        __syncthreads();                  // the previous block's readers are done (and Qs is written before the first use)
        for (int i = threadIdx.x; i < KB * hd; i += 256) {
            const int t = i / hd, d = i - t * hd;
            const bool ok = k0 + t < Tk;
            Ks[t * hp + d] = ok ? occ_load_f32(base + (size_t)(k0 + t) * ld_qkv + D + d) : 0.f;
            Vs[t * hp + d] = ok ? occ_load_f32(base + (size_t)(k0 + t) * ld_qkv + 2 * D + d) : 0.f;
        }
        __syncthreads();
        const bool valid = k0 + lane < Tk;
#pragma unroll
        for (int j = 0; j < QW; ++j) {
            const float* qw = Qs + (wave * QW + j) * hd;
            const float* kr = Ks + lane * hp;
            float sc = 0.f;
            for (int d = 0; d < hd; ++d) sc = fmaf(qw[d], kr[d], sc);
            const float mn = fmaxf(m[j], wave_max(valid ? sc : -3.0e38f));
            const float alpha = expf(m[j] - mn);
            const float p = valid ? expf(sc - mn) : 0.f;
            l[j] = l[j] * alpha + wave_sum(p);
            m[j] = mn;
            pw[lane] = p;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int d = lane + e * 64;
                if (d < hd) {
                    float a = o[j][e] * alpha;
                    for (int key = 0; key < KB; ++key) a = fmaf(pw[key], Vs[key * hp + d], a);
                    o[j][e] = a;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
#pragma unroll
    for (int j = 0; j < QW; ++j) {
        const int qi = q0 + wave * QW + j;
        if (qi >= Tn) continue;
        const float inv = 1.0f / l[j];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int d = lane + e * 64;
            if (d < hd) occ_store_f32(out + ((size_t)b * Tn + qi) * ld_out + (size_t)h * hd + d, o[j][e] * inv);
        }
    }
}


// ------------------------------------------------------------------------------------------------
// bf16 MFMA attention for short sequences (T <= 32*NP <= 256, head_dim 64): one workgroup per
// (batch, head, 64-query block), one wave per 16 queries.  K of the head sits in LDS as swizzled
// 128-byte rows, V transposed ([d][key]) so both MFMA operands are 8/16-byte LDS reads.
//   S^T = K.Q^T  (v_mfma_f32_16x16x32_bf16, A = K rows, B = Q): a lane holds 4 consecutive keys of ONE
//   query, so the softmax row reduction is registers + two xor-shuffles, and the packed probabilities
//   are already the A operand of O = P.V with the key slots permuted identically on the V^T operand.
typedef __attribute__((ext_vector_type(4))) float af32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 abf16x8;

template <int NP>
__global__ __launch_bounds__(256) void attention_mfma_kernel(const unsigned short* __restrict__ qkv, unsigned short* __restrict__ out,
                                                            int Tn, int H, long long ld_qkv, long long ld_out, float scale, float* __restrict__ lse) {
    constexpr int NK = NP * 32;            // padded key count
    constexpr int VS = NK + 4;             // V^T row stride (elements): 8-byte aligned, bank-spreading
    __shared__ uint4 Ks[NK * 8];
    __shared__ __attribute__((aligned(16))) unsigned short Vt[64 * VS];
    const int D = H * 64;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int qchunk = (((Tn + 15) / 16 + gridDim.y - 1) / gridDim.y) * 16;      // queries of this workgroup: a multiple of 16
    const int q_begin = blockIdx.y * qchunk, q_end = q_begin + qchunk < Tn ? q_begin + qchunk : Tn;
    const unsigned short* base = qkv + (size_t)b * Tn * ld_qkv + (size_t)h * 64;
    {   // all 2*NP loads of this thread are issued before the first LDS write: one exposed memory latency instead of NP
        uint4 kv[NP], vv[NP];
        const int ch = threadIdx.x & 7;
#pragma unroll
        for (int it = 0; it < NP; ++it) {
            const int key = it * 32 + (threadIdx.x >> 3);
            const int kc = key < Tn ? key : Tn - 1;
            kv[it] = *reinterpret_cast<const uint4*>(base + (size_t)kc * ld_qkv + D + ch * 8);
            vv[it] = *reinterpret_cast<const uint4*>(base + (size_t)kc * ld_qkv + 2 * D + ch * 8);
            if (key >= Tn) { kv[it] = make_uint4(0, 0, 0, 0); vv[it] = kv[it]; }
        }
#pragma unroll
        for (int it = 0; it < NP; ++it) {
            const int key = it * 32 + (threadIdx.x >> 3);
            Ks[key * 8 + (ch ^ (key & 7))] = kv[it];
            const unsigned w[4] = {vv[it].x, vv[it].y, vv[it].z, vv[it].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                Vt[(ch * 8 + 2 * e) * VS + key] = (unsigned short)(w[e] & 0xffff);
                Vt[(ch * 8 + 2 * e + 1) * VS + key] = (unsigned short)(w[e] >> 16);
            }
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, g = lane >> 4;
    // K / V of the head are staged once per workgroup; each wave then walks 16-query blocks (staging was ~2/3 of the run time
    // when every 64 queries re-staged the head)
    for (int q0 = q_begin + wave * 16; q0 < q_end; q0 += 64) {
    const int qrow = q0 + fr;
    const int qld = qrow < Tn ? qrow : Tn - 1;
    uint4 qf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) qf[s] = *reinterpret_cast<const uint4*>(base + (size_t)qld * ld_qkv + s * 32 + g * 8);

    af32x4 sc[2 * NP];
#pragma unroll
    for (int t = 0; t < 2 * NP; ++t) {
        sc[t] = (af32x4){0.f, 0.f, 0.f, 0.f};
        const int kr = t * 16 + fr;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint4 kf = Ks[kr * 8 + ((s * 4 + g) ^ (kr & 7))];
            sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<abf16x8*>(&kf), *reinterpret_cast<abf16x8*>(&qf[s]), sc[t], 0, 0, 0);
        }
    }
    const float c = scale * 1.44269504088896340736f;
    float mx = -3.0e38f;
#pragma unroll
    for (int t = 0; t < 2 * NP; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = t * 16 + g * 4 + r;
            const float v = key < Tn ? sc[t][r] * c : -3.0e38f;
            sc[t][r] = v;
            mx = fmaxf(mx, v);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 2 * NP; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = __builtin_amdgcn_exp2f(sc[t][r] - mx);
            sc[t][r] = p;
            sum += p;
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (lse && g == 0 && qrow < Tn) lse[(size_t)bh * Tn + qrow] = mx + __builtin_amdgcn_logf(sum);   // log2-sum-exp2 of the scaled scores (for backward)

    af32x4 oacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt] = (af32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        unsigned pw[4];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            pw[e] = pack_bf16x2(sc[2 * u][2 * e], sc[2 * u][2 * e + 1]);
            pw[2 + e] = pack_bf16x2(sc[2 * u + 1][2 * e], sc[2 * u + 1][2 * e + 1]);
        }
        uint4 pf = make_uint4(pw[0], pw[1], pw[2], pw[3]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const unsigned short* vr = Vt + (dt * 16 + fr) * VS + u * 32 + g * 4;
            const uint2 lo = *reinterpret_cast<const uint2*>(vr);
            const uint2 hi = *reinterpret_cast<const uint2*>(vr + 16);
            uint4 vf = make_uint4(lo.x, lo.y, hi.x, hi.y);
            oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<abf16x8*>(&pf), *reinterpret_cast<abf16x8*>(&vf), oacc[dt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ql = g * 4 + r;
        const float iv = __shfl(inv, ql, 64);
        const int q = q0 + ql;
        if (q < Tn) {
            unsigned short* orow = out + ((size_t)b * Tn + q) * ld_out + (size_t)h * 64 + fr;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) orow[dt * 16] = f32_to_bf16_bits(oacc[dt][r] * iv);
        }
    }
    }
}

// Long sequences (evaluation utterances: T is whatever the audio gives): the same fragment scheme, keys streamed in blocks of
// 128 through LDS with the online-softmax recurrence (running max m and sum l per query; the O accumulators are rescaled by
// 2^(m_old - m_new) when a block raises the max).  The reference scores un-padded, un-masked batch-1 utterances
// (oc_classifier.py:185-193); nothing here depends on T except the loop count.
// HD = 64 (XLS-R-300M) or 80 (XLS-R-1B: 1280 / 16 heads): 80 = 2.5 MFMA k-steps -> the third Q fragment is half zero, K rows are
// 10 chunks of 16 B (a 160-byte row stride is conflict-free for ds_read_b128 as it stands, no swizzle), O has 5 column blocks.
typedef __attribute__((ext_vector_type(4))) short att_s16x4;
typedef __attribute__((address_space(3))) att_s16x4 att_lds_s16x4;
// DROP: fairseq's attention_dropout (MultiheadAttention: attn_probs = dropout(softmax(.))): the keep-mask is a u8 tensor [B*H, T, Tp]
// (Tp = T rounded up to 4, drawn by occ_dropout_ex or injected by a test); the normaliser l stays the sum of the UNdropped
// probabilities, the P.V product uses p * keep / (1 - p_drop).
template <int HD, bool DROP = false>
__global__ __launch_bounds__(256) void attention_mfma_long_kernel(const unsigned short* __restrict__ qkv, unsigned short* __restrict__ out,
                                                                 int Tn, int H, long long ld_qkv, long long ld_out, float scale, float* __restrict__ lse,
                                                                 const unsigned char* __restrict__ keep = nullptr, int Tp = 0, float inv_keep = 1.f) {
    constexpr int NP = 4, NK = NP * 32;
    constexpr int CH = HD / 8, KS = (HD + 31) / 32, NDT = HD / 16;     // 16-byte chunks per row, QK^T k-steps, 16-column blocks of O
    __shared__ uint4 Ks[NK * CH];
    // V stays row-major ([key][d], 160-byte rows: 128 or 160 data bytes); its MFMA fragments (8 keys per lane) are read with the
    // transposing ds_read_b64_tr_b16 -- the 8 consecutive rows a 32-lane half touches fall on disjoint bank octets at this stride
    constexpr int VROW = 160;
    __shared__ __attribute__((aligned(16))) unsigned char Vs[NK * VROW];
    const int D = H * HD;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int q0 = blockIdx.y * 64;
    const unsigned short* base = qkv + (size_t)b * Tn * ld_qkv + (size_t)h * HD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const int qrow = q0 + wave * 16 + fr;
    const bool active = q0 + wave * 16 < Tn;                 // wave-uniform; inactive waves still help staging and hit the barriers
    const int qld = qrow < Tn ? qrow : Tn - 1;
    uint4 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)
        qf[s] = s * 32 + g * 8 < HD ? *reinterpret_cast<const uint4*>(base + (size_t)qld * ld_qkv + s * 32 + g * 8) : make_uint4(0, 0, 0, 0);
    const float c = scale * 1.44269504088896340736f;
    float m_run = -3.0e38f, l_run = 0.f;
    af32x4 oacc[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) oacc[dt] = (af32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < Tn; k0 += NK) {
        __syncthreads();                                     // previous block fully consumed
        for (int idx = threadIdx.x; idx < NK * CH; idx += 256) {
            const int kl = idx / CH, ch = idx - kl * CH, key = k0 + kl;
            uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
            if (key < Tn) {
                kv = *reinterpret_cast<const uint4*>(base + (size_t)key * ld_qkv + D + ch * 8);
                vv = *reinterpret_cast<const uint4*>(base + (size_t)key * ld_qkv + 2 * D + ch * 8);
            }
            Ks[kl * CH + (HD == 64 ? (ch ^ (kl & 7)) : ch)] = kv;
            *reinterpret_cast<uint4*>(Vs + kl * VROW + ch * 16) = vv;
        }
        __syncthreads();
        if (!active) continue;
        af32x4 sc[2 * NP];
#pragma unroll
        for (int t = 0; t < 2 * NP; ++t) {
            sc[t] = (af32x4){0.f, 0.f, 0.f, 0.f};
            const int kr = t * 16 + fr;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int chq = s * 4 + g;
                uint4 kf = chq < CH ? Ks[kr * CH + (HD == 64 ? (chq ^ (kr & 7)) : chq)] : make_uint4(0, 0, 0, 0);
                sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<abf16x8*>(&kf), *reinterpret_cast<abf16x8*>(&qf[s]), sc[t], 0, 0, 0);
            }
        }
        float mx = -3.0e38f;
#pragma unroll
        for (int t = 0; t < 2 * NP; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + t * 16 + g * 4 + r;
                const float v = key < Tn ? sc[t][r] * c : -3.0e38f;
                sc[t][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);                // every block holds at least one real key, so m_new is finite
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 2 * NP; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __builtin_amdgcn_exp2f(sc[t][r] - m_new);
                sc[t][r] = pv;
                sum += pv;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        l_run = l_run * alpha + sum;
        m_run = m_new;
        if constexpr (DROP) {
            const unsigned char* krow = keep + ((size_t)bh * Tn + qld) * Tp + k0 + g * 4;
#pragma unroll
            for (int t = 0; t < 2 * NP; ++t) {
                const unsigned m4 = k0 + t * 16 + g * 4 < Tn ? *reinterpret_cast<const unsigned*>(krow + t * 16) : 0u;      // four consecutive keys of this lane's query
#pragma unroll
                for (int r = 0; r < 4; ++r) sc[t][r] = ((m4 >> (8 * r)) & 0xffu) ? sc[t][r] * inv_keep : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {                        // O rows are queries 4g + r; their factor lives in the lane with fr == 4g + r
            const float a_q = __shfl(alpha, g * 4 + r, 64);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) oacc[dt][r] *= a_q;
        }
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            unsigned pw[4];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                pw[e] = pack_bf16x2(sc[2 * u][2 * e], sc[2 * u][2 * e + 1]);
                pw[2 + e] = pack_bf16x2(sc[2 * u + 1][2 * e], sc[2 * u + 1][2 * e + 1]);
            }
            uint4 pf = make_uint4(pw[0], pw[1], pw[2], pw[3]);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                // k-slots 0-3 <-> keys u*32 + 4g + q, k-slots 4-7 <-> keys u*32 + 16 + 4g + q (the packing of P above); columns dt*16 ..
                const unsigned char* vr = Vs + (u * 32 + 4 * g + (fr >> 2)) * VROW + (dt * 16 + (fr & 3) * 4) * 2;
                const att_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((att_lds_s16x4*)vr);
                const att_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((att_lds_s16x4*)(vr + 16 * VROW));
                const abf16x8 vf = __builtin_bit_cast(abf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<abf16x8*>(&pf), vf, oacc[dt], 0, 0, 0);
            }
        }
    }
    if (!active) return;
    const float inv = 1.0f / l_run;
    if (lse && g == 0 && qrow < Tn) lse[(size_t)bh * Tn + qrow] = m_run + __builtin_amdgcn_logf(l_run);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ql = g * 4 + r;
        const float iv = __shfl(inv, ql, 64);
        const int q = q0 + wave * 16 + ql;
        if (q < Tn) {
            unsigned short* orow = out + ((size_t)b * Tn + q) * ld_out + (size_t)h * HD + fr;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) orow[dt * 16] = f32_to_bf16_bits(oacc[dt][r] * iv);
        }
    }
}

// Whole head on chip, V row-major (the fragment scheme of the kernel above, the V handling of the streaming kernel): one workgroup
// per (batch, head) stages K (swizzled 128-byte rows) and V ([key][d], 160-byte rows, fragments by ds_read_b64_tr_b16) ONCE for all
// NP*32 >= T keys, then its four waves walk the 16-query blocks with a single-pass softmax; the next block's Q rows are fetched under
// the current block's arithmetic.  At T = 199 (NP = 7) the streaming kernel staged every head four times (once per 64 queries) and
// ran two key blocks of 128 with 22 % padding.  LDS 28 + 35 KiB: two workgroups per CU.
template <int NP, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW == 8 ? 4 : 2, NW == 8 ? 4 : 2))) void attention_mfma_head_kernel(const unsigned short* __restrict__ qkv, unsigned short* __restrict__ out,
                                                                    int Tn, int H, long long ld_qkv, long long ld_out, float scale, float* __restrict__ lse) {
    constexpr int NK = NP * 32, VROW = 160;
    __shared__ uint4 Ks[NK * 8];
    __shared__ __attribute__((aligned(16))) unsigned char Vs[NK * VROW];
    const int D = H * 64;
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int qchunk = (((Tn + 15) / 16 + gridDim.y - 1) / gridDim.y) * 16;
    const int q_begin = blockIdx.y * qchunk, q_end = q_begin + qchunk < Tn ? q_begin + qchunk : Tn;
    const unsigned short* base = qkv + (size_t)b * Tn * ld_qkv + (size_t)h * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, g = lane >> 4;
    int q0 = q_begin + wave * 16;
    uint4 qf[2];
    {
        const int qr = q0 + fr < Tn ? q0 + fr : Tn - 1;
#pragma unroll
        for (int s = 0; s < 2; ++s) qf[s] = *reinterpret_cast<const uint4*>(base + (size_t)qr * ld_qkv + s * 32 + g * 8);
    }
    {   // all loads of this thread are issued before the first LDS write: one exposed memory latency
        constexpr int KPP = 8 * NW, NIT = (NK + KPP - 1) / KPP;        // keys per pass, passes
        uint4 kv[NIT], vv[NIT];
        const int ch = threadIdx.x & 7;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int key = it * KPP + (threadIdx.x >> 3);
            const int kc = key < Tn ? key : Tn - 1;
            kv[it] = *reinterpret_cast<const uint4*>(base + (size_t)kc * ld_qkv + D + ch * 8);
            vv[it] = *reinterpret_cast<const uint4*>(base + (size_t)kc * ld_qkv + 2 * D + ch * 8);
            if (key >= Tn) { kv[it] = make_uint4(0, 0, 0, 0); vv[it] = kv[it]; }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int key = it * KPP + (threadIdx.x >> 3);
            if (key < NK) {
                Ks[key * 8 + (ch ^ (key & 7))] = kv[it];
                *reinterpret_cast<uint4*>(Vs + key * VROW + ch * 16) = vv[it];
            }
        }
    }
    __syncthreads();
    const float c = scale * 1.44269504088896340736f;
    for (; q0 < q_end; q0 += 16 * NW) {                    // wave-uniform bounds: EXEC stays full for the transposing reads
        const int qrow = q0 + fr;
        uint4 qn[2] = {qf[0], qf[1]};
        if (q0 + 16 * NW < q_end) {
            const int qr = q0 + 16 * NW + fr < Tn ? q0 + 16 * NW + fr : Tn - 1;
#pragma unroll
            for (int s = 0; s < 2; ++s) qn[s] = *reinterpret_cast<const uint4*>(base + (size_t)qr * ld_qkv + s * 32 + g * 8);
        }
        af32x4 sc[2 * NP];
#pragma unroll
        for (int t = 0; t < 2 * NP; ++t) {
            sc[t] = (af32x4){0.f, 0.f, 0.f, 0.f};
            const int kr = t * 16 + fr;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint4 kf = Ks[kr * 8 + ((s * 4 + g) ^ (kr & 7))];
                sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<abf16x8*>(&kf), *reinterpret_cast<abf16x8*>(&qf[s]), sc[t], 0, 0, 0);
            }
            if (NW == 8 && (t & 1)) __builtin_amdgcn_sched_barrier(0);       // 128-register budget: keep the K fragments of at most two tiles live
        }
        // softmax on the raw scores: p = exp2(c*s - c*max) is one FMA + one v_exp per element (c > 0, so the max commutes with the scale);
        // only the last tiles can hold padded keys, and which ones is wave-uniform
        const int t_pad = Tn >> 4;                         // first 16-key tile that may contain keys >= Tn
        float mx = -3.0e38f;
#pragma unroll
        for (int t = 0; t < 2 * NP; ++t) {
            if (t >= t_pad) {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (t * 16 + g * 4 + r >= Tn) sc[t][r] = -3.0e38f;
            }
            mx = fmaxf(mx, fmaxf(fmaxf(sc[t][0], sc[t][1]), fmaxf(sc[t][2], sc[t][3])));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float nmc = -mx * c;
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 2 * NP; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(sc[t][r], c, nmc));      // masked: -3e38*c + nmc -> exp2 = 0
                sc[t][r] = pv;
                sum += pv;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        if (lse && g == 0 && qrow < Tn) lse[(size_t)bh * Tn + qrow] = mx * c + __builtin_amdgcn_logf(sum);
        af32x4 oacc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oacc[dt] = (af32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            unsigned pw[4];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                pw[e] = pack_bf16x2(sc[2 * u][2 * e], sc[2 * u][2 * e + 1]);
                pw[2 + e] = pack_bf16x2(sc[2 * u + 1][2 * e], sc[2 * u + 1][2 * e + 1]);
            }
            uint4 pf = make_uint4(pw[0], pw[1], pw[2], pw[3]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const unsigned char* vr = Vs + (u * 32 + 4 * g + (fr >> 2)) * VROW + (dt * 16 + (fr & 3) * 4) * 2;
                const att_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((att_lds_s16x4*)vr);
                const att_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((att_lds_s16x4*)(vr + 16 * VROW));
                const abf16x8 vf = __builtin_bit_cast(abf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<abf16x8*>(&pf), vf, oacc[dt], 0, 0, 0);
            }
            if (NW == 8) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ql = g * 4 + r;
            const float iv = __shfl(inv, ql, 64);
            const int q = q0 + ql;
            if (q < Tn) {
                unsigned short* orow = out + ((size_t)b * Tn + q) * ld_out + (size_t)h * 64 + fr;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) orow[dt * 16] = f32_to_bf16_bits(oacc[dt][r] * iv);
            }
        }
        qf[0] = qn[0]; qf[1] = qn[1];
    }
}

template <int NP>
void launch_attention_head(const void* qkv, void* out, int B, int T, int H, long long ld_qkv, long long ld_out, float scale, float* lse, hipStream_t s) {
    int ysplit = 1;
    while ((long long)B * H * ysplit < 512 && ysplit * 64 < T) ysplit *= 2;
    // eight waves per head (16 per CU, 114 registers with the fragment live ranges fenced) measured 20.8 us against 21.9 us for four at B = 32, T = 199
    static const int nw_env = getenv("OCC_ATT_WAVES") ? atoi(getenv("OCC_ATT_WAVES")) : 8;
    if (nw_env != 4)
        hipLaunchKernelGGL((attention_mfma_head_kernel<NP, 8>), dim3((unsigned)(B * H), (unsigned)ysplit), dim3(512), 0, s, (const unsigned short*)qkv, (unsigned short*)out, T, H,
                           ld_qkv, ld_out, scale, lse);
    else
        hipLaunchKernelGGL((attention_mfma_head_kernel<NP, 4>), dim3((unsigned)(B * H), (unsigned)ysplit), dim3(256), 0, s, (const unsigned short*)qkv, (unsigned short*)out, T, H,
                           ld_qkv, ld_out, scale, lse);
}

template <int NP>
void launch_attention_mfma(const void* qkv, void* out, int B, int T, int H, long long ld_qkv, long long ld_out, float scale, float* lse, hipStream_t s) {
    // one workgroup per head when there are enough heads to fill the chip (2 workgroups per CU fit), else split the queries
    int ysplit = 1;
    while ((long long)B * H * ysplit < 512 && ysplit * 64 < T) ysplit *= 2;
    const dim3 grid((unsigned)(B * H), (unsigned)ysplit), block(256);
    hipLaunchKernelGGL(attention_mfma_kernel<NP>, grid, block, 0, s, (const unsigned short*)qkv, (unsigned short*)out, T, H, ld_qkv, ld_out, scale, lse);
}

}  // namespace

// f32-arithmetic attention (storage f32 or bf16): the whole-head kernel while K and V of a head fit in LDS, the streaming kernel beyond
// (any T).  kv_len (optional, device int32 [B]): valid frames per utterance of a zero-padded batch.
int occ_attention_f32_mfma_launch(const float* qkv, float* out, long long B, long long T, long long H, long long hd, long long ld_qkv, long long ld_out, float scale,
                                  const int* kv_len, hipStream_t s);       // attention_f32.hip

static int attention_f32_arith(const void* qkv, void* out, int dtype, int64_t B, int64_t T, int64_t H, int64_t hd, int64_t ld_qkv, int64_t ld_out,
                               float scale, const int32_t* kv_len, hipStream_t s) {
    if (dtype != OCC_F32 && dtype != OCC_BF16) { occ_set_error("occ_attention: dtype must be f32 or bf16"); return OCC_EUNSUPPORTED; }
    // f32 storage, head dims 64 / 80 (XLS-R), 16-byte aligned rows: the f32 matrix cores (OCC_ATTN_F32_MFMA=0: the VALU kernels below)
    static const int f32_mfma = getenv("OCC_ATTN_F32_MFMA") ? atoi(getenv("OCC_ATTN_F32_MFMA")) : 1;
    if (f32_mfma && dtype == OCC_F32 && (hd == 64 || hd == 80) && ld_qkv % 4 == 0 && ld_out % 4 == 0 && ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0) {
        occ_attention_f32_mfma_launch((const float*)qkv, (float*)out, B, T, H, hd, ld_qkv, ld_out, scale, kv_len, s);
        OCC_LAUNCH_CHECK("occ_attention(f32 mfma)");
        return OCC_OK;
    }
    const size_t shm = ((size_t)2 * T * (hd + 1) + 4 * T + 4 * hd) * sizeof(float);
    hipError_t e;
    if (shm <= 160 * 1024) {
        const int qsplit = 4;
        const dim3 grid((unsigned)(B * H), qsplit), block(256);
        if (dtype == OCC_F32) {
            e = hipFuncSetAttribute((const void*)attention_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
            if (e != hipSuccess) { occ_set_error("occ_attention: cannot raise LDS limit: %s", hipGetErrorString(e)); return OCC_ELAUNCH; }
            hipLaunchKernelGGL(attention_kernel<float>, grid, block, shm, s, (const float*)qkv, (float*)out, (int)T, (int)H, (int)hd, (long long)ld_qkv, (long long)ld_out, scale, qsplit, kv_len);
        } else {
            e = hipFuncSetAttribute((const void*)attention_kernel<unsigned short>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
            if (e != hipSuccess) { occ_set_error("occ_attention: cannot raise LDS limit: %s", hipGetErrorString(e)); return OCC_ELAUNCH; }
            hipLaunchKernelGGL(attention_kernel<unsigned short>, grid, block, shm, s, (const unsigned short*)qkv, (unsigned short*)out, (int)T, (int)H, (int)hd, (long long)ld_qkv, (long long)ld_out, scale, qsplit, kv_len);
        }
        OCC_LAUNCH_CHECK("occ_attention");
        return OCC_OK;
    }
    const size_t shm2 = ((size_t)2 * 64 * (hd + 1) + 32 * hd + 4 * 64) * sizeof(float);
    const dim3 grid((unsigned)(B * H), (unsigned)((T + 31) / 32)), block(256);
    if (dtype == OCC_F32) {
        e = hipFuncSetAttribute((const void*)attention_stream_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm2);
        if (e != hipSuccess) { occ_set_error("occ_attention: cannot raise LDS limit: %s", hipGetErrorString(e)); return OCC_ELAUNCH; }
        hipLaunchKernelGGL(attention_stream_kernel<float>, grid, block, shm2, s, (const float*)qkv, (float*)out, (int)T, (int)H, (int)hd, (long long)ld_qkv, (long long)ld_out, scale, kv_len);
    } else {
        e = hipFuncSetAttribute((const void*)attention_stream_kernel<unsigned short>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm2);
        if (e != hipSuccess) { occ_set_error("occ_attention: cannot raise LDS limit: %s", hipGetErrorString(e)); return OCC_ELAUNCH; }
        hipLaunchKernelGGL(attention_stream_kernel<unsigned short>, grid, block, shm2, s, (const unsigned short*)qkv, (unsigned short*)out, (int)T, (int)H, (int)hd, (long long)ld_qkv, (long long)ld_out, scale, kv_len);
    }
    OCC_LAUNCH_CHECK("occ_attention(stream)");
    return OCC_OK;
}

int occ_conv0_mfma_launch(const float* wav, const float* w, const float* bias, const float* gamma, const float* beta, void* out, long long B, long long L,
                          long long Tout, long long stride, float eps, hipStream_t s);     // conv0_mfma.hip

extern "C" {

int occ_layernorm(const void* x, int x_dtype, void* y, int y_dtype, const float* gamma, const float* beta, int64_t rows, int64_t C,
                  float eps, int gelu, void* stream) {
    OCC_CHECK_ARG(x && y && gamma && beta, "occ_layernorm: null pointer");
    OCC_CHECK_ARG(rows >= 1 && C >= 8 && C % 8 == 0 && C <= 2048, "occ_layernorm: C must be a multiple of 8 in [8,2048] (C=%ld)", (long)C);
    hipStream_t s = (hipStream_t)stream;
    int rc = -1;
    if (x_dtype == OCC_F32 && y_dtype == OCC_F32) rc = launch_layernorm<float, float>(x, y, gamma, beta, rows, (int)C, eps, gelu, s);
    else if (x_dtype == OCC_F32 && y_dtype == OCC_BF16) rc = launch_layernorm<float, unsigned short>(x, y, gamma, beta, rows, (int)C, eps, gelu, s);
    else if (x_dtype == OCC_BF16 && y_dtype == OCC_BF16) rc = launch_layernorm<unsigned short, unsigned short>(x, y, gamma, beta, rows, (int)C, eps, gelu, s);
    else if (x_dtype == OCC_BF16 && y_dtype == OCC_F32) rc = launch_layernorm<unsigned short, float>(x, y, gamma, beta, rows, (int)C, eps, gelu, s);
    if (rc != 0) { occ_set_error("occ_layernorm: unsupported dtype pair %d -> %d", x_dtype, y_dtype); return OCC_EUNSUPPORTED; }
    OCC_LAUNCH_CHECK("occ_layernorm");
    return OCC_OK;
}

int occ_layernorm_fp8(const float* x, void* y_bf16, void* y_f8, const float* f8_scale, float* f8_amax, const float* gamma, const float* beta, int64_t rows,
                      int64_t C, float eps, void* stream) {
    OCC_CHECK_ARG(x && y_bf16 && y_f8 && f8_scale && f8_amax && gamma && beta, "occ_layernorm_fp8: null pointer");
    OCC_CHECK_ARG(rows >= 1 && C >= 8 && C % 8 == 0 && C <= 2048, "occ_layernorm_fp8: C must be a multiple of 8 in [8,2048] (C=%ld)", (long)C);
    static const long long ln8_blocks = getenv("OCC_LN8_BLOCKS") ? atoll(getenv("OCC_LN8_BLOCKS")) : 1024;      // 4 resident workgroups of 4 rows per CU
    long long nblk = occ_cdiv(rows, 4);
    if (nblk > ln8_blocks) nblk = ln8_blocks;
    const dim3 grid((unsigned)nblk), block(256);
    hipStream_t s = (hipStream_t)stream;
#define OCC_LN8(N) hipLaunchKernelGGL((layernorm_kernel<float, unsigned short, N, true>), grid, block, 0, s, x, (unsigned short*)y_bf16, gamma, beta, (long long)rows, (int)C, eps, 0, (unsigned char*)y_f8, f8_scale, f8_amax)
    switch ((C + 511) / 512) { case 1: OCC_LN8(1); break; case 2: OCC_LN8(2); break; case 3: OCC_LN8(3); break; default: OCC_LN8(4); }
#undef OCC_LN8
    OCC_LAUNCH_CHECK("occ_layernorm_fp8");
    return OCC_OK;
}

int occ_conv0_ln_gelu(const float* wav, const float* w, const float* bias, const float* gamma, const float* beta, void* out, int out_dtype,
                      int64_t B, int64_t L, int64_t Tout, int64_t C, int64_t k, int64_t stride, float eps, void* stream) {
    OCC_CHECK_ARG(wav && w && bias && gamma && beta && out, "occ_conv0_ln_gelu: null pointer");
    OCC_CHECK_ARG(C == 512, "occ_conv0_ln_gelu: C must be 512 (got %ld)", (long)C);
    OCC_CHECK_ARG(k >= 1 && k <= C0_MAXK && stride >= 1 && stride <= 16, "occ_conv0_ln_gelu: k in [1,16], stride in [1,16]");
    OCC_CHECK_ARG(B >= 1 && B < 65536 && L >= k && Tout == (L - k) / stride + 1, "occ_conv0_ln_gelu: Tout must equal (L-k)/stride+1");
    hipStream_t s = (hipStream_t)stream;
    // bf16 output of the XLS-R geometry: the matrix-core form (conv0_mfma.hip: split-operand bf16 MFMA, f32-grade products).  The f32
    // parity path and other tap counts keep the VALU kernel below.  OCC_CONV0_MFMA=0 switches back (A/B).
    static const int c0_mfma = getenv("OCC_CONV0_MFMA") ? atoi(getenv("OCC_CONV0_MFMA")) : 1;
    if (c0_mfma && out_dtype == OCC_BF16 && k == 10 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (reinterpret_cast<uintptr_t>(gamma) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(beta) & 15) == 0) {
        occ_conv0_mfma_launch(wav, w, bias, gamma, beta, out, B, L, Tout, stride, eps, s);
        OCC_LAUNCH_CHECK("occ_conv0_ln_gelu(mfma)");
        return OCC_OK;
    }
    const dim3 grid((unsigned)occ_cdiv(Tout, C0_FRAMES), (unsigned)B), block(256);
    const int kt = k <= 10 ? 10 : C0_MAXK;
    const size_t shm = ((C0_FRAMES - 1) * stride + kt) * sizeof(float);
#define OCC_C0_LAUNCH(TO, KT) hipLaunchKernelGGL((conv0_ln_gelu_kernel<TO, KT>), grid, block, shm, s, wav, w, bias, gamma, beta, (TO*)out, (int)L, (int)Tout, (int)k, (int)stride, eps)
    if (out_dtype == OCC_F32) { if (kt == 10) OCC_C0_LAUNCH(float, 10); else OCC_C0_LAUNCH(float, C0_MAXK); }
    else if (out_dtype == OCC_BF16) { if (kt == 10) OCC_C0_LAUNCH(unsigned short, 10); else OCC_C0_LAUNCH(unsigned short, C0_MAXK); }
    else { occ_set_error("occ_conv0_ln_gelu: out dtype must be f32 or bf16"); return OCC_EUNSUPPORTED; }
#undef OCC_C0_LAUNCH
    OCC_LAUNCH_CHECK("occ_conv0_ln_gelu");
    return OCC_OK;
}

int occ_attention_dropout(const void* qkv, void* out, int64_t B, int64_t T, int64_t H, int64_t hd, int64_t ld_qkv, int64_t ld_out, float scale, float* lse,
                          const uint8_t* keep, float p, void* stream) {
    OCC_CHECK_ARG(qkv && out && keep, "occ_attention_dropout: null pointer");
    OCC_CHECK_ARG(B >= 1 && T >= 1 && H >= 1 && (hd == 64 || hd == 80) && T <= (1 << 20), "occ_attention_dropout: head_dim must be 64 or 80");
    OCC_CHECK_ARG(ld_qkv >= 3 * H * hd && ld_out >= H * hd && ld_qkv % 8 == 0 && ld_out % 8 == 0 && (reinterpret_cast<uintptr_t>(qkv) & 15) == 0, "occ_attention_dropout: leading dimensions / alignment");
    OCC_CHECK_ARG(p >= 0.f && p < 1.f && (reinterpret_cast<uintptr_t>(keep) & 3) == 0, "occ_attention_dropout: p must be in [0, 1), the mask 4-byte aligned");
    const int Tp = (int)((T + 3) / 4 * 4);
    const dim3 grid((unsigned)(B * H), (unsigned)((T + 63) / 64)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (hd == 64)
        hipLaunchKernelGGL((attention_mfma_long_kernel<64, true>), grid, block, 0, s, (const unsigned short*)qkv, (unsigned short*)out, (int)T, (int)H, (long long)ld_qkv,
                           (long long)ld_out, scale, lse, (const unsigned char*)keep, Tp, 1.0f / (1.0f - p));
    else
        hipLaunchKernelGGL((attention_mfma_long_kernel<80, true>), grid, block, 0, s, (const unsigned short*)qkv, (unsigned short*)out, (int)T, (int)H, (long long)ld_qkv,
                           (long long)ld_out, scale, lse, (const unsigned char*)keep, Tp, 1.0f / (1.0f - p));
    OCC_LAUNCH_CHECK("occ_attention_dropout");
    return OCC_OK;
}

int occ_attention(const void* qkv, void* out, int dtype, int64_t B, int64_t T, int64_t H, int64_t hd, int64_t ld_qkv, int64_t ld_out,
                  float scale, float* lse, void* stream) {
    OCC_CHECK_ARG(qkv && out, "occ_attention: null pointer");
    OCC_CHECK_ARG(B >= 1 && T >= 1 && H >= 1 && hd >= 8 && hd <= 128, "occ_attention: bad shape");
    OCC_CHECK_ARG(T <= (1 << 20), "occ_attention: T=%ld is not a plausible frame count", (long)T);
    OCC_CHECK_ARG(ld_qkv >= 3 * H * hd && ld_out >= H * hd, "occ_attention: leading dimensions too small");
    hipStream_t s = (hipStream_t)stream;
    // T > 128: the streaming kernel (98 VGPRs, 33 KiB LDS) beats holding all keys on chip (384 registers at T = 199: one wave per
    // SIMD) -- 27.1 vs 29.4 us at B = 32, T = 199, 16 heads
    if (dtype == OCC_BF16 && hd == 64 && T <= 128 && ld_qkv % 8 == 0 && (reinterpret_cast<uintptr_t>(qkv) & 15) == 0) {
        const int np = (int)((T + 31) / 32);
        if (np <= 1) launch_attention_mfma<1>(qkv, out, (int)B, (int)T, (int)H, ld_qkv, ld_out, scale, lse, s);
        else if (np <= 2) launch_attention_mfma<2>(qkv, out, (int)B, (int)T, (int)H, ld_qkv, ld_out, scale, lse, s);
        else if (np <= 4) launch_attention_mfma<4>(qkv, out, (int)B, (int)T, (int)H, ld_qkv, ld_out, scale, lse, s);
        else if (np <= 7) launch_attention_mfma<7>(qkv, out, (int)B, (int)T, (int)H, ld_qkv, ld_out, scale, lse, s);
        else launch_attention_mfma<8>(qkv, out, (int)B, (int)T, (int)H, ld_qkv, ld_out, scale, lse, s);
        OCC_LAUNCH_CHECK("occ_attention(mfma)");
        return OCC_OK;
    }
    static const int att_head = getenv("OCC_ATT_HEAD") ? atoi(getenv("OCC_ATT_HEAD")) : 1;
    if (att_head && dtype == OCC_BF16 && hd == 64 && T > 128 && T <= 224 && ld_qkv % 8 == 0 && (reinterpret_cast<uintptr_t>(qkv) & 15) == 0) {
        // 128 < T <= 224 (the 4 s training utterances: T = 199): whole head on chip, staged once
        if (T <= 160) launch_attention_head<5>(qkv, out, (int)B, (int)T, (int)H, ld_qkv, ld_out, scale, lse, s);
        else if (T <= 192) launch_attention_head<6>(qkv, out, (int)B, (int)T, (int)H, ld_qkv, ld_out, scale, lse, s);
        else launch_attention_head<7>(qkv, out, (int)B, (int)T, (int)H, ld_qkv, ld_out, scale, lse, s);
        OCC_LAUNCH_CHECK("occ_attention(mfma, head)");
        return OCC_OK;
    }
    if (dtype == OCC_BF16 && (hd == 64 || hd == 80) && ld_qkv % 8 == 0 && ld_out % 8 == 0 && (reinterpret_cast<uintptr_t>(qkv) & 15) == 0) {   // any T: keys streamed in blocks
        const dim3 grid((unsigned)(B * H), (unsigned)((T + 63) / 64)), block(256);
        if (hd == 64)
            hipLaunchKernelGGL(attention_mfma_long_kernel<64>, grid, block, 0, s, (const unsigned short*)qkv, (unsigned short*)out, (int)T, (int)H, (long long)ld_qkv,
                               (long long)ld_out, scale, lse);
        else
            hipLaunchKernelGGL(attention_mfma_long_kernel<80>, grid, block, 0, s, (const unsigned short*)qkv, (unsigned short*)out, (int)T, (int)H, (long long)ld_qkv,
                               (long long)ld_out, scale, lse);
        OCC_LAUNCH_CHECK("occ_attention(mfma, long)");
        return OCC_OK;
    }
    OCC_CHECK_ARG(!lse, "occ_attention: the log-sum-exp output needs the bf16 / head_dim 64 MFMA path");
    return attention_f32_arith(qkv, out, dtype, B, T, H, hd, ld_qkv, ld_out, scale, nullptr, s);
}

int occ_attention_varlen(const void* qkv, void* out, int dtype, int64_t B, int64_t T, int64_t H, int64_t hd, int64_t ld_qkv, int64_t ld_out,
                         float scale, const int32_t* kv_len, void* stream) {
    OCC_CHECK_ARG(qkv && out && kv_len, "occ_attention_varlen: null pointer");
    OCC_CHECK_ARG(B >= 1 && T >= 1 && H >= 1 && hd >= 8 && hd <= 128 && T <= (1 << 20), "occ_attention_varlen: bad shape");
    OCC_CHECK_ARG(ld_qkv >= 3 * H * hd && ld_out >= H * hd, "occ_attention_varlen: leading dimensions too small");
    return attention_f32_arith(qkv, out, dtype, B, T, H, hd, ld_qkv, ld_out, scale, kv_len, (hipStream_t)stream);
}

}  // extern "C"
