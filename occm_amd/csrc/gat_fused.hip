// Fused attention core of AASIST's graph layers (GraphAttentionLayer sslassist.py:58-151, HtrgGraphAttentionLayer :154-329):
//     z[i,j,:] = tanh(W (x_i o x_j) + b),  score[i,j] = aw_type(i,j) . z[i,j,:] / temp,  alpha[i,:] = softmax_j(score[i,:]),  h_i = sum_j alpha[i,j] x_j
// and its backward, without the [B,N,N,D] pairwise-product tensor and the [B,N,N,Do] projection the unfused kernels (occ_pair_mul,
// occ_gemm, occ_gat_softmax, occ_gat_dz, occ_gemm_tn, occ_pair_mul_bwd) write to HBM and read back: 1.1 MB per utterance and tensor at
// N = 66.  One wave owns a "pivot" node p and walks the other nodes q in tiles of 16:
//     Z_p[q, o] = sum_d (x_q[d] x_p[d]) W[o, d]        bf16 MFMA 16x16x32, A = the products rounded to bf16 (as the unfused bf16-compute
//                                                       mode rounds the staged pair tensor), B = rows of W from LDS, f32 accumulate
// Forward: scores -> LDS row -> softmax -> alpha row (kept for backward) and h_p.
// Backward (needs ds = d loss / d score [B,N,N] from occ_gat_dscore): z is recomputed; z[p,q,:] = z[q,p,:], so ONE recomputation serves
// the pivot as row index (ds[p,q]) and as column index (ds[q,p]):
//     c[q,o] = aw_type[o] (1 - z^2);   dzR = ds[p,q] c  (row role);   dzS = (ds[p,q] + ds[q,p]) c
//     dx_p  += sum_q ((dzS W)[q,:] o x_q)                      second MFMA product, no scatter: every dx row has one writer
//     dW    += sum_q dzR[q,:]^T (x_q o x_p),  db += sum_q dzR[q,:],  daw_type += sum_q ds[p,q] z[p,q,:]      third MFMA product; float atomics at
//                                                                                                  the end, as the unfused occ_gat_dz
// The remaining term of d loss / dx (alpha^T dh) stays with occ_bmm_alpha.  bf16-compute mode only: the exact-f32 mode keeps the unfused path.
#include "occ_common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(4))) float gf32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 gbf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned gu32x4;

__device__ __forceinline__ unsigned gpack2(float a, float b) { return pack_bf16x2(a, b); }
__device__ __forceinline__ float row16_sum(float v) {          // over the 16 lanes of a lane row (lanes with equal lane >> 4)
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
    return v;
}
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#define GAT_MFMA(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(gbf16x8, A), __builtin_bit_cast(gbf16x8, B), ACC, 0, 0, 0)

// tanh through one v_exp and one v_rcp: 1 - 2 / (e^(2x) + 1), |absolute error| ~1e-7 (libm tanhf is ~80 instructions with range branches
// and was 80 % of both kernels' run time: 48 evaluations per lane and 32 nodes)
__device__ __forceinline__ float fast_tanh(float v) {
    const float c = fminf(fmaxf(v, -15.f), 15.f);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(c * 2.8853900817779268f) + 1.f);
}

__device__ __forceinline__ int pair_type(int i, int j, int n1) { return (i < n1) ? ((j < n1) ? 0 : 2) : ((j < n1) ? 2 : 1); }

// A fragment of the pair tensor: row q (this lane's fr), elements d = d0 .. d0+7, times the pivot's values, rounded to bf16
__device__ __forceinline__ gu32x4 pair_frag(const float* xq, const float (&xp)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(xq), b = *reinterpret_cast<const float4*>(xq + 4);
    return (gu32x4){gpack2(a.x * xp[0], a.y * xp[1]), gpack2(a.z * xp[2], a.w * xp[3]), gpack2(b.x * xp[4], b.y * xp[5]), gpack2(b.z * xp[6], b.w * xp[7])};
}

template <int D, int DO>
struct GatLds {
    static constexpr int XS = D + 4;            // f32 row stride of the node features
    static constexpr int WS = D + 8;            // bf16 row stride of W  [DO][D]
    static constexpr int OS = DO + 8;           // bf16 row stride of W^T [D][DO] and of the dzS rows [q][DO]
};

// grid (ceil(N / (4 * pl)), B), 256 threads; dynamic LDS: Xf [N][XS] f32 | Wb [DO][WS] bf16 | bias [DO] | aw [3][DO] | srow [4][NP] f32
template <int D, int DO>
__global__ __launch_bounds__(256) void gat_core_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                                                           const float* __restrict__ aw3, float* __restrict__ alpha, float* __restrict__ h, int N, int n1,
                                                           float inv_temp, int pl) {
    using L = GatLds<D, DO>;
    constexpr int KS = D / 32, OT = DO / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    const int NP = (N + 15) & ~15;
    float* Xf = reinterpret_cast<float*>(gsm);
    unsigned short* Wb = reinterpret_cast<unsigned short*>(Xf + (size_t)N * L::XS);
    float* bs = reinterpret_cast<float*>(Wb + DO * L::WS);
    float* aw = bs + DO;
    float* srow_all = aw + 3 * DO;
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const float* xb = x + (size_t)b * N * D;
    for (int i = tid; i < N * (D / 4); i += 256) {
        const int r = i / (D / 4), c = i - r * (D / 4);
        *reinterpret_cast<float4*>(Xf + r * L::XS + c * 4) = *reinterpret_cast<const float4*>(xb + (size_t)r * D + c * 4);
    }
    for (int i = tid; i < DO * (D / 2); i += 256) {
        const int o = i / (D / 2), c = i - o * (D / 2);
        const float2 w2 = *reinterpret_cast<const float2*>(W + (size_t)o * D + c * 2);
        *reinterpret_cast<unsigned*>(Wb + o * L::WS + c * 2) = gpack2(w2.x, w2.y);
    }
    for (int i = tid; i < DO; i += 256) bs[i] = bias[i];
    for (int i = tid; i < 3 * DO; i += 256) aw[i] = aw3[i];
    __syncthreads();
    float* srow = srow_all + wave * NP;
    const int NT = NP >> 4;
    for (int pp = 0; pp < pl; ++pp) {
        const int p = (blockIdx.x * 4 + wave) * pl + pp;
        if (p >= N) break;                                       // wave-uniform
        float xp[KS][8];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float4 a = *reinterpret_cast<const float4*>(Xf + p * L::XS + ks * 32 + fq * 8), c = *reinterpret_cast<const float4*>(Xf + p * L::XS + ks * 32 + fq * 8 + 4);
            xp[ks][0] = a.x; xp[ks][1] = a.y; xp[ks][2] = a.z; xp[ks][3] = a.w; xp[ks][4] = c.x; xp[ks][5] = c.y; xp[ks][6] = c.z; xp[ks][7] = c.w;
        }
        for (int qt = 0; qt < NT; ++qt) {
            const int q = qt * 16 + fr, qc = q < N ? q : N - 1;
            gf32x4 acc[OT];
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) acc[ot] = (gf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const gu32x4 af = pair_frag(Xf + qc * L::XS + ks * 32 + fq * 8, xp[ks]);
#pragma unroll
                for (int ot = 0; ot < OT; ++ot) {
                    const gu32x4 bf = *reinterpret_cast<const gu32x4*>(Wb + (ot * 16 + fr) * L::WS + ks * 32 + fq * 8);
                    GAT_MFMA(acc[ot], af, bf);
                }
            }
            // acc[ot][r]: node q = qt*16 + 4*fq + r, output o = ot*16 + fr
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qr = qt * 16 + 4 * fq + r;
                const float* awt = aw + pair_type(p, qr, n1) * DO;
                float part = 0.f;
#pragma unroll
                for (int ot = 0; ot < OT; ++ot) part += awt[ot * 16 + fr] * fast_tanh(acc[ot][r] + bs[ot * 16 + fr]);
                part = row16_sum(part);
                if (fr == 0) srow[qr] = qr < N ? part * inv_temp : -3.0e38f;
            }
        }
        wave_lds_fence();
        float mx = -3.0e38f;
        for (int q = lane; q < N; q += 64) mx = fmaxf(mx, srow[q]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int q = lane; q < N; q += 64) { const float e = expf(srow[q] - mx); srow[q] = e; sum += e; }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        float* arow = alpha + ((size_t)b * N + p) * N;
        for (int q = lane; q < N; q += 64) { const float a = srow[q] * inv; srow[q] = a; arow[q] = a; }
        wave_lds_fence();
        // h_p[d] = sum_q alpha[q] x_q[d]: lanes over d (D = 32: the two lane halves take even / odd q)
        constexpr int HALVES = 64 / D;
        const int d = lane % D, part_id = lane / D;
        float hv = 0.f;
        for (int q = part_id; q < N; q += HALVES) hv += srow[q] * Xf[q * L::XS + d];
        if (HALVES == 2) hv += __shfl_xor(hv, 32, 64);
        if (lane < D) h[((size_t)b * N + p) * D + d] = hv;
        wave_lds_fence();                                        // srow is rewritten by the next pivot
    }
}

// grid (ceil(N / (8 * pl)), B), 512 threads (eight waves, one pivot at a time each); dynamic LDS:
//   Xf [N][XS] f32 | XT [D][QS] bf16 (x transposed, QS = NP32 + 8) | Wb [DO][WS] bf16 | WT [D][OS] bf16 | bias [DO] | aw [3][DO]
//   per wave: dsr [NP32] f32, dsc [NP32] f32, ZA [32][OS] bf16 (dzS rows of the current 32 nodes), ZB [DO][ZQ] bf16 (their dzR, transposed)
// The nodes q are walked 32 at a time: z, dzS / dzR of the 32 -> LDS -> the two dependent MFMA products of those 32 -> next 32.
template <int D, int DO>
__global__ __launch_bounds__(512) void gat_core_bwd_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                                                           const float* __restrict__ aw3, const float* __restrict__ ds, float* __restrict__ dx,
                                                           float* __restrict__ partial, int N, int n1, int pl) {
    using L = GatLds<D, DO>;
    constexpr int KS = D / 32, OT = DO / 16, DT = D / 16, KO = DO / 32, ZQ = 32 + 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    const int NP32 = (N + 31) & ~31, QS = NP32 + 8, KQ = NP32 >> 5;
    float* Xf = reinterpret_cast<float*>(gsm);
    unsigned short* XT = reinterpret_cast<unsigned short*>(Xf + (size_t)N * L::XS);
    unsigned short* Wb = XT + (size_t)D * QS;
    unsigned short* WT = Wb + DO * L::WS;
    float* bs = reinterpret_cast<float*>(WT + D * L::OS);
    float* aw = bs + DO;
    unsigned char* per_wave = reinterpret_cast<unsigned char*>(aw + 3 * DO);
    const size_t wave_bytes = (size_t)2 * NP32 * 4 + (size_t)32 * L::OS * 2 + (size_t)DO * ZQ * 2;
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const float* xb = x + (size_t)b * N * D;
    for (int i = tid; i < N * (D / 4); i += 512) {
        const int r = i / (D / 4), c = i - r * (D / 4);
        *reinterpret_cast<float4*>(Xf + r * L::XS + c * 4) = *reinterpret_cast<const float4*>(xb + (size_t)r * D + c * 4);
    }
    for (int i = tid; i < DO * D; i += 512) {
        const int o = i / D, d = i - o * D;
        const unsigned short wv = f32_to_bf16_bits(W[i]);
        Wb[o * L::WS + d] = wv; WT[d * L::OS + o] = wv;
    }
    for (int i = tid; i < DO; i += 512) bs[i] = bias[i];
    for (int i = tid; i < 3 * DO; i += 512) aw[i] = aw3[i];
    __syncthreads();
    for (int i = tid; i < D * QS; i += 512) {                    // XT[d][q] = bf16(x[q][d]) from the LDS copy, zero beyond N
        const int d = i / QS, q = i - d * QS;
        XT[i] = q < N ? f32_to_bf16_bits(Xf[q * L::XS + d]) : (unsigned short)0;
    }
    __syncthreads();
    float* dsr = reinterpret_cast<float*>(per_wave + wave * wave_bytes);
    float* dsc = dsr + NP32;
    unsigned short* ZA = reinterpret_cast<unsigned short*>(dsc + NP32);
    unsigned short* ZB = ZA + (size_t)32 * L::OS;
    gf32x4 dWacc[OT][DT];
    float dbacc[OT], dawacc[3][OT];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
        dbacc[ot] = 0.f; dawacc[0][ot] = 0.f; dawacc[1][ot] = 0.f; dawacc[2][ot] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) dWacc[ot][dt] = (gf32x4){0.f, 0.f, 0.f, 0.f};
    }
    const float* dsb = ds + (size_t)b * N * N;
    for (int pp = 0; pp < pl; ++pp) {
        const int p = (blockIdx.x * 8 + wave) * pl + pp;
        if (p >= N) break;                                       // wave-uniform
        for (int q = lane; q < NP32; q += 64) { dsr[q] = q < N ? dsb[(size_t)p * N + q] : 0.f; dsc[q] = q < N ? dsb[(size_t)q * N + p] : 0.f; }
        float xp[KS][8], xpd[DT], dxp[DT];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float4 a = *reinterpret_cast<const float4*>(Xf + p * L::XS + ks * 32 + fq * 8), c = *reinterpret_cast<const float4*>(Xf + p * L::XS + ks * 32 + fq * 8 + 4);
            xp[ks][0] = a.x; xp[ks][1] = a.y; xp[ks][2] = a.z; xp[ks][3] = a.w; xp[ks][4] = c.x; xp[ks][5] = c.y; xp[ks][6] = c.z; xp[ks][7] = c.w;
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { xpd[dt] = Xf[p * L::XS + dt * 16 + fr]; dxp[dt] = 0.f; }
        wave_lds_fence();
        for (int kq = 0; kq < KQ; ++kq) {
            // ---- (1) z of nodes kq*32 .. +31 recomputed; dzS rows -> ZA, dzR transposed -> ZB; bias / att_weight sums
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                const int q = kq * 32 + half * 16 + fr, qc = q < N ? q : N - 1;
                gf32x4 acc[OT];
#pragma unroll
                for (int ot = 0; ot < OT; ++ot) acc[ot] = (gf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const gu32x4 af = pair_frag(Xf + qc * L::XS + ks * 32 + fq * 8, xp[ks]);
#pragma unroll
                    for (int ot = 0; ot < OT; ++ot) {
                        const gu32x4 bf = *reinterpret_cast<const gu32x4*>(Wb + (ot * 16 + fr) * L::WS + ks * 32 + fq * 8);
                        GAT_MFMA(acc[ot], af, bf);
                    }
                }
                float dzr[OT][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ql = half * 16 + 4 * fq + r, qr = kq * 32 + ql;      // node index inside the group of 32 / in the graph
                    const int ty = pair_type(p, qr, n1);
                    const float gr = dsr[qr], gs = gr + dsc[qr];                  // zero for nodes beyond N
#pragma unroll
                    for (int ot = 0; ot < OT; ++ot) {
                        const int o = ot * 16 + fr;
                        const float z = fast_tanh(acc[ot][r] + bs[o]);
                        const float c = aw[ty * DO + o] * (1.f - z * z);
                        dzr[ot][r] = gr * c;
                        dbacc[ot] += gr * c;
                        const float gz = gr * z;
                        dawacc[0][ot] += ty == 0 ? gz : 0.f; dawacc[1][ot] += ty == 1 ? gz : 0.f; dawacc[2][ot] += ty == 2 ? gz : 0.f;
                        ZA[ql * L::OS + o] = f32_to_bf16_bits(gs * c);
                    }
                }
#pragma unroll
                for (int ot = 0; ot < OT; ++ot)
                    *reinterpret_cast<uint2*>(ZB + (ot * 16 + fr) * ZQ + half * 16 + 4 * fq) = make_uint2(gpack2(dzr[ot][0], dzr[ot][1]), gpack2(dzr[ot][2], dzr[ot][3]));
            }
            wave_lds_fence();
            // ---- (2) G = dzS W of the 32 nodes (rows q, columns d);  dx_p[d] += sum_q G[q,d] x_q[d]
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                gu32x4 za[KO];
#pragma unroll
                for (int ko = 0; ko < KO; ++ko) za[ko] = *reinterpret_cast<const gu32x4*>(ZA + (half * 16 + fr) * L::OS + ko * 32 + fq * 8);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    gf32x4 g4 = (gf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ko = 0; ko < KO; ++ko) {
                        const gu32x4 wt = *reinterpret_cast<const gu32x4*>(WT + (dt * 16 + fr) * L::OS + ko * 32 + fq * 8);
                        GAT_MFMA(g4, za[ko], wt);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {                 // G[q = kq*32 + half*16 + 4*fq + r][d = dt*16 + fr]
                        const int qr = kq * 32 + half * 16 + 4 * fq + r;
                        if (qr < N) dxp[dt] += g4[r] * Xf[qr * L::XS + dt * 16 + fr];
                    }
                }
            }
            // ---- (3) dW[o,d] += sum over the 32 nodes of dzR[q,o] x_q[d] x_p[d]: rows o from ZB, "rows" d from x transposed (times the pivot's value)
            {
                gu32x4 zb[OT];
#pragma unroll
                for (int ot = 0; ot < OT; ++ot) zb[ot] = *reinterpret_cast<const gu32x4*>(ZB + (ot * 16 + fr) * ZQ + fq * 8);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const gu32x4 xr = *reinterpret_cast<const gu32x4*>(XT + (size_t)(dt * 16 + fr) * QS + kq * 32 + fq * 8);
                    gu32x4 xf;
#pragma unroll
                    for (int e = 0; e < 4; ++e) xf[e] = gpack2(__uint_as_float(xr[e] << 16) * xpd[dt], __uint_as_float(xr[e] & 0xffff0000u) * xpd[dt]);
#pragma unroll
                    for (int ot = 0; ot < OT; ++ot) GAT_MFMA(dWacc[ot][dt], zb[ot], xf);
                }
            }
            wave_lds_fence();                                    // ZA / ZB are rewritten by the next 32 nodes
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            float v = dxp[dt];
            v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            if (fq == 0) dx[((size_t)b * N + p) * D + dt * 16 + fr] += v;          // every dx row has exactly one writer (its pivot wave)
        }
    }
    // ---- weight gradients of this workgroup's pivots.  Float atomics from every wave onto the same Do*D addresses ran at the contended
    // atomic rate (2560 adders per address: ~150 of the kernel's 200 us); instead the eight waves add up through LDS (pairwise tree),
    // wave 0 stores one partial record per workgroup and gat_bwd_finalize_kernel sums the records in workgroup order (reproducible).
    // record: dW [DO][D] | db [DO] | daw [3][DO];  dWacc[ot][dt][r] = dW[o = ot*16 + 4*fq + r][d = dt*16 + fr]
    constexpr int RW = DO * D + 4 * DO;
    float dbv[OT], dawv[3][OT];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
        float v = dbacc[ot];
        v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        dbv[ot] = v;
#pragma unroll
        for (int t = 0; t < 3; ++t) { float a = dawacc[t][ot]; a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64); dawv[t][ot] = a; }
    }
    __syncthreads();                                             // every wave is done with the LDS images
    float* red = reinterpret_cast<float*>(gsm);                  // [4][RW]
    auto visit = [&](auto&& fn) {                                // v = fn(index in the record, v) over every value this lane owns
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) dWacc[ot][dt][r] = fn((ot * 16 + 4 * fq + r) * D + dt * 16 + fr, dWacc[ot][dt][r]);
            if (fq == 0) {
                dbv[ot] = fn(DO * D + ot * 16 + fr, dbv[ot]);
#pragma unroll
                for (int t = 0; t < 3; ++t) dawv[t][ot] = fn(DO * D + DO + t * DO + ot * 16 + fr, dawv[t][ot]);
            }
        }
    };
    for (int half = 4; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2 * half) visit([&](int idx, float v) { red[(wave - half) * RW + idx] = v; return v; });
        __syncthreads();
        if (wave < half) visit([&](int idx, float v) { return v + red[wave * RW + idx]; });
        __syncthreads();
    }
    if (wave == 0) {
        float* rec = partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * RW;
        visit([&](int idx, float v) { rec[idx] = v; return v; });
    }
}

// d_att_w / d_att_b / d_aw3 += sum over the workgroup records of gat_core_bwd_kernel, in record order
__global__ __launch_bounds__(256) void gat_bwd_finalize_kernel(const float* __restrict__ partial, int nrec, int RW, int DOD, int DO, float* __restrict__ dW,
                                                               float* __restrict__ dbias, float* __restrict__ daw3) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= RW) return;
    float s = 0.f;
    int r = 0;
    for (; r + 8 <= nrec; r += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(r + u) * RW + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; r < nrec; ++r) s += partial[(size_t)r * RW + i];
    if (i < DOD) dW[i] += s;
    else if (i < DOD + DO) dbias[i - DOD] += s;
    else daw3[i - DOD - DO] += s;
}

template <int D, int DO> size_t gat_fwd_lds(int N) {
    using L = GatLds<D, DO>;
    const int NP = (N + 15) & ~15;
    return (size_t)N * L::XS * 4 + (size_t)DO * L::WS * 2 + (size_t)DO * 4 + (size_t)3 * DO * 4 + (size_t)4 * NP * 4;
}
template <int D, int DO> size_t gat_bwd_lds(int N) {
    using L = GatLds<D, DO>;
    const int NP32 = (N + 31) & ~31, QS = NP32 + 8;
    const size_t wave_bytes = (size_t)2 * NP32 * 4 + (size_t)32 * L::OS * 2 + (size_t)DO * 40 * 2;
    const size_t images = (size_t)N * L::XS * 4 + (size_t)D * QS * 2 + (size_t)DO * L::WS * 2 + (size_t)D * L::OS * 2 + (size_t)DO * 4 + (size_t)3 * DO * 4 + 8 * wave_bytes;
    const size_t reduce = (size_t)4 * (DO * D + 4 * DO) * 4;     // the end-of-kernel tree over the eight waves reuses the same LDS
    return images > reduce ? images : reduce;
}

template <int D, int DO>
int launch_fwd(const float* x, const float* W, const float* bias, const float* aw3, float* alpha, float* h, int B, int N, int n1, float inv_temp, hipStream_t s) {
    const size_t shm = gat_fwd_lds<D, DO>(N);
    if (shm > 160 * 1024) { occ_set_error("occ_gat_core_fwd: N=%d needs %zu B of LDS", N, shm); return OCC_EINVAL; }
    hipError_t e = hipFuncSetAttribute((const void*)gat_core_fwd_kernel<D, DO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) { occ_set_error("occ_gat_core_fwd: %s", hipGetErrorString(e)); return OCC_ELAUNCH; }
    const int pl = 2;                                            // pivots per wave: the workgroup's setup (x, W into LDS) amortises over 8 pivots
    hipLaunchKernelGGL((gat_core_fwd_kernel<D, DO>), dim3((unsigned)occ_cdiv(N, 4 * pl), (unsigned)B), dim3(256), shm, s, x, W, bias, aw3, alpha, h, N, n1, inv_temp, pl);
    return OCC_OK;
}
template <int D, int DO>
int launch_bwd(const float* x, const float* W, const float* bias, const float* aw3, const float* ds, float* dx, float* dW, float* db, float* daw, int B, int N, int n1,
               float* ws, long long ws_floats, hipStream_t s) {
    const size_t shm = gat_bwd_lds<D, DO>(N);
    if (shm > 160 * 1024) { occ_set_error("occ_gat_core_bwd: N=%d needs %zu B of LDS", N, shm); return OCC_EINVAL; }
    static const int pl_env = getenv("OCC_GAT_PL") ? atoi(getenv("OCC_GAT_PL")) : 0;
    const int pl = pl_env > 0 ? pl_env : 2;                      // pivots per wave: 16 pivots per workgroup
    const int gx = (int)occ_cdiv(N, 8 * pl), RW = DO * D + 4 * DO;
    if (!ws || ((uintptr_t)ws & 15) || ws_floats < (long long)gx * B * RW) { occ_set_error("occ_gat_core_bwd: needs %lld floats of 16-byte aligned scratch", (long long)gx * B * RW); return OCC_EINVAL; }
    hipError_t e = hipFuncSetAttribute((const void*)gat_core_bwd_kernel<D, DO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) { occ_set_error("occ_gat_core_bwd: %s", hipGetErrorString(e)); return OCC_ELAUNCH; }
    hipLaunchKernelGGL((gat_core_bwd_kernel<D, DO>), dim3((unsigned)gx, (unsigned)B), dim3(512), shm, s, x, W, bias, aw3, ds, dx, ws, N, n1, pl);
    hipLaunchKernelGGL(gat_bwd_finalize_kernel, dim3((unsigned)occ_cdiv(RW, 256)), dim3(256), 0, s, (const float*)ws, gx * B, RW, DO * D, DO, dW, db, daw);
    return OCC_OK;
}

}  // namespace

extern "C" {

int occ_gat_core_fwd(const float* x, const float* att_w, const float* att_b, const float* aw3, float* alpha, float* h, int64_t B, int64_t N, int64_t D, int64_t Do,
                     int64_t n1, float inv_temp, void* stream) {
    OCC_CHECK_ARG(x && att_w && att_b && aw3 && alpha && h, "occ_gat_core_fwd: null pointer");
    OCC_CHECK_ARG(B >= 1 && B < 65536 && N >= 1 && N <= 512 && n1 >= 0 && n1 <= N, "occ_gat_core_fwd: bad shape (B=%ld N=%ld n1=%ld)", (long)B, (long)N, (long)n1);
    int rc;
    hipStream_t s = (hipStream_t)stream;
    if (D == 64 && Do == 64) rc = launch_fwd<64, 64>(x, att_w, att_b, aw3, alpha, h, (int)B, (int)N, (int)n1, inv_temp, s);
    else if (D == 64 && Do == 32) rc = launch_fwd<64, 32>(x, att_w, att_b, aw3, alpha, h, (int)B, (int)N, (int)n1, inv_temp, s);
    else if (D == 32 && Do == 32) rc = launch_fwd<32, 32>(x, att_w, att_b, aw3, alpha, h, (int)B, (int)N, (int)n1, inv_temp, s);
    else { occ_set_error("occ_gat_core_fwd: (D, Do) must be (64,64), (64,32) or (32,32) -- the AASIST layers (got %ld, %ld)", (long)D, (long)Do); return OCC_EUNSUPPORTED; }
    if (rc != OCC_OK) return rc;
    OCC_LAUNCH_CHECK("occ_gat_core_fwd");
    return OCC_OK;
}

int occ_gat_core_bwd(const float* x, const float* att_w, const float* att_b, const float* aw3, const float* ds, float* dx, float* d_att_w, float* d_att_b,
                     float* d_aw3, int64_t B, int64_t N, int64_t D, int64_t Do, int64_t n1, float* ws, int64_t ws_floats, void* stream) {
    OCC_CHECK_ARG(x && att_w && att_b && aw3 && ds && dx && d_att_w && d_att_b && d_aw3, "occ_gat_core_bwd: null pointer");
    OCC_CHECK_ARG(B >= 1 && B < 65536 && N >= 1 && N <= 512 && n1 >= 0 && n1 <= N, "occ_gat_core_bwd: bad shape (B=%ld N=%ld n1=%ld)", (long)B, (long)N, (long)n1);
    int rc;
    hipStream_t s = (hipStream_t)stream;
    if (D == 64 && Do == 64) rc = launch_bwd<64, 64>(x, att_w, att_b, aw3, ds, dx, d_att_w, d_att_b, d_aw3, (int)B, (int)N, (int)n1, ws, (long long)ws_floats, s);
    else if (D == 64 && Do == 32) rc = launch_bwd<64, 32>(x, att_w, att_b, aw3, ds, dx, d_att_w, d_att_b, d_aw3, (int)B, (int)N, (int)n1, ws, (long long)ws_floats, s);
    else if (D == 32 && Do == 32) rc = launch_bwd<32, 32>(x, att_w, att_b, aw3, ds, dx, d_att_w, d_att_b, d_aw3, (int)B, (int)N, (int)n1, ws, (long long)ws_floats, s);
    else { occ_set_error("occ_gat_core_bwd: (D, Do) must be (64,64), (64,32) or (32,32) (got %ld, %ld)", (long)D, (long)Do); return OCC_EUNSUPPORTED; }
    if (rc != OCC_OK) return rc;
    OCC_LAUNCH_CHECK("occ_gat_core_bwd");
    return OCC_OK;
}

}  // extern "C"
