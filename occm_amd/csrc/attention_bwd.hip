// Backward of fairseq MultiheadAttention's softmax(scale q.k^T) v inside XLS-R (reached from sslassist.py:48 when the front-end is
// fine-tuned, oc_training.py:324) for any sequence length and head dims 64 (XLS-R-300M) and 80 (XLS-R-1B).
//
// One workgroup = 8 waves (2 per SIMD) = one (batch, head) x one block of 256 keys.  Keys live on the MFMA lane: every wave owns up to
// two 16-key tiles (tile w and w + 8 of the block) and keeps their K / V row fragments and their dK^T / dV^T accumulators in
// registers while the workgroup sweeps the queries in steps of 32:
//     S = Q.K^T and dP = dO.V^T           (A = Q / dO rows from the LDS slice, B = K / V fragments)     -> lane: 4 queries x 1 key
//     P = exp2(S*c - lse[q]),  dS = P * (dP - delta[q])         (lse from the forward, delta = rowsum(dO * O) computed per slice)
//     dV^T += dO^T.P,  dK^T += Q^T.dS     (A = transposing reads of the dO / Q slice, B = the packed P / dS registers)
//     dS^T -> LDS [keys][32];  dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q]    (A, B = transposing reads of the K image / the dS^T rows)
// k-slot (g, u, j) of a 32-deep contraction <-> row 16u + 4g + j on both operands of the three transposed products.
// The Q / dO slices are row-major [32][HD] (next slice prefetched into registers under the current step); 128-byte rows (HD = 64)
// are XOR-swizzled so that both the row reads (ds_read_b128) and the transposing reads are bank-conflict free.
// dQ: with one key block (T <= 256) it is complete inside the workgroup and stored as bf16; with more key blocks every workgroup adds
// its part to the caller's f32 accumulator (atomics, different key blocks of one head) and attention_dq_finish_kernel rounds it.
#include "occ_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ u32x2 ab_tr_read(unsigned addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

template <int HD> struct AbCfg {
    static constexpr int KS = (HD + 31) / 32;          // 32-deep contraction steps over the head dim (80 -> 3, zero padded)
    static constexpr int DT = HD / 16;                 // 16-wide d tiles
    static constexpr int CH = HD / 8;                  // 16-byte chunks per row
    static constexpr int ROWB = HD == 64 ? 128 : 208;  // LDS row bytes of the Q / dO slices and the K image (96 elements + pad for HD = 80)
    static constexpr bool SWZ = HD == 64;
};
constexpr int AB_KEYS = 256;
// dS of a 32-query step is staged TRANSPOSED, [key][32 queries] bf16 = 64-byte rows: the lane that computed dS for 4 consecutive queries of
// one key stores them with ONE 8-byte write (it was four 2-byte writes into [query][key] rows), and dQ's B operand (column = query,
// k = keys) comes back through the same transposing read that serves the K image.  The two 32-byte halves (query tiles) of a row swap
// places every four rows, so the eight rows a 32-lane pass of the transposing read touches fall on disjoint banks.
constexpr int AB_DS_ROWB = 64, AB_DS_BYTES = AB_KEYS * AB_DS_ROWB;
__device__ __forceinline__ unsigned ab_ds_off(int key, int qt) { return (unsigned)(key * AB_DS_ROWB + ((qt ^ ((key >> 2) & 1)) << 5)); }

// byte offset of 16-byte chunk c of row r in a slice / image
template <int HD> __device__ __forceinline__ unsigned ab_off(int r, int c) {
    if (AbCfg<HD>::SWZ) return (unsigned)(r * 128 + ((c ^ (((r >> 1) & 3) << 1)) << 4));
    return (unsigned)(r * AbCfg<HD>::ROWB + c * 16);
}

// DROP (attention_dropout, see occ_attention_dropout): with O = (P o keep / (1-p)) V the forward's O already carries the mask, so
// delta = rowsum(dO o O) is unchanged; dV uses the dropped probabilities, dP is masked and scaled the same way before dS = P o (dP - delta).
template <int HD, bool DROP = false, bool BIAS = false>
__global__ __launch_bounds__(512, 2) void attention_bwd2_kernel(const unsigned short* __restrict__ qkv, const unsigned short* __restrict__ o,
                                                               const unsigned short* __restrict__ dout, const float* __restrict__ lse,
                                                               unsigned short* __restrict__ dqkv, float* __restrict__ dq_accum, int Tn, int H,
                                                               long long ld_qkv, long long ld_o, float scale,
                                                               const unsigned char* __restrict__ keep = nullptr, int Tp = 0, float inv_keep = 1.f,
                                                               float* __restrict__ bsum = nullptr) {
    using C = AbCfg<HD>;
    constexpr int KS = C::KS, DT = C::DT, CH = C::CH, ROWB = C::ROWB;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    unsigned char* Kimg = sm;                                        // [256][ROWB]
    unsigned char* Qs = Kimg + AB_KEYS * ROWB;                       // [2][32][ROWB]
    unsigned char* dOs = Qs + 2 * 32 * ROWB;                         // [2][32][ROWB]
    unsigned char* dSs0 = dOs + 2 * 32 * ROWB;                                          // [2][256 keys][64 B]: step st writes buffer st & 1
    float* lse_s = reinterpret_cast<float*>(dSs0 + 2 * AB_DS_BYTES);                    // [2][32]
    float* dlt_s = lse_s + 64;                                                          // [2][32]
    const int D = H * HD;
    const int bh = blockIdx.x, b = bh / H, h = bh % H, kb = blockIdx.y;
    const int key0 = kb * AB_KEYS;
    const int nkeys = Tn - key0 < AB_KEYS ? Tn - key0 : AB_KEYS;     // keys of this block (>= 1)
    const int nkt = (nkeys + 15) >> 4;                               // 16-key tiles in use
    const unsigned short* qbase = qkv + (size_t)b * Tn * ld_qkv + (size_t)h * HD;
    const unsigned short* kbase = qbase + D;
    const unsigned short* vbase = qbase + 2 * D;
    const unsigned short* obase = o + (size_t)b * Tn * ld_o + (size_t)h * HD;
    const unsigned short* dobase = dout + (size_t)b * Tn * ld_o + (size_t)h * HD;
    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void*)sm;

    // ---- K image of this key block (rows beyond the last key and the pad columns of HD = 80 are zero)
    for (int idx = tid; idx < AB_KEYS * (ROWB / 16); idx += 512) {
        const int r = idx / (ROWB / 16), c = idx - r * (ROWB / 16);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r < nkeys && c < CH) v = *reinterpret_cast<const uint4*>(kbase + (size_t)(key0 + r) * ld_qkv + c * 8);
        if (C::SWZ) { if (c < CH) *reinterpret_cast<uint4*>(Kimg + ab_off<HD>(r, c)) = v; }
        else *reinterpret_cast<uint4*>(Kimg + r * ROWB + c * 16) = v;
    }
    // ---- this wave's key tiles: K and V row fragments in registers (B operands of S and dP: column = key fr, k = 8 consecutive d)
    u32x4 kf[2][KS], vf[2][KS];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int key = (wave + 8 * kk) * 16 + fr;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            kf[kk][s] = (u32x4){0, 0, 0, 0}; vf[kk][s] = kf[kk][s];
            const int d0 = s * 32 + g * 8;
            if (key < nkeys && d0 < HD) {
                const uint4 kv = *reinterpret_cast<const uint4*>(kbase + (size_t)(key0 + key) * ld_qkv + d0);
                const uint4 vv = *reinterpret_cast<const uint4*>(vbase + (size_t)(key0 + key) * ld_qkv + d0);
                kf[kk][s] = (u32x4){kv.x, kv.y, kv.z, kv.w}; vf[kk][s] = (u32x4){vv.x, vv.y, vv.z, vv.w};
            }
        }
    }
    f32x4 dvacc[DT][2], dkacc[DT][2];                  // [d tile][key tile]: rows d = 16*dt + 4g + r, column key = 16*tile + fr
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { dvacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; dkacc[i][j] = dvacc[i][j]; }
    const float c2 = scale * 1.44269504088896340736f;

    // ---- slice staging: thread -> (row, chunk) items of the 32 x CH chunk grid; the lower half of the workgroup carries Q (+ lse),
    // the upper half dO and O (+ delta = rowsum(dO * O), added up per row with LDS float atomics: CH adds per row and step)
    const int half = tid >> 8, ht = tid & 255;
    constexpr int NIT = (32 * CH + 255) / 256;
    uint4 pa[NIT], pb[NIT];                            // prefetched chunks: Q (lower half) or dO / O (upper half)
    float plse = 0.f;
    auto prefetch = [&](int q0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = ht + it * 256, r = idx / CH, c = idx - r * CH;
            pa[it] = make_uint4(0, 0, 0, 0); pb[it] = pa[it];
            if (idx < 32 * CH && q0 + r < Tn) {
                if (half == 0) pa[it] = *reinterpret_cast<const uint4*>(qbase + (size_t)(q0 + r) * ld_qkv + c * 8);
                else { pa[it] = *reinterpret_cast<const uint4*>(dobase + (size_t)(q0 + r) * ld_o + c * 8); pb[it] = *reinterpret_cast<const uint4*>(obase + (size_t)(q0 + r) * ld_o + c * 8); }
            }
        }
        if (half == 0 && ht < 32) plse = q0 + ht < Tn ? lse[(size_t)bh * Tn + q0 + ht] : 1.0e30f;      // rows beyond Tn: P = exp2(. - 1e30) = 0
    };
    auto commit = [&](int buf) {                       // registers -> LDS slice `buf`
        unsigned char* dst = (half == 0 ? Qs : dOs) + buf * 32 * ROWB;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = ht + it * 256, r = idx / CH, c = idx - r * CH;
            if (idx < 32 * CH) {
                *reinterpret_cast<uint4*>(dst + ab_off<HD>(r, c)) = pa[it];
                if (half == 1) {
                    const unsigned x[4] = {pa[it].x, pa[it].y, pa[it].z, pa[it].w}, y[4] = {pb[it].x, pb[it].y, pb[it].z, pb[it].w};
                    float dl = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        dl += __uint_as_float(x[e] << 16) * __uint_as_float(y[e] << 16) + __uint_as_float(x[e] & 0xffff0000u) * __uint_as_float(y[e] & 0xffff0000u);
                    if (CH == 8) {                     // a row's 8 chunks sit in 8 consecutive lanes: three shuffle steps instead of 8 same-address LDS atomics
                        dl += __shfl_xor(dl, 1, 64); dl += __shfl_xor(dl, 2, 64); dl += __shfl_xor(dl, 4, 64);
                        if (c == 0) dlt_s[buf * 32 + r] = dl;
                    } else {
                        atomicAdd(&dlt_s[buf * 32 + r], dl);
                    }
                }
            }
        }
        if (half == 0 && ht < 32) lse_s[buf * 32 + ht] = plse;
    };
    if (!C::SWZ) {                                      // HD = 80: the pad columns (elements 80 .. 95) of both slice buffers are read as zeros
        for (int idx = tid; idx < 2 * 2 * 32 * 2; idx += 512) {
            const int which = idx >> 7, r = (idx >> 1) & 63, c = CH + (idx & 1);
            *reinterpret_cast<uint4*>((which ? dOs : Qs) + r * ROWB + c * 16) = make_uint4(0, 0, 0, 0);
        }
    }
    if (tid < 64) dlt_s[tid] = 0.f;
    // dS columns of key tiles that are not in use are never written but can be read by the last 32-key step of dQ (against zero K rows)
    for (int idx = tid; idx < 2 * AB_DS_BYTES / 16; idx += 512) reinterpret_cast<uint4*>(dSs0)[idx] = make_uint4(0, 0, 0, 0);
    prefetch(0);
    __syncthreads();
    commit(0);
    const int nstep = (Tn + 31) >> 5;
    // transposing-read lane constants: lane (g, q4 = fr >> 2, p4 = lane & 3) supplies row 4g + q4 of a 16-row group, 8 bytes at p4*8 of a 32-byte block
    const int q4 = fr >> 2, p4 = lane & 3;
    // dQ^T[d][q] = scale * sum_key K^T[d][key] dS^T[key][q] of one step: 2 q tiles x DT d tiles, dealt round-robin to the 8 waves.  It runs one
    // step late (after the barrier that opens the next step, from the other dS buffer), so a step has ONE workgroup barrier: while some
    // waves add up dQ of step st-1, others are already multiplying S / dP of step st.
    // bsum (one key block only): per (batch, head) column sums of the stored (bf16-rounded) dq / dk / dv -- the q|k|v bias gradients,
    // which were a separate pass over dqkv.  dq: this wave's (q tile, d tile) pairs accumulate over the steps.
    f32x4 dqs[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    auto dq_phase = [&](const int q0, const unsigned dss) {        // dss: LDS byte address of the step's dS buffer
        int slot = 0;
        for (int pr = wave; pr < 2 * DT; pr += 8, ++slot) {
            const int qt = pr / DT, dt = pr - qt * DT;
            f32x4 qacc = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int nks = (nkeys + 31) >> 5;
            // dS^T rows ks*32 + 4g + q4 (+16): the half swap depends on (row >> 2) & 1 = g & 1 only, so the address is a lane constant + ks * 2 KiB
            const unsigned dsq = dss + (unsigned)((4 * g + q4) * AB_DS_ROWB + ((qt ^ (g & 1)) << 5) + p4 * 8);
            constexpr int UB = HD == 64 ? 4 : 2;       // 32-key steps per wait (steps beyond the block read zero K rows: harmless); HD = 80 has no registers for four
            for (int ks0 = 0; ks0 < nks; ks0 += UB) {
                u32x2 kr[UB][2], sr[UB][2];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int ks = ks0 + u < 8 ? ks0 + u : 7;
                    const int r0 = ks * 32 + 4 * g + q4;
                    unsigned offa, offb;
                    if (C::SWZ) { offa = (unsigned)(r0 * 128 + ((dt ^ ((r0 >> 1) & 3)) << 5) + p4 * 8); offb = (unsigned)((r0 + 16) * 128 + ((dt ^ (((r0 + 16) >> 1) & 3)) << 5) + p4 * 8); }
                    else { offa = (unsigned)(r0 * ROWB + dt * 32 + p4 * 8); offb = offa + 16 * ROWB; }
                    kr[u][0] = ab_tr_read(lds0 + offa); kr[u][1] = ab_tr_read(lds0 + offb);
                    sr[u][0] = ab_tr_read(dsq + ks * (32 * AB_DS_ROWB)); sr[u][1] = ab_tr_read(dsq + ks * (32 * AB_DS_ROWB) + 16 * AB_DS_ROWB);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    if (ks0 + u >= nks) continue;
                    const u32x4 ka = (u32x4){kr[u][0][0], kr[u][0][1], kr[u][1][0], kr[u][1][1]}, sbv = (u32x4){sr[u][0][0], sr[u][0][1], sr[u][1][0], sr[u][1][1]};
                    qacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ka), __builtin_bit_cast(bf16x8, sbv), qacc, 0, 0, 0);
                }
            }
            const int q = q0 + qt * 16 + fr;           // D rows = d (4g + e), column = query fr
            if (q < Tn) {
                if (gridDim.y == 1) {
                    uint2 ov;
                    ov.x = pack_bf16x2(qacc[0] * scale, qacc[1] * scale);
                    ov.y = pack_bf16x2(qacc[2] * scale, qacc[3] * scale);
                    *reinterpret_cast<uint2*>(dqkv + ((size_t)b * Tn + q) * ld_qkv + (size_t)h * HD + dt * 16 + g * 4) = ov;
                    if (BIAS) {
                        dqs[slot & 1][0] += __uint_as_float(ov.x << 16); dqs[slot & 1][1] += __uint_as_float(ov.x & 0xffff0000u);
                        dqs[slot & 1][2] += __uint_as_float(ov.y << 16); dqs[slot & 1][3] += __uint_as_float(ov.y & 0xffff0000u);
                    }
                } else {
                    float* ap = dq_accum + ((size_t)b * Tn + q) * D + (size_t)h * HD + dt * 16 + g * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd(ap + e, qacc[e] * scale);
                }
            }
        }
    };
    for (int st = 0; st < nstep; ++st) {
        const int q0 = st * 32, buf = st & 1;
        unsigned char* dSs = dSs0 + buf * AB_DS_BYTES;
        if (st + 1 < nstep) prefetch(q0 + 32);
        __syncthreads();                               // slice `buf` (Q, dO, lse, delta) and the previous step's dS are complete
        if (CH != 8 && tid < 32) dlt_s[(buf ^ 1) * 32 + tid] = 0.f;  // (atomics form only) the other buffer's delta is re-accumulated by the next commit
        if (st > 0) dq_phase(q0 - 32, lds0 + (unsigned)(dSs0 - sm) + (buf ^ 1) * AB_DS_BYTES);
        const unsigned qs0 = lds0 + (unsigned)(Qs - sm) + buf * 32 * ROWB, ds0 = lds0 + (unsigned)(dOs - sm) + buf * 32 * ROWB;
        unsigned pp[2][4], ds[2][4];                   // [key tile][packed bf16 pairs]: slots j < 4 from q-tile 0, j >= 4 from q-tile 1
        float lq[2][4], dq_[2][4];                     // lse / delta of the lane's eight query rows: read once per step, not once per key tile
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + buf * 32 + qt * 16 + g * 4), d4 = *reinterpret_cast<const f32x4*>(dlt_s + buf * 32 + qt * 16 + g * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) { lq[qt][r] = l4[r]; dq_[qt][r] = d4[r]; }
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int tile = wave + 8 * kk;
            if (tile >= nkt) { pp[kk][0] = pp[kk][1] = pp[kk][2] = pp[kk][3] = 0; ds[kk][0] = ds[kk][1] = ds[kk][2] = ds[kk][3] = 0; continue; }   // wave-uniform
            const int key = tile * 16 + fr;
            float pv[2][4], dsv[2][4];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                f32x4 sacc = (f32x4){0.f, 0.f, 0.f, 0.f}, dpacc = sacc;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const uint4 qa = *reinterpret_cast<const uint4*>(Qs + buf * 32 * ROWB + ab_off<HD>(qt * 16 + fr, s * 4 + g));
                    const uint4 da = *reinterpret_cast<const uint4*>(dOs + buf * 32 * ROWB + ab_off<HD>(qt * 16 + fr, s * 4 + g));
                    sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, qa), __builtin_bit_cast(bf16x8, kf[kk][s]), sacc, 0, 0, 0);
                    dpacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, da), __builtin_bit_cast(bf16x8, vf[kk][s]), dpacc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    [[maybe_unused]] const int ql = qt * 16 + g * 4 + r;
                    // (one fused multiply-add for the exponent's argument; head dim 80 keeps the two-instruction form: the fused one spilled two registers there)
                    const float ea = HD == 64 ? fmaf(sacc[r], c2, -lq[qt][r]) : sacc[r] * c2 - lq[qt][r];
                    const float p = key < nkeys ? __builtin_amdgcn_exp2f(ea) : 0.f;
                    float pd = p, dpd = dpacc[r];
                    if constexpr (DROP) {
                        const int qg = q0 + ql < Tn ? q0 + ql : Tn - 1, kg = key < nkeys ? key0 + key : Tn - 1;
                        const float mk = keep[((size_t)bh * Tn + qg) * Tp + kg] ? inv_keep : 0.f;
                        pd *= mk; dpd *= mk;
                    }
                    pv[qt][r] = pd;
                    dsv[qt][r] = p * (dpd - dq_[qt][r]);
                }
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                pp[kk][e] = pack_bf16x2(pv[0][2 * e], pv[0][2 * e + 1]);
                pp[kk][2 + e] = pack_bf16x2(pv[1][2 * e], pv[1][2 * e + 1]);
                ds[kk][e] = pack_bf16x2(dsv[0][2 * e], dsv[0][2 * e + 1]);
                ds[kk][2 + e] = pack_bf16x2(dsv[1][2 * e], dsv[1][2 * e + 1]);
            }
            // dS^T[key][queries 4g .. 4g+3 of q-tile qt]: the packed pairs as they are
            // ((key >> 2) & 1 = (fr >> 2) & 1: the two halves' places are lane constants)
            *reinterpret_cast<uint2*>(dSs + key * AB_DS_ROWB + (((fr >> 2) & 1) << 5) + g * 8) = make_uint2(ds[kk][0], ds[kk][1]);
            *reinterpret_cast<uint2*>(dSs + key * AB_DS_ROWB + ((((fr >> 2) & 1) ^ 1) << 5) + g * 8) = make_uint2(ds[kk][2], ds[kk][3]);
        }
        // dV^T += dO^T . P ; dK^T += Q^T . dS   (contraction over the 32 queries of the step: slot (g, u, j) <-> query 16u + 4g + j)
        if (wave < nkt) {
            // all transposing reads of the step first (one wait), then the MFMAs: block dt (32 bytes = 16 d) of rows 4g + q4 (+16); the
            // swizzle moves whole 32-byte blocks, so the 8-byte piece p4 stays in place
            // (HD = 80: in two batches of d tiles -- 40 fragment registers at once do not fit beside the 128 accumulator / operand registers)
            constexpr int DB = HD == 64 ? DT : 3;
#pragma unroll
            for (int d0 = 0; d0 < DT; d0 += DB) {
                u32x2 dfr[DB][2], qfr[DB][2];
#pragma unroll
                for (int i = 0; i < DB; ++i) {
                    const int dt = d0 + i < DT ? d0 + i : DT - 1;
                    unsigned offa, offb;
                    if (C::SWZ) { const int r0 = 4 * g + q4; offa = (unsigned)(r0 * 128 + ((dt ^ ((r0 >> 1) & 3)) << 5) + p4 * 8); offb = (unsigned)((r0 + 16) * 128 + ((dt ^ (((r0 + 16) >> 1) & 3)) << 5) + p4 * 8); }
                    else { offa = (unsigned)((4 * g + q4) * ROWB + dt * 32 + p4 * 8); offb = offa + 16 * ROWB; }
                    dfr[i][0] = ab_tr_read(ds0 + offa); dfr[i][1] = ab_tr_read(ds0 + offb); qfr[i][0] = ab_tr_read(qs0 + offa); qfr[i][1] = ab_tr_read(qs0 + offb);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < DB; ++i) {
                    const int dt = d0 + i;
                    if (dt >= DT) continue;
                    const u32x4 da = (u32x4){dfr[i][0][0], dfr[i][0][1], dfr[i][1][0], dfr[i][1][1]}, qa = (u32x4){qfr[i][0][0], qfr[i][0][1], qfr[i][1][0], qfr[i][1][1]};
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        if (wave + 8 * kk >= nkt) continue;
                        const u32x4 pbv = (u32x4){pp[kk][0], pp[kk][1], pp[kk][2], pp[kk][3]}, sbv = (u32x4){ds[kk][0], ds[kk][1], ds[kk][2], ds[kk][3]};
                        dvacc[dt][kk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, da), __builtin_bit_cast(bf16x8, pbv), dvacc[dt][kk], 0, 0, 0);
                        dkacc[dt][kk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, qa), __builtin_bit_cast(bf16x8, sbv), dkacc[dt][kk], 0, 0, 0);
                    }
                }
            }
        }
        // next slice into the other buffer: its last readers were step st-1's phases, all before this step's barrier.  (The atomics form of
        // delta, HD = 80, needs its zeroing above ordered before these adds: one more barrier there.)
        if (CH != 8) __syncthreads();
        if (st + 1 < nstep) commit(buf ^ 1);
    }
    __syncthreads();
    dq_phase((nstep - 1) * 32, lds0 + (unsigned)(dSs0 - sm) + ((nstep - 1) & 1) * AB_DS_BYTES);
    // ---- dK, dV of this wave's keys: lane holds 4 consecutive d of one key
    f32x4 dks[BIAS ? DT : 1], dvs[BIAS ? DT : 1];
#pragma unroll
    for (int dt = 0; dt < (BIAS ? DT : 1); ++dt) { dks[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dvs[dt] = dks[dt]; }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int key = (wave + 8 * kk) * 16 + fr;
        if (key >= nkeys) continue;
        unsigned short* dkrow = dqkv + ((size_t)b * Tn + key0 + key) * ld_qkv + D + (size_t)h * HD;
        unsigned short* dvrow = dqkv + ((size_t)b * Tn + key0 + key) * ld_qkv + 2 * D + (size_t)h * HD;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            uint2 kk2, vv;
            kk2.x = pack_bf16x2(dkacc[dt][kk][0] * scale, dkacc[dt][kk][1] * scale);
            kk2.y = pack_bf16x2(dkacc[dt][kk][2] * scale, dkacc[dt][kk][3] * scale);
            vv.x = pack_bf16x2(dvacc[dt][kk][0], dvacc[dt][kk][1]);
            vv.y = pack_bf16x2(dvacc[dt][kk][2], dvacc[dt][kk][3]);
            *reinterpret_cast<uint2*>(dkrow + dt * 16 + g * 4) = kk2;
            *reinterpret_cast<uint2*>(dvrow + dt * 16 + g * 4) = vv;
            if (BIAS) {
                dks[dt][0] += __uint_as_float(kk2.x << 16); dks[dt][1] += __uint_as_float(kk2.x & 0xffff0000u);
                dks[dt][2] += __uint_as_float(kk2.y << 16); dks[dt][3] += __uint_as_float(kk2.y & 0xffff0000u);
                dvs[dt][0] += __uint_as_float(vv.x << 16); dvs[dt][1] += __uint_as_float(vv.x & 0xffff0000u);
                dvs[dt][2] += __uint_as_float(vv.y << 16); dvs[dt][3] += __uint_as_float(vv.y & 0xffff0000u);
            }
        }
    }
    if (BIAS) {
        // sums over the 16 keys / queries of a lane row (xor 1, 2, 4, 8), then over the waves through LDS (the slices are free: every
        // phase that read them lies before the barrier below), added in wave order: reproducible
        __syncthreads();
        float* red = reinterpret_cast<float*>(Qs);         // [8 waves][3][HD]
        for (int idx = tid; idx < 8 * 3 * HD; idx += 512) red[idx] = 0.f;
        __syncthreads();
        auto row16 = [](float v) { v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); return v; };
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sk = row16(dks[dt][r]), sv = row16(dvs[dt][r]);
                if (fr == 0) { red[(wave * 3 + 1) * HD + dt * 16 + g * 4 + r] = sk; red[(wave * 3 + 2) * HD + dt * 16 + g * 4 + r] = sv; }
            }
        {
            int slot = 0;
            for (int pr = wave; pr < 2 * DT; pr += 8, ++slot) {
                const int dt = pr % DT;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float sq = row16(dqs[slot & 1][e]);
                    if (fr == 0) red[(wave * 3 + 0) * HD + dt * 16 + g * 4 + e] += sq;      // a wave's two pairs may share a d tile only for DT < 4 (not the case: DT = 4, 5)
                }
            }
        }
        __syncthreads();
        for (int idx = tid; idx < 3 * HD; idx += 512) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) t += red[w * 3 * HD + idx];
            bsum[((size_t)bh * gridDim.y + kb) * 3 * HD + idx] = t;
        }
    }
}

// qkv bias gradients from the per-(batch, head) sums attention_bwd2_kernel leaves: dbias[which*D + h*HD + d] += sum_b bsum[b*H + h][which][d].
// One workgroup per (which, head): 256 / HD groups of HD threads walk the B * nkb records with independent loads (one thread per output
// walking all 64 records took 19 us of pure load latency, 24 times per step), joined through LDS in group order: reproducible.
__global__ __launch_bounds__(256) void attention_bias_finalize_kernel(const float* __restrict__ bsum, float* __restrict__ dbias, int B, int H, int HD, int nkb) {
    __shared__ float part[256];
    const int which = blockIdx.x / H, h = blockIdx.x - which * H;
    const int ngr = 256 / HD, gr = threadIdx.x / HD, d = threadIdx.x - gr * HD;
    const int nrec = B * nkb;
    float t = 0.f;
    if (gr < ngr) {
        const float* src = bsum + ((size_t)h * nkb) * 3 * HD + which * HD + d;            // record (bb, k) of head h: + bb * H * nkb * 3 * HD + k * 3 * HD
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
        int r = gr;
        for (; r + 3 * ngr < nrec; r += 4 * ngr) {
            const int r1 = r + ngr, r2 = r + 2 * ngr, r3 = r + 3 * ngr;
            t0 += src[((size_t)(r / nkb) * H * nkb + r % nkb) * 3 * HD];
            t1 += src[((size_t)(r1 / nkb) * H * nkb + r1 % nkb) * 3 * HD];
            t2 += src[((size_t)(r2 / nkb) * H * nkb + r2 % nkb) * 3 * HD];
            t3 += src[((size_t)(r3 / nkb) * H * nkb + r3 % nkb) * 3 * HD];
        }
        for (; r < nrec; r += ngr) t0 += src[((size_t)(r / nkb) * H * nkb + r % nkb) * 3 * HD];
        t = (t0 + t1) + (t2 + t3);
    }
    part[threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x < HD) {
        float sum = 0.f;
        for (int g2 = 0; g2 < ngr; ++g2) sum += part[g2 * HD + threadIdx.x];
        dbias[(size_t)which * H * HD + (size_t)h * HD + threadIdx.x] += sum;
    }
}

// several key blocks: dq (bf16, the q columns of dqkv) = the f32 sums of their parts
__global__ void attention_dq_finish_kernel(const float* __restrict__ acc, unsigned short* __restrict__ dqkv, long long rows, int D, long long ld_qkv) {
    const long long n = rows * (D / 4);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / (D / 4); const int c = (int)(i - r * (D / 4)) * 4;
        const float4 v = *reinterpret_cast<const float4*>(acc + r * D + c);
        uint2 ov;
        ov.x = pack_bf16x2(v.x, v.y);
        ov.y = pack_bf16x2(v.z, v.w);
        *reinterpret_cast<uint2*>(dqkv + r * ld_qkv + c) = ov;
    }
}

template <int HD, bool DROP = false, bool BIAS = false>
int launch_attention_bwd2(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, float* dq_accum, int64_t B, int64_t T, int64_t H,
                          int64_t ld_qkv, int64_t ld_o, float scale, hipStream_t s, const unsigned char* keep = nullptr, float p = 0.f, float* dbias = nullptr,
                          float* bias_ws = nullptr, int defer = 0) {
    using C = AbCfg<HD>;
    const size_t shm = (size_t)AB_KEYS * C::ROWB + 4 * 32 * C::ROWB + 2 * AB_DS_BYTES + 128 * 4;
    hipError_t e = hipFuncSetAttribute((const void*)attention_bwd2_kernel<HD, DROP, BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) { occ_set_error("occ_attention_bwd: cannot raise LDS limit: %s", hipGetErrorString(e)); return OCC_ELAUNCH; }
    const int64_t nkb = occ_cdiv(T, AB_KEYS);
    if (nkb > 1 && hipMemsetAsync(dq_accum, 0, (size_t)(B * T * H * HD) * sizeof(float), s) != hipSuccess) { occ_set_error("occ_attention_bwd: memset failed"); return OCC_ELAUNCH; }
    hipLaunchKernelGGL((attention_bwd2_kernel<HD, DROP, BIAS>), dim3((unsigned)(B * H), (unsigned)nkb), dim3(512), shm, s, (const unsigned short*)qkv, (const unsigned short*)o,
                       (const unsigned short*)dout, lse, (unsigned short*)dqkv, dq_accum, (int)T, (int)H, (long long)ld_qkv, (long long)ld_o, scale, keep,
                       (int)((T + 3) / 4 * 4), 1.0f / (1.0f - p), dbias ? bias_ws : nullptr);
    if (dbias && !defer)                              // (defer: the records stay in bias_ws for occ_finalize_batch)
        hipLaunchKernelGGL(attention_bias_finalize_kernel, dim3((unsigned)(3 * H)), dim3(256), 0, s, (const float*)bias_ws, dbias, (int)B, (int)H, HD, (int)nkb);
    if (nkb > 1) {
        long long blocks = occ_cdiv(B * T * (H * HD / 4), 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(attention_dq_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dq_accum, (unsigned short*)dqkv, (long long)(B * T), (int)(H * HD), (long long)ld_qkv);
    }
    return OCC_OK;
}

}  // namespace

extern "C" int occ_attention_bwd_bias(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, int64_t B, int64_t T, int64_t H, int64_t hd,
                                      int64_t ld_qkv, int64_t ld_o, float scale, float* dbias, float* bias_ws, int64_t bias_ws_floats, int defer, void* stream) {
    OCC_CHECK_ARG(qkv && o && dout && lse && dqkv && dbias && bias_ws, "occ_attention_bwd_bias: null pointer");
    OCC_CHECK_ARG((hd == 64 || hd == 80) && T >= 1 && T <= AB_KEYS && B >= 1 && H >= 1 && B * H < (1ll << 31), "occ_attention_bwd_bias: head_dim 64 or 80, T <= %d (one key block)", AB_KEYS);
    OCC_CHECK_ARG(ld_qkv % 8 == 0 && ld_o % 8 == 0 && ld_qkv >= 3 * H * hd && ld_o >= H * hd && bias_ws_floats >= B * H * 3 * hd, "occ_attention_bwd_bias: leading dimensions / scratch");
    int rc;
    if (hd == 64) rc = launch_attention_bwd2<64, false, true>(qkv, o, dout, lse, dqkv, nullptr, B, T, H, ld_qkv, ld_o, scale, (hipStream_t)stream, nullptr, 0.f, dbias, bias_ws, defer);
    else rc = launch_attention_bwd2<80, false, true>(qkv, o, dout, lse, dqkv, nullptr, B, T, H, ld_qkv, ld_o, scale, (hipStream_t)stream, nullptr, 0.f, dbias, bias_ws, defer);
    if (rc != OCC_OK) return rc;
    OCC_LAUNCH_CHECK("occ_attention_bwd_bias");
    return OCC_OK;
}

extern "C" int occ_attention_bwd_dropout(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, int64_t B, int64_t T, int64_t H, int64_t hd,
                                         int64_t ld_qkv, int64_t ld_o, float scale, float* dq_accum, const uint8_t* keep, float p, void* stream) {
    OCC_CHECK_ARG(qkv && o && dout && lse && dqkv && keep, "occ_attention_bwd_dropout: null pointer");
    OCC_CHECK_ARG((hd == 64 || hd == 80) && T >= 1 && B >= 1 && H >= 1 && B * H < (1ll << 31) && T < (1ll << 24), "occ_attention_bwd_dropout: head_dim must be 64 or 80 (T=%ld hd=%ld)", (long)T, (long)hd);
    OCC_CHECK_ARG(ld_qkv % 8 == 0 && ld_o % 8 == 0 && ld_qkv >= 3 * H * hd && ld_o >= H * hd && p >= 0.f && p < 1.f, "occ_attention_bwd_dropout: leading dimensions / p");
    OCC_CHECK_ARG(T <= AB_KEYS || (dq_accum && ((uintptr_t)dq_accum & 15) == 0), "occ_attention_bwd_dropout: T > %d needs the f32 dq accumulator [B*T, H*hd]", AB_KEYS);
    int rc;
    if (hd == 64) rc = launch_attention_bwd2<64, true>(qkv, o, dout, lse, dqkv, dq_accum, B, T, H, ld_qkv, ld_o, scale, (hipStream_t)stream, (const unsigned char*)keep, p);
    else rc = launch_attention_bwd2<80, true>(qkv, o, dout, lse, dqkv, dq_accum, B, T, H, ld_qkv, ld_o, scale, (hipStream_t)stream, (const unsigned char*)keep, p);
    if (rc != OCC_OK) return rc;
    OCC_LAUNCH_CHECK("occ_attention_bwd_dropout");
    return OCC_OK;
}

extern "C" int occ_attention_bwd(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, int64_t B, int64_t T, int64_t H, int64_t hd,
                                 int64_t ld_qkv, int64_t ld_o, float scale, float* dq_accum, void* stream) {
    OCC_CHECK_ARG(qkv && o && dout && lse && dqkv, "occ_attention_bwd: null pointer");
    OCC_CHECK_ARG((hd == 64 || hd == 80) && T >= 1 && B >= 1 && H >= 1 && B * H < (1ll << 31) && T < (1ll << 24), "occ_attention_bwd: head_dim must be 64 or 80 (T=%ld hd=%ld)", (long)T, (long)hd);
    OCC_CHECK_ARG(ld_qkv % 8 == 0 && ld_o % 8 == 0 && ld_qkv >= 3 * H * hd && ld_o >= H * hd, "occ_attention_bwd: leading dimensions");
    OCC_CHECK_ARG(T <= AB_KEYS || (dq_accum && ((uintptr_t)dq_accum & 15) == 0), "occ_attention_bwd: T > %d needs the f32 dq accumulator [B*T, H*hd]", AB_KEYS);
    int rc;
    if (hd == 64) rc = launch_attention_bwd2<64>(qkv, o, dout, lse, dqkv, dq_accum, B, T, H, ld_qkv, ld_o, scale, (hipStream_t)stream);
    else rc = launch_attention_bwd2<80>(qkv, o, dout, lse, dqkv, dq_accum, B, T, H, ld_qkv, ld_o, scale, (hipStream_t)stream);
    if (rc != OCC_OK) return rc;
    OCC_LAUNCH_CHECK("occ_attention_bwd");
    return OCC_OK;
}
