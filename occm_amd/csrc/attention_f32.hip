// f32 self-attention on the f32 matrix cores (v_mfma_f32_16x16x4_f32): softmax(scale q.k^T) v of fairseq's MultiheadAttention inside XLS-R
// on the f32 scoring path (oc_classifier.py:182-186, 256-261 -> models/xlsr.py:35-46), any sequence length, head dims 64 / 80, optional
// per-utterance key counts of a zero-padded batch (kv_len).  The VALU kernels it replaces (attention_kernel / attention_stream_kernel in
// frontend.hip: lanes = keys, one fmaf per multiply) took 30 % of a batch-16 scoring pass.
//
// One workgroup = 4 waves = one (batch, head) x 64 queries; a wave owns 16 queries.  Keys are streamed through LDS in blocks of 64
// with the online-softmax recurrence, K and V rows row-major as [key][hd + 4] floats: a stride of 68 floats puts 8 consecutive rows on
// the 32 banks exactly once (the 16-byte K fragments), and the 4-byte V reads of a wave (16 consecutive d of four keys) touch every
// bank twice -- the minimum for 64 lanes.
//     S^T[key][q] = sum_d K[key][d] Q[q][d]        A = K fragment, B = the wave's Q fragments (registers, pre-scaled)
//         -> lane (q = lane & 15, g = lane >> 4) holds keys 4g .. 4g+3 of every 16-key tile: the row maximum / sum of a query is
//            registers + two cross-lane steps (xor 16, 32)
//     O^T[d][q]  += sum_key V^T[d][key] P^T[key][q]   A = V[key 4g + m][d = lane & 15] (one 4-byte read per MFMA: V stays un-transposed),
//                                                      B = the probabilities as they lie in the registers
// A 16x16x4 MFMA takes ONE f32 per lane and operand: lane group g supplies contraction index 4g + m in step m on BOTH operands (any
// order of the contraction index is allowed as long as the operands agree), so one 16-byte fragment feeds four MFMAs.
// Arithmetic is f32 throughout (IEEE FMA in the matrix core, v_exp_f32 on f32 arguments): no operand is rounded.
#include "occ_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int HD>
__global__ __launch_bounds__(256) void attention_f32_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ out, int Tn, int H, long long ld_qkv,
                                                                long long ld_out, float scale, const int* __restrict__ kv_len) {
    constexpr int DT = HD / 16, KB = 64, KP = HD + 4;
    __shared__ __attribute__((aligned(16))) float Ks[KB * KP];          // [key][d]
    __shared__ __attribute__((aligned(16))) float Vs[KB * KP];          // [key][d]
    const int D = H * HD;
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const int Tk = kv_len ? min(max(kv_len[b], 1), Tn) : Tn;
    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* base = qkv + (size_t)b * Tn * ld_qkv + (size_t)h * HD;
    const int q0 = blockIdx.y * 64;
    const int q = q0 + wave * 16 + fr;                                   // this lane's query (column of S^T and O^T)
    float* orow = out + ((size_t)b * Tn + q) * ld_out + (size_t)h * HD;
    if (q0 >= Tk) {                                                      // padding rows of a shorter utterance: nothing reads them; keep them finite
        if (q < Tn) {
#pragma unroll
            for (int c = 0; c < DT; ++c) *reinterpret_cast<f32x4*>(orow + c * 16 + g * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        return;
    }
    // Q fragments: element m of qf[c] = scale * Q[q][16c + 4g + m]
    f32x4 qf[DT];
#pragma unroll
    for (int c = 0; c < DT; ++c) {
        qf[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (q < Tn) qf[c] = *reinterpret_cast<const f32x4*>(base + (size_t)q * ld_qkv + c * 16 + g * 4) * scale;
    }
    f32x4 o[DT];
#pragma unroll
    for (int c = 0; c < DT; ++c) o[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrun = -3.0e38f, lrun = 0.f;                                   // running maximum (all four lane groups of a query agree) and this lane's part of the sum
    constexpr int CH = HD / 4;                                           // 16-byte chunks per row
    for (int k0 = 0; k0 < Tk; k0 += KB) {
        __syncthreads();                                                 // the previous block's fragment reads are done
        for (int idx = tid; idx < KB * CH; idx += 256) {
            const int r = idx / CH, c4 = (idx - r * CH) * 4;
            f32x4 kv = (f32x4){0.f, 0.f, 0.f, 0.f}, vv = kv;
            if (k0 + r < Tk) {
                kv = *reinterpret_cast<const f32x4*>(base + (size_t)(k0 + r) * ld_qkv + D + c4);
                vv = *reinterpret_cast<const f32x4*>(base + (size_t)(k0 + r) * ld_qkv + 2 * D + c4);
            }
            *reinterpret_cast<f32x4*>(Ks + r * KP + c4) = kv;
            *reinterpret_cast<f32x4*>(Vs + r * KP + c4) = vv;
        }
        __syncthreads();
        // ---- scores of the block: s[kt][r] = S^T[key k0 + 16 kt + 4g + r][q]
        f32x4 s[4];
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < DT; ++c) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + (kt * 16 + fr) * KP + c * 16 + g * 4);
#pragma unroll
                for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[m], qf[c][m], acc, 0, 0, 0);
            }
            s[kt] = acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) if (k0 + kt * 16 + g * 4 + r < Tk) mx = fmaxf(mx, acc[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(mrun, mx);                              // (a block has at least one valid key: k0 < Tk)
        const float alpha = expf(mrun - mnew);
        mrun = mnew;
        float psum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = k0 + kt * 16 + g * 4 + r < Tk ? expf(s[kt][r] - mnew) : 0.f;
                s[kt][r] = p; psum += p;
            }
        lrun = lrun * alpha + psum;
#pragma unroll
        for (int c = 0; c < DT; ++c) o[c] = o[c] * alpha;
        // ---- O^T += V^T . P^T
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int c = 0; c < DT; ++c) {
#pragma unroll
                for (int m = 0; m < 4; ++m) o[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(Vs[(kt * 16 + g * 4 + m) * KP + c * 16 + fr], s[kt][m], o[c], 0, 0, 0);
            }
    }
    lrun += __shfl_xor(lrun, 16, 64); lrun += __shfl_xor(lrun, 32, 64);
    if (q < Tn) {
        const float inv = 1.0f / lrun;
#pragma unroll
        for (int c = 0; c < DT; ++c) *reinterpret_cast<f32x4*>(orow + c * 16 + g * 4) = o[c] * inv;       // O^T rows d = 16c + 4g + r of column q
    }
}

}  // namespace

// qkv / out f32, 16-byte aligned rows (ld % 4 == 0), head_dim 64 or 80: checked by the caller (attention_f32_arith in frontend.hip)
int occ_attention_f32_mfma_launch(const float* qkv, float* out, long long B, long long T, long long H, long long hd, long long ld_qkv, long long ld_out, float scale,
                                  const int* kv_len, hipStream_t s) {
    const dim3 grid((unsigned)(B * H), (unsigned)((T + 63) / 64)), block(256);
    if (hd == 64) hipLaunchKernelGGL(attention_f32_mfma_kernel<64>, grid, block, 0, s, qkv, out, (int)T, (int)H, ld_qkv, ld_out, scale, kv_len);
    else hipLaunchKernelGGL(attention_f32_mfma_kernel<80>, grid, block, 0, s, qkv, out, (int)T, (int)H, ld_qkv, ld_out, scale, kv_len);
    return 0;
}
