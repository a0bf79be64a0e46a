/* libocc_hip.so -- C ABI of the MI355X (gfx950) implementation of the occm training hot path.
 *
 * The reference (nguyenvulong/occm) is pure Python on PyTorch and has no FFI layer of its own;
 * each entry point below names the reference Python function (file:line in the upstream tree)
 * whose arithmetic it replaces.  INTEGRATION.md shows the ctypes binding a maintainer of the
 * reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless the name ends in _host;
 *   - shapes are int64_t, row-major, innermost dimension contiguous unless a stride is passed;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued asynchronously on it;
 *   - return value: 0 on success, a negative occ_status otherwise; occ_last_error() returns a
 *     thread-local message for the most recent failure on the calling thread;
 *   - the library keeps no global mutable state and never allocates device memory.
 */
#ifndef OCC_HIP_H
#define OCC_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum occ_status { OCC_OK = 0, OCC_EINVAL = -1, OCC_ELAUNCH = -2, OCC_EUNSUPPORTED = -3 };
enum occ_dtype { OCC_F32 = 0, OCC_BF16 = 1, OCC_F64 = 2, OCC_F32_AS_BF16 = 3 /* occ_gemm ab_dtype only: f32 operands in memory, rounded to bf16 on the way into LDS, bf16 MFMA */,
                 OCC_AF32_WBF16 = 4 /* occ_gemm ab_dtype only: A f32 in memory (rounded to bf16 while staged), W bf16 */,
                 OCC_FP8_E4M3 = 5, OCC_FP8_E5M2 = 6 /* OCP 8-bit floats (e4m3fn / e5m2), one byte per element: occ_fp8_quantize output;
                    as occ_gemm ab_dtype: A in that format, W e4m3, products on v_mfma_scale_f32_16x16x128_f8f6f4, f32 accumulate */,
                 OCC_F32X3 = 7 /* occ_gemm ab_dtype only: f32 operands in memory, each split into bf16 hi + lo while staged; three bf16 MFMAs per
                    block (Wh.Xh + Wl.Xh + Wh.Xl, f32 accumulate): products exact to 2^-16 relative at 3/16 of the exact-f32 MFMA's cycles --
                    the accurate-and-fast arithmetic of the scoring path (K %% 8 == 0) */ };
enum occ_act { OCC_ACT_NONE = 0, OCC_ACT_GELU = 1, OCC_ACT_SELU = 2, OCC_ACT_RELU = 3, OCC_ACT_TANH = 4,
               OCC_ACT_GELU_GRAD = 5 /* occ_gemm epilogue: (acc+bias) * gelu'(aux) */,
               /* The same pair with the derivative computed ONCE, in the forward epilogue, where the exponential is already at hand:
                * KEEP_GRAD: C = gelu(acc+bias), aux <- bf16(gelu'(acc+bias));  MUL_AUX: C = (acc+bias) * aux (no transcendental in the
                * input-gradient epilogue, which is VALU-bound at N = 4096).                                                         */
               OCC_ACT_GELU_KEEP_GRAD = 6, OCC_ACT_MUL_AUX = 7 };

const char* occ_last_error(void);
int occ_version(void);
const char* occ_arch(void);            /* "gfx950" */

/* ---------------------------------------------------------------- RawBoost (RawBoost.py) -- */
/* y[b,j] = sum_i sum_k coef[b,i,k] * pow(x[b, j + (ntaps[b,i]+1)/2 - k], i+1 if powers else 1),
 * zero outside [0,L): filterFIR (RawBoost.py:51-56) summed over the N_f branches of
 * LnL_convolutive_noise (:59-66).  x: f32 or f64 [B,L]; y: f64 [B,L]; coef: f64 [B,n_filt,max_taps];
 * ntaps: i32 [B,n_filt] (odd, <= max_taps <= 1024).                                             */
int occ_rawboost_fir_bank(const void* x, int x_dtype, double* y, const double* coef, const int32_t* ntaps,
                          int64_t B, int64_t L, int64_t n_filt, int64_t max_taps, int powers, void* stream);
/* In place on y f64 [B,L]: optional mean removal, then peak normalisation (mode 1: only when the
 * peak exceeds 1 -- normWav(x,0); mode 2: always -- normWav(x,1); mode 0: none).  RawBoost.py:20-25,
 * 67-68.  partials: f64 scratch [B, ceil(L/4096), 4].                                            */
int occ_rawboost_center_norm(double* y, int64_t B, int64_t L, int subtract_mean, int norm_mode,
                             double* partials, void* stream);
/* ISD_additive_noise scatter (RawBoost.py:78-82): y[b,pos] = y[b,pos]*(1 + g_sd*fr) for the first
 * n[b] entries of pos/fr ([B,max_n]).  Follow with occ_rawboost_center_norm(.., 0, 1, ..).        */
int occ_rawboost_isd_scatter(double* y, const int32_t* pos, const double* fr, const int32_t* n,
                             int64_t B, int64_t L, int64_t max_n, double g_sd, void* stream);
/* SSI_additive_noise mix (RawBoost.py:93-96): out = x + noise/||noise||*||x||/10^(0.05*snr[b]);
 * noise must already be filtered and peak-normalised.  partials: f64 scratch [B, ceil(L/4096), 4]. */
int occ_rawboost_ssi_mix(const double* x, const double* noise, const double* snr, double* out,
                         int64_t B, int64_t L, double* partials, void* stream);
/* HOST function (no GPU work): genNotchCoeffs (RawBoost.py:28-48) for caller-drawn (fc,bw,c)[n_bands], G.
 * Writes max_taps doubles (zero padded) to out_host and the tap count to ntaps_out_host.          */
int occ_notch_coeffs_host(const double* fc, const double* bw, const int32_t* c, int64_t n_bands, double G, double fs,
                          double* out_host, int32_t* ntaps_out_host, int64_t max_taps);
/* out = a + b (f64): the parallel branch sum of RawBoost algo 8 (data_utils_SSL.py:160-165).         */
int occ_add_f64(const double* a, const double* b, double* out, int64_t n, void* stream);
/* Device version for a batch of filters: fc, bw f64 [n_filters, n_bands], c i32 [n_filters, n_bands], gain f64 [n_filters] ->
 * coef f64 [n_filters, max_taps] (zero padded), ntaps i32 [n_filters].  Same arithmetic as occ_notch_coeffs_host.            */
int occ_notch_coeffs(const double* fc, const double* bw, const int32_t* c, const double* gain, int64_t n_filters, int64_t n_bands,
                     double fs, double* coef, int32_t* ntaps, int64_t max_taps, void* stream);
/* ISD_additive_noise with the randomness on the device (Philox): for each utterance b, n[b] positions drawn uniformly without
 * replacement (radix select of the n[b] smallest per-sample keys) get y *= 1 + g_sd*(2u1-1)(2u2-1).  thr_scratch: u32 [B];
 * count (optional i32 [B], pre-zeroed) receives the number of touched samples.  Follow with occ_rawboost_center_norm(.., 0, 1, ..). */
int occ_rawboost_isd_device(double* y, const int32_t* n, uint32_t* thr_scratch, int32_t* count, int64_t B, int64_t L, double g_sd,
                            uint64_t seed, uint64_t stream_id, void* stream);
int occ_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream);
/* Counter-based N(0,1) / U[0,1) fill (Philox4x32-10), element i uses counter (i/4, stream_id).   */
int occ_philox_fill(void* dst, int dtype, int64_t n, uint64_t seed, uint64_t stream_id, int normal, void* stream);

/* Fused attention core of the graph layers (GraphAttentionLayer sslassist.py:102-130, HtrgGraphAttentionLayer :271-300), bf16 MFMA
 * with f32 accumulate -- replaces occ_pair_mul + occ_gemm(att_proj, tanh) + occ_gat_softmax + occ_bmm_alpha and their [B,N,N,D] /
 * [B,N,N,Do] tensors:  alpha[b,i,:] = softmax_j(aw_type(i,j) . tanh(W (x_i o x_j) + b) * inv_temp),  h[b,i,:] = sum_j alpha[b,i,j] x[b,j,:].
 * x f32 [B,N,D]; att_w f32 [Do,D]; att_b [Do]; aw3 f32 [3,Do] = (w11, w22, w12) with type by (i < n1, j < n1), n1 = N for homogeneous
 * layers; alpha f32 [B,N,N] and h f32 [B,N,D] out.  (D, Do) in {(64,64), (64,32), (32,32)}.                                       */
int occ_gat_core_fwd(const float* x, const float* att_w, const float* att_b, const float* aw3, float* alpha, float* h, int64_t B, int64_t N,
                     int64_t D, int64_t Do, int64_t n1, float inv_temp, void* stream);
/* Its backward through the score path, given ds = d loss / d score f32 [B,N,N] (occ_gat_dscore): z is recomputed, nothing of size
 * N*N*D is stored.  dx f32 [B,N,D] += gradient through the pairwise products (add the alpha^T dh term with occ_bmm_alpha);
 * d_att_w [Do,D], d_att_b [Do], d_aw3 [3,Do] are accumulated from per-workgroup records in a fixed order (reproducible); ws: f32
 * scratch of at least B * ceil(N / 16) * (Do*D + 4*Do) floats, 16-byte aligned.  N such that the kernel's LDS image fits (N <= 96 for
 * (64,64)); OCC_EINVAL otherwise.                                                                                                */
int occ_gat_core_bwd(const float* x, const float* att_w, const float* att_b, const float* aw3, const float* ds, float* dx, float* d_att_w,
                     float* d_att_b, float* d_aw3, int64_t B, int64_t N, int64_t D, int64_t Do, int64_t n1, float* ws, int64_t ws_floats,
                     void* stream);
/* ---------------------------------------------------- losses (losses/custom_loss.py) ------ */
/* compactness_loss (custom_loss.py:4-29) over groups of `group` rows: rows [g*group, g*group+6) of
 * emb f32 [n_groups*group, E]; loss[0] = mean over groups of the reference's 6-row value.
 * demb (may be NULL) receives d(loss*scale)/demb (zeros for rows 6.. of each group).             */
int occ_compactness_loss(const float* emb, float* loss, float* demb, int64_t n_groups, int64_t group,
                         int64_t E, float scale, void* stream);
/* descriptiveness_loss (custom_loss.py:78-99): mean cross-entropy of logits f32 [B,C] vs labels i64 [B];
 * dlogits (may be NULL) receives d(loss*scale)/dlogits.                                          */
int occ_ce_loss(const float* logits, const int64_t* labels, float* loss, float* dlogits, int64_t B,
                int64_t C, float scale, void* stream);
/* Signed sum of pair distances with F.pairwise_distance's eps: loss[0] = act(bias + sum_k weights[k] *
 * ||emb[pair_i[k]] - emb[pair_j[k]] + 1e-6||_2), act = relu when `relu` != 0.  Replaces triplet_loss
 * (custom_loss.py:32-57: pairs (0,1) weight +1, (0,2) weight -1, bias = margin, relu) and
 * euclidean_distance_loss (custom_loss.py:59-74: pairs (0,1),(0,2),(0,3),(2,1),(2,3), weights 1/5).
 * pair_i / pair_j i32 [n_pairs] and weights f32 [n_pairs] are device arrays, n_pairs <= 16; the row indices must be
 * < R (caller-checked).  demb f32 [R,E] (may be NULL) receives d(loss*scale)/d(emb).                    */
int occ_pair_dist_loss(const float* emb, const int32_t* pair_i, const int32_t* pair_j, const float* weights,
                       int64_t n_pairs, float bias, int relu, float* loss, float* demb, int64_t R, int64_t E,
                       float scale, void* stream);
/* F.pairwise_distance(ref, emb) of oc_classifier.py:193, 261: dist[i] = ||ref - emb[i] + 1e-6||_2.  */
int occ_pairwise_dist(const float* ref, const float* emb, float* dist, int64_t N, int64_t E, void* stream);

/* ---------------------------------------------------- optimizer (oc_training.py:324, 385) -- */
/* torch.optim.Adam step (no weight decay, no amsgrad) over a list of n f32 tensors given as device
 * arrays of pointers/sizes: p -= lr * m_hat / (sqrt(v_hat) + eps).  A NULL grads[t] skips tensor t
 * (torch skips parameters whose .grad is None); steps: i32 [n] per-tensor step counters kept on the
 * device and advanced by this call (torch keeps one step per parameter).  grads are multiplied by
 * grad_scale first (1/world_size after a summing all-reduce).  bf16_copies (device array of n pointers,
 * or NULL; entries may be NULL): tensor t's updated values are also written there as bf16 in the same
 * pass (the GEMM operands of a fine-tuned front-end).                                              */
int occ_adam_multi(void* const* params, void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                   const int64_t* sizes, int32_t* steps, int64_t n_tensors, int64_t max_size, float lr,
                   float beta1, float beta2, float eps, float grad_scale, void* const* bf16_copies, void* stream);

/* ------------------------------------------------------------------ GEMM family ----------- */
/* Row addressing shared by A, C and the residual R: with r = m % rows_per_batch, row m lives at
 *     base + (m / rows_per_batch) * batch_stride + r * row_stride                      (elements)
 * or, when rows_per_line > 0 (2-D images inside padded buffers),
 *     base + (m / rows_per_batch) * batch_stride + (r / rows_per_line) * line_stride + (r % rows_per_line) * row_stride
 * which expresses plain matrices (rows_per_batch = M), strided-window implicit GEMMs (Conv1d over a
 * channels-last signal: row_stride = stride*C, K = k*C; Conv2d over channels-last images) and writes
 * into the interior of zero-padded buffers.
 * A rows may additionally be split into `a_nseg` K-segments of `a_seg_len` contiguous elements that
 * are `a_seg_stride` apart (grouped / 2-D convolutions); a_nseg*a_seg_len == K.                   */
typedef struct occ_rowmap { int64_t rows_per_batch, batch_stride, row_stride, rows_per_line, line_stride; } occ_rowmap;
typedef struct occ_gemm_desc {
    int64_t M, N, K;
    const void* A; occ_rowmap a_map; int64_t a_nseg, a_seg_len, a_seg_stride;
    const void* W; int64_t ldw;                /* W[N,K] row-major (torch Linear layout), K contiguous */
    const void* bias;                          /* f32 [N] or NULL */
    const void* R; occ_rowmap r_map; int r_dtype;   /* residual added after activation, or NULL */
    void* C; occ_rowmap c_map; int c_dtype;
    int ab_dtype;                              /* OCC_BF16: bf16 MFMA (f32 accumulate); OCC_F32: exact f32 MFMA; OCC_F32_AS_BF16 */
    int act;                                   /* occ_act applied to (acc + bias) */
    float alpha;                               /* acc scaled by alpha before bias */
    /* grouped problems (grouped Conv1d): group g uses A + g*a_group_stride, W + g*w_group_stride and
     * output/bias/residual columns shifted by g*c_group_stride; n_groups <= 1 means a single problem. */
    int64_t n_groups, a_group_stride, w_group_stride, c_group_stride;
    /* optional bf16 side tensor addressed like C (same row map): with act = OCC_ACT_GELU the pre-activation (acc+bias) is
     * stored to it (saved for backward); with act = OCC_ACT_GELU_GRAD it is read (the saved pre-activation); OCC_ACT_GELU_KEEP_GRAD
     * stores gelu'(acc+bias) to it and OCC_ACT_MUL_AUX multiplies by it.  NULL = unused. */
    void* aux;
    /* fp8 operands (ab_dtype OCC_FP8_*): device scalars 1/scale of the per-tensor quantisation of A and W; the accumulator is
     * multiplied by alpha * (*a_dequant) * (*w_dequant) before bias.  NULL = 1.                                           */
    const float* a_dequant; const float* w_dequant;
    /* Optional fp8 copy of a bf16 result, written by the same epilogue (delayed per-tensor scaling: the operand of the NEXT fp8 GEMM,
     * which otherwise costs a quantisation pass over C): c_f8 u8 [M, N] (plain rows) = fp8 of the bf16-rounded C times *c_f8_scale
     * (saturating; c_f8_fmt OCC_FP8_E4M3 or OCC_FP8_E5M2), *c_f8_amax = max(*c_f8_amax, |C|).  Needs c_dtype bf16, no residual,
     * N % 8 == 0 and a launch that takes the 256-row kernel (else OCC_EUNSUPPORTED); NULL = off.                              */
    void* c_f8; const float* c_f8_scale; float* c_f8_amax; int c_f8_fmt;
    /* Optional column sums of a bf16 result from the same epilogue: c_colsum f32 [N] += sum over rows of the (bf16-rounded) C -- the
     * bias gradient of the layer whose output gradient C is (fc1: C = the gradient through GELU), which otherwise is a pass over C.
     * c_colsum_ws: caller-owned f32 scratch, >= 2 * ceil(M / 208) * N floats (per-tile partial sums, added in a fixed order by a
     * second small launch).  Same launch restrictions as c_f8.                                                              */
    float* c_colsum; float* c_colsum_ws; int64_t c_colsum_ws_floats;
    int c_colsum_defer;                        /* 1: partial sums only, no finalize launch: see occ_finalize_batch (the caller keeps c_colsum_ws zero-initialised: a launch writes its 2 * ceil(M / tile rows) rows only) */
} occ_gemm_desc;
/* C = act(alpha * A.W^T + bias) + R.  Replaces every nn.Linear / Conv1d of the path
 * (fairseq wav2vec2 layers reached from sslassist.py:48; AModel.LL sslassist.py:509).             */
int occ_gemm(const occ_gemm_desc* d, void* stream);
/* Per-tensor fp8 quantisation with delayed scaling (E4M3 forward / E5M2 gradients).
 * dst[i] = fp8(src[i] * *scale) saturating at the format's largest finite value (src f32 or bf16, fmt OCC_FP8_E4M3 / OCC_FP8_E5M2);
 * scale: device scalar or NULL (= 1); amax: device scalar or NULL, raised to max |src[i]| (atomic max) for the NEXT step's scale. */
int occ_fp8_quantize(const void* src, int src_dtype, void* dst, int fmt, int64_t n, const float* scale, float* amax, void* stream);
/* Many bf16 tensors in ONE launch: job i quantises src[0..n) (n % 8 == 0, 16-byte aligned) into dst with *scale (NULL = 1) and raises
 * *amax (NULL = not tracked); dst == NULL measures |max| only.  The job array lives in DEVICE memory, sorted by first_chunk = number of
 * 8192-element chunks of the jobs before it; total_chunks = their sum.                                                         */
typedef struct occ_fp8_job { const void* src; void* dst; int64_t n; const float* scale; float* amax; int64_t first_chunk; } occ_fp8_job;
int occ_fp8_quantize_batch(const occ_fp8_job* jobs_dev, int64_t n_jobs, int64_t total_chunks, int fmt, void* stream);
/* amax[i] = max |src| only (current scaling: the first step of a site, weights).                                            */
int occ_fp8_amax(const void* src, int src_dtype, int64_t n, float* amax, void* stream);
/* For n sites: scale[i] = fmax / (amax[i] * margin), inv_scale[i] = 1 / scale[i] (both left unchanged when amax[i] == 0:
 * nothing was measured since the last update), then amax[i] = 0.
 * fmax: 448 (e4m3) or 57344 (e5m2).                                                                                          */
int occ_fp8_update_scales(float* amax, float* scale, float* inv_scale, int64_t n, float fmax, float margin, void* stream);

/* Tuning hook: forces one kernel of the bf16 GEMM family instead of the size heuristic (1 = heuristic, the default; 30 = the
 * 256x256 eight-phase kernel, 31 = its 224-row form, 32 = its 208-row form; 3 = 256x128 LDS-DMA tile; 14 = half-slab pipeline; 22 = in-workgroup split-K); v < 0 only queries.
 * Returns the previous value.  Initialised from OCC_GEMM_VARIANT.  Results agree across kernels up to f32 summation order.  */
int occ_gemm_variant(int v);
/* Which kernel family the calling thread's last occ_gemm call launched (-1 before the first call): lets tests pin the dispatch. */
enum { OCC_GEMM_KERNEL_OTHER = 0, OCC_GEMM_KERNEL_P8 = 8, OCC_GEMM_KERNEL_P8_FP8 = 9, OCC_GEMM_KERNEL_P8_TAIL = 10 /* eight-phase kernel + a small-tile launch for the last partial round */,
       OCC_GEMM_KERNEL_P8_224 = 11 /* eight-phase kernel on 224-row (or 208-row) tiles */,
       OCC_GEMM_KERNEL_Q4 = 12 /* four-wave 256 x 128 kernel, two workgroups per CU (csrc/gemm_q4.hip) */ };
int occ_gemm_last_kernel(void);


/* Weight-gradient GEMM (f32): C[n1,n2] += alpha * sum_m A[m,n1] * B[m,n2]; A rows [N1] and B rows [N2] go through
 * row maps, B rows may be K-segmented conv windows; C f32 [N1, ldc] is accumulated with atomics.  Backward of
 * the Linear / Conv2d weights of the AASIST back-end.                                               */
typedef struct occ_gemm_tn_desc {
    int64_t M, N1, N2;
    const void* A; occ_rowmap a_map;
    const void* B; occ_rowmap b_map; int64_t b_nseg, b_seg_len, b_seg_stride;
    void* C; int64_t ldc;
    float alpha;
    void* colsum;                      /* optional f32 [N1]: += alpha * column sums of A (the bias gradient), or NULL */
    int a_dtype, b_dtype;              /* OCC_F32 (0, default) or OCC_BF16: operand storage; products and sums are f32 either way */
    int compute;                       /* OCC_F32 (0, default): exact-f32 MFMA; OCC_BF16: operands rounded to bf16, bf16 MFMA, f32 accumulate */
    /* Optional caller-owned device scratch (16-byte aligned) for the large bf16 products: the reduction over M is cut into pieces
     * over workgroups and the pieces meet in f32 slabs here instead of float atomics on C.  Its contents are undefined between calls;
     * calls that share one workspace must be ordered on one stream.  NULL / 0: the atomics path is used.  268 MB covers every
     * weight of XLS-R-300M/1B at any batch size (256 workgroups x 256 KiB slabs + tickets).                                        */
    void* workspace; int64_t workspace_bytes;
    /* n_groups > 1: that many independent products in one launch (the weight gradient of a grouped Conv1d: group g uses
     * A + g*a_group_stride, B + g*b_group_stride, C + g*c_group_stride, strides in elements); bf16 operands / bf16 MFMA only,
     * N1 >= 64, colsum must be NULL (sum the whole A once with occ_colsum).  0 / 1: one product.                                  */
    int64_t n_groups, a_group_stride, b_group_stride, c_group_stride;
    /* Hint, non-zero: C holds zeros (a gradient buffer cleared since its last use).  The large bf16 kernels then STORE alpha * product
     * instead of reading C back and adding -- the same result (0 + x = x) without the 4 bytes per element of read traffic.  Paths that
     * accumulate with atomics ignore it.                                                                                        */
    int c_is_zero;
} occ_gemm_tn_desc;
int occ_gemm_tn(const occ_gemm_tn_desc* d, void* stream);
/* Two weight gradients with the same reduction rows M (out-proj with qkv, fc2 with fc1 of one transformer layer) in ONE launch when both
 * take the large bf16 kernel (N1, N2 multiples of 256, plain row maps, M % 64 == 0, d0's workspace): half the slab traffic and longer
 * reduction pieces than two launches.  Anything else runs as occ_gemm_tn(d0) then occ_gemm_tn(d1); results are the same either way up to
 * the f32 summation order of the reduction pieces.                                                                              */
int occ_gemm_tn_pair(const occ_gemm_tn_desc* d0, const occ_gemm_tn_desc* d1, void* stream);
/* out[n] += alpha * sum_m A[m,n] (bias gradients).                                                   */
int occ_colsum(const void* A, int a_dtype, const occ_rowmap* a_map, int64_t M, int64_t N, float* out, float alpha, void* stream);   /* A f32 or bf16 */

/* --------------------------------------------- AASIST back-end (models/sslassist.py:58-597), f32 ---- */
int occ_fill_f32(float* p, float v, int64_t n, void* stream);
int occ_axpby_f32(const float* a, const float* b, float* out, float alpha, float beta, int64_t n, void* stream);   /* b may be NULL */
/* out[i0,i1,i2,i3] (contiguous) (+)= in[offset + sum i_k*stride_k]; shape/strides are HOST arrays of 4.  Used to
 * repack weights (flip / transpose) between the checkpoint layout and the GEMM layouts.              */
int occ_copy_strided(const float* in, float* out, int64_t offset, const int64_t* shape4_host, const int64_t* strides4_host,
                     int accumulate, void* stream);
/* y[y_map(r), :C] (+)= x[x_map(r), :C]: split / concatenate node sets, move rows in and out of padded buffers. */
int occ_copy_rows(const float* x, const occ_rowmap* x_map, float* y, const occ_rowmap* y_map, int64_t rows, int64_t C,
                  int accumulate, void* stream);
/* dx = dy * act'(.) with the derivative expressed through the activation OUTPUT y (GEMM-fused SELU / tanh / relu). */
int occ_act_bwd(const float* dy, const float* y, float* dx, int act, int64_t n, void* stream);
/* nn.Dropout: y = x*mask/(1-p); generate != 0 draws the keep-mask (Philox) and stores it, else uses the given mask. */
int occ_dropout(const float* x, float* y, uint8_t* mask, int64_t n, float p, uint64_t seed, uint64_t stream_id,
                int generate, void* stream);
/* The same draw with the step counter in DEVICE memory: Philox stream id = (*step << 8) + site (site < 256), i.e. exactly
 * occ_dropout(..., stream_id = (step << 8) + site, generate = 1).  A training step captured in a HIP graph then draws fresh masks on
 * every replay; occ_add_u64 (captured with it) advances the counter.                                                       */
int occ_dropout_step(const float* x, float* y, uint8_t* mask, int64_t n, float p, uint64_t seed, const uint64_t* step, uint64_t site, void* stream);
int occ_add_u64(uint64_t* counter, uint64_t v, void* stream);
/* F.max_pool2d(LL(x).transpose(1,2).unsqueeze(1), (3,3)) (sslassist.py:512-514): y [B,T,F] -> out [B,F/3,T/3] written
 * with element stride out_c, idx = window argmax; bwd scatters dout back into dy [B,T,F] (pre-zeroed).   */
int occ_stem_pool_fwd(const float* y, float* out, uint8_t* idx, int64_t B, int64_t T, int64_t F, int64_t out_c, void* stream);
int occ_stem_pool_bwd(const float* dout, const uint8_t* idx, float* dy, int64_t B, int64_t T, int64_t F, int64_t dout_c, void* stream);
/* BatchNorm (nn.BatchNorm1d/2d on channels-last rows) split as statistics + fused normalise/activation.
 * train: batch mean / biased variance -> mean,rstd (may be NULL to only move the running stats), running stats
 * updated with momentum (unbiased variance) like torch; eval: mean,rstd from the running stats.
 * ws: f64 scratch [512*C*2]; 256 % C == 0.                                                            */
int occ_bn_stats(const float* x, const occ_rowmap* x_map, int64_t rows, int64_t C, double* ws, float* mean, float* rstd,
                 float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps, int train,
                 void* stream);
int occ_bn_act_fwd(const float* x, const occ_rowmap* x_map, const float* mean, const float* rstd, const float* gamma,
                   const float* beta, int act, float* y, const occ_rowmap* y_map, int64_t rows, int64_t C, void* stream);
/* dx = d/dx of act(BN(x)) given dy (train-mode statistics); dgamma/dbeta (may be NULL) are accumulated; sums: f32 [2C] scratch. */
int occ_bn_act_bwd(const float* dy, const occ_rowmap* dy_map, const float* x, const occ_rowmap* x_map, const float* mean,
                   const float* rstd, const float* gamma, const float* beta, int act, float* dx, const occ_rowmap* dx_map,
                   float* dgamma, float* dbeta, double* ws, float* sums, int64_t rows, int64_t C, void* stream);
/* out[o,c] = sum_r x*softmax_r(w) (+pos[o % pos_period, c]) with element (o,r,c) at
 * (o/inner_n)*outer_stride + (o%inner_n)*inner_stride + r*r_stride + c  (sslassist.py:526-538).          */
int occ_softmax_wsum_fwd(const float* x, const float* w, int64_t n_outer, int64_t inner_n, int64_t outer_stride,
                         int64_t inner_stride, int64_t R, int64_t r_stride, int64_t C, const float* pos, int64_t pos_period,
                         float* out, void* stream);
int occ_softmax_wsum_bwd(const float* x, const float* w, int64_t n_outer, int64_t inner_n, int64_t outer_stride,
                         int64_t inner_stride, int64_t R, int64_t r_stride, int64_t C, const float* dm, float* dx, float* dw,
                         int accumulate, void* stream);
/* GraphAttentionLayer / HtrgGraphAttentionLayer pieces (sslassist.py:58-151, 154-329); x [B,N,D], P/A [B,N,N,D], alpha [B,N,N]. */
int occ_pair_mul(const float* x, float* P, int64_t B, int64_t N, int64_t D, void* stream);
int occ_pair_mul_bwd(const float* dP, const float* x, float* dx, int64_t B, int64_t N, int64_t D, int accumulate, void* stream);
/* alpha = softmax_j(A . aw_type(i,j) * inv_temp); aw f32 [3][Do] = (att_weight11, att_weight22, att_weight12), n1 = #type-1 nodes
 * (homogeneous layers: n1 = N, only the first vector is used).                                           */
int occ_gat_softmax(const float* A, const float* aw, int64_t B, int64_t N, int64_t Do, int64_t n1, float inv_temp, float* alpha,
                    void* stream);
/* trans=0: out[b,i,:] (+)= sum_j alpha[b,i,j] x[b,j,:];  trans=1: out[b,j,:] (+)= sum_i alpha[b,i,j] x[b,i,:].   */
int occ_bmm_alpha(const float* alpha, const float* x, float* out, int64_t B, int64_t N, int64_t D, int trans, int accumulate,
                  void* stream);
int occ_gat_dscore(const float* alpha, const float* dh, const float* x, float* ds, int64_t B, int64_t N, int64_t D, float inv_temp,
                   void* stream);
/* in place A <- dZ = ds*aw_type*(1-A^2); daw [3][Do] += sum ds*A.                                          */
int occ_gat_dz(float* A, const float* ds, const float* aw, int64_t B, int64_t N, int64_t Do, int64_t n1, float* daw, void* stream);
/* master-node update of HtrgGraphAttentionLayer (sslassist.py:234-239, 255-270, 310-316).                */
typedef struct occ_master_desc {
    int64_t B, N, D, Do;
    const float* x;                    /* [B,N,D] concatenated, dropped-out node features */
    const float* master; int64_t master_bstride;   /* [B,D] (stride D) or the shared parameter [1,D] (stride 0) */
    const float *att_projM_w, *att_projM_b, *att_weightM, *proj_with_attM_w, *proj_with_attM_b, *proj_without_attM_w,
        *proj_without_attM_b;
    float inv_temp;
    float* out;                        /* [B,Do] new master */
    float *am, *agg;                   /* saved for backward: [B,N], [B,D] */
} occ_master_desc;
typedef struct occ_master_grads {
    const float* dout;                 /* [B,Do] */
    float* dx; int dx_accumulate;      /* [B,N,D] */
    float* dmaster; int64_t dmaster_bstride;       /* accumulated atomically */
    float *d_att_projM_w, *d_att_projM_b, *d_att_weightM, *d_proj_with_attM_w, *d_proj_with_attM_b, *d_proj_without_attM_w,
        *d_proj_without_attM_b;        /* accumulated atomically */
} occ_master_grads;
int occ_master_fwd(const occ_master_desc* d, void* stream);
int occ_master_bwd(const occ_master_desc* d, const occ_master_grads* g, void* stream);
/* GraphPool (sslassist.py:332-368): scores = sigmoid(proj(drop(h))), keep the k best nodes, out = h*score in rank order. */
int occ_graph_pool_fwd(const float* h, const uint8_t* mask, float drop_p, const float* w, const float* bias, int64_t B, int64_t N,
                       int64_t D, int64_t k, float* out, int32_t* idx, float* scores, void* stream);
int occ_graph_pool_bwd(const float* h, const uint8_t* mask, float drop_p, const float* w, const float* scores, const int32_t* idx,
                       const float* dout, int64_t B, int64_t N, int64_t D, int64_t k, float* dh, float* dw, float* dbias, void* stream);
/* read-out + classifier (sslassist.py:573-597); masks are keep-masks (NULL = no dropout).                 */
typedef struct occ_readout_desc {
    int64_t B, Nt, Ns, Dg, n_classes;
    const float *T1, *T2, *S1, *S2, *M1, *M2;
    const uint8_t *mask_T1, *mask_T2, *mask_S1, *mask_S2, *mask_M1, *mask_M2, *mask_last;
    float p_way, p_last;
    const float *out_w, *out_b;
    float *emb, *logits;
} occ_readout_desc;
typedef struct occ_readout_grads {
    const float *demb, *dlogits;
    float *dT1, *dT2, *dS1, *dS2, *dM1, *dM2;
    float *d_out_w, *d_out_b;
} occ_readout_grads;
int occ_readout_fwd(const occ_readout_desc* d, void* stream);
int occ_readout_bwd(const occ_readout_desc* d, const occ_readout_grads* g, void* stream);

/* ---------------------------------------------------- SE-ResNet34 back-end pieces (models/senet.py:13-156), f32 channels-last ---- */
/* nn.MaxPool2d(3, stride 2, padding 1): x [B,H,W,C] -> y (rows through y_map) + argmax taps; bwd scatter-adds into a pre-zeroed dx.   */
int occ_maxpool3s2_fwd(const float* x, float* y, const occ_rowmap* y_map, uint8_t* idx, int64_t B, int64_t H, int64_t W, int64_t C, void* stream);
int occ_maxpool3s2_bwd(const float* dy, const occ_rowmap* dy_map, const uint8_t* idx, float* dx, int64_t B, int64_t H, int64_t W, int64_t C, void* stream);
/* out[b,c] = alpha * sum_r x[b,r,c] (AdaptiveAvgPool2d(1) of SELayer / the network head); x rows (b*R + r) through a row map.         */
int occ_batch_colsum(const float* x, const occ_rowmap* x_map, int64_t B, int64_t R, int64_t C, float alpha, float* out, void* stream);
/* SELayer.fc (senet.py:17-22): g = sigmoid(W2 relu(W1 s)); z [B,Cr] saved; bwd accumulates dW1, dW2 and returns ds.                   */
int occ_se_gate_fwd(const float* s, const float* W1, const float* W2, int64_t B, int64_t C, int64_t Cr, float* z, float* g, void* stream);
int occ_se_gate_bwd(const float* s, const float* z, const float* g, const float* dg, const float* W1, const float* W2, int64_t B, int64_t C,
                    int64_t Cr, float* dW1, float* dW2, float* ds, void* stream);
/* SEBasicBlock tail (senet.py:53-61): out = relu(y*gate[b,c] + res); bwd gives dy, dres (write or accumulate) and dgate (accumulated). */
int occ_se_scale_add_relu(const float* y, const float* gate, const float* res, const occ_rowmap* res_map, float* out, const occ_rowmap* out_map,
                          int64_t B, int64_t R, int64_t C, void* stream);
int occ_se_scale_add_relu_bwd(const float* dout, const occ_rowmap* dout_map, const float* out, const occ_rowmap* out_map, const float* y,
                              const float* gate, float* dy, float* dres, const occ_rowmap* dres_map, int dres_accumulate, float* dgate, int64_t B,
                              int64_t R, int64_t C, void* stream);
/* x[b,r,c] += v[b,c] (gradient of a global average pool).                                                                     */
int occ_add_batch_vec(float* x, const occ_rowmap* x_map, const float* v, int64_t B, int64_t R, int64_t C, void* stream);
/* Input gradient of the stem conv (senet.py:73: Conv2d(1,16,7,stride 2,padding 3)): dy [B,Ho,Wo,16], w [16][7][7][4] (input channel padded
 * to 4, as the forward implicit GEMM uses it) -> dx [B,H,W]; needed only when the front-end under the SE-ResNet is fine-tuned.            */
int occ_conv7s2_dgrad_c1(const float* dy, const float* w, float* dx, int64_t B, int64_t H, int64_t W, void* stream);

/* ---------------------------------------------------- LCNN back-end pieces (models/lcnn.py:121-241), f32 channels-last ---- */
/* Max-Feature-Map (mfm.forward, lcnn.py:133-136): y[r, c] = max(x[r, c], x[r, C + c]), x rows of 2C contiguous floats, y rows through y_map.
 * bwd: torch.maximum's derivative -- the larger half takes dy, a tie splits it evenly; dx [rows, 2C] is written completely.            */
int occ_mfm_fwd(const float* x, float* y, const occ_rowmap* y_map, int64_t rows, int64_t C, void* stream);
int occ_mfm_bwd(const float* dy, const occ_rowmap* dy_map, const float* x, float* dx, int64_t rows, int64_t C, void* stream);
/* MFM followed by nn.MaxPool2d(2, 2) (lcnn.py:154-166) in one pass: x [B,H,W,2C] -> y [B,H/2,W/2,C] (rows through y_map), idx u8
 * [B,H/2,W/2,C] = winning window slot | winning half << 2; bwd writes all of dx [B,H,W,2C] (rows of 2C floats through dx_map, e.g. the
 * interior of the zero-bordered buffer the input-gradient correlation reads).                                                         */
int occ_mfm_pool2_fwd(const float* x, float* y, const occ_rowmap* y_map, uint8_t* idx, int64_t B, int64_t H, int64_t W, int64_t C, void* stream);
int occ_mfm_pool2_bwd(const float* dy, const occ_rowmap* dy_map, const uint8_t* idx, float* dx, const occ_rowmap* dx_map, int64_t B, int64_t H, int64_t W, int64_t C,
                      void* stream);
/* nn.AdaptiveAvgPool2d((1, Wout)) + flatten (lcnn.py:169, 191-194): x [B,H,W,C] -> out [B, C*Wout] with out[b, c*Wout + i] = mean over
 * all rows and torch's adaptive column bin i.                                                                                         */
int occ_adaptive_avgpool_1xw_fwd(const float* x, float* out, int64_t B, int64_t H, int64_t W, int64_t C, int64_t Wout, void* stream);
int occ_adaptive_avgpool_1xw_bwd(const float* dout, float* dx, int64_t B, int64_t H, int64_t W, int64_t C, int64_t Wout, void* stream);

/* ------------------------------------------------------------- front-end row kernels ------- */
/* y = LayerNorm(x) * gamma + beta, optional GELU, over rows of width C (C % 64 == 0, C <= 8192).
 * fairseq LayerNorm / Fp32LayerNorm + GELU of the conv blocks and transformer layers.            */
int occ_layernorm(const void* x, int x_dtype, void* y, int y_dtype, const float* gamma, const float* beta,
                  int64_t rows, int64_t C, float eps, int gelu, void* stream);
/* First conv block of the wav2vec2 feature extractor fused: Conv1d(1->C,k,stride,bias) -> LayerNorm(C)
 * -> GELU.  wav f32 [B,L] -> out [B,Tout,C] (channels-last).  w f32 [C,k]; C == 512, k <= 16.       */
int occ_conv0_ln_gelu(const float* wav, const float* w, const float* bias, const float* gamma,
                      const float* beta, void* out, int out_dtype, int64_t B, int64_t L, int64_t Tout,
                      int64_t C, int64_t k, int64_t stride, float eps, void* stream);
/* Multi-head self-attention core softmax(Q.K^T).V (un-masked, as the reference runs it: fairseq MultiheadAttention reached from
 * sslassist.py:48; evaluation scores un-padded batch-1 utterances of any length, oc_classifier.py:185-193).
 * qkv: [B*T, 3*D] rows (q | k | v); scores are scale * q.k (fairseq scales q by hd^-0.5); out: [B*T, D].
 * bf16, head_dim 64: MFMA kernels, any T (T <= 256: whole key set on chip; longer: keys streamed in blocks of 128 with the
 * online-softmax recurrence).  f32 (parity path) or other head dims: f32-arithmetic LDS kernels, any T (the whole head on chip while
 * it fits 160 KiB of LDS, ~300 frames at hd 64; keys streamed in blocks of 64 with the online-softmax recurrence beyond).
 * lse (optional, f32 [B*H, T], MFMA kernels only): log2-sum-exp2 of the scaled scores, kept for backward.                */
int occ_attention(const void* qkv, void* out, int dtype, int64_t B, int64_t T, int64_t H, int64_t hd,
                  int64_t ld_qkv, int64_t ld_out, float scale, float* lse, void* stream);
/* The same on a zero-padded batch of utterances of unequal length (scoring in length-sorted batches instead of the reference's
 * one-utterance loop, oc_classifier.py:182-186, 256-261): kv_len int32 [B] (device) = valid frames of each utterance; keys at or past
 * it are masked, so rows [0, kv_len[b]) equal the un-padded single-utterance result; rows past it are not meaningful.
 * f32-arithmetic kernels (storage f32 or bf16), any T.                                                                    */
/* f32 -> three bf16 K-panels [rows, 3K] for an f32-grade product on the bf16 MFMA as ONE GEMM of depth 3K (no reference counterpart: the
 * reference scores in fp32, oc_classifier.py:159-202; this is the arithmetic that keeps its accuracy class at the bf16 kernels' rate):
 * mode 0 (activations) [hi | lo | hi], mode 1 (weights) [hi | hi | lo], hi = bf16(v), lo = bf16(v - hi).  x rows through x_map, K %% 8 == 0. */
int occ_split3_bf16(const float* x, const occ_rowmap* x_map, void* out, int64_t rows, int64_t K, int mode, void* stream);
int occ_attention_varlen(const void* qkv, void* out, int dtype, int64_t B, int64_t T, int64_t H, int64_t hd,
                         int64_t ld_qkv, int64_t ld_out, float scale, const int32_t* kv_len, void* stream);
/* ---- backward of the transformer encoder (fine-tuning; autograd of fairseq's pre-LN TransformerSentenceEncoderLayer) ---- */
/* dst[c, r] = bf16(src[r, c]) (src f32 or bf16 [rows, ld_src], dst bf16 [cols, ld_dst >= rows]): K-contiguous operands for the
 * weight-gradient GEMMs dW = dY^T X, which then run on the same bf16 MFMA GEMM as the forward.                */
/* colsum (optional f32 [cols]): += column sums of src -- the bias gradient comes for free with the dY^T operand.           */
int occ_transpose_bf16(const void* src, int src_dtype, void* dst, int64_t rows, int64_t cols, int64_t ld_src, int64_t ld_dst,
                       float* colsum, void* stream);
/* n_jobs transposes dst[c, r] = bf16(src[r, c]) in ONE launch.  The job array lives in DEVICE memory (the caller builds it once: the
 * operands of a model do not move), sorted by first_tile = number of 64x64 tiles of the jobs before it; total_tiles = their sum.
 * src_dtype OCC_F32 or OCC_BF16.                                                                                           */
typedef struct occ_transpose_job { const void* src; void* dst; int64_t rows, cols, ld_src, ld_dst, first_tile; int64_t src_dtype; } occ_transpose_job;
int occ_transpose_bf16_batch(const occ_transpose_job* jobs_dev, int64_t n_jobs, int64_t total_tiles, void* stream);
/* dx = LayerNorm'(x)^T dy (+ dres, the residual-branch gradient; dx may alias dres); dgamma += sum dy*xhat; dbeta += sum dy. */
/* Same with the source rows addressed through a row map (strided conv windows, interiors of padded buffers).               */
int occ_transpose_bf16_rows(const void* src, int src_dtype, const occ_rowmap* src_map, void* dst, int64_t rows, int64_t cols,
                            int64_t ld_dst, float* colsum, void* stream);
/* General form: x f32 or bf16; gelu != 0 treats dy as the gradient wrt gelu(LN(x)) (conv blocks of the feature extractor, needs
 * beta); dx (f32, contiguous) and dx_bf16 (through dx_bf16_map, NULL = contiguous) are each optional.
 * scratch (optional, caller-owned, 16-byte aligned, 512*C floats cover every case): the per-workgroup dgamma / dbeta sums go
 * through it (plain stores + a small second launch) instead of global float atomics; NULL = atomics.                       */
int occ_layernorm_bwd_ex(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* gamma, const float* beta,
                         const float* dres, float* dx, void* dx_bf16, const occ_rowmap* dx_bf16_map, float* dgamma, float* dbeta,
                         int64_t rows, int64_t C, float eps, int gelu, float* scratch, int64_t scratch_floats, void* stream);
/* y = residual + scale * x * keep / (1 - p), x / y f32 or bf16 (may alias), residual f32 or NULL, mask u8 [n] or NULL (= no dropout):
 * the train-mode dropouts of fairseq's Wav2Vec2Model (dropout_input, dropout, activation_dropout; active because the reference calls
 * aasist.train(), oc_training.py:351) and their backward (generate = 0 with the stored mask).  generate != 0 draws the mask (Philox).   */
int occ_dropout_ex(const void* x, int x_dtype, void* y, int y_dtype, uint8_t* mask, const float* residual, int64_t n, float p, float scale,
                   uint64_t seed, uint64_t stream_id, int generate, void* stream);
/* out (bf16, through out_map) = dy (f32 [rows,C]) * gelu'(u (bf16 [rows,C])): gradient through the positional conv's GELU.   */
int occ_gelu_bwd_rows(const float* dy, const void* u, void* out, const occ_rowmap* out_map, int64_t rows, int64_t C, void* stream);
/* Backward of occ_conv0_ln_gelu (recomputes the block from the waveform): dw [C,k], dbias, dgamma, dbeta accumulated.       */
int occ_conv0_ln_gelu_bwd(const float* wav, const float* w, const float* bias, const float* gamma, const float* beta, const void* dact,
                          int dact_dtype, float* dw, float* dbias, float* dgamma, float* dbeta, int64_t B, int64_t L, int64_t Tout,
                          int64_t C, int64_t k, int64_t stride, float eps, void* stream);
/* weight_norm(dim=2) of fairseq's pos_conv: v f32 [O,I,K], g f32 [K] -> bf16 GEMM operands w_fwd [G][O/G][K][I] and (optional)
 * w_bwd [G][I][K][O/G] (taps reversed, for the input gradient), norms f32 [K]; bwd maps a packed-layout weight gradient to dv, dg.
 * scratch (optional, caller-owned): with at least 512*K floats (pack) / (O+1)*K floats (bwd) and 256 % K == 0 the coalesced kernels run
 * (LDS transposes, fixed-order partial sums); NULL = one workgroup per tap with strided reads.                                   */
int occ_weight_norm_pack(const float* v, const float* g, void* w_fwd, void* w_bwd, float* norms, int64_t O, int64_t I, int64_t K,
                         int64_t G, float* scratch, int64_t scratch_floats, void* stream);
int occ_weight_norm_bwd(const float* v, const float* g, const float* norms, const float* dw_packed, float* dv, float* dg, int64_t O,
                        int64_t I, int64_t K, int64_t G, float* scratch, int64_t scratch_floats, void* stream);
/* dx_bf16 (optional): a bf16 copy of dx for the next input-gradient GEMMs.                                             */
int occ_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* gamma, const float* dres, float* dx, void* dx_bf16,
                      float* dgamma, float* dbeta, int64_t rows, int64_t C, float eps, float* scratch, int64_t scratch_floats, void* stream);
/* LayerNorm (f32 rows in, bf16 out) that also writes the fp8 operand of the GEMM consuming it: y_f8 (u8 [rows, C]) = e4m3 of the
 * bf16-rounded output times *f8_scale (saturating), *f8_amax = max(*f8_amax, |y_bf16|) -- delayed per-tensor scaling, see occ_fp8_quantize. */
int occ_layernorm_fp8(const float* x, void* y_bf16, void* y_f8, const float* f8_scale, float* f8_amax, const float* gamma, const float* beta,
                      int64_t rows, int64_t C, float eps, void* stream);
/* occ_layernorm_bwd with the work of two stand-alone passes folded in (transformer layers of the fine-tuned front-end, where this
 * LayerNorm's input is x = residual + Linear(.)): dbias (optional f32 [C]) += column sums of the OUTPUT dx = the bias gradient of that
 * Linear; dx_f8 (optional u8 [rows, C]) = e5m2 of the bf16-rounded dx times *f8_scale (saturating), *f8_amax = max(*f8_amax, |dx_bf16|)
 * -- the operand and the next-step statistics of the fp8 input-gradient GEMM.  rows >= 2048, C <= 1536, scratch >= 768*C floats.    */
int occ_layernorm_bwd_fused(const void* dy, int dy_dtype, const float* x, const float* gamma, const float* dres, float* dx, void* dx_bf16,
                            float* dgamma, float* dbeta, float* dbias, void* dx_f8, const float* f8_scale, float* f8_amax, int64_t rows,
                            int64_t C, float eps, float* scratch, int64_t scratch_floats, int defer, void* stream);
/* defer (here, in occ_gemm_desc.c_colsum_defer and in occ_attention_bwd_bias): 1 = leave the per-workgroup partial sums in the scratch buffer
 * (which must then be this call site's own until occ_finalize_batch has run) and skip the finalize launch; the caller adds every site's sums
 * with ONE occ_finalize_batch launch after the backward pass.  Job layouts:
 *   kind 0 (LayerNorm / column sums): partials [n0 rows][n1], out[i] += sum of the rows; i < n2 -> out0, < 2 n2 -> out1, else out2 (may be NULL).
 *           LayerNorm: n0 = min(256, ceil(rows / 48)), n1 = 3 C, n2 = C, outs = dgamma, dbeta, dbias.  GEMM column sums: n0 = 2 ceil(M / tile rows)
 *           (208, 224 or 256, the dispatcher's choice: give n0 = 2 * ceil(M / 208) and keep the buffer zero-initialised), n1 = n2 = N, out0 = c_colsum.
 *   kind 1 (attention q|k|v bias, one key block): partials [n0 = B][n1 = H][3][n2 = hd], out0 = dbias [3 * H * hd].
 * n_blocks: kind 0 ceil(n1 / 64), kind 1 3 * n1; first_block = running sum, ascending.                                                */
typedef struct occ_finalize_job {
    const float* partials; float* out0; float* out1; float* out2;
    int32_t kind, n0, n1, n2, first_block, n_blocks;
} occ_finalize_job;
int occ_finalize_batch(const occ_finalize_job* jobs_dev, int64_t n_jobs, int64_t total_blocks, void* stream);
/* The keep-mask of occ_dropout_ex(generate = 1) on its own: mask[i] = 1 with probability 1 - p, Philox4x32-10 counter (i / 4, stream_id),
 * key seed -- the same bytes occ_dropout_ex writes for the same (n, p, seed, stream_id).                                          */
int occ_dropout_mask(uint8_t* mask, int64_t n, float p, uint64_t seed, uint64_t stream_id, void* stream);
/* The same attention core with fairseq's attention_dropout (MultiheadAttention: attn_probs = dropout(softmax(.)), live in the
 * reference because XLS-R stays in train mode: oc_training.py:352, sslassist.py:20-29): keep is a u8 mask [B*H, T, Tp], Tp = T rounded
 * up to a multiple of 4, 4-byte aligned (non-zero = keep; draw it with occ_dropout_ex(generate = 1) or pass your own), p the drop
 * probability.  bf16 only, head_dim 64 or 80, any T; lse (f32 [B*H, T], may be NULL) is the log-sum-exp of the UNdropped scores.
 * The backward takes the same mask.                                                                                             */
int occ_attention_dropout(const void* qkv, void* out, int64_t B, int64_t T, int64_t H, int64_t hd, int64_t ld_qkv, int64_t ld_out,
                          float scale, float* lse, const uint8_t* keep, float p, void* stream);
int occ_attention_bwd_dropout(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, int64_t B, int64_t T,
                              int64_t H, int64_t hd, int64_t ld_qkv, int64_t ld_o, float scale, float* dq_accum, const uint8_t* keep,
                              float p, void* stream);
/* occ_attention_bwd for T <= 256 that also accumulates the q|k|v bias gradient: dbias f32 [3D] += column sums of the (bf16) dqkv it
 * writes -- the separate pass over dqkv folded into the kernel that produces it.  bias_ws: f32 scratch >= B*H*3*hd floats.       */
int occ_attention_bwd_bias(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, int64_t B, int64_t T, int64_t H,
                           int64_t hd, int64_t ld_qkv, int64_t ld_o, float scale, float* dbias, float* bias_ws, int64_t bias_ws_floats,
                           int defer, void* stream);
/* dq|dk|dv (bf16 [B*T, 3D], same layout as qkv) of softmax(scale q.k^T) v given o (forward output), dout and the forward's lse.
 * head_dim 64 or 80, any T.  T > 256 needs dq_accum: caller-owned f32 scratch [B*T, H*hd] (16-byte aligned) in which the key blocks
 * of a head meet; it may be NULL for T <= 256.                                                                   */
int occ_attention_bwd(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, int64_t B, int64_t T, int64_t H,
                      int64_t hd, int64_t ld_qkv, int64_t ld_o, float scale, float* dq_accum, void* stream);

/* ---- LFCC front-end (replaces utils.py:127-138 extract_lfcc -> spafe lfcc; parity unpinned: oracle/lfcc_ref.py) ----
 * The DFT, the linear filter bank and the DCT run as f32 occ_gemm calls; these are the stages between them.
 * frames [B*F, ld] = window * pre-emphasised samples (pe(n) = x[n] - c x[n-1]; zeros past the signal, columns frame_len..ld-1 zero).   */
int occ_lfcc_frames(const float* wav, const float* window, float* frames, int64_t B, int64_t L, int64_t F, int64_t frame_len, int64_t hop,
                    int64_t ld, float pre_emph, void* stream);
/* spec rows = [re(0..nbins-1) | im(0..nbins-1)]; power[r,k] = scale (re^2 + im^2), columns nbins..ld_power-1 zeroed (GEMM K padding).      */
int occ_lfcc_power(const float* spec, float* power, int64_t rows, int64_t nbins, int64_t ld_spec, int64_t ld_power, float scale, void* stream);
/* x = log(x == 0 ? eps : x) in place (spafe zero_handling + numpy.log).                                                                    */
int occ_log_eps(float* x, int64_t n, float eps, void* stream);
/* normalize="mvn": per utterance and coefficient over its F frames, out = (x - mean) / population std.                                     */
int occ_mvn_frames(const float* x, float* out, int64_t B, int64_t F, int64_t C, int64_t ld_x, int64_t ld_out, void* stream);

/* ---- input pipeline: FLAC decode on the host (replaces librosa.load -> libsndfile at oc_training.py:214, 234,
 * oc_classifier.py:93, data_utils_SSL.py:66, 91).  No GPU work.  buf = the whole file in host memory.
 * occ_flac_info: info[0..3] = sample rate, channels, bits per sample, 1 if STREAMINFO carries an MD5; total = samples per channel
 * (0 = unknown); md5 (optional) = the 16 signature bytes.  occ_flac_decode: out = interleaved int32 [capacity * channels],
 * *decoded = samples per channel written; frame-header CRC-8 and frame CRC-16 are verified (OCC_EINVAL + occ_last_error on mismatch).   */
int occ_flac_info(const uint8_t* buf, int64_t n, int32_t* info, int64_t* total, uint8_t* md5);
int occ_flac_decode(const uint8_t* buf, int64_t n, int32_t* out, int64_t capacity, int64_t* decoded);

#ifdef __cplusplus
}
#endif
#endif
