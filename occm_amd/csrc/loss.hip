// One-class losses, distance scoring and the Adam step.
// Reference: losses/custom_loss.py:4-29, 78-99; oc_classifier.py:193, 261; oc_training.py:324, 385.
#include "occ_common.h"

namespace {

constexpr int LOSS_THREADS = 256;
constexpr float PAIR_EPS = 1e-6f;       // F.pairwise_distance eps, added to the difference

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x / OCC_WAVE, l = threadIdx.x % OCC_WAVE;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// One workgroup; groups are walked in order so the sum is reproducible.
__global__ __launch_bounds__(LOSS_THREADS) void compactness_kernel(const float* __restrict__ emb, float* __restrict__ loss,
                                                                  float* __restrict__ demb, int n_groups, int group, int E, float scale) {
    __shared__ float red[4];
    __shared__ float inv_dist[6];
    constexpr int NBMAX = 6;                             // custom_loss.py:15: batch_embeddings[:6]
    const int NB = group < NBMAX ? group : NBMAX;        // a shorter batch simply has fewer rows
    float total = 0.f;
    for (int g = 0; g < n_groups; ++g) {
        const float* e = emb + (size_t)g * group * E;
        float d2[NBMAX] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int c = threadIdx.x; c < E; c += LOSS_THREADS) {
            float v[NBMAX], S = 0.f;
#pragma unroll
            for (int i = 0; i < NBMAX; ++i) { v[i] = i < NB ? e[(size_t)i * E + c] : 0.f; S += v[i]; }
#pragma unroll
            for (int i = 0; i < NBMAX; ++i) {
                const float u = v[i] - (S - v[i]) / (float)(NB - 1) + PAIR_EPS;
                if (i < NB) d2[i] += u * u;
            }
        }
        float gl = 0.f;
#pragma unroll
        for (int i = 0; i < NBMAX; ++i) {
            const float dist = sqrtf(block_sum_256(d2[i], red));
            if (i < NB) gl += dist;
            if (threadIdx.x == 0) inv_dist[i] = (i < NB && dist > 0.f) ? 1.f / dist : 0.f;
        }
        total += gl / (float)NB;
        if (demb) {
            __syncthreads();
            float* de = demb + (size_t)g * group * E;
            const float k = scale / ((float)NB * (float)n_groups);
            for (int c = threadIdx.x; c < E; c += LOSS_THREADS) {
                float v[NBMAX], S = 0.f, uh[NBMAX], uh_sum = 0.f;
#pragma unroll
                for (int i = 0; i < NBMAX; ++i) { v[i] = i < NB ? e[(size_t)i * E + c] : 0.f; S += v[i]; }
#pragma unroll
                for (int i = 0; i < NBMAX; ++i) {
                    uh[i] = (v[i] - (S - v[i]) / (float)(NB - 1) + PAIR_EPS) * inv_dist[i];
                    uh_sum += uh[i];
                }
#pragma unroll
                for (int i = 0; i < NBMAX; ++i)
                    if (i < NB) de[(size_t)i * E + c] = k * ((float)NB / (float)(NB - 1) * uh[i] - uh_sum / (float)(NB - 1));
                for (int i = NB; i < group; ++i) de[(size_t)i * E + c] = 0.f;
            }
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) loss[0] = total / (float)n_groups;
}

__global__ __launch_bounds__(LOSS_THREADS) void ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                         float* __restrict__ loss, float* __restrict__ dlogits, int B, int C, float scale) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int r = threadIdx.x; r < B; r += LOSS_THREADS) {
        const float* x = logits + (size_t)r * C;
        float m = x[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(x[c] - m);
        const float lse = m + logf(s);
        const int y = (int)labels[r];
        acc += lse - x[y];
        if (dlogits)
            for (int c = 0; c < C; ++c)
                dlogits[(size_t)r * C + c] = (expf(x[c] - lse) - (c == y ? 1.f : 0.f)) * scale / (float)B;
    }
    const float tot = block_sum_256(acc, red);
    if (threadIdx.x == 0) loss[0] = tot / (float)B;       // custom_loss.py:99: sum / len
}

__global__ void pairwise_dist_kernel(const float* __restrict__ ref, const float* __restrict__ emb, float* __restrict__ dist, int N, int E) {
    const int row = blockIdx.x * (blockDim.x / OCC_WAVE) + threadIdx.x / OCC_WAVE;
    const int lane = threadIdx.x % OCC_WAVE;
    if (row >= N) return;
    float s = 0.f;
    for (int c = lane; c < E; c += OCC_WAVE) {
        const float d = ref[c] - emb[(size_t)row * E + c] + PAIR_EPS;
        s += d * d;
    }
    s = wave_sum(s);
    if (lane == 0) dist[row] = sqrtf(s);
}

// Signed sum of pair distances: loss = act(bias + sum_k w[k] * ||e[pi[k]] - e[pj[k]] + eps||), act = relu or identity
// (triplet_loss custom_loss.py:32-57: pairs (0,1,+1), (0,2,-1), bias = margin, relu; euclidean_distance_loss :59-74: five pairs of
// weight 1/5).  One workgroup, pairs walked in order (reproducible); demb = d(loss * scale)/d(emb), zero where the relu is shut.
constexpr int PAIR_MAX = 16;
__global__ __launch_bounds__(LOSS_THREADS) void pair_dist_loss_kernel(const float* __restrict__ emb, const int32_t* __restrict__ pi,
                                                                       const int32_t* __restrict__ pj, const float* __restrict__ w, int n_pairs,
                                                                       float bias, int relu, float* __restrict__ loss, float* __restrict__ demb,
                                                                       int R, int E, float scale) {
    __shared__ float red[4];
    __shared__ float coef[PAIR_MAX];
    float total = bias;
    for (int k = 0; k < n_pairs; ++k) {
        const float* a = emb + (size_t)pi[k] * E;
        const float* b = emb + (size_t)pj[k] * E;
        float s = 0.f;
        for (int c = threadIdx.x; c < E; c += LOSS_THREADS) { const float d = a[c] - b[c] + PAIR_EPS; s += d * d; }
        const float dist = sqrtf(block_sum_256(s, red));
        total += w[k] * dist;
        if (threadIdx.x == 0) coef[k] = dist > 0.f ? w[k] / dist : 0.f;
    }
    const bool open_ = !relu || total > 0.f;
    if (threadIdx.x == 0) loss[0] = open_ ? total : 0.f;
    if (!demb) return;
    __syncthreads();
    for (int i = threadIdx.x; i < R * E; i += LOSS_THREADS) demb[i] = 0.f;
    __syncthreads();
    if (!open_) return;
    for (int k = 0; k < n_pairs; ++k) {                  // pairs in order, one column per thread: no two threads touch one element
        const int ia = pi[k], ib = pj[k];
        for (int c = threadIdx.x; c < E; c += LOSS_THREADS) {
            const float g = scale * coef[k] * (emb[(size_t)ia * E + c] - emb[(size_t)ib * E + c] + PAIR_EPS);
            demb[(size_t)ia * E + c] += g;
            demb[(size_t)ib * E + c] -= g;
        }
    }
}

// torch.optim.Adam keeps one step counter per parameter and skips parameters whose grad is None.
__global__ void adam_bump_steps_kernel(void* const* __restrict__ grads, int32_t* __restrict__ steps, int n) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n && grads[t] != nullptr) steps[t] += 1;
}

// bf16 (optional, per tensor): a bf16 copy of the updated parameters written in the same pass -- the GEMM operands of a fine-tuned
// front-end, which a separate cast kernel would otherwise re-read 1.26 GB of f32 for every step.
__global__ __launch_bounds__(256) void adam_multi_kernel(void* const* __restrict__ params, void* const* __restrict__ grads,
                                                         void* const* __restrict__ m1, void* const* __restrict__ m2,
                                                         const int64_t* __restrict__ sizes, const int32_t* __restrict__ steps,
                                                         float lr, float b1, float b2, float eps, float gscale, void* const* __restrict__ bf16) {
    const int t = blockIdx.y;
    const float stepf = (float)steps[t];
    const float bc1 = 1.f - powf(b1, stepf);
    const float bc2_sqrt = sqrtf(1.f - powf(b2, stepf));
    const int64_t n = sizes[t];
    float* p = (float*)params[t];
    const float* g = (const float*)grads[t];
    float* m = (float*)m1[t];
    float* v = (float*)m2[t];
    unsigned short* pb = bf16 ? (unsigned short*)bf16[t] : nullptr;
    if (g == nullptr) return;                              // parameter without gradient (dead bn1, quirk 1)
    const float step_size = lr / bc1;
    auto upd = [&](float pi, float gr, float& mi, float& vi) {
        const float gi = gr * gscale;
        mi = mi * b1 + (1.f - b1) * gi;                     // exp_avg.lerp_(grad, 1-beta1)
        vi = vi * b2 + (1.f - b2) * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        return pi - step_size * (mi / denom);
    };
    const bool vec = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0 && (((uintptr_t)pb) & 7) == 0;
    const int64_t n4 = vec ? n / 4 : 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 pv = reinterpret_cast<float4*>(p)[i], mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        pv.x = upd(pv.x, gv.x, mv.x, vv.x); pv.y = upd(pv.y, gv.y, mv.y, vv.y); pv.z = upd(pv.z, gv.z, mv.z, vv.z); pv.w = upd(pv.w, gv.w, mv.w, vv.w);
        reinterpret_cast<float4*>(p)[i] = pv; reinterpret_cast<float4*>(m)[i] = mv; reinterpret_cast<float4*>(v)[i] = vv;
        if (pb) {
            uint2 o;
            o.x = pack_bf16x2(pv.x, pv.y);
            o.y = pack_bf16x2(pv.z, pv.w);
            reinterpret_cast<uint2*>(pb)[i] = o;
        }
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float mi = m[i], vi = v[i];
        const float pn = upd(p[i], g[i], mi, vi);
        p[i] = pn; m[i] = mi; v[i] = vi;
        if (pb) pb[i] = f32_to_bf16_bits(pn);
    }
}

}  // namespace

extern "C" {

int occ_compactness_loss(const float* emb, float* loss, float* demb, int64_t n_groups, int64_t group, int64_t E, float scale, void* stream) {
    OCC_CHECK_ARG(emb && loss, "occ_compactness_loss: null pointer");
    OCC_CHECK_ARG(n_groups >= 1 && group >= 2 && E >= 1, "occ_compactness_loss: needs n_groups>=1, group>=2 (got %ld, %ld)", (long)n_groups, (long)group);
    hipLaunchKernelGGL(compactness_kernel, dim3(1), dim3(LOSS_THREADS), 0, (hipStream_t)stream, emb, loss, demb, (int)n_groups, (int)group, (int)E, scale);
    OCC_LAUNCH_CHECK("occ_compactness_loss");
    return OCC_OK;
}

int occ_ce_loss(const float* logits, const int64_t* labels, float* loss, float* dlogits, int64_t B, int64_t C, float scale, void* stream) {
    OCC_CHECK_ARG(logits && labels && loss, "occ_ce_loss: null pointer");
    OCC_CHECK_ARG(B >= 1 && C >= 2 && C <= 1024, "occ_ce_loss: bad shape");
    hipLaunchKernelGGL(ce_kernel, dim3(1), dim3(LOSS_THREADS), 0, (hipStream_t)stream, logits, labels, loss, dlogits, (int)B, (int)C, scale);
    OCC_LAUNCH_CHECK("occ_ce_loss");
    return OCC_OK;
}

int occ_pairwise_dist(const float* ref, const float* emb, float* dist, int64_t N, int64_t E, void* stream) {
    OCC_CHECK_ARG(ref && emb && dist && N >= 1 && E >= 1, "occ_pairwise_dist: bad argument");
    hipLaunchKernelGGL(pairwise_dist_kernel, dim3((unsigned)occ_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, ref, emb, dist, (int)N, (int)E);
    OCC_LAUNCH_CHECK("occ_pairwise_dist");
    return OCC_OK;
}

int occ_pair_dist_loss(const float* emb, const int32_t* pair_i, const int32_t* pair_j, const float* weights, int64_t n_pairs, float bias, int relu,
                       float* loss, float* demb, int64_t R, int64_t E, float scale, void* stream) {
    OCC_CHECK_ARG(emb && pair_i && pair_j && weights && loss, "occ_pair_dist_loss: null pointer");
    OCC_CHECK_ARG(n_pairs >= 1 && n_pairs <= PAIR_MAX && R >= 1 && E >= 1, "occ_pair_dist_loss: needs 1..%d pairs (got %ld)", PAIR_MAX, (long)n_pairs);
    hipLaunchKernelGGL(pair_dist_loss_kernel, dim3(1), dim3(LOSS_THREADS), 0, (hipStream_t)stream, emb, pair_i, pair_j, weights, (int)n_pairs, bias, relu,
                       loss, demb, (int)R, (int)E, scale);
    OCC_LAUNCH_CHECK("occ_pair_dist_loss");
    return OCC_OK;
}

int occ_adam_multi(void* const* params, void* const* grads, void* const* exp_avg, void* const* exp_avg_sq, const int64_t* sizes,
                   int32_t* steps, int64_t n_tensors, int64_t max_size, float lr, float beta1, float beta2, float eps, float grad_scale,
                   void* const* bf16_copies, void* stream) {
    OCC_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && sizes && steps, "occ_adam_multi: null pointer");
    OCC_CHECK_ARG(n_tensors >= 1 && n_tensors < 65536 && max_size >= 1, "occ_adam_multi: bad argument");
    int64_t bx = occ_cdiv(max_size, 256 * 4 * 4);
    if (bx > 2048) bx = 2048;
    if (bx < 1) bx = 1;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_bump_steps_kernel, dim3((unsigned)occ_cdiv(n_tensors, 256)), dim3(256), 0, s, grads, steps, (int)n_tensors);
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)bx, (unsigned)n_tensors), dim3(256), 0, s, params, grads, exp_avg,
                       exp_avg_sq, sizes, (const int32_t*)steps, lr, beta1, beta2, eps, grad_scale, bf16_copies);
    OCC_LAUNCH_CHECK("occ_adam_multi");
    return OCC_OK;
}

}  // extern "C"
