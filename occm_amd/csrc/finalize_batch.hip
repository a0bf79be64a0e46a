// One launch for the partial-sum finalizes of a whole backward pass.  Three producers of the fine-tuning step leave per-workgroup partial
// sums that only the optimizer needs: the fused LayerNorm backward (dgamma | dbeta | dbias, [block][3][C]), the GELU' epilogue's column
// sums (fc1's bias gradient, [half tile][N]) and the attention backward's q|k|v bias sums ([batch][head][3][hd]).  Each used to be followed
// by its own 3-8 us finalize launch -- 104 launches, 0.6 ms per step on the critical path (autograd's bias / LayerNorm-parameter gradients,
// oc_training.py:384).  With `defer` set the producers keep their partial sums in caller-owned per-site buffers and occ_finalize_batch adds
// them all up from a device-resident job table, each in a fixed order (run-to-run bit-equal; the LayerNorm and attention sums in the very order
// of their per-site kernels, the column sums with the four waves taking every fourth partial row where colsum_finalize_kernel walks them in turn).
#include "occ_common.h"

namespace {

struct Job {                 // mirror of occ_finalize_job (include/occ_hip.h)
    const float* partials; float* out0; float* out1; float* out2;
    int kind, n0, n1, n2, first_block, n_blocks;
};

// kind 0: out[i] += sum_p partials[p * tot + i], tot = n1 (= nsets * C), n0 = number of partial rows, n2 = C: i < C -> out0, < 2C -> out1, else out2.
//         64 columns per workgroup, the four waves take every fourth partial row (layernorm_bwd_finalize_kernel's order).
// kind 1: attention bias records: n0 = B, n1 = H, n2 = hd; one workgroup per (which, head) (attention_bias_finalize_kernel's order; one key block).
__global__ __launch_bounds__(256) void finalize_batch_kernel(const Job* __restrict__ jobs, int n_jobs) {
    __shared__ float red[256];
    int lo = 0, hi = n_jobs - 1;                     // the job whose block range holds blockIdx.x (first_block ascending)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const Job j = jobs[lo];
    const int blk = (int)blockIdx.x - j.first_block;
    if (blk >= j.n_blocks) return;
    if (j.kind == 0) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int tot = j.n1, C = j.n2, nparts = j.n0;
        const int i = blk * 64 + lane;
        float s = 0.f;
        if (i < tot) {
            int b = wave;
            for (; b + 28 < nparts; b += 32) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = j.partials[(long long)(b + 4 * u) * tot + i];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            for (; b < nparts; b += 4) s += j.partials[(long long)b * tot + i];
        }
        red[threadIdx.x] = s;
        __syncthreads();
        if (wave == 0 && i < tot) {
            const float t = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane];
            if (i < C) j.out0[i] += t; else if (i < 2 * C) j.out1[i - C] += t; else if (j.out2) j.out2[i - 2 * C] += t;
        }
        return;
    }
    const int B = j.n0, H = j.n1, HD = j.n2;
    const int which = blk / H, h = blk - which * H;
    const int ngr = 256 / HD, gr = threadIdx.x / HD, d = threadIdx.x - gr * HD;
    float t = 0.f;
    if (gr < ngr) {
        const float* src = j.partials + (size_t)h * 3 * HD + which * HD + d;          // record bb of head h: + bb * H * 3 * HD
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
        int r = gr;
        for (; r + 3 * ngr < B; r += 4 * ngr) {
            t0 += src[(size_t)r * H * 3 * HD]; t1 += src[(size_t)(r + ngr) * H * 3 * HD];
            t2 += src[(size_t)(r + 2 * ngr) * H * 3 * HD]; t3 += src[(size_t)(r + 3 * ngr) * H * 3 * HD];
        }
        for (; r < B; r += ngr) t0 += src[(size_t)r * H * 3 * HD];
        t = (t0 + t1) + (t2 + t3);
    }
    red[threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x < HD) {
        float sum = 0.f;
        for (int g2 = 0; g2 < ngr; ++g2) sum += red[g2 * HD + threadIdx.x];
        j.out0[(size_t)which * H * HD + (size_t)h * HD + threadIdx.x] += sum;
    }
}

}  // namespace

extern "C" int occ_finalize_batch(const occ_finalize_job* jobs_dev, int64_t n_jobs, int64_t total_blocks, void* stream) {
    static_assert(sizeof(Job) == sizeof(occ_finalize_job), "job table layout");
    OCC_CHECK_ARG(jobs_dev && n_jobs >= 1 && n_jobs <= 4096 && total_blocks >= 1 && total_blocks < (1ll << 24), "occ_finalize_batch: bad job table");
    hipLaunchKernelGGL(finalize_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const Job*>(jobs_dev), (int)n_jobs);
    OCC_LAUNCH_CHECK("occ_finalize_batch");
    return OCC_OK;
}
