// LCNN back-end pieces (models/lcnn.py:121-241), f32 channels-last: Max-Feature-Map (mfm.forward :133-136: split the 2C output
// channels of a conv / linear in halves, elementwise max), MFM fused with the MaxPool2d(2, 2) that follows every conv group
// (:154-166), and AdaptiveAvgPool2d((1, 64)) + flatten (:169, :191-194).  The convolutions, linears, BatchNorm and dropout around them
// are the library's occ_gemm / occ_gemm_tn / occ_bn_* / occ_dropout.  All of this is HBM-bound byte shuffling (2C <= 64 channels on
// millions of rows): one thread per output element, whole cache lines used by every wave.
#include "occ_common.h"

namespace {

__device__ __forceinline__ long long lc_row_off(const occ_rowmap& m, long long row) {
    const long long b = row / m.rows_per_batch, r = row - b * m.rows_per_batch;
    if (m.rows_per_line > 0) {
        const long long l = r / m.rows_per_line;
        return b * m.batch_stride + l * m.line_stride + (r - l * m.rows_per_line) * m.row_stride;
    }
    return b * m.batch_stride + r * m.row_stride;
}

__global__ __launch_bounds__(256) void mfm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, const occ_rowmap ym, long long rows, int C) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * C) return;
    const long long r = i / C;
    const int c = (int)(i - r * C);
    const float a = x[r * 2 * C + c], b = x[r * 2 * C + C + c];
    y[lc_row_off(ym, r) + c] = a > b || a != a ? a : b;              // torch.maximum propagates NaN
}

// torch.maximum's derivative (derivatives.yaml): the larger side takes dy, a tie splits it evenly
__global__ __launch_bounds__(256) void mfm_bwd_kernel(const float* __restrict__ dy, const occ_rowmap dm, const float* __restrict__ x, float* __restrict__ dx,
                                                       long long rows, int C) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * C) return;
    const long long r = i / C;
    const int c = (int)(i - r * C);
    const float a = x[r * 2 * C + c], b = x[r * 2 * C + C + c], g = dy[lc_row_off(dm, r) + c];
    dx[r * 2 * C + c] = a > b ? g : (a == b ? 0.5f * g : 0.f);
    dx[r * 2 * C + C + c] = b > a ? g : (a == b ? 0.5f * g : 0.f);
}

// y[b, ho, wo, c] = max over the 2x2 window of max(x[.., c], x[.., C + c]); idx = window slot (dh*2 + dw, the FIRST maximum in scan
// order, as max_pool2d's backward uses) | side << 2 (0: first half, 1: second half, 2: tie)
__global__ __launch_bounds__(256) void mfm_pool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, const occ_rowmap ym, unsigned char* __restrict__ idx,
                                                            long long B, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * Ho * Wo * C) return;
    const int c = (int)(i % C);
    const long long p = i / C;
    const int wo = (int)(p % Wo), ho = (int)((p / Wo) % Ho);
    const long long b = p / ((long long)Wo * Ho);
    float best = 0.f; int slot = 0, side = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const float* px = x + (((b * H + 2 * ho + (s >> 1)) * W + 2 * wo + (s & 1)) * 2 * C);
        const float a = px[c], bb = px[C + c];
        const float m = a > bb || a != a ? a : bb;
        if (s == 0 || m > best || m != m) { best = m; slot = s; side = a > bb ? 0 : (a == bb ? 2 : 1); }
    }
    y[lc_row_off(ym, p) + c] = best;
    idx[i] = (unsigned char)(slot | (side << 2));
}

// dx [B,H,W,2C] (rows through xm: the interior of a zero-bordered buffer when a 3x3 / 5x5 input-gradient correlation reads it next) is
// written completely: positions outside every window -- an odd last row / column -- and the losing entries get 0
__global__ __launch_bounds__(256) void mfm_pool_bwd_kernel(const float* __restrict__ dy, const occ_rowmap dm, const unsigned char* __restrict__ idx,
                                                            float* __restrict__ dx, const occ_rowmap xm, long long B, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * H * W * C) return;
    const int c = (int)(i % C);
    const long long p = i / C;
    const int w = (int)(p % W), h = (int)((p / W) % H);
    const long long b = p / ((long long)W * H);
    float ga = 0.f, gb = 0.f;
    const int ho = h >> 1, wo = w >> 1;
    if (ho < Ho && wo < Wo) {
        const long long q = (b * Ho + ho) * Wo + wo;
        const int code = idx[q * C + c];
        if ((code & 3) == ((h & 1) * 2 + (w & 1))) {
            const float g = dy[lc_row_off(dm, q) + c];
            const int side = code >> 2;
            ga = side == 0 ? g : (side == 2 ? 0.5f * g : 0.f);
            gb = side == 1 ? g : (side == 2 ? 0.5f * g : 0.f);
        }
    }
    const long long xo = lc_row_off(xm, p);
    dx[xo + c] = ga;
    dx[xo + C + c] = gb;
}

// adaptive bins of torch: [floor(i*W/Wout), ceil((i+1)*W/Wout))
__global__ __launch_bounds__(256) void adaptive_pool_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, long long B, int H, int W, int C, int Wout) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * C * Wout) return;
    const int c = (int)(i % C);
    const int wo = (int)((i / C) % Wout);
    const long long b = i / ((long long)C * Wout);
    const int w0 = (int)(((long long)wo * W) / Wout), w1 = (int)((((long long)wo + 1) * W + Wout - 1) / Wout);
    float s = 0.f;
    for (int h = 0; h < H; ++h)
        for (int w = w0; w < w1; ++w) s += x[((b * H + h) * W + w) * C + c];
    out[b * C * Wout + (long long)c * Wout + wo] = s / (float)(H * (w1 - w0));
}

__global__ __launch_bounds__(256) void adaptive_pool_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx, long long B, int H, int W, int C, int Wout) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * H * W * C) return;
    const int c = (int)(i % C);
    const int w = (int)((i / C) % W);
    const long long b = i / ((long long)C * W * H);
    // bins that contain column w: wo with floor(wo*W/Wout) <= w < ceil((wo+1)*W/Wout)
    int lo = (int)(((long long)w * Wout) / W), hi = (int)((((long long)w + 1) * Wout + W - 1) / W);
    if (lo > 0) --lo;
    if (hi < Wout) ++hi;
    float g = 0.f;
    for (int wo = lo; wo < hi; ++wo) {
        const int w0 = (int)(((long long)wo * W) / Wout), w1 = (int)((((long long)wo + 1) * W + Wout - 1) / Wout);
        if (w >= w0 && w < w1) g += dout[b * C * Wout + (long long)c * Wout + wo] / (float)(H * (w1 - w0));
    }
    dx[i] = g;
}

inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" int occ_mfm_fwd(const float* x, float* y, const occ_rowmap* y_map, int64_t rows, int64_t C, void* stream) {
    OCC_CHECK_ARG(x && y && y_map && rows >= 1 && C >= 1 && y_map->rows_per_batch >= 1 && rows * C < (1ll << 39), "occ_mfm_fwd: bad argument");
    hipLaunchKernelGGL(mfm_fwd_kernel, dim3(blocks_for(rows * C)), dim3(256), 0, (hipStream_t)stream, x, y, *y_map, (long long)rows, (int)C);
    OCC_LAUNCH_CHECK("occ_mfm_fwd");
    return OCC_OK;
}

extern "C" int occ_mfm_bwd(const float* dy, const occ_rowmap* dy_map, const float* x, float* dx, int64_t rows, int64_t C, void* stream) {
    OCC_CHECK_ARG(dy && dy_map && x && dx && rows >= 1 && C >= 1 && dy_map->rows_per_batch >= 1 && rows * C < (1ll << 39), "occ_mfm_bwd: bad argument");
    hipLaunchKernelGGL(mfm_bwd_kernel, dim3(blocks_for(rows * C)), dim3(256), 0, (hipStream_t)stream, dy, *dy_map, x, dx, (long long)rows, (int)C);
    OCC_LAUNCH_CHECK("occ_mfm_bwd");
    return OCC_OK;
}

extern "C" int occ_mfm_pool2_fwd(const float* x, float* y, const occ_rowmap* y_map, uint8_t* idx, int64_t B, int64_t H, int64_t W, int64_t C, void* stream) {
    OCC_CHECK_ARG(x && y && y_map && idx && B >= 1 && H >= 2 && W >= 2 && C >= 1 && y_map->rows_per_batch >= 1 && B * H * W * C < (1ll << 39),
                  "occ_mfm_pool2_fwd: bad argument (needs H, W >= 2)");
    hipLaunchKernelGGL(mfm_pool_fwd_kernel, dim3(blocks_for(B * (H / 2) * (W / 2) * C)), dim3(256), 0, (hipStream_t)stream, x, y, *y_map, idx, (long long)B, (int)H,
                       (int)W, (int)C);
    OCC_LAUNCH_CHECK("occ_mfm_pool2_fwd");
    return OCC_OK;
}

extern "C" int occ_mfm_pool2_bwd(const float* dy, const occ_rowmap* dy_map, const uint8_t* idx, float* dx, const occ_rowmap* dx_map, int64_t B, int64_t H, int64_t W,
                                 int64_t C, void* stream) {
    OCC_CHECK_ARG(dy && dy_map && idx && dx && dx_map && B >= 1 && H >= 2 && W >= 2 && C >= 1 && dy_map->rows_per_batch >= 1 && dx_map->rows_per_batch >= 1 &&
                      B * H * W * C < (1ll << 39), "occ_mfm_pool2_bwd: bad argument");
    hipLaunchKernelGGL(mfm_pool_bwd_kernel, dim3(blocks_for(B * H * W * C)), dim3(256), 0, (hipStream_t)stream, dy, *dy_map, idx, dx, *dx_map, (long long)B, (int)H,
                       (int)W, (int)C);
    OCC_LAUNCH_CHECK("occ_mfm_pool2_bwd");
    return OCC_OK;
}

extern "C" int occ_adaptive_avgpool_1xw_fwd(const float* x, float* out, int64_t B, int64_t H, int64_t W, int64_t C, int64_t Wout, void* stream) {
    OCC_CHECK_ARG(x && out && B >= 1 && H >= 1 && W >= 1 && C >= 1 && Wout >= 1, "occ_adaptive_avgpool_1xw_fwd: bad argument");
    hipLaunchKernelGGL(adaptive_pool_fwd_kernel, dim3(blocks_for(B * C * Wout)), dim3(256), 0, (hipStream_t)stream, x, out, (long long)B, (int)H, (int)W, (int)C, (int)Wout);
    OCC_LAUNCH_CHECK("occ_adaptive_avgpool_1xw_fwd");
    return OCC_OK;
}

extern "C" int occ_adaptive_avgpool_1xw_bwd(const float* dout, float* dx, int64_t B, int64_t H, int64_t W, int64_t C, int64_t Wout, void* stream) {
    OCC_CHECK_ARG(dout && dx && B >= 1 && H >= 1 && W >= 1 && C >= 1 && Wout >= 1, "occ_adaptive_avgpool_1xw_bwd: bad argument");
    hipLaunchKernelGGL(adaptive_pool_bwd_kernel, dim3(blocks_for(B * H * W * C)), dim3(256), 0, (hipStream_t)stream, dout, dx, (long long)B, (int)H, (int)W, (int)C, (int)Wout);
    OCC_LAUNCH_CHECK("occ_adaptive_avgpool_1xw_bwd");
    return OCC_OK;
}
