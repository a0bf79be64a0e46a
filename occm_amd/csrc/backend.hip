// AASIST back-end kernels (everything that is not a GEMM): stem max-pool, train-mode BatchNorm with
// fused activation (forward + backward), softmax-weighted pooling over the spectral / temporal axis,
// graph-attention pieces, heterogeneous master-node update, graph pooling, read-out, dropout, and a
// strided copy used to repack weights.  Reference: models/sslassist.py:58-597.
// All tensors are f32, channels-last ([rows, C] with C contiguous).
#include "occ_common.h"

namespace {

constexpr int BT = 256;

__device__ __forceinline__ float block_sum(float v, float* red) {       // red: >= 4 floats of LDS
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
    return s;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) s = fmaxf(s, red[i]);
    return s;
}

// ---------------------------------------------------------------------------- small utilities --
__global__ void fill_f32_kernel(float* __restrict__ p, float v, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void axpby_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, float alpha, float beta, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        o[i] = alpha * a[i] + (b ? beta * b[i] : 0.f);
}
// out[i0,i1,i2,i3] (contiguous) (+)= in[off + i0*s0 + i1*s1 + i2*s2 + i3*s3]
__global__ void copy_strided_kernel(const float* __restrict__ in, float* __restrict__ out, long long off, int n0, int n1, int n2, int n3,
                                    long long s0, long long s1, long long s2, long long s3, int accumulate) {
    const long long n = (long long)n0 * n1 * n2 * n3;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const int i3 = (int)(r % n3); r /= n3;
        const int i2 = (int)(r % n2); r /= n2;
        const int i1 = (int)(r % n1); r /= n1;
        const int i0 = (int)r;
        const float v = in[off + i0 * s0 + i1 * s1 + i2 * s2 + i3 * s3];
        out[i] = accumulate ? out[i] + v : v;
    }
}

// y[ymap(r)][c] = x[xmap(r)][c]   (split / concatenate node sets, gather interiors of padded buffers)
__global__ void copy_rows_kernel(const float* __restrict__ x, RowMapI xmap, float* __restrict__ y, RowMapI ymap, long long rows, int C, int accumulate) {
    const long long n = rows * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / C; const int c = (int)(i - r * C);
        const float v = x[row_off(xmap, r) + c];
        float* o = y + row_off(ymap, r) + c;
        *o = accumulate ? *o + v : v;
    }
}
// dx = dy * act'(.) with the derivative taken from the activation OUTPUT y (SELU: y>0 ? s : y + s*a; tanh: 1-y^2; relu: y>0)
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, int act, long long n) {
    const float a = 1.6732632423543772848170429916717f, s = 1.0507009873554804934193349852946f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float yv = y[i];
        float g = 1.f;
        if (act == OCC_ACT_SELU) g = yv > 0.f ? s : yv + s * a;
        else if (act == OCC_ACT_TANH) g = 1.f - yv * yv;
        else if (act == OCC_ACT_RELU) g = yv > 0.f ? 1.f : 0.f;
        dx[i] = dy[i] * g;
    }
}

// ------------------------------------------------------------------------------------ dropout --
__device__ __forceinline__ void philox_r(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
// mask[i] = keep ? 1 : 0 (generated when gen != 0), y = x * mask / (1-p)
// step (optional, device): the Philox stream id becomes (*step << 8) + sid, so a captured HIP graph draws fresh masks on every replay
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ mask, long long n, float p,
                               uint64_t seed, uint64_t sid, int gen, const uint64_t* __restrict__ step) {
    if (step) sid += *step << 8;
    const float sc = 1.0f / (1.0f - p);
    const long long nq = (n + 3) / 4;
    for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < nq; q += (long long)gridDim.x * blockDim.x) {
        uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)sid, (uint32_t)(sid >> 32)};
        if (gen) {
            uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
            for (int r = 0; r < 10; ++r) { philox_r(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const long long i = q * 4 + h;
            if (i < n) {
                unsigned char m;
                if (gen) { m = ((float)c[h] * (1.0f / 4294967296.0f)) >= p ? 1 : 0; mask[i] = m; }
                else m = mask[i];
                y[i] = m ? x[i] * sc : 0.f;
            }
        }
    }
}

// ------------------------------------------------------------------------------ stem max-pool --
// y [B,T,F] (LL output) -> out [B,Hp,Wp] = max over 3x3 windows of y^T [F,T] (floor mode), idx = argmax 0..8
__global__ void stem_pool_fwd_kernel(const float* __restrict__ y, float* __restrict__ out, unsigned char* __restrict__ idx, int B, int T, int F,
                                     int Hp, int Wp, int out_c) {
    const long long n = (long long)B * Hp * Wp;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int w = (int)(i % Wp), h = (int)((i / Wp) % Hp), b = (int)(i / ((long long)Wp * Hp));
        float best = -3.4e38f; int bi = 0;
#pragma unroll
        for (int dh = 0; dh < 3; ++dh)                         // torch scans the window row-major over (feat, time)
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const float v = y[((size_t)b * T + (3 * w + dw)) * F + 3 * h + dh];
                if (v > best) { best = v; bi = dh * 3 + dw; }
            }
        out[i * out_c] = best;
        idx[i] = (unsigned char)bi;
    }
}
__global__ void stem_pool_bwd_kernel(const float* __restrict__ dout, const unsigned char* __restrict__ idx, float* __restrict__ dy, int B, int T,
                                     int F, int Hp, int Wp, int dout_c) {
    const long long n = (long long)B * Hp * Wp;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int w = (int)(i % Wp), h = (int)((i / Wp) % Hp), b = (int)(i / ((long long)Wp * Hp));
        const int bi = idx[i];
        const float g = dout[i * dout_c];
#pragma unroll
        for (int dh = 0; dh < 3; ++dh)
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) dy[((size_t)b * T + (3 * w + dw)) * F + 3 * h + dh] = (dh * 3 + dw == bi) ? g : 0.f;
    }
}

// --------------------------------------------------------------------------------- BatchNorm --
// Column statistics of x [rows, C] (rows through a row map), 256 % C == 0.  ws: f64 [nchunk][C][2].
__global__ __launch_bounds__(BT) void bn_partial_kernel(const float* __restrict__ x, RowMapI map, long long rows, int C, long long rows_per_chunk,
                                                        double* __restrict__ ws) {
    __shared__ double r0[BT], r1[BT];
    const int col = threadIdx.x % C, rl = threadIdx.x / C, rstep = BT / C;
    const long long begin = (long long)blockIdx.x * rows_per_chunk;
    const long long end = begin + rows_per_chunk < rows ? begin + rows_per_chunk : rows;
    double s = 0.0, q = 0.0;
    for (long long r = begin + rl; r < end; r += rstep) {
        const double v = (double)x[row_off(map, r) + col];
        s += v; q += v * v;
    }
    r0[threadIdx.x] = s; r1[threadIdx.x] = q;
    __syncthreads();
    if (threadIdx.x < C) {
        for (int k = 1; k < rstep; ++k) { s += r0[threadIdx.x + k * C]; q += r1[threadIdx.x + k * C]; }
        ws[((size_t)blockIdx.x * C + threadIdx.x) * 2] = s;
        ws[((size_t)blockIdx.x * C + threadIdx.x) * 2 + 1] = q;
    }
}
// mean/rstd from the partials (train) or from the running statistics (eval); running stats updated in train mode.
// One wave per channel: lanes stride over the chunks, then a wave reduction.
__global__ __launch_bounds__(BT) void bn_finalize_kernel(const double* __restrict__ ws, int nchunk, int C, long long rows, float* __restrict__ mean,
                                                         float* __restrict__ rstd, float* __restrict__ run_mean, float* __restrict__ run_var,
                                                         long long* __restrict__ nbt, float momentum, float eps, int train) {
    const int c = blockIdx.x * (BT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    if (train) {
        double s = 0.0, q = 0.0;
        for (int k = lane; k < nchunk; k += 64) { s += ws[((size_t)k * C + c) * 2]; q += ws[((size_t)k * C + c) * 2 + 1]; }
        s = wave_sum(s); q = wave_sum(q);
        if (lane != 0) return;
        const double m = s / (double)rows;
        double var = q / (double)rows - m * m;
        if (var < 0.0) var = 0.0;
        if (mean) { mean[c] = (float)m; rstd[c] = (float)(1.0 / sqrt(var + (double)eps)); }
        if (run_mean) {
            const double unb = rows > 1 ? var * (double)rows / (double)(rows - 1) : var;
            run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)m;
            run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
        }
        if (nbt && c == 0) nbt[0] += 1;
    } else if (lane == 0) {
        mean[c] = run_mean[c];
        rstd[c] = 1.0f / sqrtf(run_var[c] + eps);
    }
}
template <int ACT>
__global__ void bn_act_fwd_kernel(const float* __restrict__ x, RowMapI xmap, const float* __restrict__ mean, const float* __restrict__ rstd,
                                  const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y, RowMapI ymap,
                                  long long rows, int C) {
    const long long n = rows * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / C; const int c = (int)(i - r * C);
        const float z = (x[row_off(xmap, r) + c] - mean[c]) * rstd[c] * gamma[c] + beta[c];
        y[row_off(ymap, r) + c] = occ_apply_act<ACT>(z);
    }
}
template <int ACT> __device__ __forceinline__ float act_grad(float z) {
    if (ACT == OCC_ACT_SELU) return selu_grad_from_in(z);
    if (ACT == OCC_ACT_RELU) return z > 0.f ? 1.f : 0.f;
    return 1.f;
}
// partial sums of dz and dz*xhat per column; dz = dy * act'(z)
template <int ACT>
__global__ __launch_bounds__(BT) void bn_bwd_partial_kernel(const float* __restrict__ dy, RowMapI dymap, const float* __restrict__ x, RowMapI xmap,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, long long rows, int C,
                                                            long long rows_per_chunk, double* __restrict__ ws) {
    __shared__ double r0[BT], r1[BT];
    const int col = threadIdx.x % C, rl = threadIdx.x / C, rstep = BT / C;
    const long long begin = (long long)blockIdx.x * rows_per_chunk;
    const long long end = begin + rows_per_chunk < rows ? begin + rows_per_chunk : rows;
    const float mu = mean[col], rs = rstd[col], g = gamma[col], be = beta[col];
    double s = 0.0, q = 0.0;
    for (long long r = begin + rl; r < end; r += rstep) {
        const float xh = (x[row_off(xmap, r) + col] - mu) * rs;
        const float dz = dy[row_off(dymap, r) + col] * act_grad<ACT>(xh * g + be);
        s += (double)dz; q += (double)dz * (double)xh;
    }
    r0[threadIdx.x] = s; r1[threadIdx.x] = q;
    __syncthreads();
    if (threadIdx.x < C) {
        for (int k = 1; k < rstep; ++k) { s += r0[threadIdx.x + k * C]; q += r1[threadIdx.x + k * C]; }
        ws[((size_t)blockIdx.x * C + threadIdx.x) * 2] = s;
        ws[((size_t)blockIdx.x * C + threadIdx.x) * 2 + 1] = q;
    }
}
// sums[c] = {sum dz, sum dz*xhat}; dgamma/dbeta accumulated.  One wave per channel.
__global__ __launch_bounds__(BT) void bn_bwd_finalize_kernel(const double* __restrict__ ws, int nchunk, int C, float* __restrict__ sums,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * (BT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int k = lane; k < nchunk; k += 64) { s += ws[((size_t)k * C + c) * 2]; q += ws[((size_t)k * C + c) * 2 + 1]; }
    s = wave_sum(s); q = wave_sum(q);
    if (lane != 0) return;
    sums[2 * c] = (float)s; sums[2 * c + 1] = (float)q;
    if (dgamma) dgamma[c] += (float)q;
    if (dbeta) dbeta[c] += (float)s;
}
template <int ACT>
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, RowMapI dymap, const float* __restrict__ x, RowMapI xmap, const float* __restrict__ mean,
                                    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ sums, float* __restrict__ dx, RowMapI dxmap, long long rows, int C) {
    const long long n = rows * C;
    const float invn = 1.0f / (float)rows;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / C; const int c = (int)(i - r * C);
        const float xh = (x[row_off(xmap, r) + c] - mean[c]) * rstd[c];
        const float dz = dy[row_off(dymap, r) + c] * act_grad<ACT>(xh * gamma[c] + beta[c]);
        dx[row_off(dxmap, r) + c] = gamma[c] * rstd[c] * (dz - sums[2 * c] * invn - xh * sums[2 * c + 1] * invn);
    }
}

// ---- float4 forms of the four BatchNorm kernels for C % 4 == 0 (every layer but the 1-channel stem): 16-byte accesses, a
// thread owns one 4-column group, 32-bit index math; same partial-sum workspace layout as the scalar kernels.
__global__ __launch_bounds__(BT) void bn_partial_vec_kernel(const float* __restrict__ x, RowMapI map, long long rows, int C, long long rows_per_chunk,
                                                            double* __restrict__ ws) {
    __shared__ double red[BT][8];
    const int cq = C >> 2, g = threadIdx.x % cq, rl = threadIdx.x / cq, rstep = BT / cq;
    const long long begin = (long long)blockIdx.x * rows_per_chunk;
    const long long end = begin + rows_per_chunk < rows ? begin + rows_per_chunk : rows;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    for (long long r = begin + rl; r < end; r += rstep) {
        const float4 v = *reinterpret_cast<const float4*>(x + row_off(map, r) + 4 * g);
        s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
        q[0] += (double)v.x * v.x; q[1] += (double)v.y * v.y; q[2] += (double)v.z * v.z; q[3] += (double)v.w * v.w;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[threadIdx.x][e] = s[e]; red[threadIdx.x][4 + e] = q[e]; }
    __syncthreads();
    if (threadIdx.x < C) {                       // thread = one column: gather its group's partials over the row lanes
        const int gg = threadIdx.x >> 2, e = threadIdx.x & 3;
        double ss = 0.0, qq = 0.0;
        for (int k = 0; k < rstep; ++k) { ss += red[gg + k * cq][e]; qq += red[gg + k * cq][4 + e]; }
        ws[((size_t)blockIdx.x * C + threadIdx.x) * 2] = ss;
        ws[((size_t)blockIdx.x * C + threadIdx.x) * 2 + 1] = qq;
    }
}
template <int ACT>
__global__ void bn_act_fwd_vec_kernel(const float* __restrict__ x, RowMapI xmap, const float* __restrict__ mean, const float* __restrict__ rstd,
                                      const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y, RowMapI ymap,
                                      long long rows, int C) {
    const unsigned cq = (unsigned)C >> 2;
    const long long n = rows * cq;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = n < (1ll << 31) ? (long long)((unsigned)i / cq) : i / cq;
        const int c = (int)(i - r * cq) * 4;
        const float4 v = *reinterpret_cast<const float4*>(x + row_off(xmap, r) + c);
        const float4 mu = *reinterpret_cast<const float4*>(mean + c), rs = *reinterpret_cast<const float4*>(rstd + c);
        const float4 ga = *reinterpret_cast<const float4*>(gamma + c), be = *reinterpret_cast<const float4*>(beta + c);
        float4 o;
        o.x = occ_apply_act<ACT>((v.x - mu.x) * rs.x * ga.x + be.x); o.y = occ_apply_act<ACT>((v.y - mu.y) * rs.y * ga.y + be.y);
        o.z = occ_apply_act<ACT>((v.z - mu.z) * rs.z * ga.z + be.z); o.w = occ_apply_act<ACT>((v.w - mu.w) * rs.w * ga.w + be.w);
        *reinterpret_cast<float4*>(y + row_off(ymap, r) + c) = o;
    }
}
template <int ACT>
__global__ __launch_bounds__(BT) void bn_bwd_partial_vec_kernel(const float* __restrict__ dy, RowMapI dymap, const float* __restrict__ x, RowMapI xmap,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta, long long rows, int C,
                                                                long long rows_per_chunk, double* __restrict__ ws) {
    __shared__ double red[BT][8];
    const int cq = C >> 2, g = threadIdx.x % cq, rl = threadIdx.x / cq, rstep = BT / cq;
    const long long begin = (long long)blockIdx.x * rows_per_chunk;
    const long long end = begin + rows_per_chunk < rows ? begin + rows_per_chunk : rows;
    const float4 mu = *reinterpret_cast<const float4*>(mean + 4 * g), rs = *reinterpret_cast<const float4*>(rstd + 4 * g);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + 4 * g), be = *reinterpret_cast<const float4*>(beta + 4 * g);
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    for (long long r = begin + rl; r < end; r += rstep) {
        const float4 xv = *reinterpret_cast<const float4*>(x + row_off(xmap, r) + 4 * g);
        const float4 dv = *reinterpret_cast<const float4*>(dy + row_off(dymap, r) + 4 * g);
        const float xh[4] = {(xv.x - mu.x) * rs.x, (xv.y - mu.y) * rs.y, (xv.z - mu.z) * rs.z, (xv.w - mu.w) * rs.w};
        const float dz[4] = {dv.x * act_grad<ACT>(xh[0] * ga.x + be.x), dv.y * act_grad<ACT>(xh[1] * ga.y + be.y),
                             dv.z * act_grad<ACT>(xh[2] * ga.z + be.z), dv.w * act_grad<ACT>(xh[3] * ga.w + be.w)};
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[e] += (double)dz[e]; q[e] += (double)dz[e] * (double)xh[e]; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[threadIdx.x][e] = s[e]; red[threadIdx.x][4 + e] = q[e]; }
    __syncthreads();
    if (threadIdx.x < C) {
        const int gg = threadIdx.x >> 2, e = threadIdx.x & 3;
        double ss = 0.0, qq = 0.0;
        for (int k = 0; k < rstep; ++k) { ss += red[gg + k * cq][e]; qq += red[gg + k * cq][4 + e]; }
        ws[((size_t)blockIdx.x * C + threadIdx.x) * 2] = ss;
        ws[((size_t)blockIdx.x * C + threadIdx.x) * 2 + 1] = qq;
    }
}
template <int ACT>
__global__ void bn_bwd_apply_vec_kernel(const float* __restrict__ dy, RowMapI dymap, const float* __restrict__ x, RowMapI xmap,
                                        const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                        const float* __restrict__ beta, const float* __restrict__ sums, float* __restrict__ dx, RowMapI dxmap,
                                        long long rows, int C) {
    const unsigned cq = (unsigned)C >> 2;
    const long long n = rows * cq;
    const float invn = 1.0f / (float)rows;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = n < (1ll << 31) ? (long long)((unsigned)i / cq) : i / cq;
        const int c = (int)(i - r * cq) * 4;
        const float4 xv = *reinterpret_cast<const float4*>(x + row_off(xmap, r) + c);
        const float4 dv = *reinterpret_cast<const float4*>(dy + row_off(dymap, r) + c);
        const float xa[4] = {xv.x, xv.y, xv.z, xv.w}, da[4] = {dv.x, dv.y, dv.z, dv.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float rs = rstd[c + e], ga = gamma[c + e];
            const float xh = (xa[e] - mean[c + e]) * rs;
            const float dz = da[e] * act_grad<ACT>(xh * ga + beta[c + e]);
            o[e] = ga * rs * (dz - sums[2 * (c + e)] * invn - xh * sums[2 * (c + e) + 1] * invn);
        }
        *reinterpret_cast<float4*>(dx + row_off(dxmap, r) + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// ------------------------------------------------------- softmax-weighted sum along one axis --
// element (o, r, c) of x / w lives at (o / inner_n) * outer_stride + (o % inner_n) * inner_stride + r * r_stride + c.
// out[o, c] = sum_r x * softmax_r(w) (+ pos[(o % pos_period), c]).          sslassist.py:526-538
struct AxisMap { long long inner_n, outer_stride, inner_stride, r_stride; };
__device__ __forceinline__ long long axis_base(const AxisMap& m, long long o) { return (o / m.inner_n) * m.outer_stride + (o % m.inner_n) * m.inner_stride; }

__global__ __launch_bounds__(BT) void softmax_wsum_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, AxisMap am, int R, int C,
                                                              const float* __restrict__ pos, int pos_period, float* __restrict__ out) {
    __shared__ float sm[3][BT];
    const long long o = blockIdx.x;
    const long long base = axis_base(am, o);
    const int c = threadIdx.x % C, part = threadIdx.x / C, nparts = BT / C;
    float mx = -3.4e38f, den = 0.f, num = 0.f;
    for (int r = part; r < R; r += nparts) {
        const float wv = w[base + (long long)r * am.r_stride + c], xv = x[base + (long long)r * am.r_stride + c];
        if (wv > mx) { const float f = expf(mx - wv); den *= f; num *= f; mx = wv; }
        const float e = expf(wv - mx);
        den += e; num += e * xv;
    }
    sm[0][threadIdx.x] = mx; sm[1][threadIdx.x] = den; sm[2][threadIdx.x] = num;
    __syncthreads();
    if (threadIdx.x < C) {
        float M = mx;
        for (int k = 1; k < nparts; ++k) M = fmaxf(M, sm[0][threadIdx.x + k * C]);
        float D = 0.f, N = 0.f;
        for (int k = 0; k < nparts; ++k) {
            const float f = expf(sm[0][threadIdx.x + k * C] - M);
            D += sm[1][threadIdx.x + k * C] * f; N += sm[2][threadIdx.x + k * C] * f;
        }
        float v = N / D;
        if (pos) v += pos[(size_t)(o % pos_period) * C + c];
        out[o * C + c] = v;
    }
}
// dx (+)= dm * p ; dw (+)= dm * p * (x - m)
__global__ __launch_bounds__(BT) void softmax_wsum_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w, AxisMap am, int R, int C,
                                                              const float* __restrict__ dm, float* __restrict__ dx, float* __restrict__ dw, int accumulate) {
    __shared__ float sm[3][BT];
    __shared__ float fin[3][BT];
    const long long o = blockIdx.x;
    const long long base = axis_base(am, o);
    const int c = threadIdx.x % C, part = threadIdx.x / C, nparts = BT / C;
    float mx = -3.4e38f, den = 0.f, num = 0.f;
    for (int r = part; r < R; r += nparts) {
        const float wv = w[base + (long long)r * am.r_stride + c], xv = x[base + (long long)r * am.r_stride + c];
        if (wv > mx) { const float f = expf(mx - wv); den *= f; num *= f; mx = wv; }
        const float e = expf(wv - mx);
        den += e; num += e * xv;
    }
    sm[0][threadIdx.x] = mx; sm[1][threadIdx.x] = den; sm[2][threadIdx.x] = num;
    __syncthreads();
    if (threadIdx.x < C) {
        float M = mx;
        for (int k = 1; k < nparts; ++k) M = fmaxf(M, sm[0][threadIdx.x + k * C]);
        float D = 0.f, N = 0.f;
        for (int k = 0; k < nparts; ++k) {
            const float f = expf(sm[0][threadIdx.x + k * C] - M);
            D += sm[1][threadIdx.x + k * C] * f; N += sm[2][threadIdx.x + k * C] * f;
        }
        fin[0][c] = M; fin[1][c] = 1.0f / D; fin[2][c] = N / D;
    }
    __syncthreads();
    const float M = fin[0][c], invD = fin[1][c], m = fin[2][c], g = dm[o * C + c];
    for (int r = part; r < R; r += nparts) {
        const long long a = base + (long long)r * am.r_stride + c;
        const float xv = x[a];
        const float p = expf(w[a] - M) * invD;
        const float gx = g * p, gw = g * p * (xv - m);
        if (accumulate) { dx[a] += gx; dw[a] += gw; } else { dx[a] = gx; dw[a] = gw; }
    }
}

// ----------------------------------------------------------------- graph attention pieces --
// P[b,i,j,:] = x[b,i,:] * x[b,j,:]                                   sslassist.py:102-114
__global__ void pair_mul_kernel(const float* __restrict__ x, float* __restrict__ P, int B, int N, int D) {
    const int d4 = D / 4;
    const long long n = (long long)B * N * N * d4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const int c = (int)(r % d4); r /= d4;
        const int j = (int)(r % N); r /= N;
        const int ii = (int)(r % N); const int b = (int)(r / N);
        const float4 a = reinterpret_cast<const float4*>(x + ((size_t)b * N + ii) * D)[c];
        const float4 v = reinterpret_cast<const float4*>(x + ((size_t)b * N + j) * D)[c];
        reinterpret_cast<float4*>(P)[i] = make_float4(a.x * v.x, a.y * v.y, a.z * v.z, a.w * v.w);
    }
}
// dx[b,i,:] (+)= sum_j (dP[b,i,j,:] + dP[b,j,i,:]) * x[b,j,:]
__global__ __launch_bounds__(BT) void pair_mul_bwd_kernel(const float* __restrict__ dP, const float* __restrict__ x, float* __restrict__ dx, int N,
                                                          int D, int accumulate) {
    __shared__ float red[BT];
    const int b = blockIdx.x / N, i = blockIdx.x % N;
    const int d = threadIdx.x % D, part = threadIdx.x / D, nparts = BT / D;
    float s = 0.f;
    for (int j = part; j < N; j += nparts) {
        const float g = dP[(((size_t)b * N + i) * N + j) * D + d] + dP[(((size_t)b * N + j) * N + i) * D + d];
        s += g * x[((size_t)b * N + j) * D + d];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < D) {
        for (int k = 1; k < nparts; ++k) s += red[threadIdx.x + k * D];
        float* o = dx + ((size_t)b * N + i) * D + d;
        *o = accumulate ? *o + s : s;
    }
}
// alpha[b,i,:] = softmax_j( A[b,i,j,:] . aw_type(i,j) * inv_temp )     sslassist.py:125-130, 282-300
// aw: [3][Do] = (w11, w22, w12); homogeneous layers pass n1 = N (only w11 is used).
__global__ __launch_bounds__(BT) void gat_softmax_kernel(const float* __restrict__ A, const float* __restrict__ aw, int N, int Do, int n1, float inv_temp,
                                                         float* __restrict__ alpha) {
    extern __shared__ float sc[];               // [N] + 8
    float* red = sc + N;
    const int b = blockIdx.x / N, i = blockIdx.x % N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* Ai = A + ((size_t)b * N + i) * N * Do;
    for (int j = wave; j < N; j += BT / 64) {
        const int type = (i < n1) ? ((j < n1) ? 0 : 2) : ((j < n1) ? 2 : 1);
        float s = 0.f;
        for (int o = lane; o < Do; o += 64) s += Ai[(size_t)j * Do + o] * aw[type * Do + o];
        s = wave_sum(s);
        if (lane == 0) sc[j] = s * inv_temp;
    }
    __syncthreads();
    float mx = -3.4e38f;
    for (int j = threadIdx.x; j < N; j += BT) mx = fmaxf(mx, sc[j]);
    mx = block_max(mx, red);
    float sum = 0.f;
    for (int j = threadIdx.x; j < N; j += BT) { const float e = expf(sc[j] - mx); sc[j] = e; sum += e; }
    sum = block_sum(sum, red + 4);
    const float inv = 1.0f / sum;
    for (int j = threadIdx.x; j < N; j += BT) alpha[((size_t)b * N + i) * N + j] = sc[j] * inv;
}
// trans == 0: out[b,i,:] = sum_j alpha[b,i,j] x[b,j,:]      trans == 1: out[b,j,:] (+)= sum_i alpha[b,i,j] x[b,i,:]
__global__ __launch_bounds__(BT) void bmm_alpha_kernel(const float* __restrict__ alpha, const float* __restrict__ x, float* __restrict__ out, int N, int D,
                                                       int trans, int accumulate) {
    __shared__ float red[BT];
    const int b = blockIdx.x / N, i = blockIdx.x % N;
    const int d = threadIdx.x % D, part = threadIdx.x / D, nparts = BT / D;
    float s = 0.f;
    for (int j = part; j < N; j += nparts) {
        const float a = trans ? alpha[((size_t)b * N + j) * N + i] : alpha[((size_t)b * N + i) * N + j];
        s += a * x[((size_t)b * N + j) * D + d];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < D) {
        for (int k = 1; k < nparts; ++k) s += red[threadIdx.x + k * D];
        float* o = out + ((size_t)b * N + i) * D + d;
        *o = accumulate ? *o + s : s;
    }
}
// ds[b,i,j] = alpha * (dalpha - sum_j alpha*dalpha) * inv_temp,  dalpha[i,j] = dh[b,i,:] . x[b,j,:]
__global__ __launch_bounds__(BT) void gat_dscore_kernel(const float* __restrict__ alpha, const float* __restrict__ dh, const float* __restrict__ x,
                                                        float* __restrict__ ds, int N, int D, float inv_temp) {
    extern __shared__ float sm[];               // dh_i [D] + da [N] + 8
    float* dhi = sm; float* da = sm + D; float* red = da + N;
    const int b = blockIdx.x / N, i = blockIdx.x % N;
    for (int d = threadIdx.x; d < D; d += BT) dhi[d] = dh[((size_t)b * N + i) * D + d];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = wave; j < N; j += BT / 64) {
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s += dhi[d] * x[((size_t)b * N + j) * D + d];
        s = wave_sum(s);
        if (lane == 0) da[j] = s;
    }
    __syncthreads();
    float dot = 0.f;
    for (int j = threadIdx.x; j < N; j += BT) dot += alpha[((size_t)b * N + i) * N + j] * da[j];
    dot = block_sum(dot, red);
    for (int j = threadIdx.x; j < N; j += BT) {
        const size_t a = ((size_t)b * N + i) * N + j;
        ds[a] = alpha[a] * (da[j] - dot) * inv_temp;
    }
}
// in place: A <- dZ = ds * aw_type * (1 - A^2);  daw[type] += sum ds * A
__global__ __launch_bounds__(BT) void gat_dz_kernel(float* __restrict__ A, const float* __restrict__ ds, const float* __restrict__ aw, int N, int Do, int n1,
                                                    float* __restrict__ daw) {
    __shared__ float acc[3][64];
    const int b = blockIdx.x / N, i = blockIdx.x % N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 192) acc[threadIdx.x / 64][threadIdx.x % 64] = 0.f;
    __syncthreads();
    float* Ai = A + ((size_t)b * N + i) * N * Do;
    float loc[3] = {0.f, 0.f, 0.f};
    for (int j = wave; j < N; j += BT / 64) {
        const int type = (i < n1) ? ((j < n1) ? 0 : 2) : ((j < n1) ? 2 : 1);
        const float g = ds[((size_t)b * N + i) * N + j];
        for (int o = lane; o < Do; o += 64) {             // Do <= 64: one element per lane
            const float a = Ai[(size_t)j * Do + o];
            Ai[(size_t)j * Do + o] = g * aw[type * Do + o] * (1.f - a * a);
            loc[type] += g * a;
        }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) if (lane < Do) atomicAdd(&acc[t][lane], loc[t]);
    __syncthreads();
    if (threadIdx.x < 192) {
        const int t = threadIdx.x / 64, o = threadIdx.x % 64;
        if (o < Do && acc[t][o] != 0.f) atomicAdd(daw + t * Do + o, acc[t][o]);
    }
}

// ---------------------------------------------------------- heterogeneous master-node update --
// sslassist.py:234-239, 255-270, 310-316.  One workgroup per utterance, everything in LDS.
struct MasterP {
    const float *x, *master; long long master_bstride;             // x [B,N,D]; master [B or 1, D]
    const float *Wm, *bm, *awm, *Wwa, *bwa, *Wwo, *bwo;             // att_projM [Do,D], att_weightM [Do], proj_with_attM, proj_without_attM
    float *out, *am, *agg;                                          // out [B,Do]; saved: am [B,N], agg [B,D]
    int N, D, Do; float inv_temp;
};
__global__ __launch_bounds__(BT) void master_fwd_kernel(const MasterP p) {
    extern __shared__ float sm[];
    float* xs = sm;                         // [N*D]
    float* ms = xs + p.N * p.D;             // [D]
    float* ts = ms + p.D;                   // [N*Do]
    float* ss = ts + p.N * p.Do;            // [N]
    float* ag = ss + p.N;                   // [D]
    float* red = ag + p.D;                  // [8]
    const int b = blockIdx.x, N = p.N, D = p.D, Do = p.Do;
    for (int i = threadIdx.x; i < N * D; i += BT) xs[i] = p.x[(size_t)b * N * D + i];
    for (int i = threadIdx.x; i < D; i += BT) ms[i] = p.master[(size_t)b * p.master_bstride + i];
    __syncthreads();
    for (int i = threadIdx.x; i < N * Do; i += BT) {
        const int n = i / Do, o = i % Do;
        float s = p.bm[o];
        for (int d = 0; d < D; ++d) s = fmaf(p.Wm[o * D + d], xs[n * D + d] * ms[d], s);
        ts[i] = tanhf(s);
    }
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += BT) {
        float s = 0.f;
        for (int o = 0; o < Do; ++o) s = fmaf(ts[n * Do + o], p.awm[o], s);
        ss[n] = s * p.inv_temp;
    }
    __syncthreads();
    float mx = -3.4e38f;
    for (int n = threadIdx.x; n < N; n += BT) mx = fmaxf(mx, ss[n]);
    mx = block_max(mx, red);
    float sum = 0.f;
    for (int n = threadIdx.x; n < N; n += BT) { const float e = expf(ss[n] - mx); ss[n] = e; sum += e; }
    sum = block_sum(sum, red + 4);
    for (int n = threadIdx.x; n < N; n += BT) { ss[n] /= sum; p.am[(size_t)b * N + n] = ss[n]; }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += BT) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s = fmaf(ss[n], xs[n * D + d], s);
        ag[d] = s; p.agg[(size_t)b * D + d] = s;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < Do; o += BT) {
        float s = p.bwa[o] + p.bwo[o];
        for (int d = 0; d < D; ++d) s += p.Wwa[o * D + d] * ag[d] + p.Wwo[o * D + d] * ms[d];
        p.out[(size_t)b * Do + o] = s;
    }
}
struct MasterBP {
    MasterP f;
    const float* dout;                                               // [B,Do]
    float *dx; int dx_accumulate;                                    // [B,N,D]
    float *dmaster; long long dmaster_bstride;                       // [B or 1, D] (atomic accumulate)
    float *dWm, *dbm, *dawm, *dWwa, *dbwa, *dWwo, *dbwo;             // atomic accumulate
};
__global__ __launch_bounds__(BT) void master_bwd_kernel(const MasterBP q) {
    extern __shared__ float sm[];
    const MasterP& p = q.f;
    const int b = blockIdx.x, N = p.N, D = p.D, Do = p.Do;
    float* xs = sm;                         // [N*D]
    float* ms = xs + N * D;                 // [D]
    float* ts = ms + D;                     // [N*Do]  t, later du
    float* am = ts + N * Do;                // [N]
    float* dsn = am + N;                    // [N]
    float* dag = dsn + N;                   // [D]
    float* dms = dag + D;                   // [D]
    float* dos = dms + D;                   // [Do]
    float* red = dos + Do;                  // [8]
    for (int i = threadIdx.x; i < N * D; i += BT) xs[i] = p.x[(size_t)b * N * D + i];
    for (int i = threadIdx.x; i < D; i += BT) ms[i] = p.master[(size_t)b * p.master_bstride + i];
    for (int i = threadIdx.x; i < N; i += BT) am[i] = p.am[(size_t)b * N + i];
    for (int i = threadIdx.x; i < Do; i += BT) dos[i] = q.dout[(size_t)b * Do + i];
    __syncthreads();
    for (int i = threadIdx.x; i < N * Do; i += BT) {                 // recompute t
        const int n = i / Do, o = i % Do;
        float s = p.bm[o];
        for (int d = 0; d < D; ++d) s = fmaf(p.Wm[o * D + d], xs[n * D + d] * ms[d], s);
        ts[i] = tanhf(s);
    }
    // projection grads + dagg, dm(from proj_without)
    for (int i = threadIdx.x; i < Do * D; i += BT) {
        const int o = i / D, d = i % D;
        atomicAdd(q.dWwa + i, dos[o] * p.agg[(size_t)b * D + d]);
        atomicAdd(q.dWwo + i, dos[o] * ms[d]);
    }
    for (int o = threadIdx.x; o < Do; o += BT) { atomicAdd(q.dbwa + o, dos[o]); atomicAdd(q.dbwo + o, dos[o]); }
    for (int d = threadIdx.x; d < D; d += BT) {
        float a = 0.f, m = 0.f;
        for (int o = 0; o < Do; ++o) { a = fmaf(dos[o], p.Wwa[o * D + d], a); m = fmaf(dos[o], p.Wwo[o * D + d], m); }
        dag[d] = a; dms[d] = m;
    }
    __syncthreads();
    // dam[n] = dagg . x[n];  ds = am * (dam - sum am*dam) * inv_temp
    float part = 0.f;
    for (int n = threadIdx.x; n < N; n += BT) {
        float s = 0.f;
        for (int d = 0; d < D; ++d) s = fmaf(dag[d], xs[n * D + d], s);
        dsn[n] = s; part += am[n] * s;
    }
    const float dot = block_sum(part, red);
    for (int n = threadIdx.x; n < N; n += BT) dsn[n] = am[n] * (dsn[n] - dot) * p.inv_temp;
    __syncthreads();
    // dawm[o] += sum_n ds[n] t[n,o];  du = ds*awm*(1-t^2) (in place);  dbm[o] += sum_n du
    for (int o = threadIdx.x; o < Do; o += BT) {
        float a = 0.f, s = 0.f;
        for (int n = 0; n < N; ++n) {
            const float t = ts[n * Do + o];
            a = fmaf(dsn[n], t, a);
            const float du = dsn[n] * p.awm[o] * (1.f - t * t);
            ts[n * Do + o] = du; s += du;
        }
        atomicAdd(q.dawm + o, a); atomicAdd(q.dbm + o, s);
    }
    __syncthreads();
    // dWm[o,d] += sum_n du[n,o] x[n,d] m[d]
    for (int i = threadIdx.x; i < Do * D; i += BT) {
        const int o = i / D, d = i % D;
        float s = 0.f;
        for (int n = 0; n < N; ++n) s = fmaf(ts[n * Do + o], xs[n * D + d], s);
        atomicAdd(q.dWm + i, s * ms[d]);
    }
    // dxm[n,d] = sum_o du[n,o] Wm[o,d];  dx[n,d] (+)= am[n]*dagg[d] + dxm*m[d];  dm[d] += sum_n dxm*x[n,d]
    for (int i = threadIdx.x; i < N * D; i += BT) {
        const int n = i / D, d = i % D;
        float s = 0.f;
        for (int o = 0; o < Do; ++o) s = fmaf(ts[n * Do + o], p.Wm[o * D + d], s);
        const float g = am[n] * dag[d] + s * ms[d];
        float* o_ = q.dx + (size_t)b * N * D + i;
        *o_ = q.dx_accumulate ? *o_ + g : g;
        atomicAdd(&dms[d], s * xs[i]);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += BT) atomicAdd(q.dmaster + (size_t)b * q.dmaster_bstride + d, dms[d]);
}

// ------------------------------------------------------------------------------ graph pooling --
// sslassist.py:341-368.  scores = sigmoid(w . drop(h) + bias); keep the k best; out = h*score (rank order).
__global__ __launch_bounds__(BT) void graph_pool_fwd_kernel(const float* __restrict__ h, const unsigned char* __restrict__ mask, float drop_scale,
                                                            const float* __restrict__ w, const float* __restrict__ bias, int N, int D, int k,
                                                            float* __restrict__ out, int* __restrict__ idx, float* __restrict__ scores) {
    extern __shared__ float sm[];            // sc [N] + node_of_rank [k] (int)
    int* node_of = reinterpret_cast<int*>(sm + N);
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int n = wave; n < N; n += BT / 64) {
        float s = 0.f;
        for (int d = lane; d < D; d += 64) {
            const size_t a = ((size_t)b * N + n) * D + d;
            const float hv = mask ? (mask[a] ? h[a] * drop_scale : 0.f) : h[a];
            s += hv * w[d];
        }
        s = wave_sum(s);
        if (lane == 0) { const float z = s + bias[0]; sm[n] = 1.0f / (1.0f + expf(-z)); }
    }
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += BT) {
        const float s = sm[n];
        scores[(size_t)b * N + n] = s;
        int rank = 0;                        // descending score, ties by node index (torch.topk order is unspecified for ties)
        for (int j = 0; j < N; ++j) rank += (sm[j] > s || (sm[j] == s && j < n)) ? 1 : 0;
        if (rank < k) { node_of[rank] = n; idx[(size_t)b * k + rank] = n; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < k * D; i += BT) {
        const int r = i / D, d = i % D, node = node_of[r];
        out[((size_t)b * k + r) * D + d] = h[((size_t)b * N + node) * D + d] * sm[node];
    }
}
// dh [B,N,D] written in full; dw, dbias accumulated atomically
__global__ __launch_bounds__(BT) void graph_pool_bwd_kernel(const float* __restrict__ h, const unsigned char* __restrict__ mask, float drop_scale,
                                                            const float* __restrict__ w, const float* __restrict__ scores, const int* __restrict__ idx,
                                                            const float* __restrict__ dout, int N, int D, int k, float* __restrict__ dh,
                                                            float* __restrict__ dw, float* __restrict__ dbias) {
    extern __shared__ float sm[];            // dlogit [N] + rank_of [N] (as int)
    float* dlog = sm; int* rank_of = reinterpret_cast<int*>(sm + N);
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int n = threadIdx.x; n < N; n += BT) rank_of[n] = -1;
    __syncthreads();
    for (int r = threadIdx.x; r < k; r += BT) rank_of[idx[(size_t)b * k + r]] = r;
    __syncthreads();
    for (int n = wave; n < N; n += BT / 64) {
        const int r = rank_of[n];
        float s = 0.f;
        if (r >= 0)
            for (int d = lane; d < D; d += 64) s += dout[((size_t)b * k + r) * D + d] * h[((size_t)b * N + n) * D + d];
        s = wave_sum(s);
        if (lane == 0) { const float sc = scores[(size_t)b * N + n]; dlog[n] = s * sc * (1.f - sc); }
    }
    __syncthreads();
    float db = 0.f;
    for (int n = threadIdx.x; n < N; n += BT) db += dlog[n];
    db = wave_sum(db);
    if (lane == 0 && db != 0.f) atomicAdd(dbias, db);
    for (int i = threadIdx.x; i < N * D; i += BT) {
        const int n = i / D, d = i % D;
        const size_t a = ((size_t)b * N + n) * D + d;
        const int r = rank_of[n];
        const float keep = mask ? (mask[a] ? drop_scale : 0.f) : 1.f;
        float g = dlog[n] * w[d] * keep;
        if (r >= 0) g += dout[((size_t)b * k + r) * D + d] * scores[(size_t)b * N + n];
        dh[a] = g;
    }
    for (int d = threadIdx.x; d < D; d += BT) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) {
            const size_t a = ((size_t)b * N + n) * D + d;
            const float hv = mask ? (mask[a] ? h[a] * drop_scale : 0.f) : h[a];
            s += dlog[n] * hv;
        }
        atomicAdd(dw + d, s);
    }
}

// ----------------------------------------------------------------------------------- read-out --
// sslassist.py:573-597: drop_way x6 -> branch max -> (|.|max, mean) over nodes -> concat(5*Dg) -> dropout -> out_layer.
struct ReadoutP {
    const float *T1, *T2, *S1, *S2, *M1, *M2;                        // [B,Nt,Dg] x2, [B,Ns,Dg] x2, [B,Dg] x2
    const unsigned char *mT1, *mT2, *mS1, *mS2, *mM1, *mM2, *mLast;  // keep masks or NULL
    float way_scale, last_scale;
    const float *W, *bias;                                           // out_layer [ncls, 5*Dg], [ncls]
    float *emb, *logits;                                             // [B,5*Dg] (dropped when mLast), [B,ncls]
    int Nt, Ns, Dg, ncls;
};
__device__ __forceinline__ float rd_val(const float* t, const unsigned char* m, size_t a, float sc) { return m ? (m[a] ? t[a] * sc : 0.f) : t[a]; }
__global__ __launch_bounds__(BT) void readout_fwd_kernel(const ReadoutP p) {
    extern __shared__ float es[];            // emb [5*Dg]
    const int b = blockIdx.x, Dg = p.Dg, E = 5 * Dg;
    for (int c = threadIdx.x; c < 2 * Dg; c += BT) {
        const bool isT = c < Dg;
        const int d = isT ? c : c - Dg, Nn = isT ? p.Nt : p.Ns;
        const float *a1 = isT ? p.T1 : p.S1, *a2 = isT ? p.T2 : p.S2;
        const unsigned char *m1 = isT ? p.mT1 : p.mS1, *m2 = isT ? p.mT2 : p.mS2;
        float amax = -1.f, sum = 0.f;
        for (int n = 0; n < Nn; ++n) {
            const size_t a = ((size_t)b * Nn + n) * Dg + d;
            const float v = fmaxf(rd_val(a1, m1, a, p.way_scale), rd_val(a2, m2, a, p.way_scale));
            amax = fmaxf(amax, fabsf(v)); sum += v;
        }
        es[(isT ? 0 : 2 * Dg) + d] = amax;
        es[(isT ? Dg : 3 * Dg) + d] = sum / (float)Nn;
    }
    for (int d = threadIdx.x; d < Dg; d += BT) {
        const size_t a = (size_t)b * Dg + d;
        es[4 * Dg + d] = fmaxf(rd_val(p.M1, p.mM1, a, p.way_scale), rd_val(p.M2, p.mM2, a, p.way_scale));
    }
    __syncthreads();
    for (int c = threadIdx.x; c < E; c += BT) {
        float v = es[c];
        if (p.mLast) v = p.mLast[(size_t)b * E + c] ? v * p.last_scale : 0.f;
        es[c] = v; p.emb[(size_t)b * E + c] = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = wave; k < p.ncls; k += BT / 64) {
        float s = 0.f;
        for (int c = lane; c < E; c += 64) s += p.W[(size_t)k * E + c] * es[c];
        s = wave_sum(s);
        if (lane == 0) p.logits[(size_t)b * p.ncls + k] = s + p.bias[k];
    }
}
struct ReadoutBP {
    ReadoutP f;
    const float *demb, *dlogits;                                     // loss gradients wrt returned emb / logits (demb may be NULL)
    float *dT1, *dT2, *dS1, *dS2, *dM1, *dM2;                        // written in full
    float *dW, *dbias;                                               // atomic accumulate
};
__global__ __launch_bounds__(BT) void readout_bwd_kernel(const ReadoutBP q) {
    extern __shared__ float gs[];            // dpre [5*Dg]
    const ReadoutP& p = q.f;
    const int b = blockIdx.x, Dg = p.Dg, E = 5 * Dg;
    for (int c = threadIdx.x; c < E; c += BT) {
        float g = q.demb ? q.demb[(size_t)b * E + c] : 0.f;
        for (int k = 0; k < p.ncls; ++k) g += q.dlogits[(size_t)b * p.ncls + k] * p.W[(size_t)k * E + c];
        if (p.mLast) g = p.mLast[(size_t)b * E + c] ? g * p.last_scale : 0.f;
        gs[c] = g;
        for (int k = 0; k < p.ncls; ++k) atomicAdd(q.dW + (size_t)k * E + c, q.dlogits[(size_t)b * p.ncls + k] * p.emb[(size_t)b * E + c]);
    }
    for (int k = threadIdx.x; k < p.ncls; k += BT) atomicAdd(q.dbias + k, q.dlogits[(size_t)b * p.ncls + k]);
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * Dg; c += BT) {
        const bool isT = c < Dg;
        const int d = isT ? c : c - Dg, Nn = isT ? p.Nt : p.Ns;
        const float *a1 = isT ? p.T1 : p.S1, *a2 = isT ? p.T2 : p.S2;
        const unsigned char *m1 = isT ? p.mT1 : p.mS1, *m2 = isT ? p.mT2 : p.mS2;
        float *g1 = isT ? q.dT1 : q.dS1, *g2 = isT ? q.dT2 : q.dS2;
        const float gmax = gs[(isT ? 0 : 2 * Dg) + d], gavg = gs[(isT ? Dg : 3 * Dg) + d] / (float)Nn;
        float amax = -1.f; int arg = 0;
        for (int n = 0; n < Nn; ++n) {
            const size_t a = ((size_t)b * Nn + n) * Dg + d;
            const float v = fmaxf(rd_val(a1, m1, a, p.way_scale), rd_val(a2, m2, a, p.way_scale));
            if (fabsf(v) > amax) { amax = fabsf(v); arg = n; }
        }
        for (int n = 0; n < Nn; ++n) {
            const size_t a = ((size_t)b * Nn + n) * Dg + d;
            const float v1 = rd_val(a1, m1, a, p.way_scale), v2 = rd_val(a2, m2, a, p.way_scale);
            const float v = fmaxf(v1, v2);
            float g = gavg;
            if (n == arg) g += v > 0.f ? gmax : (v < 0.f ? -gmax : 0.f);
            const bool first = v1 >= v2;
            const float k1 = m1 ? (m1[a] ? p.way_scale : 0.f) : 1.f, k2 = m2 ? (m2[a] ? p.way_scale : 0.f) : 1.f;
            g1[a] = first ? g * k1 : 0.f;
            g2[a] = first ? 0.f : g * k2;
        }
    }
    for (int d = threadIdx.x; d < Dg; d += BT) {
        const size_t a = (size_t)b * Dg + d;
        const float v1 = rd_val(p.M1, p.mM1, a, p.way_scale), v2 = rd_val(p.M2, p.mM2, a, p.way_scale);
        const float g = gs[4 * Dg + d];
        const bool first = v1 >= v2;
        const float k1 = p.mM1 ? (p.mM1[a] ? p.way_scale : 0.f) : 1.f, k2 = p.mM2 ? (p.mM2[a] ? p.way_scale : 0.f) : 1.f;
        q.dM1[a] = first ? g * k1 : 0.f;
        q.dM2[a] = first ? 0.f : g * k2;
    }
}

// ----------------------------------------------------------------- SE-ResNet34 pieces (models/senet.py) --
// MaxPool2d(3, stride 2, padding 1) on channels-last x [B,H,W,C] -> y [B,Ho,Wo,C] (through a row map: the next conv's
// zero-bordered input), idx = argmax tap 0..8 (255: window empty).                                   senet.py:76, 124
__global__ void maxpool3s2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, RowMapI ymap, unsigned char* __restrict__ idx, int B, int H, int W,
                                      int C, int Ho, int Wo) {
    const long long n = (long long)B * Ho * Wo * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const int c = (int)(r % C); r /= C;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho); const int b = (int)(r / Ho);
        float best = -3.4e38f; int bi = 255;
#pragma unroll
        for (int dh = 0; dh < 3; ++dh)
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int h = 2 * ho - 1 + dh, w = 2 * wo - 1 + dw;
                if (h >= 0 && h < H && w >= 0 && w < W) {
                    const float v = x[(((size_t)b * H + h) * W + w) * C + c];
                    if (v > best) { best = v; bi = dh * 3 + dw; }
                }
            }
        y[row_off(ymap, i / C) + c] = best;
        idx[i] = (unsigned char)bi;
    }
}
// dx (pre-zeroed, [B,H,W,C]) += scatter of dy (through a row map) to the argmax positions
__global__ void maxpool3s2_bwd_kernel(const float* __restrict__ dy, RowMapI dymap, const unsigned char* __restrict__ idx, float* __restrict__ dx, int B, int H,
                                      int W, int C, int Ho, int Wo) {
    const long long n = (long long)B * Ho * Wo * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const int c = (int)(r % C); r /= C;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho); const int b = (int)(r / Ho);
        const int bi = idx[i];
        if (bi == 255) continue;
        const int h = 2 * ho - 1 + bi / 3, w = 2 * wo - 1 + bi % 3;
        atomicAdd(dx + (((size_t)b * H + h) * W + w) * C + c, dy[row_off(dymap, i / C) + c]);
    }
}
// out[b,c] = alpha * sum_r x[b, r, c]  (x rows through a row map whose batch level is b): global average pools
__global__ __launch_bounds__(BT) void batch_colsum_kernel(const float* __restrict__ x, RowMapI xmap, int R, int C, float alpha, float* __restrict__ out) {
    __shared__ float red[BT];
    const int b = blockIdx.x;
    const int c = threadIdx.x % C, part = threadIdx.x / C, nparts = BT / C;
    float s = 0.f;
    for (int r = part; r < R; r += nparts) s += x[row_off(xmap, (long long)b * R + r) + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < C) {
        for (int k = 1; k < nparts; ++k) s += red[threadIdx.x + k * C];
        out[(size_t)b * C + c] = s * alpha;
    }
}
// SELayer gate (senet.py:13-28): g = sigmoid(W2 relu(W1 s)), s [B,C], W1 [Cr,C], W2 [C,Cr] (no biases); z [B,Cr] is kept.
__global__ __launch_bounds__(BT) void se_gate_fwd_kernel(const float* __restrict__ s, const float* __restrict__ W1, const float* __restrict__ W2, int C, int Cr,
                                                         float* __restrict__ z, float* __restrict__ g) {
    extern __shared__ float sm[];            // s [C] + z [Cr]
    float* ss = sm; float* zs = sm + C;
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += BT) ss[c] = s[(size_t)b * C + c];
    __syncthreads();
    for (int j = threadIdx.x; j < Cr; j += BT) {
        float a = 0.f;
        for (int c = 0; c < C; ++c) a = fmaf(W1[j * C + c], ss[c], a);
        a = a > 0.f ? a : 0.f;
        zs[j] = a; z[(size_t)b * Cr + j] = a;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += BT) {
        float a = 0.f;
        for (int j = 0; j < Cr; ++j) a = fmaf(W2[c * Cr + j], zs[j], a);
        g[(size_t)b * C + c] = 1.0f / (1.0f + expf(-a));
    }
}
// given dg [B,C]: dW2 += (dg*g*(1-g)) z^T ; dz = W2^T (..) masked by z>0 ; dW1 += dz s^T ; ds = W1^T dz
__global__ __launch_bounds__(BT) void se_gate_bwd_kernel(const float* __restrict__ s, const float* __restrict__ z, const float* __restrict__ g,
                                                         const float* __restrict__ dg, const float* __restrict__ W1, const float* __restrict__ W2, int C, int Cr,
                                                         float* __restrict__ dW1, float* __restrict__ dW2, float* __restrict__ ds) {
    extern __shared__ float sm[];            // da [C] + dz [Cr] + s [C] + z [Cr]
    float* da = sm; float* dz = da + C; float* ss = dz + Cr; float* zs = ss + C;
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += BT) {
        const float gv = g[(size_t)b * C + c];
        da[c] = dg[(size_t)b * C + c] * gv * (1.f - gv);
        ss[c] = s[(size_t)b * C + c];
    }
    for (int j = threadIdx.x; j < Cr; j += BT) zs[j] = z[(size_t)b * Cr + j];
    __syncthreads();
    for (int i = threadIdx.x; i < C * Cr; i += BT) atomicAdd(dW2 + i, da[i / Cr] * zs[i % Cr]);
    for (int j = threadIdx.x; j < Cr; j += BT) {
        float a = 0.f;
        for (int c = 0; c < C; ++c) a = fmaf(W2[c * Cr + j], da[c], a);
        dz[j] = zs[j] > 0.f ? a : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < Cr * C; i += BT) atomicAdd(dW1 + i, dz[i / C] * ss[i % C]);
    for (int c = threadIdx.x; c < C; c += BT) {
        float a = 0.f;
        for (int j = 0; j < Cr; ++j) a = fmaf(W1[j * C + c], dz[j], a);
        ds[(size_t)b * C + c] = a;
    }
}
// out = relu(y * gate[b,c] + res)   (SEBasicBlock tail, senet.py:53-61); rows = B*R, batch b = row / R
__global__ void se_scale_add_relu_kernel(const float* __restrict__ y, const float* __restrict__ gate, const float* __restrict__ res, RowMapI rmap,
                                         float* __restrict__ out, RowMapI omap, long long rows, int R, int C) {
    const long long n = rows * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / C; const int c = (int)(i - r * C);
        const float v = y[i] * gate[(r / R) * C + c] + res[row_off(rmap, r) + c];
        out[row_off(omap, r) + c] = v > 0.f ? v : 0.f;
    }
}
// dpre = dout * (out > 0); dres = dpre (through a row map, accumulate or write); dy = dpre * gate; dgate[b,c] += sum dpre*y
__global__ __launch_bounds__(BT) void se_scale_add_relu_bwd_kernel(const float* __restrict__ dout, RowMapI domap, const float* __restrict__ out, RowMapI omap,
                                                                   const float* __restrict__ y, const float* __restrict__ gate, float* __restrict__ dy,
                                                                   float* __restrict__ dres, RowMapI drmap, int dres_accumulate, float* __restrict__ dgate,
                                                                   int R, int C) {
    __shared__ float red[BT];
    const int b = blockIdx.y;
    const int c = threadIdx.x % C, part = threadIdx.x / C, nparts = BT / C;
    const int chunk = (R + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * chunk, r1 = r0 + chunk < R ? r0 + chunk : R;
    float acc = 0.f;
    for (int rr = r0 + part; rr < r1; rr += nparts) {
        const long long r = (long long)b * R + rr;
        const float o = out[row_off(omap, r) + c];
        const float dp = o > 0.f ? dout[row_off(domap, r) + c] : 0.f;
        const float yv = y[r * C + c];
        dy[r * C + c] = dp * gate[(size_t)b * C + c];
        float* dr = dres + row_off(drmap, r) + c;
        *dr = dres_accumulate ? *dr + dp : dp;
        acc += dp * yv;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < C) {
        for (int k = 1; k < nparts; ++k) acc += red[threadIdx.x + k * C];
        atomicAdd(dgate + (size_t)b * C + c, acc);
    }
}
// x[b, r, c] += v[b, c]   (gradient of a global average pool, v already divided by R)
__global__ void add_batch_vec_kernel(float* __restrict__ x, RowMapI xmap, const float* __restrict__ v, long long rows, int R, int C) {
    const long long n = rows * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / C; const int c = (int)(i - r * C);
        x[row_off(xmap, r) + c] += v[(r / R) * C + c];
    }
}

// Input gradient of the SE-ResNet stem conv (1 input channel, 16 output channels, 7x7, stride 2, padding 3; senet.py:73):
// dx[b,h,w] = sum_{kh,kw,co} dy[b,(h+3-kh)/2,(w+3-kw)/2,co] * w[co,kh,kw] over the taps with even offsets that land inside dy.
// w is the engine's [16][7][7][4] layout (input channel padded to 4); one thread per input pixel, <= 16 taps x 16 channels.
__global__ __launch_bounds__(BT) void conv7s2_dgrad_c1_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int B, int H, int W,
                                                              int Ho, int Wo) {
    __shared__ float wl[49 * 16];               // [kh][kw][co]
    for (int i = threadIdx.x; i < 49 * 16; i += BT) { const int co = i & 15, t = i >> 4; wl[i] = w[(co * 49 + t) * 4]; }
    __syncthreads();
    const long long n = (long long)B * H * W;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W); const long long r = i / W; const int h = (int)(r % H); const int b = (int)(r / H);
        float acc = 0.f;
        for (int kh = (h + 3) & 1; kh < 7; kh += 2) {
            const int ho = (h + 3 - kh) >> 1;
            if (h + 3 - kh < 0 || ho >= Ho) continue;
            for (int kw = (x + 3) & 1; kw < 7; kw += 2) {
                const int wo = (x + 3 - kw) >> 1;
                if (x + 3 - kw < 0 || wo >= Wo) continue;
                const float4* d = reinterpret_cast<const float4*>(dy + (((size_t)b * Ho + ho) * Wo + wo) * 16);
                const float* ww = wl + (kh * 7 + kw) * 16;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = d[q];
                    acc = fmaf(v.x, ww[4 * q], acc); acc = fmaf(v.y, ww[4 * q + 1], acc); acc = fmaf(v.z, ww[4 * q + 2], acc); acc = fmaf(v.w, ww[4 * q + 3], acc);
                }
            }
        }
        dx[i] = acc;
    }
}

int grid_for(long long n, int per = BT) { long long g = occ_cdiv(n, per); return (int)(g < 8192 ? (g < 1 ? 1 : g) : 8192); }

}  // namespace

extern "C" {

int occ_fill_f32(float* p, float v, int64_t n, void* stream) {
    OCC_CHECK_ARG(p && n >= 0, "occ_fill_f32: bad argument");
    if (n == 0) return OCC_OK;
    hipLaunchKernelGGL(fill_f32_kernel, dim3(grid_for(n)), dim3(BT), 0, (hipStream_t)stream, p, v, (long long)n);
    OCC_LAUNCH_CHECK("occ_fill_f32");
    return OCC_OK;
}
int occ_axpby_f32(const float* a, const float* b, float* out, float alpha, float beta, int64_t n, void* stream) {
    OCC_CHECK_ARG(a && out && n >= 0, "occ_axpby_f32: bad argument");
    if (n == 0) return OCC_OK;
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(BT), 0, (hipStream_t)stream, a, b, out, alpha, beta, (long long)n);
    OCC_LAUNCH_CHECK("occ_axpby_f32");
    return OCC_OK;
}
int occ_copy_strided(const float* in, float* out, int64_t offset, const int64_t* shape4_host, const int64_t* strides4_host, int accumulate, void* stream) {
    OCC_CHECK_ARG(in && out && shape4_host && strides4_host, "occ_copy_strided: null pointer");
    const int64_t* n = shape4_host; const int64_t* s = strides4_host;
    OCC_CHECK_ARG(n[0] >= 1 && n[1] >= 1 && n[2] >= 1 && n[3] >= 1 && n[0] * n[1] * n[2] * n[3] < (1ll << 40), "occ_copy_strided: bad shape");
    hipLaunchKernelGGL(copy_strided_kernel, dim3(grid_for(n[0] * n[1] * n[2] * n[3])), dim3(BT), 0, (hipStream_t)stream, in, out, (long long)offset,
                       (int)n[0], (int)n[1], (int)n[2], (int)n[3], (long long)s[0], (long long)s[1], (long long)s[2], (long long)s[3], accumulate);
    OCC_LAUNCH_CHECK("occ_copy_strided");
    return OCC_OK;
}
int occ_copy_rows(const float* x, const occ_rowmap* x_map, float* y, const occ_rowmap* y_map, int64_t rows, int64_t C, int accumulate, void* stream) {
    OCC_CHECK_ARG(x && x_map && y && y_map && rows >= 1 && C >= 1 && x_map->rows_per_batch >= 1 && y_map->rows_per_batch >= 1, "occ_copy_rows: bad argument");
    hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for(rows * C)), dim3(BT), 0, (hipStream_t)stream, x, to_rowmap(*x_map), y, to_rowmap(*y_map), (long long)rows, (int)C, accumulate);
    OCC_LAUNCH_CHECK("occ_copy_rows");
    return OCC_OK;
}
int occ_act_bwd(const float* dy, const float* y, float* dx, int act, int64_t n, void* stream) {
    OCC_CHECK_ARG(dy && y && dx && n >= 0, "occ_act_bwd: bad argument");
    if (n == 0) return OCC_OK;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n)), dim3(BT), 0, (hipStream_t)stream, dy, y, dx, act, (long long)n);
    OCC_LAUNCH_CHECK("occ_act_bwd");
    return OCC_OK;
}
int occ_dropout(const float* x, float* y, uint8_t* mask, int64_t n, float p, uint64_t seed, uint64_t stream_id, int generate, void* stream) {
    OCC_CHECK_ARG(x && y && mask && n >= 0 && p >= 0.f && p < 1.f, "occ_dropout: bad argument");
    if (n == 0) return OCC_OK;
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(occ_cdiv(n, 4))), dim3(BT), 0, (hipStream_t)stream, x, y, mask, (long long)n, p, seed, stream_id, generate,
                       (const uint64_t*)nullptr);
    OCC_LAUNCH_CHECK("occ_dropout");
    return OCC_OK;
}

__global__ void add_u64_kernel(uint64_t* p, uint64_t v) { *p += v; }

int occ_dropout_step(const float* x, float* y, uint8_t* mask, int64_t n, float p, uint64_t seed, const uint64_t* step, uint64_t site, void* stream) {
    OCC_CHECK_ARG(x && y && mask && step && n >= 0 && p >= 0.f && p < 1.f && site < 256, "occ_dropout_step: bad argument");
    if (n == 0) return OCC_OK;
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(occ_cdiv(n, 4))), dim3(BT), 0, (hipStream_t)stream, x, y, mask, (long long)n, p, seed, site, 1, step);
    OCC_LAUNCH_CHECK("occ_dropout_step");
    return OCC_OK;
}

int occ_add_u64(uint64_t* counter, uint64_t v, void* stream) {
    OCC_CHECK_ARG(counter, "occ_add_u64: null pointer");
    hipLaunchKernelGGL(add_u64_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter, v);
    OCC_LAUNCH_CHECK("occ_add_u64");
    return OCC_OK;
}
int occ_stem_pool_fwd(const float* y, float* out, uint8_t* idx, int64_t B, int64_t T, int64_t F, int64_t out_c, void* stream) {
    OCC_CHECK_ARG(y && out && idx && B >= 1 && T >= 3 && F >= 3 && out_c >= 1, "occ_stem_pool_fwd: bad argument");
    const long long n = B * (F / 3) * (T / 3);
    hipLaunchKernelGGL(stem_pool_fwd_kernel, dim3(grid_for(n)), dim3(BT), 0, (hipStream_t)stream, y, out, idx, (int)B, (int)T, (int)F, (int)(F / 3), (int)(T / 3), (int)out_c);
    OCC_LAUNCH_CHECK("occ_stem_pool_fwd");
    return OCC_OK;
}
int occ_stem_pool_bwd(const float* dout, const uint8_t* idx, float* dy, int64_t B, int64_t T, int64_t F, int64_t dout_c, void* stream) {
    OCC_CHECK_ARG(dout && idx && dy && B >= 1 && T >= 3 && F >= 3 && dout_c >= 1, "occ_stem_pool_bwd: bad argument");
    const long long n = B * (F / 3) * (T / 3);
    hipLaunchKernelGGL(stem_pool_bwd_kernel, dim3(grid_for(n)), dim3(BT), 0, (hipStream_t)stream, dout, idx, dy, (int)B, (int)T, (int)F, (int)(F / 3), (int)(T / 3), (int)dout_c);
    OCC_LAUNCH_CHECK("occ_stem_pool_bwd");
    return OCC_OK;
}

static bool bn_vec(const occ_rowmap& m, int64_t C, const void* p) {
    return C % 4 == 0 && m.row_stride % 4 == 0 && m.batch_stride % 4 == 0 && m.line_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0;
}
// chunks of the row range for the partial sums; the f64 workspace [nchunk][C][2] holds at most 512*256*2 entries
static int bn_chunks(int64_t rows, int64_t C, long long* rpc) {
    long long nchunk = occ_cdiv(rows, 64);
    const long long cap = C % 4 == 0 ? (512 * 256) / C : 512;
    if (nchunk > cap) nchunk = cap;
    if (nchunk > 512) nchunk = 512;             // two workgroups per CU feed the partial kernel; the finalize wave walks nchunk / 64 strided rows (10 us at 1386)
    *rpc = occ_cdiv(rows, nchunk);
    return (int)occ_cdiv(rows, *rpc);
}
int occ_bn_stats(const float* x, const occ_rowmap* x_map, int64_t rows, int64_t C, double* ws, float* mean, float* rstd, float* running_mean,
                 float* running_var, int64_t* num_batches_tracked, float momentum, float eps, int train, void* stream) {
    OCC_CHECK_ARG(x && x_map && ws && rows >= 1 && C >= 1 && C <= 256 && 256 % C == 0, "occ_bn_stats: C must divide 256 (C=%ld)", (long)C);
    OCC_CHECK_ARG(train || (mean && rstd && running_mean && running_var), "occ_bn_stats: eval mode needs running statistics and outputs");
    hipStream_t s = (hipStream_t)stream;
    int nchunk = 0;
    if (train) {
        long long rpc; nchunk = bn_chunks(rows, C, &rpc);
        if (bn_vec(*x_map, C, x)) hipLaunchKernelGGL(bn_partial_vec_kernel, dim3(nchunk), dim3(BT), 0, s, x, to_rowmap(*x_map), (long long)rows, (int)C, rpc, ws);
        else hipLaunchKernelGGL(bn_partial_kernel, dim3(nchunk), dim3(BT), 0, s, x, to_rowmap(*x_map), (long long)rows, (int)C, rpc, ws);
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)occ_cdiv(C, BT / 64)), dim3(BT), 0, s, (const double*)ws, nchunk, (int)C, (long long)rows, mean, rstd,
                       running_mean, running_var, (long long*)num_batches_tracked, momentum, eps, train);
    OCC_LAUNCH_CHECK("occ_bn_stats");
    return OCC_OK;
}
int occ_bn_act_fwd(const float* x, const occ_rowmap* x_map, const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                   float* y, const occ_rowmap* y_map, int64_t rows, int64_t C, void* stream) {
    OCC_CHECK_ARG(x && x_map && mean && rstd && gamma && beta && y && y_map && rows >= 1 && C >= 1, "occ_bn_act_fwd: bad argument");
    const dim3 grid(grid_for(rows * C)), block(BT);
    hipStream_t s = (hipStream_t)stream;
    const RowMapI xm = to_rowmap(*x_map), ym = to_rowmap(*y_map);
    if (bn_vec(*x_map, C, x) && bn_vec(*y_map, C, y) && ((uintptr_t)mean & 15) == 0 && ((uintptr_t)rstd & 15) == 0 && ((uintptr_t)gamma & 15) == 0 &&
        ((uintptr_t)beta & 15) == 0 && (act == OCC_ACT_SELU || act == OCC_ACT_RELU || act == OCC_ACT_NONE)) {
        const dim3 gv(grid_for(rows * C / 4));
        if (act == OCC_ACT_SELU) hipLaunchKernelGGL(bn_act_fwd_vec_kernel<OCC_ACT_SELU>, gv, block, 0, s, x, xm, mean, rstd, gamma, beta, y, ym, (long long)rows, (int)C);
        else if (act == OCC_ACT_RELU) hipLaunchKernelGGL(bn_act_fwd_vec_kernel<OCC_ACT_RELU>, gv, block, 0, s, x, xm, mean, rstd, gamma, beta, y, ym, (long long)rows, (int)C);
        else hipLaunchKernelGGL(bn_act_fwd_vec_kernel<OCC_ACT_NONE>, gv, block, 0, s, x, xm, mean, rstd, gamma, beta, y, ym, (long long)rows, (int)C);
        OCC_LAUNCH_CHECK("occ_bn_act_fwd");
        return OCC_OK;
    }
    if (act == OCC_ACT_SELU) hipLaunchKernelGGL(bn_act_fwd_kernel<OCC_ACT_SELU>, grid, block, 0, s, x, xm, mean, rstd, gamma, beta, y, ym, (long long)rows, (int)C);
    else if (act == OCC_ACT_RELU) hipLaunchKernelGGL(bn_act_fwd_kernel<OCC_ACT_RELU>, grid, block, 0, s, x, xm, mean, rstd, gamma, beta, y, ym, (long long)rows, (int)C);
    else if (act == OCC_ACT_NONE) hipLaunchKernelGGL(bn_act_fwd_kernel<OCC_ACT_NONE>, grid, block, 0, s, x, xm, mean, rstd, gamma, beta, y, ym, (long long)rows, (int)C);
    else { occ_set_error("occ_bn_act_fwd: unsupported activation %d", act); return OCC_EUNSUPPORTED; }
    OCC_LAUNCH_CHECK("occ_bn_act_fwd");
    return OCC_OK;
}
int occ_bn_act_bwd(const float* dy, const occ_rowmap* dy_map, const float* x, const occ_rowmap* x_map, const float* mean, const float* rstd,
                   const float* gamma, const float* beta, int act, float* dx, const occ_rowmap* dx_map, float* dgamma, float* dbeta, double* ws,
                   float* sums, int64_t rows, int64_t C, void* stream) {
    OCC_CHECK_ARG(dy && dy_map && x && x_map && mean && rstd && gamma && beta && dx && dx_map && ws && sums, "occ_bn_act_bwd: null pointer");
    OCC_CHECK_ARG(rows >= 1 && C >= 1 && C <= 256 && 256 % C == 0, "occ_bn_act_bwd: C must divide 256");
    hipStream_t s = (hipStream_t)stream;
    long long rpc; const int nchunk = bn_chunks(rows, C, &rpc);
    const RowMapI dym = to_rowmap(*dy_map), xm = to_rowmap(*x_map), dxm = to_rowmap(*dx_map);
    const dim3 grid(grid_for(rows * C)), block(BT);
    if (bn_vec(*dy_map, C, dy) && bn_vec(*x_map, C, x) && bn_vec(*dx_map, C, dx) && ((uintptr_t)mean & 15) == 0 && ((uintptr_t)rstd & 15) == 0 &&
        ((uintptr_t)gamma & 15) == 0 && ((uintptr_t)beta & 15) == 0 && (act == OCC_ACT_SELU || act == OCC_ACT_RELU || act == OCC_ACT_NONE)) {
        const dim3 gv(grid_for(rows * C / 4));
#define OCC_BN_BWD_V(A)                                                                                                                    \
    hipLaunchKernelGGL(bn_bwd_partial_vec_kernel<A>, dim3(nchunk), block, 0, s, dy, dym, x, xm, mean, rstd, gamma, beta, (long long)rows, (int)C, rpc, ws); \
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)occ_cdiv(C, BT / 64)), dim3(BT), 0, s, (const double*)ws, nchunk, (int)C, sums, dgamma, dbeta); \
    hipLaunchKernelGGL(bn_bwd_apply_vec_kernel<A>, gv, block, 0, s, dy, dym, x, xm, mean, rstd, gamma, beta, (const float*)sums, dx, dxm, (long long)rows, (int)C)
        if (act == OCC_ACT_SELU) { OCC_BN_BWD_V(OCC_ACT_SELU); }
        else if (act == OCC_ACT_RELU) { OCC_BN_BWD_V(OCC_ACT_RELU); }
        else { OCC_BN_BWD_V(OCC_ACT_NONE); }
#undef OCC_BN_BWD_V
        OCC_LAUNCH_CHECK("occ_bn_act_bwd");
        return OCC_OK;
    }
#define OCC_BN_BWD(A)                                                                                                                      \
    hipLaunchKernelGGL(bn_bwd_partial_kernel<A>, dim3(nchunk), block, 0, s, dy, dym, x, xm, mean, rstd, gamma, beta, (long long)rows, (int)C, rpc, ws); \
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)occ_cdiv(C, BT / 64)), dim3(BT), 0, s, (const double*)ws, nchunk, (int)C, sums, dgamma, dbeta); \
    hipLaunchKernelGGL(bn_bwd_apply_kernel<A>, grid, block, 0, s, dy, dym, x, xm, mean, rstd, gamma, beta, (const float*)sums, dx, dxm, (long long)rows, (int)C)
    if (act == OCC_ACT_SELU) { OCC_BN_BWD(OCC_ACT_SELU); }
    else if (act == OCC_ACT_RELU) { OCC_BN_BWD(OCC_ACT_RELU); }
    else if (act == OCC_ACT_NONE) { OCC_BN_BWD(OCC_ACT_NONE); }
    else { occ_set_error("occ_bn_act_bwd: unsupported activation %d", act); return OCC_EUNSUPPORTED; }
#undef OCC_BN_BWD
    OCC_LAUNCH_CHECK("occ_bn_act_bwd");
    return OCC_OK;
}

int occ_softmax_wsum_fwd(const float* x, const float* w, int64_t n_outer, int64_t inner_n, int64_t outer_stride, int64_t inner_stride, int64_t R,
                         int64_t r_stride, int64_t C, const float* pos, int64_t pos_period, float* out, void* stream) {
    OCC_CHECK_ARG(x && w && out && n_outer >= 1 && inner_n >= 1 && R >= 1 && C >= 1 && C <= 256 && 256 % C == 0, "occ_softmax_wsum_fwd: bad argument");
    AxisMap am{inner_n, outer_stride, inner_stride, r_stride};
    hipLaunchKernelGGL(softmax_wsum_fwd_kernel, dim3((unsigned)n_outer), dim3(BT), 0, (hipStream_t)stream, x, w, am, (int)R, (int)C, pos,
                       (int)(pos_period > 0 ? pos_period : 1), out);
    OCC_LAUNCH_CHECK("occ_softmax_wsum_fwd");
    return OCC_OK;
}
int occ_softmax_wsum_bwd(const float* x, const float* w, int64_t n_outer, int64_t inner_n, int64_t outer_stride, int64_t inner_stride, int64_t R,
                         int64_t r_stride, int64_t C, const float* dm, float* dx, float* dw, int accumulate, void* stream) {
    OCC_CHECK_ARG(x && w && dm && dx && dw && n_outer >= 1 && inner_n >= 1 && R >= 1 && C >= 1 && C <= 256 && 256 % C == 0, "occ_softmax_wsum_bwd: bad argument");
    AxisMap am{inner_n, outer_stride, inner_stride, r_stride};
    hipLaunchKernelGGL(softmax_wsum_bwd_kernel, dim3((unsigned)n_outer), dim3(BT), 0, (hipStream_t)stream, x, w, am, (int)R, (int)C, dm, dx, dw, accumulate);
    OCC_LAUNCH_CHECK("occ_softmax_wsum_bwd");
    return OCC_OK;
}

int occ_pair_mul(const float* x, float* P, int64_t B, int64_t N, int64_t D, void* stream) {
    OCC_CHECK_ARG(x && P && B >= 1 && N >= 1 && D >= 4 && D % 4 == 0, "occ_pair_mul: bad argument");
    hipLaunchKernelGGL(pair_mul_kernel, dim3(grid_for(B * N * N * D / 4)), dim3(BT), 0, (hipStream_t)stream, x, P, (int)B, (int)N, (int)D);
    OCC_LAUNCH_CHECK("occ_pair_mul");
    return OCC_OK;
}
int occ_pair_mul_bwd(const float* dP, const float* x, float* dx, int64_t B, int64_t N, int64_t D, int accumulate, void* stream) {
    OCC_CHECK_ARG(dP && x && dx && B >= 1 && N >= 1 && D >= 1 && D <= 256 && 256 % D == 0, "occ_pair_mul_bwd: bad argument");
    hipLaunchKernelGGL(pair_mul_bwd_kernel, dim3((unsigned)(B * N)), dim3(BT), 0, (hipStream_t)stream, dP, x, dx, (int)N, (int)D, accumulate);
    OCC_LAUNCH_CHECK("occ_pair_mul_bwd");
    return OCC_OK;
}
int occ_gat_softmax(const float* A, const float* aw, int64_t B, int64_t N, int64_t Do, int64_t n1, float inv_temp, float* alpha, void* stream) {
    OCC_CHECK_ARG(A && aw && alpha && B >= 1 && N >= 1 && N <= 4096 && Do >= 1 && n1 >= 0 && n1 <= N, "occ_gat_softmax: bad argument");
    hipLaunchKernelGGL(gat_softmax_kernel, dim3((unsigned)(B * N)), dim3(BT), (N + 8) * sizeof(float), (hipStream_t)stream, A, aw, (int)N, (int)Do, (int)n1, inv_temp, alpha);
    OCC_LAUNCH_CHECK("occ_gat_softmax");
    return OCC_OK;
}
int occ_bmm_alpha(const float* alpha, const float* x, float* out, int64_t B, int64_t N, int64_t D, int trans, int accumulate, void* stream) {
    OCC_CHECK_ARG(alpha && x && out && B >= 1 && N >= 1 && D >= 1 && D <= 256 && 256 % D == 0, "occ_bmm_alpha: bad argument");
    hipLaunchKernelGGL(bmm_alpha_kernel, dim3((unsigned)(B * N)), dim3(BT), 0, (hipStream_t)stream, alpha, x, out, (int)N, (int)D, trans, accumulate);
    OCC_LAUNCH_CHECK("occ_bmm_alpha");
    return OCC_OK;
}
int occ_gat_dscore(const float* alpha, const float* dh, const float* x, float* ds, int64_t B, int64_t N, int64_t D, float inv_temp, void* stream) {
    OCC_CHECK_ARG(alpha && dh && x && ds && B >= 1 && N >= 1 && N <= 4096 && D >= 1 && D <= 1024, "occ_gat_dscore: bad argument");
    hipLaunchKernelGGL(gat_dscore_kernel, dim3((unsigned)(B * N)), dim3(BT), (D + N + 8) * sizeof(float), (hipStream_t)stream, alpha, dh, x, ds, (int)N, (int)D, inv_temp);
    OCC_LAUNCH_CHECK("occ_gat_dscore");
    return OCC_OK;
}
int occ_gat_dz(float* A, const float* ds, const float* aw, int64_t B, int64_t N, int64_t Do, int64_t n1, float* daw, void* stream) {
    OCC_CHECK_ARG(A && ds && aw && daw && B >= 1 && N >= 1 && Do >= 1 && Do <= 64 && n1 >= 0 && n1 <= N, "occ_gat_dz: bad argument (Do <= 64)");
    hipLaunchKernelGGL(gat_dz_kernel, dim3((unsigned)(B * N)), dim3(BT), 0, (hipStream_t)stream, A, ds, aw, (int)N, (int)Do, (int)n1, daw);
    OCC_LAUNCH_CHECK("occ_gat_dz");
    return OCC_OK;
}

static MasterP make_master(const occ_master_desc* d) {
    MasterP p;
    p.x = d->x; p.master = d->master; p.master_bstride = d->master_bstride;
    p.Wm = d->att_projM_w; p.bm = d->att_projM_b; p.awm = d->att_weightM; p.Wwa = d->proj_with_attM_w; p.bwa = d->proj_with_attM_b;
    p.Wwo = d->proj_without_attM_w; p.bwo = d->proj_without_attM_b;
    p.out = d->out; p.am = d->am; p.agg = d->agg; p.N = (int)d->N; p.D = (int)d->D; p.Do = (int)d->Do; p.inv_temp = d->inv_temp;
    return p;
}
int occ_master_fwd(const occ_master_desc* d, void* stream) {
    OCC_CHECK_ARG(d && d->x && d->master && d->out && d->am && d->agg && d->B >= 1 && d->N >= 1 && d->D >= 1 && d->Do >= 1, "occ_master_fwd: bad argument");
    const size_t shm = ((size_t)d->N * d->D + d->D + (size_t)d->N * d->Do + d->N + d->D + 8) * sizeof(float);
    OCC_CHECK_ARG(shm <= 64 * 1024, "occ_master_fwd: graph too large for LDS (%zu B)", shm);
    hipLaunchKernelGGL(master_fwd_kernel, dim3((unsigned)d->B), dim3(BT), shm, (hipStream_t)stream, make_master(d));
    OCC_LAUNCH_CHECK("occ_master_fwd");
    return OCC_OK;
}
int occ_master_bwd(const occ_master_desc* d, const occ_master_grads* g, void* stream) {
    OCC_CHECK_ARG(d && g && d->x && d->master && d->am && d->agg && g->dout && g->dx && g->dmaster, "occ_master_bwd: null pointer");
    OCC_CHECK_ARG(g->d_att_projM_w && g->d_att_projM_b && g->d_att_weightM && g->d_proj_with_attM_w && g->d_proj_with_attM_b &&
                  g->d_proj_without_attM_w && g->d_proj_without_attM_b, "occ_master_bwd: null gradient buffer");
    MasterBP q;
    q.f = make_master(d);
    q.dout = g->dout; q.dx = g->dx; q.dx_accumulate = g->dx_accumulate; q.dmaster = g->dmaster; q.dmaster_bstride = g->dmaster_bstride;
    q.dWm = g->d_att_projM_w; q.dbm = g->d_att_projM_b; q.dawm = g->d_att_weightM; q.dWwa = g->d_proj_with_attM_w; q.dbwa = g->d_proj_with_attM_b;
    q.dWwo = g->d_proj_without_attM_w; q.dbwo = g->d_proj_without_attM_b;
    const size_t shm = ((size_t)d->N * d->D + d->D + (size_t)d->N * d->Do + 2 * d->N + 2 * d->D + d->Do + 8) * sizeof(float);
    OCC_CHECK_ARG(shm <= 64 * 1024, "occ_master_bwd: graph too large for LDS (%zu B)", shm);
    hipLaunchKernelGGL(master_bwd_kernel, dim3((unsigned)d->B), dim3(BT), shm, (hipStream_t)stream, q);
    OCC_LAUNCH_CHECK("occ_master_bwd");
    return OCC_OK;
}

int occ_graph_pool_fwd(const float* h, const uint8_t* mask, float drop_p, const float* w, const float* bias, int64_t B, int64_t N, int64_t D, int64_t k,
                       float* out, int32_t* idx, float* scores, void* stream) {
    OCC_CHECK_ARG(h && w && bias && out && idx && scores && B >= 1 && N >= 1 && N <= 2048 && D >= 1 && k >= 1 && k <= N, "occ_graph_pool_fwd: bad argument");
    hipLaunchKernelGGL(graph_pool_fwd_kernel, dim3((unsigned)B), dim3(BT), (N + k) * sizeof(float), (hipStream_t)stream, h, mask, 1.0f / (1.0f - drop_p), w, bias,
                       (int)N, (int)D, (int)k, out, (int*)idx, scores);
    OCC_LAUNCH_CHECK("occ_graph_pool_fwd");
    return OCC_OK;
}
int occ_graph_pool_bwd(const float* h, const uint8_t* mask, float drop_p, const float* w, const float* scores, const int32_t* idx, const float* dout,
                       int64_t B, int64_t N, int64_t D, int64_t k, float* dh, float* dw, float* dbias, void* stream) {
    OCC_CHECK_ARG(h && w && scores && idx && dout && dh && dw && dbias && B >= 1 && N >= 1 && N <= 2048 && D >= 1 && k >= 1 && k <= N, "occ_graph_pool_bwd: bad argument");
    hipLaunchKernelGGL(graph_pool_bwd_kernel, dim3((unsigned)B), dim3(BT), 2 * N * sizeof(float), (hipStream_t)stream, h, mask, 1.0f / (1.0f - drop_p), w, scores,
                       (const int*)idx, dout, (int)N, (int)D, (int)k, dh, dw, dbias);
    OCC_LAUNCH_CHECK("occ_graph_pool_bwd");
    return OCC_OK;
}

static ReadoutP make_readout(const occ_readout_desc* d) {
    ReadoutP p;
    p.T1 = d->T1; p.T2 = d->T2; p.S1 = d->S1; p.S2 = d->S2; p.M1 = d->M1; p.M2 = d->M2;
    p.mT1 = d->mask_T1; p.mT2 = d->mask_T2; p.mS1 = d->mask_S1; p.mS2 = d->mask_S2; p.mM1 = d->mask_M1; p.mM2 = d->mask_M2; p.mLast = d->mask_last;
    p.way_scale = 1.0f / (1.0f - d->p_way); p.last_scale = 1.0f / (1.0f - d->p_last);
    p.W = d->out_w; p.bias = d->out_b; p.emb = d->emb; p.logits = d->logits;
    p.Nt = (int)d->Nt; p.Ns = (int)d->Ns; p.Dg = (int)d->Dg; p.ncls = (int)d->n_classes;
    return p;
}
int occ_readout_fwd(const occ_readout_desc* d, void* stream) {
    OCC_CHECK_ARG(d && d->T1 && d->T2 && d->S1 && d->S2 && d->M1 && d->M2 && d->out_w && d->out_b && d->emb && d->logits, "occ_readout_fwd: null pointer");
    OCC_CHECK_ARG(d->B >= 1 && d->Nt >= 1 && d->Ns >= 1 && d->Dg >= 1 && d->Dg <= 512 && d->n_classes >= 1, "occ_readout_fwd: bad shape");
    hipLaunchKernelGGL(readout_fwd_kernel, dim3((unsigned)d->B), dim3(BT), 5 * d->Dg * sizeof(float), (hipStream_t)stream, make_readout(d));
    OCC_LAUNCH_CHECK("occ_readout_fwd");
    return OCC_OK;
}
int occ_readout_bwd(const occ_readout_desc* d, const occ_readout_grads* g, void* stream) {
    OCC_CHECK_ARG(d && g && g->dlogits && g->dT1 && g->dT2 && g->dS1 && g->dS2 && g->dM1 && g->dM2 && g->d_out_w && g->d_out_b, "occ_readout_bwd: null pointer");
    ReadoutBP q;
    q.f = make_readout(d);
    q.demb = g->demb; q.dlogits = g->dlogits; q.dT1 = g->dT1; q.dT2 = g->dT2; q.dS1 = g->dS1; q.dS2 = g->dS2; q.dM1 = g->dM1; q.dM2 = g->dM2;
    q.dW = g->d_out_w; q.dbias = g->d_out_b;
    hipLaunchKernelGGL(readout_bwd_kernel, dim3((unsigned)d->B), dim3(BT), 5 * d->Dg * sizeof(float), (hipStream_t)stream, q);
    OCC_LAUNCH_CHECK("occ_readout_bwd");
    return OCC_OK;
}

int occ_maxpool3s2_fwd(const float* x, float* y, const occ_rowmap* y_map, uint8_t* idx, int64_t B, int64_t H, int64_t W, int64_t C, void* stream) {
    OCC_CHECK_ARG(x && y && y_map && idx && B >= 1 && H >= 1 && W >= 1 && C >= 1 && y_map->rows_per_batch >= 1, "occ_maxpool3s2_fwd: bad argument");
    const int Ho = (int)((H + 2 - 3) / 2 + 1), Wo = (int)((W + 2 - 3) / 2 + 1);
    hipLaunchKernelGGL(maxpool3s2_fwd_kernel, dim3(grid_for(B * Ho * Wo * C)), dim3(BT), 0, (hipStream_t)stream, x, y, to_rowmap(*y_map), idx, (int)B, (int)H, (int)W,
                       (int)C, Ho, Wo);
    OCC_LAUNCH_CHECK("occ_maxpool3s2_fwd");
    return OCC_OK;
}
int occ_maxpool3s2_bwd(const float* dy, const occ_rowmap* dy_map, const uint8_t* idx, float* dx, int64_t B, int64_t H, int64_t W, int64_t C, void* stream) {
    OCC_CHECK_ARG(dy && dy_map && idx && dx && B >= 1 && H >= 1 && W >= 1 && C >= 1 && dy_map->rows_per_batch >= 1, "occ_maxpool3s2_bwd: bad argument");
    const int Ho = (int)((H + 2 - 3) / 2 + 1), Wo = (int)((W + 2 - 3) / 2 + 1);
    hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3(grid_for(B * Ho * Wo * C)), dim3(BT), 0, (hipStream_t)stream, dy, to_rowmap(*dy_map), idx, dx, (int)B, (int)H, (int)W,
                       (int)C, Ho, Wo);
    OCC_LAUNCH_CHECK("occ_maxpool3s2_bwd");
    return OCC_OK;
}
int occ_batch_colsum(const float* x, const occ_rowmap* x_map, int64_t B, int64_t R, int64_t C, float alpha, float* out, void* stream) {
    OCC_CHECK_ARG(x && x_map && out && B >= 1 && R >= 1 && C >= 1 && C <= 256 && 256 % C == 0 && x_map->rows_per_batch >= 1, "occ_batch_colsum: bad argument");
    hipLaunchKernelGGL(batch_colsum_kernel, dim3((unsigned)B), dim3(BT), 0, (hipStream_t)stream, x, to_rowmap(*x_map), (int)R, (int)C, alpha, out);
    OCC_LAUNCH_CHECK("occ_batch_colsum");
    return OCC_OK;
}
int occ_se_gate_fwd(const float* s, const float* W1, const float* W2, int64_t B, int64_t C, int64_t Cr, float* z, float* g, void* stream) {
    OCC_CHECK_ARG(s && W1 && W2 && z && g && B >= 1 && C >= 1 && Cr >= 1, "occ_se_gate_fwd: bad argument");
    hipLaunchKernelGGL(se_gate_fwd_kernel, dim3((unsigned)B), dim3(BT), (C + Cr) * sizeof(float), (hipStream_t)stream, s, W1, W2, (int)C, (int)Cr, z, g);
    OCC_LAUNCH_CHECK("occ_se_gate_fwd");
    return OCC_OK;
}
int occ_se_gate_bwd(const float* s, const float* z, const float* g, const float* dg, const float* W1, const float* W2, int64_t B, int64_t C, int64_t Cr,
                    float* dW1, float* dW2, float* ds, void* stream) {
    OCC_CHECK_ARG(s && z && g && dg && W1 && W2 && dW1 && dW2 && ds && B >= 1 && C >= 1 && Cr >= 1, "occ_se_gate_bwd: bad argument");
    hipLaunchKernelGGL(se_gate_bwd_kernel, dim3((unsigned)B), dim3(BT), 2 * (C + Cr) * sizeof(float), (hipStream_t)stream, s, z, g, dg, W1, W2, (int)C, (int)Cr, dW1,
                       dW2, ds);
    OCC_LAUNCH_CHECK("occ_se_gate_bwd");
    return OCC_OK;
}
int occ_se_scale_add_relu(const float* y, const float* gate, const float* res, const occ_rowmap* res_map, float* out, const occ_rowmap* out_map, int64_t B,
                          int64_t R, int64_t C, void* stream) {
    OCC_CHECK_ARG(y && gate && res && res_map && out && out_map && B >= 1 && R >= 1 && C >= 1, "occ_se_scale_add_relu: bad argument");
    hipLaunchKernelGGL(se_scale_add_relu_kernel, dim3(grid_for(B * R * C)), dim3(BT), 0, (hipStream_t)stream, y, gate, res, to_rowmap(*res_map), out,
                       to_rowmap(*out_map), (long long)(B * R), (int)R, (int)C);
    OCC_LAUNCH_CHECK("occ_se_scale_add_relu");
    return OCC_OK;
}
int occ_se_scale_add_relu_bwd(const float* dout, const occ_rowmap* dout_map, const float* out, const occ_rowmap* out_map, const float* y, const float* gate,
                              float* dy, float* dres, const occ_rowmap* dres_map, int dres_accumulate, float* dgate, int64_t B, int64_t R, int64_t C,
                              void* stream) {
    OCC_CHECK_ARG(dout && dout_map && out && out_map && y && gate && dy && dres && dres_map && dgate, "occ_se_scale_add_relu_bwd: null pointer");
    OCC_CHECK_ARG(B >= 1 && B < 65536 && R >= 1 && C >= 1 && C <= 256 && 256 % C == 0, "occ_se_scale_add_relu_bwd: bad shape");
    long long gx = occ_cdiv(R, 64);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(se_scale_add_relu_bwd_kernel, dim3((unsigned)gx, (unsigned)B), dim3(BT), 0, (hipStream_t)stream, dout, to_rowmap(*dout_map), out,
                       to_rowmap(*out_map), y, gate, dy, dres, to_rowmap(*dres_map), dres_accumulate, dgate, (int)R, (int)C);
    OCC_LAUNCH_CHECK("occ_se_scale_add_relu_bwd");
    return OCC_OK;
}
int occ_add_batch_vec(float* x, const occ_rowmap* x_map, const float* v, int64_t B, int64_t R, int64_t C, void* stream) {
    OCC_CHECK_ARG(x && x_map && v && B >= 1 && R >= 1 && C >= 1, "occ_add_batch_vec: bad argument");
    hipLaunchKernelGGL(add_batch_vec_kernel, dim3(grid_for(B * R * C)), dim3(BT), 0, (hipStream_t)stream, x, to_rowmap(*x_map), v, (long long)(B * R), (int)R, (int)C);
    OCC_LAUNCH_CHECK("occ_add_batch_vec");
    return OCC_OK;
}
int occ_conv7s2_dgrad_c1(const float* dy, const float* w, float* dx, int64_t B, int64_t H, int64_t W, void* stream) {
    OCC_CHECK_ARG(dy && w && dx && B >= 1 && H >= 1 && W >= 1 && (((uintptr_t)dy) & 15) == 0, "occ_conv7s2_dgrad_c1: bad argument");
    const int Ho = (int)((H + 6 - 7) / 2 + 1), Wo = (int)((W + 6 - 7) / 2 + 1);
    hipLaunchKernelGGL(conv7s2_dgrad_c1_kernel, dim3(grid_for(B * H * W)), dim3(BT), 0, (hipStream_t)stream, dy, w, dx, (int)B, (int)H, (int)W, Ho, Wo);
    OCC_LAUNCH_CHECK("occ_conv7s2_dgrad_c1");
    return OCC_OK;
}

}  // extern "C"
