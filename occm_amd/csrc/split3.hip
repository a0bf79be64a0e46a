// Split-operand copies for f32-grade products on the bf16 matrix cores (the accurate-and-fast arithmetic of the scoring path,
// oc_classifier.py:159-202, 243-265 run in batches): an f32 operand v is written as three bf16 K-panels so that ONE bf16 GEMM of depth 3K
// computes  xh.wh + xl.wh + xh.wl  (v = vh + vl, vh = bf16(v), vl = bf16(v - vh); the dropped xl.wl term is 2^-16 relative):
//     activations (mode 0):  [ xh | xl | xh ]          weights (mode 1):  [ wh | wh | wl ]
// so the LDS-DMA GEMM kernels (gemm_p8.hip and the 128 x 128 family) run it at their bf16 rate: an f32-grade product at a third of it.
#include "occ_common.h"

namespace {

__device__ __forceinline__ unsigned pk2(float a, float b) { return pack_bf16x2(a, b); }
__device__ __forceinline__ float lo_of(float v) { return v - bf16_bits_to_f32(f32_to_bf16_bits(v)); }

// one thread = 8 consecutive elements of a row: two float4 in, three uint4 out
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, RowMapI xmap, unsigned short* __restrict__ out, long long rows, int K, int mode) {
    const int kc = K >> 3;
    const long long n = rows * kc;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / kc; const int c = (int)(i - r * kc) * 8;
        const float* p = x + row_off(xmap, r) + c;
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        const uint4 hi = make_uint4(pk2(a.x, a.y), pk2(a.z, a.w), pk2(b.x, b.y), pk2(b.z, b.w));
        const uint4 lo = make_uint4(pk2(lo_of(a.x), lo_of(a.y)), pk2(lo_of(a.z), lo_of(a.w)), pk2(lo_of(b.x), lo_of(b.y)), pk2(lo_of(b.z), lo_of(b.w)));
        unsigned short* o = out + r * 3 * K + c;
        *reinterpret_cast<uint4*>(o) = hi;
        *reinterpret_cast<uint4*>(o + K) = mode == 0 ? lo : hi;
        *reinterpret_cast<uint4*>(o + 2 * K) = mode == 0 ? hi : lo;
    }
}

}  // namespace

extern "C" int occ_split3_bf16(const float* x, const occ_rowmap* x_map, void* out, int64_t rows, int64_t K, int mode, void* stream) {
    OCC_CHECK_ARG(x && x_map && out && rows >= 1 && K >= 8 && K % 8 == 0 && (mode == 0 || mode == 1), "occ_split3_bf16: bad argument (K %% 8 == 0, mode 0 / 1)");
    OCC_CHECK_ARG(x_map->rows_per_batch >= 1 && x_map->row_stride % 4 == 0 && x_map->batch_stride % 4 == 0 && x_map->line_stride % 4 == 0 &&
                  (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0, "occ_split3_bf16: rows must be 16-byte aligned");
    long long blocks = occ_cdiv(rows * (K / 8), 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, to_rowmap(*x_map), (unsigned short*)out, (long long)rows, (int)K, mode);
    OCC_LAUNCH_CHECK("occ_split3_bf16");
    return OCC_OK;
}
