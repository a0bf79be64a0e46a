// Diagnostic build of the eight-phase GEMM (NOT part of libocc_hip.so): the kernel source compiled with -DP8_DIAG stamps the shader
// clock at workgroup entry, after the prologue, after the K loop and at exit.  Prints, per shape, the median share of a workgroup's
// life spent in each part, and when workgroups start relative to the launch (rounds).  Read the SHARES, not the absolute run time.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DP8_DIAG -I occm_amd/csrc scripts/diag_p8.hip -o /tmp/diag_p8 && /tmp/diag_p8
#include "../occm_amd/csrc/gemm_p8.hip"
#include <algorithm>
#include <vector>
#include <cstdio>
void occ_set_error(const char*, ...) {}
using namespace occ_gemm_detail;
static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
int main() {
    struct S { const char* name; long long M, N, K; int gelu, resid, cbf; } shapes[] = {
        {"fc1 ", 12736, 4096, 1024, 1, 0, 1}, {"qkv ", 12736, 3072, 1024, 0, 0, 1}, {"out ", 12736, 1024, 1024, 0, 1, 0}, {"fc2 ", 12736, 1024, 4096, 0, 1, 0},
        {"sq4k", 4096, 4096, 4096, 0, 0, 1}, {"sq8k", 8192, 8192, 8192, 0, 0, 1}};
    for (auto& sh : shapes) {
        unsigned short *X, *W; char* C; float *bias, *R; unsigned long long* dg;
        hipMalloc(&X, sh.M * sh.K * 2); hipMalloc(&W, sh.N * sh.K * 2); hipMalloc(&C, sh.M * sh.N * 4); hipMalloc(&bias, sh.N * 4); hipMalloc(&R, sh.M * sh.N * 4);
        std::vector<unsigned short> hx(sh.M * sh.K), hw(sh.N * sh.K);
        unsigned s = 12345;
        for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3c00 + ((s >> 9) & 0x3ff) + ((s >> 31) << 15)); }
        for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3a00 + ((s >> 9) & 0x3ff) + ((s >> 31) << 15)); }
        hipMemcpy(X, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
        hipMemset(bias, 0, sh.N * 4); hipMemset(R, 0, sh.M * sh.N * 4);
        GemmArgs a{};
        a.M = sh.M; a.N = sh.N; a.K = sh.K; a.X = (const char*)X; a.xmap = occ_make_rowmap(sh.M, 0, sh.K, 0, 0); a.nseg = 1; a.seg_len = sh.K;
        a.W = (const char*)W; a.ldw = sh.K; a.bias = bias; a.R = sh.resid ? (const char*)R : nullptr; a.rmap = occ_make_rowmap(sh.M, 0, sh.N, 0, 0); a.r_dtype = OCC_F32;
        a.C = C; a.cmap = occ_make_rowmap(sh.M, 0, sh.N, 0, 0); a.c_dtype = sh.cbf ? OCC_BF16 : OCC_F32; a.act = sh.gelu ? OCC_ACT_GELU : OCC_ACT_NONE; a.alpha = 1.f;
        a.nbm = (int)((sh.M + 255) / 256); a.nbn = (int)((sh.N + 255) / 256); a.group_m = a.nbm >= 8 ? 8 : 0;
        const int nwg = a.nbm * a.nbn;
        hipMalloc(&dg, nwg * 6 * 8);
        a.diag = dg;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(gemm_p8_kernel<0>, dim3(nwg), dim3(512), 0, 0, a);
        hipEventRecord(e0);
        for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(gemm_p8_kernel<0>, dim3(nwg), dim3(512), 0, 0, a);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(nwg * 6);
        hipMemcpy(h.data(), dg, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> pro, loop, epi, life, clk, start;
        unsigned long long rmin = ~0ull, rmax = 0;
        for (int i = 0; i < nwg; ++i) { rmin = std::min(rmin, h[i * 6 + 4]); rmax = std::max(rmax, h[i * 6 + 5]); }
        for (int i = 0; i < nwg; ++i) {
            const unsigned long long* o = &h[i * 6];
            pro.push_back(double(o[1] - o[0])); loop.push_back(double(o[2] - o[1])); epi.push_back(double(o[3] - o[2])); life.push_back(double(o[3] - o[0]));
            clk.push_back(double(o[3] - o[0]) / (double(o[5] - o[4]) * 10.0));      // cycles per ns (realtime ticks at 100 MHz)
            start.push_back(double(o[4] - rmin) * 0.01);
        }
        std::sort(start.begin(), start.end());
        const double us = ms * 100.0, tf = 2.0 * sh.M * sh.N * sh.K / (us * 1e-6) / 1e12;
        printf("%s M=%lld N=%lld K=%lld: %.1f us/launch %.0f TF | wgs %d | per-wg cycles: prologue %.0f  loop %.0f (%.0f / K-tile)  epilogue %.0f  life %.0f | clock %.2f GHz | kernel span %.1f us | wg starts (us): p10 %.1f p50 %.1f p90 %.1f max %.1f\n",
               sh.name, sh.M, sh.N, sh.K, us, tf, nwg, med(pro), med(loop), med(loop) / (sh.K / 64), med(epi), med(life), med(clk), double(rmax - rmin) * 0.01,
               start[nwg / 10], start[nwg / 2], start[nwg * 9 / 10], start.back());
        hipFree(X); hipFree(W); hipFree(C); hipFree(bias); hipFree(R); hipFree(dg);
    }
    return 0;
}
