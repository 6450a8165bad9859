// Development-only entry points (not part of include/yolo2_hip.h): time one convolution
// shape with a chosen kernel variant, on buffers allocated here.
#include <stdio.h>
#include <stdlib.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <vector>
#include "kernels.h"

namespace y2 {
hipError_t launch_conv_igemm_variant(int variant, const ConvArgs& a, hipStream_t s, int* bp, int* bc);
hipError_t launch_conv_halo_variant(int variant, const ConvArgs& a, hipStream_t s, int* bp);
hipError_t launch_wgrad9_variant(int variant, const WgradArgs& a, hipStream_t s);
hipError_t launch_wgrad_variant(int variant, const WgradArgs& a, hipStream_t s);
hipError_t launch_conv_haloq_variant(int variant, const ConvArgs& a, hipStream_t s, int* bp);
static hipError_t run_variant(int variant, const ConvArgs& a, hipStream_t s, int* bp, int* bc) {
    if (variant == 100) return launch_conv(1, a, s, bp);   // the product policy
    if (variant >= 118 && variant < 150 && a.taps == 9) return launch_conv_haloq_variant(variant, a, s, bp);
    const bool halo_id = (variant >= 23 && variant < 100) || (variant >= 110 && variant < 118);
    if (halo_id && a.taps == 9) return launch_conv_halo_variant(variant, a, s, bp);
    if (halo_id) variant = 0;
    return launch_conv_igemm_variant(variant, a, s, bp, bc);
}
}
namespace y2 { hipError_t rf_read_stamps(unsigned long long* dst); }
using namespace y2;

extern "C" __attribute__((visibility("default"))) int y2dev_rf_stamps(unsigned long long* dst) {
    return rf_read_stamps(dst) == hipSuccess ? 0 : -1;
}


extern "C" __attribute__((visibility("default"))) int y2dev_bench_wgrad(int N, int H, int W, int Cin, int Cout, int k, int variant, int splitk, int iters,
                                 float* ms_out) {
    const size_t sz = 2;
    const size_t pix = (size_t)N * (H + 1) * (W + 1) + 4 * (W + 3) + 2048;
    void *x = nullptr, *dy = nullptr;
    float* dw = nullptr;
    if (hipMalloc(&x, pix * Cin * sz) != hipSuccess) return -1;
    if (hipMalloc(&dy, pix * Cout * sz) != hipSuccess) return -1;
    if (hipMalloc(&dw, (size_t)k * k * Cin * Cout * 4) != hipSuccess) return -1;
    std::vector<unsigned short> hx(pix * Cin), hy(pix * Cout);
    unsigned int r = 777;
    for (auto& v : hx) { r = r * 1664525u + 1013904223u; v = (unsigned short)(((r >> 16) & 0x83FF) | 0x3800); }
    for (auto& v : hy) { r = r * 1664525u + 1013904223u; v = (unsigned short)(((r >> 16) & 0x83FF) | 0x2C00); }
    hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dy, hy.data(), hy.size() * 2, hipMemcpyHostToDevice);
    hipMemset(dw, 0, (size_t)k * k * Cin * Cout * 4);
    float* slab = nullptr;
    const size_t slab_floats = (size_t)1024 * 18432 * 2;
    if (!getenv("Y2DEV_NO_SLAB") && hipMalloc(&slab, slab_floats * 4) != hipSuccess) return -1;
    WgradArgs g{};
    g.slab = slab; g.slab_floats = slab ? slab_floats : 0;
    g.x = (char*)x + (size_t)(W + 3) * Cin * sz; g.dy = (char*)dy + (size_t)(W + 3) * Cout * sz; g.dW = dw;
    g.N = N; g.H = H; g.W = W; g.M = N * H * W; g.Cin = Cin; g.Cdy = Cout; g.Cout = Cout; g.taps = k * k;
    g.splitk = splitk; g.scale = 1.f;
    g.xcd = getenv("Y2_XCD_WGRAD") ? atoi(getenv("Y2_XCD_WGRAD")) : 1;
    auto run = [&]() {
        if (variant >= 100) return launch_wgrad_variant(variant, g, 0);
        return variant >= 2 ? launch_wgrad9_variant(variant, g, 0) : (variant == 1 ? launch_wgrad9(1, g, 0) : launch_wgrad(1, g, 0));
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i)
        if (run() != hipSuccess) return -2;
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) run();
    hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) return -3;
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    *ms_out = ms / iters;
    hipFree(x); hipFree(dy); hipFree(dw); if (slab) hipFree(slab);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return 0;
}

extern "C" __attribute__((visibility("default"))) int y2dev_bench_conv(int N, int H, int W, int Cin, int Cout, int k, int variant, int iters, float* ms_out) {
    const size_t sz = 2;
    const size_t xpix = (size_t)N * (H + 1) * (W + 1) + 4 * (W + 3) + 2048;
    const int taps = k * k;
    const int cout_pad = (Cout + 255) / 256 * 256;
    void *x = nullptr, *w = nullptr, *y = nullptr;
    float *bias = nullptr, *part = nullptr;
    // Y2DEV_BENCH_ROT=n: rotate over n input/output buffer sets (defeats the 256 MB Infinity Cache);
    // Y2DEV_BENCH_STATS=1: write the batch-norm partial records too
    const int rot = getenv("Y2DEV_BENCH_ROT") ? atoi(getenv("Y2DEV_BENCH_ROT")) : 1;
    const bool stats = getenv("Y2DEV_BENCH_STATS") != nullptr;
    const size_t xbytes = (xpix * Cin * sz + 255) / 256 * 256, ybytes = ((size_t)N * H * W * Cout * sz + 4096 + 255) / 256 * 256;
    if (hipMalloc(&x, xbytes * rot) != hipSuccess) return -1;
    if (hipMalloc(&w, (size_t)cout_pad * taps * Cin * sz) != hipSuccess) return -1;
    if (hipMalloc(&y, ybytes * rot) != hipSuccess) return -1;
    if (hipMalloc(&bias, Cout * 4) != hipSuccess) return -1;
    const size_t prow = (size_t)(N * H * W + 127) / 128;
    if (hipMalloc(&part, (prow * (2 * cout_pad + 1)) * 4) != hipSuccess) return -1;
    // pseudo-random f16 contents (finite, sign-varying)
    std::vector<unsigned short> hx(xpix * Cin), hw((size_t)cout_pad * taps * Cin);
    unsigned int r = 12345;
    for (auto& v : hx) { r = r * 1664525u + 1013904223u; v = (unsigned short)(((r >> 16) & 0x83FF) | 0x3800); }
    for (auto& v : hw) { r = r * 1664525u + 1013904223u; v = (unsigned short)(((r >> 16) & 0x83FF) | 0x2C00); }
    for (int i = 0; i < rot; ++i) hipMemcpy((char*)x + i * xbytes, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemset(bias, 0, Cout * 4);
    ConvArgs a{};
    if (stats) { a.part_cnt = part; a.part_mean = part + prow; a.part_m2 = part + prow * (1 + cout_pad); }
    a.x = (char*)x + (size_t)(W + 3) * Cin * sz; a.w = w; a.y = y; a.bias = bias;
    a.N = N; a.H = H; a.W = W; a.C = Cin; a.M = N * H * W; a.Cout = Cout; a.ldy = Cout; a.taps = taps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int bp, bc;
    for (int i = 0; i < 3; ++i)
        if (run_variant(variant, a, 0, &bp, &bc) != hipSuccess) return -2;
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) {
        a.x = (char*)x + (i % rot) * xbytes + (size_t)(W + 3) * Cin * sz;
        a.y = (char*)y + (i % rot) * ybytes;
        run_variant(variant, a, 0, &bp, &bc);
    }
    hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) return -3;
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    *ms_out = ms / iters;
    hipFree(x); hipFree(w); hipFree(y); hipFree(bias); hipFree(part);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return 0;
}
