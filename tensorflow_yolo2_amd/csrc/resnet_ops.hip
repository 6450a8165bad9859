// Operators of the ResNet-50 backbone swap that the Darknet stacks do not have:
// the reference swaps slim's resnet_v1_50 in front of the YOLO grid head
// (src/yolo2_nets/tf_resnet.py:12-32, src/pascal/pascal_train_resnet.py:37-50).  These are graph-level
// operators on fp32 NHWC tensors (the 1x1 / 3x3 convolutions of the bottleneck units are the MFMA kernels behind
// y2_conv2d / y2_conv2d_backward; a stride-2 unit runs the stride-1 convolution and subsamples, which is slim's
// own definition of conv2d_same, resnet_utils.py:77-122):
//   * slim.batch_norm (decay 0.997, eps 1e-5, scale) + optional residual add + optional ReLU, forward (batch or
//     moving statistics) and backward            (resnet_utils.py:230-257, resnet_v1.py:99-112)
//   * subsample = max_pool2d([1,1], stride)        (resnet_utils.py:60-75)
//   * max_pool2d([3,3], stride 2, 'SAME')          (resnet_v1.py:198)
//   * conv2d_same(64, 7, stride 2) on the 3-channel image, forward + filter gradient (resnet_v1.py:197)
//   * bias + ReLU of slim.fully_connected, dropout(0.5) (pascal_train_resnet.py:41-46)
// Per-channel reductions (BN statistics, BN backward sums, the root filter gradient) are two-level: the rows are cut
// into slices so that ~2000 workgroups sweep the tensor at HBM rate, every workgroup leaves one double-precision
// partial per channel in a per-stream scratch buffer, and a finalize kernel adds the partials in a fixed order
// (deterministic run to run).  Round 2 had one workgroup per 32 channels sweep ALL rows: 2 workgroups on 256 CUs at
// 64 channels, 4.5 ms per BN backward, 352 of 375 ms of the batch-32 step.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>

#include <map>
#include <mutex>
#include <utility>

#include "../../include/yolo2_hip.h"
#include "common.h"
#include "kernels.h"

namespace y2 {
int set_error(int code, const char* msg);
}
using namespace y2;
static int rfail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return set_error(code, buf);
}
#define RCHK(expr)                                                                               \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) return rfail(Y2_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

static inline unsigned grid_for(size_t total, unsigned cap = 16384) {
    size_t nb = (total + 255) / 256;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return (unsigned)nb;
}

// per-(device, stream) scratch for the partial sums.  Grows on demand and NEVER frees or moves a buffer it has handed
// out: ResNet50Yolo(graph=True) bakes these pointers into a captured HIP graph, and stream handles come from a small
// round-robin pool, so another user of the same hipStream_t may ask for more later.  A larger request gets
// a NEW buffer; the old ones stay alive until the process ends (they are a few MB).  Growth during a stream capture
// is refused (hipMalloc is not capturable and the graph would keep the too-small pointer): the caller fails loudly.
namespace y2 {
// A null return leaves the reason in the library's error state (y2_last_error) and its code in op_scratch_error():
// callers return THAT code and do not overwrite the message ("scratch would have to grow during a stream
// capture" used to be replaced by a generic "no scratch memory").
static thread_local int g_scratch_err = Y2_OK;
int op_scratch_error() { return g_scratch_err != Y2_OK ? g_scratch_err : set_error(Y2_ERR_HIP, "operator scratch: allocation failed"); }
void* op_scratch(hipStream_t s, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, std::pair<void*, size_t>> pool;
    int dev = 0;
    g_scratch_err = Y2_OK;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        g_scratch_err = set_error(Y2_ERR_HIP, "operator scratch: no current device");
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(mu);
    auto& e = pool[std::make_pair(dev, s)];
    if (e.second < bytes) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
        if (cap != hipStreamCaptureStatusNone) {
            g_scratch_err = set_error(Y2_ERR_STATE, "operator scratch would have to grow during a stream capture: run one eager step first");
            return nullptr;
        }
        // generous first size: the largest request of the ResNet-50 swap at batch 32 is < 8 MB
        const size_t want = bytes < (16u << 20) ? (16u << 20) : bytes + bytes / 2;
        void* p = nullptr;
        if (hipMalloc(&p, want) != hipSuccess) {
            (void)hipGetLastError();
            g_scratch_err = set_error(Y2_ERR_HIP, "operator scratch: hipMalloc failed (out of device memory)");
            return nullptr;
        }
        e.first = p;          // the previous buffer (if any) is deliberately leaked: a graph may still replay into it
        e.second = want;
    }
    return e.first;
}
// per-(device, stream) counters of the in-kernel split-K sums of the op-level weight gradients (wgrad_finish.h): zeroed
// once when they are allocated, self-cleaning afterwards (every launch leaves them at zero)
int* op_counters(hipStream_t s, size_t ints) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, std::pair<int*, size_t>> pool;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    auto& e = pool[std::make_pair(dev, s)];
    if (e.second < ints) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
        if (cap != hipStreamCaptureStatusNone) return nullptr;      // the caller falls back to the separate sum kernel
        int* p = nullptr;
        if (hipMalloc((void**)&p, ints * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (hipMemset(p, 0, ints * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        e.first = p;
        e.second = ints;
    }
    return e.first;
}
}  // namespace y2

// ---------------------------------------------------------------------------
// batch norm over [M][C] fp32.  Partial sums: grid (channel groups of 32, row slices); 256 threads = RPP rows x 32
// channels per pass (VEC = 4: eight lanes x float4 per 128-byte row segment, 32 rows per pass; VEC = 1 for channel
// counts that are not multiples of 4).  Forward: shifted sums about the first row.  Backward: S1 = sum dz,
// S2 = sum dz * xhat with dz = dy * [y > 0].  part[0][slice][c], part[1][slice][c] in double.
// ---------------------------------------------------------------------------
struct RnSlices {
    int groups, slices;
    size_t rows_per_slice;
};
static RnSlices rn_slices(size_t M, int C, int vec) {
    const int rpp = 256 / (32 / vec);
    RnSlices r;
    r.groups = (C + 31) / 32;
    size_t want = 2048 / (size_t)r.groups;
    if (want < 1) want = 1;
    const size_t maxs = (M + (size_t)rpp * 4 - 1) / ((size_t)rpp * 4);   // at least four passes per workgroup
    if (want > maxs) want = maxs;
    size_t rps = (M + want - 1) / want;
    rps = (rps + rpp - 1) / rpp * rpp;
    r.rows_per_slice = rps;
    r.slices = (int)((M + rps - 1) / rps);
    return r;
}

template <int VEC, bool BWD>
__global__ __launch_bounds__(256) void rn_bn_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ y, size_t M, int C,
                                                            const float* __restrict__ mean, const float* __restrict__ var,
                                                            float eps, int relu, size_t rows_per_slice,
                                                            double* __restrict__ part) {
    constexpr int LPR = 32 / VEC, RPP = 256 / LPR;
    __shared__ double r1[RPP][33], r2[RPP][33];
    const int cl = (threadIdx.x % LPR) * VEC, row0 = threadIdx.x / LPR;
    const int c = blockIdx.x * 32 + cl;
    const bool cv = c < C;                       // VEC = 4 is only launched with C % 4 == 0
    size_t m0 = (size_t)blockIdx.y * rows_per_slice, m1 = m0 + rows_per_slice;
    if (m1 > M) m1 = M;
    double s1[VEC], s2[VEC];
    float a[VEC], b[VEC];                        // forward: pivot, -; backward: mean, invstd
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        s1[v] = 0.0;
        s2[v] = 0.0;
        a[v] = 0.f;
        b[v] = 1.f;
        if (cv) {
            if (BWD) { a[v] = mean[c + v]; b[v] = 1.0f / sqrtf(var[c + v] + eps); }
            else a[v] = x[c + v];
        }
    }
    if (cv) {
#pragma unroll 4
        for (size_t m = m0 + row0; m < m1; m += RPP) {
            const size_t i = m * C + c;
            float xv[VEC], dv[VEC], yv[VEC];
            if constexpr (VEC == 4) {
                const float4 t = *(const float4*)(x + i);
                xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
                if constexpr (BWD) {
                    const float4 d = *(const float4*)(dy + i);
                    dv[0] = d.x; dv[1] = d.y; dv[2] = d.z; dv[3] = d.w;
                    if (relu) {
                        const float4 q = *(const float4*)(y + i);
                        yv[0] = q.x; yv[1] = q.y; yv[2] = q.z; yv[3] = q.w;
                    }
                }
            } else {
                xv[0] = x[i];
                if constexpr (BWD) {
                    dv[0] = dy[i];
                    if (relu) yv[0] = y[i];
                }
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                if constexpr (BWD) {
                    const float dz = (relu && !(yv[v] > 0.f)) ? 0.f : dv[v];
                    s1[v] += (double)dz;
                    s2[v] += (double)dz * (double)((xv[v] - a[v]) * b[v]);
                } else {
                    const double d = (double)(xv[v] - a[v]);
                    s1[v] += d;
                    s2[v] += d * d;
                }
            }
        }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        r1[row0][cl + v] = s1[v];
        r2[row0][cl + v] = s2[v];
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        const int cc = blockIdx.x * 32 + threadIdx.x;
        if (cc < C) {
            double p = 0.0, q = 0.0;
            for (int k = 0; k < RPP; ++k) { p += r1[k][threadIdx.x]; q += r2[k][threadIdx.x]; }
            const size_t o = (size_t)blockIdx.y * C + cc;
            part[o] = p;
            part[(size_t)gridDim.y * C + o] = q;
        }
    }
}
// one workgroup per 32 channels: eight slice lanes per channel add every eighth slice, then one thread adds the
// eight sums -- the same order every run (a single thread walking 1024 slices was latency-bound: 50 us)
template <bool BWD>
__global__ __launch_bounds__(256) void rn_bn_finalize_kernel(const double* __restrict__ part, int slices, int C, size_t M,
                                                             const float* __restrict__ x, float* o1, float* o2,
                                                             float* mm, float* mv, float decay) {
    __shared__ double r1[8][33], r2[8][33];
    const int cl = threadIdx.x & 31, k0 = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double p = 0.0, q = 0.0;
    if (c < C) {
#pragma unroll 4
        for (int k = k0; k < slices; k += 8) {
            p += part[(size_t)k * C + c];
            q += part[((size_t)slices + k) * C + c];
        }
    }
    r1[k0][cl] = p;
    r2[k0][cl] = q;
    __syncthreads();
    if (k0 != 0 || c >= C) return;
    p = 0.0;
    q = 0.0;
    for (int k = 0; k < 8; ++k) { p += r1[k][cl]; q += r2[k][cl]; }
    if (BWD) {
        o2[c] = (float)p;     // dbeta
        o1[c] = (float)q;     // dgamma
    } else {
        const double md = p / (double)M;
        o1[c] = (float)((double)x[c] + md);
        const double v = q / (double)M - md * md;
        o2[c] = (float)(v > 0 ? v : 0);
        if (mm) {      // moving <- decay moving + (1 - decay) batch, with the values as stored (biased variance, see below)
            mm[c] = decay * mm[c] + (1.0f - decay) * o1[c];
            mv[c] = decay * mv[c] + (1.0f - decay) * o2[c];
        }
    }
}
template <bool BWD>
static hipError_t rn_bn_reduce(const float* x, const float* dy, const float* y, size_t M, int C, const float* mean,
                               const float* var, float eps, int relu, float* o1, float* o2, hipStream_t s,
                               float* mm = nullptr, float* mv = nullptr, float decay = 0.f) {
    const uintptr_t al = (uintptr_t)x | (uintptr_t)dy | (uintptr_t)y;
    const int vec = (C % 4 == 0 && al % 16 == 0) ? 4 : 1;
    const RnSlices sl = rn_slices(M, C, vec);
    double* part = (double*)op_scratch(s, (size_t)2 * sl.slices * C * sizeof(double));
    if (!part) return hipErrorOutOfMemory;
    const dim3 grid(sl.groups, sl.slices);
    if (vec == 4)
        hipLaunchKernelGGL((rn_bn_partial_kernel<4, BWD>), grid, dim3(256), 0, s, x, dy, y, M, C, mean, var, eps, relu,
                           sl.rows_per_slice, part);
    else
        hipLaunchKernelGGL((rn_bn_partial_kernel<1, BWD>), grid, dim3(256), 0, s, x, dy, y, M, C, mean, var, eps, relu,
                           sl.rows_per_slice, part);
    hipLaunchKernelGGL((rn_bn_finalize_kernel<BWD>), dim3(sl.groups), dim3(256), 0, s, part, sl.slices, C, M, x, o1, o2, mm,
                       mv, decay);
    return hipGetLastError();
}

// y = act(gamma (x - mean) invstd + beta + res); moving <- decay moving + (1 - decay) batch when update
__global__ void rn_bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ res, float* __restrict__ y,
                                   size_t total, int C, const float* mean, const float* var, const float* gamma,
                                   const float* beta, float eps, int relu) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const float inv = 1.0f / sqrtf(var[c] + eps);
        float v = (x[i] - mean[c]) * inv * gamma[c] + beta[c];
        if (res) v += res[i];
        y[i] = relu ? fmaxf(v, 0.f) : v;
    }
}
// four channels per thread (C % 4 == 0, 16-byte aligned tensors).  The launcher makes the grid stride a multiple of
// C / 4 where it can (FIXED): a thread then stays on its four channels and derives their coefficients once
// (1 / sqrt and the products per element were as expensive as the memory traffic)
template <bool FIXED>
__global__ void rn_bn_apply4_kernel(const float* __restrict__ x, const float* __restrict__ res, float* __restrict__ y,
                                    size_t total4, int C, const float* mean, const float* var, const float* gamma,
                                    const float* beta, float eps, int relu) {
    float m[4], a[4], b[4];
    auto coef = [&](int c) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            m[e] = mean[c + e];
            a[e] = 1.0f / sqrtf(var[c + e] + eps);
            b[e] = beta[c + e];
        }
    };
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (FIXED && i0 < total4) coef((int)((i0 * 4) % C));
    float g[4];
    if (FIXED && i0 < total4) {
        const int c = (int)((i0 * 4) % C);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = gamma[c + e];
    }
    for (size_t i = i0; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        if (!FIXED) {
            const int c = (int)((i * 4) % C);
            coef(c);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = gamma[c + e];
        }
        const float4 xv = ((const float4*)x)[i];
        float o[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (o[e] - m[e]) * a[e] * g[e] + b[e];
        if (res) {
            const float4 r = ((const float4*)res)[i];
            o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w;
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
        }
        ((float4*)y)[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// dx = gamma invstd (dz - dbeta/M - xhat dgamma/M)   (training)   |   gamma invstd dz   (moving statistics)
// dres = dz (the residual branch enters before the activation)
__global__ void rn_bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                       const float* __restrict__ x, float* __restrict__ dx, float* __restrict__ dres,
                                       size_t total, size_t M, int C, const float* mean, const float* var,
                                       const float* gamma, float eps, int relu, int training, const float* dgamma,
                                       const float* dbeta) {
    const float invM = 1.0f / (float)M;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const float inv = 1.0f / sqrtf(var[c] + eps);
        const float dz = (relu && !(y[i] > 0.f)) ? 0.f : dy[i];
        if (dres) dres[i] = dz;
        const float xh = (x[i] - mean[c]) * inv;
        dx[i] = training ? gamma[c] * inv * (dz - dbeta[c] * invM - xh * dgamma[c] * invM) : gamma[c] * inv * dz;
    }
}

template <bool FIXED>
__global__ void rn_bn_bwd_apply4_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                        const float* __restrict__ x, float* __restrict__ dx, float* __restrict__ dres,
                                        size_t total4, size_t M, int C, const float* mean, const float* var,
                                        const float* gamma, float eps, int relu, int training, const float* dgamma,
                                        const float* dbeta) {
    const float invM = 1.0f / (float)M;
    float m[4], inv[4], gi[4], kb[4], kg[4];
    auto coef = [&](int c) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            m[e] = mean[c + e];
            inv[e] = 1.0f / sqrtf(var[c + e] + eps);
            gi[e] = gamma[c + e] * inv[e];
            kb[e] = dbeta[c + e] * invM;
            kg[e] = dgamma[c + e] * invM;
        }
    };
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (FIXED && i0 < total4) coef((int)((i0 * 4) % C));
    for (size_t i = i0; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        if (!FIXED) coef((int)((i * 4) % C));
        const float4 d4 = ((const float4*)dy)[i], x4 = ((const float4*)x)[i];
        float dzv[4] = {d4.x, d4.y, d4.z, d4.w};
        const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
        if (relu) {
            const float4 y4 = ((const float4*)y)[i];
            const float yv[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (!(yv[e] > 0.f)) dzv[e] = 0.f;
        }
        if (dres) ((float4*)dres)[i] = make_float4(dzv[0], dzv[1], dzv[2], dzv[3]);
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (xv[e] - m[e]) * inv[e];
            o[e] = training ? gi[e] * (dzv[e] - kb[e] - xh * kg[e]) : gi[e] * dzv[e];
        }
        ((float4*)dx)[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// ---------------------------------------------------------------------------
__global__ void rn_subsample_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C, int s,
                                    int forward) {
    const int Ho = (H + s - 1) / s, Wo = (W + s - 1) / s;
    const size_t total = forward ? (size_t)N * Ho * Wo * C : (size_t)N * H * W * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t p = i / C;
        if (forward) {
            const int wo = (int)(p % Wo), ho = (int)((p / Wo) % Ho), n = (int)(p / ((size_t)Wo * Ho));
            y[i] = x[(((size_t)n * H + ho * s) * W + wo * s) * C + c];
        } else {   // x = gradient at the coarse grid [N,Ho,Wo,C], y = gradient at the fine grid
            const int w = (int)(p % W), h = (int)((p / W) % H), n = (int)(p / ((size_t)W * H));
            y[i] = (h % s == 0 && w % s == 0) ? x[(((size_t)n * Ho + h / s) * Wo + w / s) * C + c] : 0.f;
        }
    }
}

// max_pool2d 3x3 stride 2 'SAME' (TF: pad_total = max((Ho-1)*2 + 3 - H, 0), pad_beg = pad_total / 2)
__global__ void rn_maxpool3_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int ph = ((Ho - 1) * 2 + 3 - H > 0 ? (Ho - 1) * 2 + 3 - H : 0) / 2;
    const int pw = ((Wo - 1) * 2 + 3 - W > 0 ? (Wo - 1) * 2 + 3 - W : 0) / 2;
    const size_t total = (size_t)N * Ho * Wo * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t p = i / C;
        const int wo = (int)(p % Wo), ho = (int)((p / Wo) % Ho), n = (int)(p / ((size_t)Wo * Ho));
        float best = -INFINITY;
        size_t arg = 0;
        for (int dh = 0; dh < 3; ++dh)
            for (int dw = 0; dw < 3; ++dw) {
                const int h = ho * 2 + dh - ph, w = wo * 2 + dw - pw;
                if (h < 0 || h >= H || w < 0 || w >= W) continue;
                const size_t j = (((size_t)n * H + h) * W + w) * C + c;
                if (x[j] > best) { best = x[j]; arg = j; }     // first maximum in row-major window order
            }
        if (y) y[i] = best;
    }
}
// backward as a GATHER in two passes (the windows overlap, stride 2 < 3: the scatter form added into dx with float
// atomics, the one order-dependent sum left in the ResNet swap).  Pass 1: every window's first maximum in row-major
// order, as the forward pass finds it, as a byte 3 * dh + dw.  Pass 2: every input element looks at the <= 4 windows
// that contain it and takes dy where the window's byte names it; contributions added in window order.  No pre-zeroed dx.
__global__ void rn_maxpool3_arg_kernel(const float* __restrict__ x, unsigned char* __restrict__ arg, int N, int H, int W, int C) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int ph = ((Ho - 1) * 2 + 3 - H > 0 ? (Ho - 1) * 2 + 3 - H : 0) / 2;
    const int pw = ((Wo - 1) * 2 + 3 - W > 0 ? (Wo - 1) * 2 + 3 - W : 0) / 2;
    const size_t total = (size_t)N * Ho * Wo * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t p = i / C;
        const int wo = (int)(p % Wo), ho = (int)((p / Wo) % Ho), n = (int)(p / ((size_t)Wo * Ho));
        float best = -INFINITY;
        int a = 0;
        for (int dh = 0; dh < 3; ++dh)
            for (int dw = 0; dw < 3; ++dw) {
                const int h = ho * 2 + dh - ph, w = wo * 2 + dw - pw;
                if (h < 0 || h >= H || w < 0 || w >= W) continue;
                const float v = x[(((size_t)n * H + h) * W + w) * C + c];
                if (v > best) { best = v; a = 3 * dh + dw; }
            }
        arg[i] = (unsigned char)a;
    }
}
__global__ void rn_maxpool3_bwd_kernel(const unsigned char* __restrict__ arg, const float* __restrict__ dy,
                                       float* __restrict__ dx, int N, int H, int W, int C) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int ph = ((Ho - 1) * 2 + 3 - H > 0 ? (Ho - 1) * 2 + 3 - H : 0) / 2;
    const int pw = ((Wo - 1) * 2 + 3 - W > 0 ? (Wo - 1) * 2 + 3 - W : 0) / 2;
    const size_t total = (size_t)N * H * W * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t p = i / C;
        const int w = (int)(p % W), h = (int)((p / W) % H), n = (int)(p / ((size_t)W * H));
        float g = 0.f;
        const int ho_lo = (h + ph - 2 + 1) >> 1, ho_hi = (h + ph) >> 1;     // ceil((h + ph - 2) / 2) .. floor((h + ph) / 2)
        const int wo_lo = (w + pw - 2 + 1) >> 1, wo_hi = (w + pw) >> 1;
        for (int ho = ho_lo < 0 ? 0 : ho_lo; ho <= ho_hi && ho < Ho; ++ho)
            for (int wo = wo_lo < 0 ? 0 : wo_lo; wo <= wo_hi && wo < Wo; ++wo) {
                const size_t o = (((size_t)n * Ho + ho) * Wo + wo) * C + c;
                const int a = arg[o];
                if (ho * 2 + a / 3 - ph == h && wo * 2 + a % 3 - pw == w) g += dy[o];
            }
        dx[i] = g;
    }
}

// ---------------------------------------------------------------------------
// root convolution: conv2d_same(64, 7, stride 2) on [N,H,W,3]: pad 3 before / 3 after, VALID stride 2
// (resnet_utils.py:77-122 -- explicit padding, not TF 'SAME').  One workgroup per output row (n, ho), persistent over
// rows: the seven zero-padded input rows of that output row sit in LDS ([7][(W + 6) * 3] floats, 19 KB at 224), a
// lane owns one filter (co), and the 21 (kw, c) taps of one filter row are CONTIGUOUS in the staged row -- the patch
// reads are wave-uniform (LDS broadcast, no bank conflicts).
//   forward: wave = slice of the output columns, all 147 taps of filter co in registers;
//   filter gradient: wave = kh, 21 accumulators per lane, dy[p][co] read coalesced; every workgroup leaves its partial
//   dW in the scratch buffer and a second kernel adds the partials in order (deterministic).
// ---------------------------------------------------------------------------
constexpr int kC7Threads = 448;   // seven waves

constexpr int kC7Pad = 16;        // zero columns behind a staged row: a group of eight output pixels may overhang
Y2_DEV void rn_conv7_stage(const float* __restrict__ x, float* xs, int n, int ho, int H, int W) {
    const int rowf = (W + 6 + kC7Pad) * 3;
    for (int kh = 0; kh < 7; ++kh) {
        const int h = ho * 2 + kh - 3;
        const bool hv = h >= 0 && h < H;
        const float* src = x + ((size_t)n * H + (hv ? h : 0)) * W * 3;
        float* dst = xs + kh * rowf;
        for (int i = threadIdx.x; i < rowf; i += kC7Threads) {
            const int j = i - 9;
            dst[i] = (hv && j >= 0 && j < W * 3) ? src[j] : 0.f;
        }
    }
}

// forward: one output pixel per wave and iteration, the 147 taps of filter co in registers.  Bound by the LDS issue rate
// (147 wave-uniform reads per 147 FMAs: 0.38 ms at batch 32); groups of 4 / 8 pixels that read each input column
// once spill the filter registers (616 / 1292 bytes per lane) and are not faster.
__global__ __launch_bounds__(kC7Threads) void rn_conv7_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                  float* __restrict__ y, int N, int H, int W, int Co) {
    extern __shared__ float xs[];
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;     // (H + 6 - 7) / 2 + 1
    const int rowf = (W + 6 + kC7Pad) * 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int co0 = 0; co0 < Co; co0 += 64) {
        const int co = co0 + lane;
        float wr[147];
#pragma unroll
        for (int t = 0; t < 147; ++t) wr[t] = co < Co ? w[(size_t)t * Co + co] : 0.f;
        for (int row = blockIdx.x; row < N * Ho; row += gridDim.x) {
            const int n = row / Ho, ho = row % Ho;
            __syncthreads();
            rn_conv7_stage(x, xs, n, ho, H, W);
            __syncthreads();
            for (int wo = wave; wo < Wo; wo += 7) {
                float acc = 0.f;
#pragma unroll
                for (int kh = 0; kh < 7; ++kh) {
                    const float* xp = xs + kh * rowf + wo * 6;
#pragma unroll
                    for (int j = 0; j < 21; ++j) acc = fmaf(xp[j], wr[kh * 21 + j], acc);
                }
                if (co < Co) y[((size_t)row * Wo + wo) * Co + co] = acc;
            }
        }
    }
}

__global__ __launch_bounds__(kC7Threads) void rn_conv7_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                    float* __restrict__ part, int N, int H, int W, int Co) {
    extern __shared__ float xs[];
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int rowf = (W + 6 + kC7Pad) * 3;
    const int co = threadIdx.x & 63, kh = threadIdx.x >> 6;
    float acc[21];
#pragma unroll
    for (int j = 0; j < 21; ++j) acc[j] = 0.f;
    for (int row = blockIdx.x; row < N * Ho; row += gridDim.x) {
        const int n = row / Ho, ho = row % Ho;
        __syncthreads();
        rn_conv7_stage(x, xs, n, ho, H, W);
        __syncthreads();
        const float* dyr = dy + (size_t)row * Wo * Co + co;
        const float* xr = xs + kh * rowf;
#pragma unroll 4
        for (int wo = 0; wo < Wo; ++wo) {
            const float d = co < Co ? dyr[(size_t)wo * Co] : 0.f;
            const float* xp = xr + wo * 6;
#pragma unroll
            for (int j = 0; j < 21; ++j) acc[j] = fmaf(xp[j], d, acc[j]);
        }
    }
    if (co < Co) {
        float* o = part + (size_t)blockIdx.x * 147 * Co;
#pragma unroll
        for (int j = 0; j < 21; ++j) o[(size_t)(kh * 21 + j) * Co + co] = acc[j];
    }
}
// out[i] = sum over the partials in a fixed order: 64 outputs x 4 partial lanes per workgroup (one thread walking 512
// partials was latency-bound: 120 us)
__global__ __launch_bounds__(256) void rn_sum_partials_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                              int nparts, int n) {
    __shared__ float red[4][64];
    const int il = threadIdx.x & 63, k0 = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + il;
    float s = 0.f;
    if (i < n) {
#pragma unroll 4
        for (int k = k0; k < nparts; k += 4) s += part[(size_t)k * n + i];
    }
    red[k0][il] = s;
    __syncthreads();
    if (k0 == 0 && i < n) out[i] = (red[0][il] + red[1][il]) + (red[2][il] + red[3][il]);
}

// ---------------------------------------------------------------------------
// fully connected tail: y = relu(x + b) in place form and its backward; dropout
// ---------------------------------------------------------------------------
__global__ void rn_bias_relu_kernel(float* __restrict__ y, const float* __restrict__ b, size_t total, int C, int relu) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float v = y[i] + b[i % C];
        y[i] = relu ? fmaxf(v, 0.f) : v;
    }
}
// dz = dy * [y > 0]; db = column sums of dz (one block per 256 columns, rows swept)
__global__ void rn_bias_relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dz,
                                        float* __restrict__ db, size_t M, int C, int relu) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (size_t m = 0; m < M; ++m) {
        const size_t i = m * C + c;
        const float v = (relu && !(y[i] > 0.f)) ? 0.f : dy[i];
        dz[i] = v;
        s += v;
    }
    db[c] = s;
}
Y2_DEV uint64_t rn_mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
// tf.nn.dropout(x, keep_prob): kept elements are scaled by 1 / keep_prob; the mask is a pure function of
// (seed, element index), so the backward pass regenerates it
__global__ void rn_dropout_kernel(const float* __restrict__ x, float* __restrict__ y, size_t total, float keep,
                                  uint64_t seed, const uint64_t* __restrict__ seed_dev) {
    if (seed_dev) seed = *seed_dev;     // the seed lives in device memory when the step is replayed from a HIP graph
    const float inv = 1.0f / keep;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float u = (float)(rn_mix64(seed * 0xD1B54A32D192ED03ull + i) >> 40) * (1.0f / 16777216.0f);
        y[i] = u < keep ? x[i] * inv : 0.f;
    }
}

template <typename... P>
static bool rn_vec4(int channels, P... ptrs) {
    uintptr_t al = 0;
    for (const void* q : {(const void*)ptrs...}) al |= (uintptr_t)q;
    return channels % 4 == 0 && al % 16 == 0;
}
// grid of the four-channel kernels: at least four elements per thread where the tensor allows, and a stride
// (grid * 256) that is a multiple of C / 4 so that a thread keeps its channels (*fixed)
static unsigned rn_grid4(size_t total4, int channels, bool* fixed) {
    size_t nb = (total4 + 1023) / 1024;
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    const size_t q = (size_t)channels / 4;
    *fixed = false;
    for (size_t t = nb; t >= 1 && t + 8 > nb; --t)       // a nearby grid whose stride divides
        if ((t * 256) % q == 0) { nb = t; *fixed = true; break; }
    return (unsigned)nb;
}
static void rn_bn_apply(const float* x, const float* res, float* y, size_t total, int channels, const float* mean,
                        const float* var, const float* gamma, const float* beta, float eps, int relu, hipStream_t s) {
    if (rn_vec4(channels, x, res, y)) {
        bool fixed;
        const unsigned nb = rn_grid4(total / 4, channels, &fixed);
        if (fixed)
            hipLaunchKernelGGL(rn_bn_apply4_kernel<true>, dim3(nb), dim3(256), 0, s, x, res, y, total / 4, channels, mean, var,
                               gamma, beta, eps, relu);
        else
            hipLaunchKernelGGL(rn_bn_apply4_kernel<false>, dim3(nb), dim3(256), 0, s, x, res, y, total / 4, channels, mean, var,
                               gamma, beta, eps, relu);
    } else
        hipLaunchKernelGGL(rn_bn_apply_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, res, y, total, channels, mean, var,
                           gamma, beta, eps, relu);
}

// The join's backward between LINKED stacks (y2_link, round 5): g = (d1 + d2) * [out > 0] in the arithmetic type T.
// out: the unit's output as the consumer stack holds it (bordered [N][H+1][W+1][C], cell 0); d1 [M][C] of T: the input
// gradient of the main branch of the unit above; d2 [M][C]: the other addend -- of T (that unit's shortcut branch or its
// own g) or fp32 (d2_f32: the g of a run's top unit, which still arrives through the fp32 operators); g [M][C] of T.
// Eight channels per thread.
template <typename T>
__global__ void rn_join_bwd_t_kernel(const T* __restrict__ outb, const T* __restrict__ d1, const void* __restrict__ d2,
                                     int d2_f32, T* __restrict__ g, int N, int H, int W, int C) {
    const int cg = C / 8;
    const size_t total = (size_t)N * H * W * cg;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cg) * 8;
        const size_t m = i / cg;
        const int w = (int)(m % W), h = (int)((m / W) % H), n = (int)(m / ((size_t)W * H));
        const y2::Chunk<T> o = y2::ld_chunk<T>(outb + y2::bpix(n, h, w, H, W) * C + c);
        const y2::Chunk<T> a = y2::ld_chunk<T>(d1 + m * C + c);
        float b[8];
        if (d2_f32) {
            const float4 b0 = *(const float4*)((const float*)d2 + m * C + c), b1 = *(const float4*)((const float*)d2 + m * C + c + 4);
            b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
        } else {
            const y2::Chunk<T> bt = y2::ld_chunk<T>((const T*)d2 + m * C + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) b[e] = y2::Elem<T>::to_f32(bt.v[e]);
        }
        y2::Chunk<T> r;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            r.v[e] = y2::Elem<T>::from_f32(y2::Elem<T>::to_f32(o.v[e]) > 0.f ? y2::Elem<T>::to_f32(a.v[e]) + b[e] : 0.f);
        y2::st_chunk<T>(g + m * C + c, r);
    }
}
// The same join one level down a STRIDE-2 unit (y2_join_backward_s2): the second addend is the gradient of the unit's
// identity shortcut subsample(x) (resnet_v1.py:99-101) and lives on the unit's OUTPUT grid [N][H/2][W/2][C]: it reaches
// the even rows / columns of this [N][H][W] grid, the other positions take d1 alone.
template <typename T>
__global__ void rn_join_bwd_s2_kernel(const T* __restrict__ outb, const T* __restrict__ d1, const void* __restrict__ d2,
                                      int d2_f32, T* __restrict__ g, int N, int H, int W, int C) {
    const int cg = C / 8, Hs = H / 2, Ws = W / 2;
    const size_t total = (size_t)N * H * W * cg;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cg) * 8;
        const size_t m = i / cg;
        const int w = (int)(m % W), h = (int)((m / W) % H), n = (int)(m / ((size_t)W * H));
        const y2::Chunk<T> o = y2::ld_chunk<T>(outb + y2::bpix(n, h, w, H, W) * C + c);
        const y2::Chunk<T> a = y2::ld_chunk<T>(d1 + m * C + c);
        float b[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (!(h & 1) && !(w & 1)) {
            const size_t ms = ((size_t)n * Hs + (h >> 1)) * Ws + (w >> 1);
            if (d2_f32) {
                const float4 b0 = *(const float4*)((const float*)d2 + ms * C + c), b1 = *(const float4*)((const float*)d2 + ms * C + c + 4);
                b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
            } else {
                const y2::Chunk<T> bt = y2::ld_chunk<T>((const T*)d2 + ms * C + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) b[e] = y2::Elem<T>::to_f32(bt.v[e]);
            }
        }
        y2::Chunk<T> r;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            r.v[e] = y2::Elem<T>::from_f32(y2::Elem<T>::to_f32(o.v[e]) > 0.f ? y2::Elem<T>::to_f32(a.v[e]) + b[e] : 0.f);
        y2::st_chunk<T>(g + m * C + c, r);
    }
}
// resnet_utils.subsample(x, 2) between two bordered tensors of the arithmetic type (y2_subsample_bordered): the identity
// shortcut of a stride-2 unit inside a linked run; 16-byte chunks (8 channels)
template <typename T>
__global__ void rn_subsample_bordered_kernel(const T* __restrict__ src, T* __restrict__ dst, int N, int H, int W, int C) {
    const int cg = C / 8, Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)N * Ho * Wo * cg;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cg) * 8;
        const size_t m = i / cg;
        const int wo = (int)(m % Wo), ho = (int)((m / Wo) % Ho), n = (int)(m / ((size_t)Wo * Ho));
        y2::st_chunk<T>(dst + y2::bpix(n, ho, wo, Ho, Wo) * C + c, y2::ld_chunk<T>(src + y2::bpix(n, 2 * ho, 2 * wo, H, W) * C + c));
    }
}
// ---------------------------------------------------------------------------
// Round 5: the root convolution on the matrix pipe in the half-precision modes (y2_conv7x7s2_t).  The scalar kernel above
// is bound by its LDS issue rate (147 wave-uniform reads per 147 FMAs: 0.40 ms at batch 32, 3.6 % of the ResNet step).
// Here the seven input rows of an output row are staged as T with FOUR channels per pixel (8 bytes: the window of output
// pixel wo starts at byte 16 wo, every fragment read is one aligned ds_read_b128), a filter row is 7 taps x 4 channels
// + one zero tap = 32 k-slots = two 32x32x16 steps, D[cout][pixel], fp32 accumulation and output; the filters are packed
// in fragment order by a one-block kernel in front ([cout tile][kh][k-group][lane][8]).  A wave takes 32 output pixels x
// 64 filters (28 matrix instructions) and writes its tile through a [pixel][cout] patch as whole 256-byte rows.
// ---------------------------------------------------------------------------
constexpr int kC7mThreads = 256;
template <typename T>
__global__ void rn_conv7_pack_kernel(const float* __restrict__ w, T* __restrict__ wp, int Co, int total) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int lane = idx & 63, g = (idx >> 6) & 1, kh = (idx >> 7) % 7, ct = idx / (128 * 7);
    const int co = ct * 32 + (lane & 31), hh = lane >> 5;
    T o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 16 * g + 8 * hh + j, kw = k >> 2, c = k & 3;
        o[j] = Elem<T>::from_f32((kw < 7 && c < 3 && co < Co) ? w[(size_t)((kh * 7 + kw) * 3 + c) * Co + co] : 0.f);
    }
    *(u32x4*)(wp + (size_t)idx * 8) = *(const u32x4*)o;
}
template <typename T>
__global__ __launch_bounds__(kC7mThreads) void rn_conv7_mfma_kernel(const float* __restrict__ x, const T* __restrict__ wp,
                                                                    float* __restrict__ y, int N, int H, int W, int Co, int PW) {
    typedef typename Elem<T>::frag frag_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PROW = 64 + 4;                     // floats per patch row (16-byte pad)
    char* const rows = smem;                         // [7][PW] pixels of 4 T
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, hh = lane >> 5;
    float* const patch = (float*)(smem + (size_t)7 * PW * 8) + wave * 32 * PROW;
    frag_t fw[2][7][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int kh = 0; kh < 7; ++kh)
#pragma unroll
            for (int g = 0; g < 2; ++g) fw[ct][kh][g] = *(const frag_t*)(wp + ((size_t)((ct * 7 + kh) * 2 + g) * 64 + lane) * 8);
    for (int row = blockIdx.x; row < N * Ho; row += gridDim.x) {
        const int n = row / Ho, ho = row - n * Ho;
        __syncthreads();
        for (int i = tid; i < 7 * PW; i += kC7mThreads) {
            const int kh = i / PW, pc = i - kh * PW;
            const int hi = 2 * ho + kh - 3, wi = pc - 3;
            T o[4] = {Elem<T>::from_f32(0.f), Elem<T>::from_f32(0.f), Elem<T>::from_f32(0.f), Elem<T>::from_f32(0.f)};
            if (hi >= 0 && hi < H && wi >= 0 && wi < W) {
                const float* px = x + ((size_t)(n * H + hi) * W + wi) * 3;
                o[0] = Elem<T>::from_f32(px[0]); o[1] = Elem<T>::from_f32(px[1]); o[2] = Elem<T>::from_f32(px[2]);
            }
            *(u32x2*)(rows + (size_t)i * 8) = *(const u32x2*)o;
        }
        __syncthreads();
        for (int grp = wave; grp * 32 < Wo; grp += kC7mThreads / 64) {
            const int wo = grp * 32 + r32;
            f32x16 acc[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[ct][q] = 0.f;
            const char* win = rows + (size_t)wo * 16 + 16 * hh;        // the window's first pixel is input column 2 wo (padded)
#pragma unroll
            for (int kh = 0; kh < 7; ++kh)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const frag_t fb = *(const frag_t*)(win + (size_t)kh * PW * 8 + 32 * g);
                    mma32(acc[0], fw[0][kh][g], fb);
                    mma32(acc[1], fw[1][kh][g], fb);
                }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
                    *(f32x4*)(patch + r32 * PROW + ct * 32 + 8 * q4 + 4 * hh) =
                        f32x4{acc[ct][4 * q4], acc[ct][4 * q4 + 1], acc[ct][4 * q4 + 2], acc[ct][4 * q4 + 3]};
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int p = it * 4 + (lane >> 4), ch = lane & 15;
                if (grp * 32 + p < Wo && ch * 4 < Co)
                    *(f32x4*)(y + ((size_t)row * Wo + grp * 32 + p) * Co + ch * 4) = *(const f32x4*)(patch + p * PROW + ch * 4);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
    }
}
// The root's filter gradient on the matrix pipe (y2_conv7x7s2_backward_filter_t): dW[co][kh][slot] = sum over the output
// pixels of dy[p][co] * window_kh(p)[slot], slot = 4 kw + c -- K of the matrix products is the output pixel.  Per output
// row the seven input rows are staged as in the forward kernel and the dy row as [wo][64] of T; wave kh reads BOTH operands
// with hardware-transposed LDS reads (k = 16 output pixels per step: dy^T rows = filters, window columns = slots; the
// window of pixel wo starts 16 bytes behind that of wo - 1) into two 32x32 accumulators (64 filters x 32 slots).  Block
// partials in the scalar kernel's format, summed in a fixed order by rn_sum_partials_kernel.
template <typename T>
__global__ __launch_bounds__(kC7Threads) void rn_conv7_wgrad_mfma_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                         float* __restrict__ part, int N, int H, int W, int Co,
                                                                         int PW, int WoP) {
    typedef typename Elem<T>::frag frag_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const rows = smem;                                 // [7][PW] pixels of 4 T
    char* const dyl = smem + (size_t)7 * PW * 8;             // [WoP][64] T
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int tid = threadIdx.x, lane = tid & 63, kh = tid >> 6;
    const int r32 = lane & 31, hh = lane >> 5;
    const int qq = (lane & 15) >> 2, pp = lane & 3, g1 = (lane >> 4) & 1;
    f32x16 acc[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[ct][q] = 0.f;
    for (int row = blockIdx.x; row < N * Ho; row += gridDim.x) {
        const int n = row / Ho, ho = row - n * Ho;
        __syncthreads();
        for (int i = tid; i < 7 * PW; i += kC7Threads) {
            const int r = i / PW, pc = i - r * PW;
            const int hi = 2 * ho + r - 3, wi = pc - 3;
            T o[4] = {Elem<T>::from_f32(0.f), Elem<T>::from_f32(0.f), Elem<T>::from_f32(0.f), Elem<T>::from_f32(0.f)};
            if (hi >= 0 && hi < H && wi >= 0 && wi < W) {
                const float* px = x + ((size_t)(n * H + hi) * W + wi) * 3;
                o[0] = Elem<T>::from_f32(px[0]); o[1] = Elem<T>::from_f32(px[1]); o[2] = Elem<T>::from_f32(px[2]);
            }
            *(u32x2*)(rows + (size_t)i * 8) = *(const u32x2*)o;
        }
        for (int i = tid; i < WoP * 16; i += kC7Threads) {      // 4 filters per thread and pixel
            const int wo = i >> 4, c4 = (i & 15) * 4;
            T o[4] = {Elem<T>::from_f32(0.f), Elem<T>::from_f32(0.f), Elem<T>::from_f32(0.f), Elem<T>::from_f32(0.f)};
            if (wo < Wo && c4 < Co) {
                const f32x4 v = *(const f32x4*)(dy + ((size_t)row * Wo + wo) * Co + c4);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = Elem<T>::from_f32(v[e]);
            }
            *(u32x2*)(dyl + (size_t)wo * 128 + c4 * 2) = *(const u32x2*)o;
        }
        __syncthreads();
        const char* xr = rows + (size_t)kh * PW * 8;
        for (int w0 = 0; w0 < WoP; w0 += 16) {
            const int k = w0 + 8 * hh + qq;
            const char* pb = xr + (size_t)k * 16 + (16 * g1 + 4 * pp) * 2;
            const frag_t fb = tr_frag<T>(pb, pb + 4 * 16);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const char* pa = dyl + (size_t)k * 128 + (ct * 32 + 16 * g1 + 4 * pp) * 2;
                const frag_t fa = tr_frag<T>(pa, pa + 4 * 128);
                mma32(acc[ct], fa, fb);
            }
        }
    }
    // D[filter (registers)][slot (lane)]: slot = 4 kw + c
    float* o = part + (size_t)blockIdx.x * 147 * Co;
    const int kw = r32 >> 2, c = r32 & 3;
    if (kw < 7 && c < 3) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int co = ct * 32 + acc_row(q, hh);
                if (co < Co) o[(size_t)(kh * 21 + kw * 3 + c) * Co + co] = acc[ct][q];
            }
    }
}
template <typename T>
static int conv7_wgrad_mfma_T(const float* x, const float* dy, float* dw, int N, int H, int W, int Cout, hipStream_t s) {
    const int Wo = (W + 1) / 2, groups = (Wo + 31) / 32, WoP = (Wo + 15) / 16 * 16;
    const int PW = 2 * groups * 32 + 8;
    const size_t lds = (size_t)7 * PW * 8 + (size_t)WoP * 128;
    if (lds > 64 * 1024) return Y2_ERR_ARG;
    const int rows = N * ((H + 1) / 2);
    const int blocks = rows < 512 ? rows : 512;
    float* part = (float*)op_scratch(s, (size_t)blocks * 147 * Cout * sizeof(float));
    if (!part) return op_scratch_error();
    hipLaunchKernelGGL(rn_conv7_wgrad_mfma_kernel<T>, dim3(blocks), dim3(kC7Threads), lds, s, x, dy, part, N, H, W, Cout, PW, WoP);
    hipLaunchKernelGGL(rn_sum_partials_kernel, dim3((147 * Cout + 63) / 64), dim3(256), 0, s, part, dw, blocks, 147 * Cout);
    RCHK(hipGetLastError());
    return Y2_OK;
}
template <typename T>
static int conv7_mfma_T(const float* x, const float* w, float* y, int N, int H, int W, int Cout, hipStream_t s) {
    const int Wo = (W + 1) / 2, groups = (Wo + 31) / 32;
    const int PW = 2 * groups * 32 + 8;               // every window of every group's 32 pixels stays inside the row
    const size_t lds = (size_t)7 * PW * 8 + (size_t)(kC7mThreads / 64) * 32 * (64 + 4) * sizeof(float);
    if (lds > 64 * 1024) return Y2_ERR_ARG;
    const int total = 2 * 7 * 2 * 64;
    T* wp = (T*)op_scratch(s, (size_t)total * 8 * sizeof(T));
    if (!wp) return op_scratch_error();
    hipLaunchKernelGGL(rn_conv7_pack_kernel<T>, dim3((total + 255) / 256), dim3(256), 0, s, w, wp, Cout, total);
    const int rows = N * ((H + 1) / 2);
    hipLaunchKernelGGL(rn_conv7_mfma_kernel<T>, dim3(rows < 1024 ? rows : 1024), dim3(kC7mThreads), lds, s, x, (const T*)wp, y, N, H, W,
                       Cout, PW);
    RCHK(hipGetLastError());
    return Y2_OK;
}

extern "C" {

int y2_batch_norm_forward(const float* x, const float* residual, float* y, size_t rows, int channels, const float* gamma,
                          const float* beta, float* moving_mean, float* moving_var, float* save_mean, float* save_var,
                          float eps, float decay, int is_training, int update_moving, int relu, void* stream) {
    if (!x || !y || !gamma || !beta || !moving_mean || !moving_var || !save_mean || !save_var || rows < 1 || channels < 1)
        return rfail(Y2_ERR_ARG, "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const size_t total = rows * channels;
    if (is_training) {
        // slim.batch_norm feeds the moving variance the batch variance of the non-fused path (tf.nn.moments +
        // assign_moving_average: biased), which is what slim of the reference's era runs; folded into the finalize
        RCHK((rn_bn_reduce<false>(x, nullptr, nullptr, rows, channels, nullptr, nullptr, eps, 0, save_mean, save_var, s,
                                  update_moving ? moving_mean : nullptr, update_moving ? moving_var : nullptr, decay)));
        rn_bn_apply(x, residual, y, total, channels, save_mean, save_var, gamma, beta, eps, relu, s);
    } else {
        RCHK(hipMemcpyAsync(save_mean, moving_mean, channels * sizeof(float), hipMemcpyDeviceToDevice, s));
        RCHK(hipMemcpyAsync(save_var, moving_var, channels * sizeof(float), hipMemcpyDeviceToDevice, s));
        rn_bn_apply(x, residual, y, total, channels, moving_mean, moving_var, gamma, beta, eps, relu, s);
    }
    RCHK(hipGetLastError());
    return Y2_OK;
}

int y2_batch_norm_backward(const float* dy, const float* y, const float* x, float* dx, float* dresidual, size_t rows,
                           int channels, const float* gamma, const float* save_mean, const float* save_var, float eps,
                           int is_training, int relu, float* dgamma, float* dbeta, void* stream) {
    if (!dy || !y || !x || !dx || !gamma || !save_mean || !save_var || !dgamma || !dbeta)
        return rfail(Y2_ERR_ARG, "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const size_t total = rows * channels;
    RCHK((rn_bn_reduce<true>(x, dy, y, rows, channels, save_mean, save_var, eps, relu, dgamma, dbeta, s)));
    if (rn_vec4(channels, dy, y, x, dx, dresidual)) {
        bool fixed;
        const unsigned nb = rn_grid4(total / 4, channels, &fixed);
        if (fixed)
            hipLaunchKernelGGL(rn_bn_bwd_apply4_kernel<true>, dim3(nb), dim3(256), 0, s, dy, y, x, dx, dresidual, total / 4,
                               rows, channels, save_mean, save_var, gamma, eps, relu, is_training, dgamma, dbeta);
        else
            hipLaunchKernelGGL(rn_bn_bwd_apply4_kernel<false>, dim3(nb), dim3(256), 0, s, dy, y, x, dx, dresidual, total / 4,
                               rows, channels, save_mean, save_var, gamma, eps, relu, is_training, dgamma, dbeta);
    } else
        hipLaunchKernelGGL(rn_bn_bwd_apply_kernel, dim3(grid_for(total)), dim3(256), 0, s, dy, y, x, dx, dresidual, total,
                           rows, channels, save_mean, save_var, gamma, eps, relu, is_training, dgamma, dbeta);
    RCHK(hipGetLastError());
    return Y2_OK;
}

// relu(a + b) and its backward g = dout * [out > 0] (the join of a bottleneck unit: resnet_v1.py:112 output =
// tf.nn.relu(shortcut + residual)); fp32, 16-byte accesses
__global__ void rn_add_relu_kernel(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ o,
                                   size_t n4, const float* as, const float* bs, float* os, size_t tail0, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 x = a[i], y = b[i];
        o[i] = make_float4(fmaxf(x.x + y.x, 0.f), fmaxf(x.y + y.y, 0.f), fmaxf(x.z + y.z, 0.f), fmaxf(x.w + y.w, 0.f));
    }
    if (blockIdx.x == 0)
        for (size_t i = tail0 + threadIdx.x; i < n; i += blockDim.x) os[i] = fmaxf(as[i] + bs[i], 0.f);
}
// d2 (nullable): a second addend of the incoming gradient (the two branches of the unit above: their sum never exists)
__global__ void rn_add_relu_bwd_kernel(const float4* __restrict__ d, const float4* __restrict__ d2,
                                       const float4* __restrict__ out, float4* __restrict__ g, size_t n4, const float* ds,
                                       const float* d2s, const float* os, float* gs, size_t tail0, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 x = d[i];
        const float4 y = out[i];
        if (d2) {
            const float4 z = d2[i];
            x.x += z.x; x.y += z.y; x.z += z.z; x.w += z.w;
        }
        g[i] = make_float4(y.x > 0.f ? x.x : 0.f, y.y > 0.f ? x.y : 0.f, y.z > 0.f ? x.z : 0.f, y.w > 0.f ? x.w : 0.f);
    }
    if (blockIdx.x == 0)
        for (size_t i = tail0 + threadIdx.x; i < n; i += blockDim.x) gs[i] = os[i] > 0.f ? ds[i] + (d2s ? d2s[i] : 0.f) : 0.f;
}
int y2_add_relu(const float* a, const float* b, float* out, size_t n, void* stream) {
    if (!a || !b || !out) return rfail(Y2_ERR_ARG, "null tensor");
    if ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) != 0) return rfail(Y2_ERR_ARG, "tensors must be 16-byte aligned");
    const size_t n4 = n / 4;
    hipLaunchKernelGGL(rn_add_relu_kernel, dim3(grid_for(n4)), dim3(256), 0, (hipStream_t)stream, (const float4*)a,
                       (const float4*)b, (float4*)out, n4, a, b, out, n4 * 4, n);
    RCHK(hipGetLastError());
    return Y2_OK;
}
int y2_add_relu_backward(const float* dout, const float* dout2, const float* out, float* g, size_t n, void* stream) {
    if (!dout || !out || !g) return rfail(Y2_ERR_ARG, "null tensor");
    if ((((uintptr_t)dout | (uintptr_t)dout2 | (uintptr_t)out | (uintptr_t)g) & 15) != 0)
        return rfail(Y2_ERR_ARG, "tensors must be 16-byte aligned");
    const size_t n4 = n / 4;
    hipLaunchKernelGGL(rn_add_relu_bwd_kernel, dim3(grid_for(n4)), dim3(256), 0, (hipStream_t)stream, (const float4*)dout,
                       (const float4*)dout2, (const float4*)out, (float4*)g, n4, dout, dout2, out, g, n4 * 4, n);
    RCHK(hipGetLastError());
    return Y2_OK;
}

int y2_join_backward(int dtype, const void* out_bordered, const void* d1, const void* d2, int d2_f32, void* g, int N, int H,
                     int W, int C, void* stream) {
    if (!out_bordered || !d1 || !d2 || !g) return rfail(Y2_ERR_ARG, "null tensor");
    if (dtype != 1 && dtype != 2) return rfail(Y2_ERR_ARG, "y2_join_backward: 16-bit arithmetic types (linked stacks)");
    if (C % 8 != 0) return rfail(Y2_ERR_ARG, "y2_join_backward: channels in multiples of 8");
    const size_t total = (size_t)N * H * W * (C / 8);
    const dim3 grid(grid_for(total)), block(256);
    if (dtype == 1)
        hipLaunchKernelGGL(rn_join_bwd_t_kernel<y2::half_t>, grid, block, 0, (hipStream_t)stream, (const y2::half_t*)out_bordered,
                           (const y2::half_t*)d1, d2, d2_f32, (y2::half_t*)g, N, H, W, C);
    else
        hipLaunchKernelGGL(rn_join_bwd_t_kernel<y2::bf16_t>, grid, block, 0, (hipStream_t)stream, (const y2::bf16_t*)out_bordered,
                           (const y2::bf16_t*)d1, d2, d2_f32, (y2::bf16_t*)g, N, H, W, C);
    RCHK(hipGetLastError());
    return Y2_OK;
}

int y2_join_backward_s2(int dtype, const void* out_bordered, const void* d1, const void* d2, int d2_f32, void* g, int N, int H,
                        int W, int C, void* stream) {
    if (!out_bordered || !d1 || !d2 || !g) return rfail(Y2_ERR_ARG, "y2_join_backward_s2: null tensor");
    if (dtype != 1 && dtype != 2) return rfail(Y2_ERR_ARG, "y2_join_backward_s2: the 16-bit arithmetic types");
    if (C % 8 != 0 || (H & 1) || (W & 1)) return rfail(Y2_ERR_ARG, "y2_join_backward_s2: channels in multiples of 8, an even map");
    const size_t total = (size_t)N * H * W * (C / 8);
    const dim3 grid(grid_for(total)), block(256);
    if (dtype == 1)
        hipLaunchKernelGGL(rn_join_bwd_s2_kernel<y2::half_t>, grid, block, 0, (hipStream_t)stream, (const y2::half_t*)out_bordered,
                           (const y2::half_t*)d1, d2, d2_f32, (y2::half_t*)g, N, H, W, C);
    else
        hipLaunchKernelGGL(rn_join_bwd_s2_kernel<y2::bf16_t>, grid, block, 0, (hipStream_t)stream, (const y2::bf16_t*)out_bordered,
                           (const y2::bf16_t*)d1, d2, d2_f32, (y2::bf16_t*)g, N, H, W, C);
    RCHK(hipGetLastError());
    return Y2_OK;
}
int y2_subsample_bordered(int dtype, const void* src_bordered, void* dst_bordered, int N, int H, int W, int C, void* stream) {
    if (!src_bordered || !dst_bordered) return rfail(Y2_ERR_ARG, "y2_subsample_bordered: null tensor");
    if (dtype != 1 && dtype != 2) return rfail(Y2_ERR_ARG, "y2_subsample_bordered: the 16-bit arithmetic types");
    if (C % 8 != 0 || (H & 1) || (W & 1)) return rfail(Y2_ERR_ARG, "y2_subsample_bordered: channels in multiples of 8, an even map");
    const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / 8);
    const dim3 grid(grid_for(total)), block(256);
    if (dtype == 1)
        hipLaunchKernelGGL(rn_subsample_bordered_kernel<y2::half_t>, grid, block, 0, (hipStream_t)stream,
                           (const y2::half_t*)src_bordered, (y2::half_t*)dst_bordered, N, H, W, C);
    else
        hipLaunchKernelGGL(rn_subsample_bordered_kernel<y2::bf16_t>, grid, block, 0, (hipStream_t)stream,
                           (const y2::bf16_t*)src_bordered, (y2::bf16_t*)dst_bordered, N, H, W, C);
    RCHK(hipGetLastError());
    return Y2_OK;
}

int y2_subsample(const float* x, float* y, int N, int H, int W, int C, int factor, int forward, void* stream) {
    if (!x || !y || factor < 1) return rfail(Y2_ERR_ARG, "bad arguments");
    const int Ho = (H + factor - 1) / factor, Wo = (W + factor - 1) / factor;
    const size_t total = forward ? (size_t)N * Ho * Wo * C : (size_t)N * H * W * C;
    hipLaunchKernelGGL(rn_subsample_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C,
                       factor, forward);
    RCHK(hipGetLastError());
    return Y2_OK;
}

int y2_maxpool3x3s2(const float* x, float* y, int N, int H, int W, int C, void* stream) {
    if (!x || !y) return rfail(Y2_ERR_ARG, "null tensor");
    const size_t total = (size_t)N * ((H + 1) / 2) * ((W + 1) / 2) * C;
    hipLaunchKernelGGL(rn_maxpool3_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C);
    RCHK(hipGetLastError());
    return Y2_OK;
}
int y2_maxpool3x3s2_backward(const float* x, const float* dy, float* dx, int N, int H, int W, int C, void* stream) {
    if (!x || !dy || !dx) return rfail(Y2_ERR_ARG, "null tensor");
    const size_t total = (size_t)N * H * W * C, outs = (size_t)N * ((H + 1) / 2) * ((W + 1) / 2) * C;
    unsigned char* arg = (unsigned char*)op_scratch((hipStream_t)stream, outs);
    if (!arg) return op_scratch_error();      // (op_scratch left the reason in the error state)
    hipLaunchKernelGGL(rn_maxpool3_arg_kernel, dim3(grid_for(outs)), dim3(256), 0, (hipStream_t)stream, x, arg, N, H, W, C);
    hipLaunchKernelGGL(rn_maxpool3_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, arg, dy, dx, N, H, W, C);
    RCHK(hipGetLastError());
    return Y2_OK;
}

int y2_conv7x7s2(const float* x, const float* w, float* y, int N, int H, int W, int Cout, void* stream) {
    if (!x || !w || !y || N < 1 || H < 1 || W < 1 || Cout < 1) return rfail(Y2_ERR_ARG, "bad arguments");
    const size_t lds = (size_t)7 * (W + 6 + kC7Pad) * 3 * sizeof(float);
    if (lds > 64 * 1024) return rfail(Y2_ERR_ARG, "image rows of %d pixels do not fit the staged window", W);
    const int rows = N * ((H + 1) / 2);
    hipLaunchKernelGGL(rn_conv7_fwd_kernel, dim3(rows < 1024 ? rows : 1024), dim3(kC7Threads), lds, (hipStream_t)stream, x, w,
                       y, N, H, W, Cout);
    RCHK(hipGetLastError());
    return Y2_OK;
}
int y2_conv7x7s2_t(const float* x, const float* w, float* y, int N, int H, int W, int Cout, int dtype, void* stream) {
    if (!x || !w || !y || N < 1 || H < 1 || W < 1 || Cout < 1) return rfail(Y2_ERR_ARG, "bad arguments");
    if ((dtype == 1 || dtype == 2) && Cout <= 64 && (Cout % 4) == 0) {
        const int rc = dtype == 1 ? conv7_mfma_T<half_t>(x, w, y, N, H, W, Cout, (hipStream_t)stream)
                                  : conv7_mfma_T<bf16_t>(x, w, y, N, H, W, Cout, (hipStream_t)stream);
        if (rc != Y2_ERR_ARG) return rc;         // (rows too long for the staged window: the fp32 kernel decides)
    }
    return y2_conv7x7s2(x, w, y, N, H, W, Cout, stream);
}
int y2_conv7x7s2_backward_filter(const float* x, const float* dy, float* dw, int N, int H, int W, int Cout, void* stream);
int y2_conv7x7s2_backward_filter_t(const float* x, const float* dy, float* dw, int N, int H, int W, int Cout, int dtype,
                                   void* stream) {
    if (!x || !dy || !dw || N < 1 || H < 1 || W < 1 || Cout < 1) return rfail(Y2_ERR_ARG, "bad arguments");
    if ((dtype == 1 || dtype == 2) && Cout <= 64 && (Cout % 4) == 0) {
        const int rc = dtype == 1 ? conv7_wgrad_mfma_T<half_t>(x, dy, dw, N, H, W, Cout, (hipStream_t)stream)
                                  : conv7_wgrad_mfma_T<bf16_t>(x, dy, dw, N, H, W, Cout, (hipStream_t)stream);
        if (rc != Y2_ERR_ARG) return rc;
    }
    return y2_conv7x7s2_backward_filter(x, dy, dw, N, H, W, Cout, stream);
}
int y2_conv7x7s2_backward_filter(const float* x, const float* dy, float* dw, int N, int H, int W, int Cout, void* stream) {
    if (!x || !dy || !dw || N < 1 || H < 1 || W < 1) return rfail(Y2_ERR_ARG, "bad arguments");
    if (Cout < 1 || Cout > 64) return rfail(Y2_ERR_ARG, "the root convolution has 64 filters (resnet_v1.py:197)");
    const size_t lds = (size_t)7 * (W + 6 + kC7Pad) * 3 * sizeof(float);
    if (lds > 64 * 1024) return rfail(Y2_ERR_ARG, "image rows of %d pixels do not fit the staged window", W);
    hipStream_t s = (hipStream_t)stream;
    const int rows = N * ((H + 1) / 2);
    const int blocks = rows < 512 ? rows : 512;
    float* part = (float*)op_scratch(s, (size_t)blocks * 147 * Cout * sizeof(float));
    if (!part) return op_scratch_error();
    hipLaunchKernelGGL(rn_conv7_wgrad_kernel, dim3(blocks), dim3(kC7Threads), lds, s, x, dy, part, N, H, W, Cout);
    hipLaunchKernelGGL(rn_sum_partials_kernel, dim3((147 * Cout + 63) / 64), dim3(256), 0, s, part, dw, blocks, 147 * Cout);
    RCHK(hipGetLastError());
    return Y2_OK;
}

int y2_bias_relu(float* y, const float* bias, size_t rows, int channels, int relu, void* stream) {
    if (!y || !bias) return rfail(Y2_ERR_ARG, "null tensor");
    hipLaunchKernelGGL(rn_bias_relu_kernel, dim3(grid_for(rows * channels)), dim3(256), 0, (hipStream_t)stream, y, bias,
                       rows * channels, channels, relu);
    RCHK(hipGetLastError());
    return Y2_OK;
}
int y2_bias_relu_backward(const float* dy, const float* y, float* dz, float* dbias, size_t rows, int channels, int relu,
                          void* stream) {
    if (!dy || !y || !dz || !dbias) return rfail(Y2_ERR_ARG, "null tensor");
    hipLaunchKernelGGL(rn_bias_relu_bwd_kernel, dim3((channels + 255) / 256), dim3(256), 0, (hipStream_t)stream, dy, y, dz,
                       dbias, rows, channels, relu);
    RCHK(hipGetLastError());
    return Y2_OK;
}
int y2_dropout(const float* x, float* y, size_t n, float keep_prob, uint64_t seed, void* stream) {
    if (!x || !y || !(keep_prob > 0.f) || keep_prob > 1.f) return rfail(Y2_ERR_ARG, "bad arguments");
    hipLaunchKernelGGL(rn_dropout_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, keep_prob, seed,
                       (const uint64_t*)nullptr);
    RCHK(hipGetLastError());
    return Y2_OK;
}
int y2_dropout_dev(const float* x, float* y, size_t n, float keep_prob, const uint64_t* seed, void* stream) {
    if (!x || !y || !seed || !(keep_prob > 0.f) || keep_prob > 1.f) return rfail(Y2_ERR_ARG, "bad arguments");
    hipLaunchKernelGGL(rn_dropout_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, keep_prob,
                       (uint64_t)0, seed);
    RCHK(hipGetLastError());
    return Y2_OK;
}

}  // extern "C"
