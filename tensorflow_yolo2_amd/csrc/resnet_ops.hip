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
// The reference trains this model at batch 4: these operators are written for correctness and coalesced access,
// not tuned -- the MFMA work is in the convolutions.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/yolo2_hip.h"
#include "common.h"

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

// ---------------------------------------------------------------------------
// batch norm over [M][C] fp32.  Statistics: one block per 32 channels sweeps all rows (256 threads = 8 rows x 32
// channels per pass: 128-byte row segments), shifted sums about the first row, double merge.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rn_bn_stats_kernel(const float* __restrict__ x, size_t M, int C, float* mean,
                                                          float* var) {
    __shared__ double r1[8][32], r2[8][32];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), row0 = threadIdx.x >> 5;
    const bool cv = c < C;
    const float piv = cv ? x[c] : 0.f;
    double s1 = 0.0, s2 = 0.0;
    if (cv)
        for (size_t m = row0; m < M; m += 8) {
            const double d = (double)(x[m * C + c] - piv);
            s1 += d;
            s2 += d * d;
        }
    r1[row0][threadIdx.x & 31] = s1;
    r2[row0][threadIdx.x & 31] = s2;
    __syncthreads();
    if (row0 == 0 && cv) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < 8; ++k) { a += r1[k][threadIdx.x]; b += r2[k][threadIdx.x]; }
        const double md = a / (double)M;
        mean[c] = (float)((double)piv + md);
        double v = b / (double)M - md * md;
        var[c] = (float)(v > 0 ? v : 0);
    }
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
__global__ void rn_bn_moving_kernel(float* mm, float* mv, const float* mean, const float* var, int C, float decay,
                                    float unbias) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mm[c] = decay * mm[c] + (1.0f - decay) * mean[c];
    mv[c] = decay * mv[c] + (1.0f - decay) * var[c] * unbias;
}

// backward: dz = dy * [y > 0] (ReLU on the stored output) ; S1 = sum dz, S2 = sum dz * xhat
__global__ __launch_bounds__(256) void rn_bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                               const float* __restrict__ x, size_t M, int C,
                                                               const float* mean, const float* var, float eps, int relu,
                                                               float* dgamma, float* dbeta) {
    __shared__ double r1[8][32], r2[8][32];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), row0 = threadIdx.x >> 5;
    const bool cv = c < C;
    double s1 = 0.0, s2 = 0.0;
    if (cv) {
        const float mu = mean[c], inv = 1.0f / sqrtf(var[c] + eps);
        for (size_t m = row0; m < M; m += 8) {
            const size_t i = m * C + c;
            const float dz = (relu && !(y[i] > 0.f)) ? 0.f : dy[i];
            s1 += (double)dz;
            s2 += (double)dz * (double)((x[i] - mu) * inv);
        }
    }
    r1[row0][threadIdx.x & 31] = s1;
    r2[row0][threadIdx.x & 31] = s2;
    __syncthreads();
    if (row0 == 0 && cv) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < 8; ++k) { a += r1[k][threadIdx.x]; b += r2[k][threadIdx.x]; }
        dbeta[c] = (float)a;
        dgamma[c] = (float)b;
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
__global__ void rn_maxpool3_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ dy,
                                   float* __restrict__ dx, int N, int H, int W, int C) {
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
        if (dx) atomicAdd(dx + arg, dy[i]);                     // windows overlap (stride 2 < 3): dx pre-zeroed
    }
}

// ---------------------------------------------------------------------------
// root convolution: conv2d_same(64, 7, stride 2) on [N,H,W,3]: pad 3 before / 3 after, VALID stride 2
// forward: one thread per (pixel, cout): 147 MACs;  filter gradient: one block per (kh, kw, c), thread = cout
// ---------------------------------------------------------------------------
__global__ void rn_conv7_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int N,
                                    int H, int W, int Co) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;     // (H + 6 - 7) / 2 + 1
    const size_t total = (size_t)N * Ho * Wo * Co;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % Co);
        const size_t p = i / Co;
        const int wo = (int)(p % Wo), ho = (int)((p / Wo) % Ho), n = (int)(p / ((size_t)Wo * Ho));
        float acc = 0.f;
        for (int kh = 0; kh < 7; ++kh) {
            const int h = ho * 2 + kh - 3;
            if (h < 0 || h >= H) continue;
            for (int kw = 0; kw < 7; ++kw) {
                const int ww = wo * 2 + kw - 3;
                if (ww < 0 || ww >= W) continue;
                const float* xp = x + (((size_t)n * H + h) * W + ww) * 3;
                const float* wp = w + ((size_t)(kh * 7 + kw) * 3) * Co + co;
                acc = fmaf(xp[0], wp[0], acc);
                acc = fmaf(xp[1], wp[Co], acc);
                acc = fmaf(xp[2], wp[2 * Co], acc);
            }
        }
        y[i] = acc;
    }
}
__global__ __launch_bounds__(256) void rn_conv7_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             float* __restrict__ dw, int N, int H, int W, int Co) {
    // block = (kh, kw, c); 256 threads = 64 couts x 4 pixel slices
    __shared__ float red[4][64];
    const int t = blockIdx.x, c = t % 3, kw = (t / 3) % 7, kh = t / 21;
    const int co = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    float acc = 0.f;
    if (co < Co)
        for (size_t p = sl; p < (size_t)N * Ho * Wo; p += 4) {
            const int wo = (int)(p % Wo), ho = (int)((p / Wo) % Ho), n = (int)(p / ((size_t)Wo * Ho));
            const int h = ho * 2 + kh - 3, ww = wo * 2 + kw - 3;
            if (h < 0 || h >= H || ww < 0 || ww >= W) continue;
            acc = fmaf(x[(((size_t)n * H + h) * W + ww) * 3 + c], dy[p * Co + co], acc);
        }
    red[sl][co] = acc;
    __syncthreads();
    if (sl == 0 && co < Co) dw[(size_t)t * Co + co] = red[0][co] + red[1][co] + red[2][co] + red[3][co];
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
                                  uint64_t seed) {
    const float inv = 1.0f / keep;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float u = (float)(rn_mix64(seed * 0xD1B54A32D192ED03ull + i) >> 40) * (1.0f / 16777216.0f);
        y[i] = u < keep ? x[i] * inv : 0.f;
    }
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
        hipLaunchKernelGGL(rn_bn_stats_kernel, dim3((channels + 31) / 32), dim3(256), 0, s, x, rows, channels, save_mean,
                           save_var);
        if (update_moving) {
            // slim.batch_norm feeds the moving variance the UNBIASED batch variance (fused_batch_norm semantics are
            // version dependent; tf.nn.moments + assign_moving_average of the non-fused path use the biased one,
            // which is what slim of the reference's era runs): biased
            hipLaunchKernelGGL(rn_bn_moving_kernel, dim3((channels + 255) / 256), dim3(256), 0, s, moving_mean, moving_var,
                               save_mean, save_var, channels, decay, 1.0f);
        }
        hipLaunchKernelGGL(rn_bn_apply_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, residual, y, total, channels,
                           save_mean, save_var, gamma, beta, eps, relu);
    } else {
        RCHK(hipMemcpyAsync(save_mean, moving_mean, channels * sizeof(float), hipMemcpyDeviceToDevice, s));
        RCHK(hipMemcpyAsync(save_var, moving_var, channels * sizeof(float), hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(rn_bn_apply_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, residual, y, total, channels,
                           moving_mean, moving_var, gamma, beta, eps, relu);
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
    hipLaunchKernelGGL(rn_bn_bwd_reduce_kernel, dim3((channels + 31) / 32), dim3(256), 0, s, dy, y, x, rows, channels,
                       save_mean, save_var, eps, relu, dgamma, dbeta);
    hipLaunchKernelGGL(rn_bn_bwd_apply_kernel, dim3(grid_for(total)), dim3(256), 0, s, dy, y, x, dx, dresidual, total, rows,
                       channels, save_mean, save_var, gamma, eps, relu, is_training, dgamma, dbeta);
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
    hipLaunchKernelGGL(rn_maxpool3_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, nullptr, nullptr,
                       N, H, W, C);
    RCHK(hipGetLastError());
    return Y2_OK;
}
int y2_maxpool3x3s2_backward(const float* x, const float* dy, float* dx, int N, int H, int W, int C, void* stream) {
    if (!x || !dy || !dx) return rfail(Y2_ERR_ARG, "null tensor");
    RCHK(hipMemsetAsync(dx, 0, (size_t)N * H * W * C * sizeof(float), (hipStream_t)stream));
    const size_t total = (size_t)N * ((H + 1) / 2) * ((W + 1) / 2) * C;
    hipLaunchKernelGGL(rn_maxpool3_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, nullptr, dy, dx, N, H,
                       W, C);
    RCHK(hipGetLastError());
    return Y2_OK;
}

int y2_conv7x7s2(const float* x, const float* w, float* y, int N, int H, int W, int Cout, void* stream) {
    if (!x || !w || !y) return rfail(Y2_ERR_ARG, "null tensor");
    const size_t total = (size_t)N * ((H + 1) / 2) * ((W + 1) / 2) * Cout;
    hipLaunchKernelGGL(rn_conv7_fwd_kernel, dim3(grid_for(total, 65536)), dim3(256), 0, (hipStream_t)stream, x, w, y, N, H, W,
                       Cout);
    RCHK(hipGetLastError());
    return Y2_OK;
}
int y2_conv7x7s2_backward_filter(const float* x, const float* dy, float* dw, int N, int H, int W, int Cout, void* stream) {
    if (!x || !dy || !dw) return rfail(Y2_ERR_ARG, "null tensor");
    if (Cout > 64) return rfail(Y2_ERR_ARG, "the root convolution has 64 filters (resnet_v1.py:197)");
    hipLaunchKernelGGL(rn_conv7_wgrad_kernel, dim3(147), dim3(256), 0, (hipStream_t)stream, x, dy, dw, N, H, W, Cout);
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
    hipLaunchKernelGGL(rn_dropout_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, keep_prob, seed);
    RCHK(hipGetLastError());
    return Y2_OK;
}

}  // extern "C"
