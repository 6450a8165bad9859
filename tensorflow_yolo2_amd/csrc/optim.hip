// Flat-buffer optimizers + initialisers.  All trainable tensors of the network
// live in ONE contiguous fp32 buffer (and so do the gradients: one RCCL
// all-reduce per step), so each optimizer is a single streaming kernel.
// Reference: tf.train.AdamOptimizer() (src/pascal/pascal_train_darknet.py:51),
// tf.train.MomentumOptimizer(0.001, 0.9) (src/imagenet/imagenet_train_darknet.py:58),
// tf.truncated_normal(stddev=0.1) / tf.constant(0.1) (src/yolo2_nets/darknet.py:10-17).
#include "common.h"
#include "kernels.h"

namespace y2 {

__global__ void adam_kernel(float4* p, float4* m, float4* v, const float4* g, size_t n4, float lr_t, float b1,
                            float b2, float eps, float gscale, float* pt, float* mt, float* vt, const float* gt,
                            size_t tail0, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 pp = p[i], mm = m[i], vv = v[i], gg = g[i];
        float* P = &pp.x; float* M = &mm.x; float* V = &vv.x; const float* G = &gg.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = G[k] * gscale;
            M[k] = b1 * M[k] + (1.0f - b1) * gk;
            V[k] = b2 * V[k] + (1.0f - b2) * gk * gk;
            P[k] = P[k] - lr_t * M[k] / (sqrtf(V[k]) + eps);
        }
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
    if (blockIdx.x == 0) {
        for (size_t i = tail0 + threadIdx.x; i < n; i += blockDim.x) {
            const float gk = gt[i] * gscale;
            const float mk = b1 * mt[i] + (1.0f - b1) * gk;
            const float vk = b2 * vt[i] + (1.0f - b2) * gk * gk;
            mt[i] = mk; vt[i] = vk;
            pt[i] = pt[i] - lr_t * mk / (sqrtf(vk) + eps);
        }
    }
}
hipError_t launch_adam(float* p, float* m, float* v, const float* g, size_t n, float lr_t, float b1, float b2,
                       float eps, float gscale, hipStream_t s) {
    const size_t n4 = n / 4;
    size_t nb = (n4 + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)nb), dim3(256), 0, s, (float4*)p, (float4*)m, (float4*)v,
                       (const float4*)g, n4, lr_t, b1, b2, eps, gscale, p, m, v, g, n4 * 4, n);
    return hipGetLastError();
}

__global__ void momentum_kernel(float* p, float* acc, const float* g, size_t n, float lr, float mom, float gscale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float a = mom * acc[i] + g[i] * gscale;
        acc[i] = a;
        p[i] = p[i] - lr * a;
    }
}
hipError_t launch_momentum(float* p, float* acc, const float* g, size_t n, float lr, float mom, float gscale,
                           hipStream_t s) {
    size_t nb = (n + 255) / 256;
    if (nb > 8192) nb = 8192;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(momentum_kernel, dim3((unsigned)nb), dim3(256), 0, s, p, acc, g, n, lr, mom, gscale);
    return hipGetLastError();
}

// counter-based generator (splitmix64 finaliser) -> Box-Muller -> reject |z| > 2
Y2_DEV uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ void trunc_normal_kernel(float* p, size_t n, float stddev, uint64_t seed, uint64_t stream_id) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float z = 0.f;
        for (uint64_t attempt = 0; attempt < 64; ++attempt) {
            const uint64_t r = mix64(mix64(seed ^ (stream_id * 0xD1B54A32D192ED03ull)) + i * 64 + attempt);
            const float u1 = ((float)((r >> 40) + 1)) * (1.0f / 16777217.0f);  // (0,1]
            const float u2 = ((float)((r >> 8) & 0xFFFFFF)) * (1.0f / 16777216.0f);
            z = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
            if (fabsf(z) <= 2.0f) break;
        }
        p[i] = z * stddev;
    }
}
hipError_t launch_init_trunc_normal(float* p, size_t n, float stddev, uint64_t seed, uint64_t stream_id,
                                    hipStream_t s) {
    size_t nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(trunc_normal_kernel, dim3((unsigned)nb), dim3(256), 0, s, p, n, stddev, seed, stream_id);
    return hipGetLastError();
}

__global__ void fill_kernel(float* p, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
hipError_t launch_fill(float* p, size_t n, float v, hipStream_t s) {
    size_t nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)nb), dim3(256), 0, s, p, n, v);
    return hipGetLastError();
}

}  // namespace y2
