// Flat-buffer optimizers + initialisers.  All trainable tensors of the network
// live in ONE contiguous fp32 buffer (and so do the gradients: one RCCL
// all-reduce per step), so each optimizer is a single streaming kernel.
// Reference: tf.train.AdamOptimizer() (src/pascal/pascal_train_darknet.py:51),
// tf.train.MomentumOptimizer(0.001, 0.9) (src/imagenet/imagenet_train_darknet.py:58),
// tf.truncated_normal(stddev=0.1) / tf.constant(0.1) (src/yolo2_nets/darknet.py:10-17).
#include "common.h"
#include "kernels.h"
#include "optim_math.h"

namespace y2 {

__global__ void adam_kernel(float4* p, float4* m, float4* v, const float4* g, size_t n4, float lr_t, float b1,
                            float b2, float eps, float gscale, float* pt, float* mt, float* vt, const float* gt,
                            size_t tail0, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 pp = p[i], mm = m[i], vv = v[i], gg = g[i];
        float* P = &pp.x; float* M = &mm.x; float* V = &vv.x; const float* G = &gg.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) adam_update(P[k], M[k], V[k], G[k] * gscale, lr_t, b1, b2, eps);
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
    if (blockIdx.x == 0) {
        for (size_t i = tail0 + threadIdx.x; i < n; i += blockDim.x) {
            float pk = pt[i], mk = mt[i], vk = vt[i];
            adam_update(pk, mk, vk, gt[i] * gscale, lr_t, b1, b2, eps);
            mt[i] = mk; vt[i] = vk; pt[i] = pk;
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
        float pk = p[i], a = acc[i];
        momentum_update(pk, a, g[i] * gscale, lr, mom);
        acc[i] = a;
        p[i] = pk;
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

// ---------------------------------------------------------------------------
// Dynamic loss scaling for the half-precision modes (the reference runs fp32 and has no such failure
// mode: an fp16 dY beyond 65504 becomes inf, then NaN in the weight gradient, and an unguarded update
// would poison params, m and v for good -- and, through the SUM all-reduce, every replica).
//   grad_check : any non-finite value in the flat gradient buffer -> ctrl.found_inf; the last block to
//                finish also advances the device-side step counter and computes TF's lr_t for it
//                (only finite steps count: a skipped step leaves t, m, v and params untouched)
//   *_guarded  : the optimizer kernels above, skipped as a whole when ctrl.found_inf is set
// No host synchronisation: the host reads ctrl one step late to adapt grad_scale.
// ---------------------------------------------------------------------------
struct OptCtrl { int found_inf, step, skipped, reserved; float lr_t; int pad[3]; };

// full scan (y2_grad_check_full: any flat buffer)
__global__ __launch_bounds__(256) void grad_check_kernel(const float4* g, size_t n4, const float* gt, size_t tail0,
                                                         size_t n, OptCtrl* ctrl) {
    // |x| < inf  <=>  exponent bits != all ones; OR the exponent tests of a whole thread, then of the block
    unsigned bad = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = g[i];
        const unsigned a = __float_as_uint(v.x), b = __float_as_uint(v.y), c = __float_as_uint(v.z), d = __float_as_uint(v.w);
        bad |= ((a & 0x7F800000u) == 0x7F800000u) | ((b & 0x7F800000u) == 0x7F800000u) |
               ((c & 0x7F800000u) == 0x7F800000u) | ((d & 0x7F800000u) == 0x7F800000u);
    }
    if (blockIdx.x == 0)
        for (size_t i = tail0 + threadIdx.x; i < n; i += blockDim.x)
            bad |= (__float_as_uint(gt[i]) & 0x7F800000u) == 0x7F800000u;
    if (__any(bad != 0) && (threadIdx.x & 63) == 0) atomicOr(&ctrl->found_inf, 1);
}
hipError_t launch_grad_check(const float* g, size_t n, void* ctrl, hipStream_t s) {
    hipError_t e = hipMemsetAsync(ctrl, 0, sizeof(int), s);   // found_inf of the previous step
    if (e != hipSuccess) return e;
    const size_t n4 = n / 4;
    size_t nb = (n4 + 1023) / 1024;
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(grad_check_kernel, dim3((unsigned)nb), dim3(256), 0, s, (const float4*)g, n4, g, n4 * 4, n,
                       (OptCtrl*)ctrl);
    return hipGetLastError();
}

// Range scan (y2_range_check; round 6): found_inf |= any element that is non-finite or beyond +-limit.  The fused FC update
// (fc.hip fc_dw_adam_kernel) rounds dz and x to the arithmetic type inside the kernel and never stores dW: a |dz| above the
// type's largest finite value is finite in fp32 -- so every stored gradient it feeds stays finite -- but inf in the product.
// Does NOT clear the flag: it runs between the scan of the stored gradients and the guarded update.
__global__ __launch_bounds__(256) void range_check_kernel(const float* x, size_t n, float limit, OptCtrl* ctrl) {
    unsigned bad = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        bad |= !(fabsf(x[i]) <= limit);          // NaN compares false: flagged
    if (__any(bad != 0) && (threadIdx.x & 63) == 0) atomicOr(&ctrl->found_inf, 1);
}
hipError_t launch_range_check(const float* x, size_t n, float limit, void* ctrl, hipStream_t s) {
    size_t nb = (n + 2047) / 2048;
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(range_check_kernel, dim3((unsigned)nb), dim3(256), 0, s, x, n, limit, (OptCtrl*)ctrl);
    return hipGetLastError();
}

// Sentinel scan (y2_grad_check): a non-finite value anywhere in the backward pass reaches the checked ranges.
//   * an inf / NaN in dA of layer l makes S1 = sum(dz) of its channel non-finite -> dbeta_l (checked), and through
//     ka = f(S1) every dy of that channel -> dW_l, and through the dgrad every layer below;
//   * a dy that overflows only at its own f16 store (finite sums) is an inf operand of dgrad_l -> dA_{l-1} -> dbeta_{l-1};
//     in the first layer (no dgrad below it) dy feeds the filter gradient directly -> dW_0 (checked whole).
// ranges: [offset, count] pairs of the flat gradient buffer: b / gamma / beta of every layer + the first filter.
// flag (nullable): a non-finite marker set by a kernel of the backward pass (ConvArgs::nonfinite); consumed here.
__global__ __launch_bounds__(256) void grad_check_ranges_kernel(const float* g, const unsigned* ranges, int nranges,
                                                                OptCtrl* ctrl, unsigned* flag) {
    unsigned bad = 0;
    if (flag && blockIdx.x == 0 && threadIdx.x == 0) {
        bad = *flag != 0;
        *flag = 0;
    }
    for (int r = blockIdx.x; r < nranges; r += gridDim.x) {     // a block per range (each is a few thousand floats)
        const float* p = g + ranges[2 * r];
        const unsigned cnt = ranges[2 * r + 1];
        for (unsigned i = threadIdx.x; i < cnt; i += blockDim.x)
            bad |= (__float_as_uint(p[i]) & 0x7F800000u) == 0x7F800000u;
    }
    if (__any(bad != 0) && (threadIdx.x & 63) == 0) atomicOr(&ctrl->found_inf, 1);
}
hipError_t launch_grad_check_ranges(const float* g, const void* ranges_dev, int nranges, void* ctrl, hipStream_t s,
                                    unsigned* flag, bool keep) {
    if (!keep) {
        hipError_t e = hipMemsetAsync(ctrl, 0, sizeof(int), s);
        if (e != hipSuccess) return e;
    }
    const int nb = nranges < 1 ? 1 : (nranges < 64 ? nranges : 64);
    hipLaunchKernelGGL(grad_check_ranges_kernel, dim3(nb), dim3(256), 0, s, g, (const unsigned*)ranges_dev, nranges,
                       (OptCtrl*)ctrl, flag);
    return hipGetLastError();
}

// after either scan: finite -> step += 1 and TF's lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t) for it; else skipped += 1
__global__ void opt_ctrl_advance_kernel(OptCtrl* ctrl, float lr, float b1, float b2) {
    if (ctrl->found_inf) {
        ctrl->skipped += 1;
    } else {
        const int t = ctrl->step + 1;
        ctrl->step = t;
        ctrl->lr_t = (float)((double)lr * sqrt(1.0 - pow((double)b2, (double)t)) / (1.0 - pow((double)b1, (double)t)));
    }
}
hipError_t launch_opt_ctrl_advance(void* ctrl, float lr, float b1, float b2, hipStream_t s) {
    hipLaunchKernelGGL(opt_ctrl_advance_kernel, dim3(1), dim3(1), 0, s, (OptCtrl*)ctrl, lr, b1, b2);
    return hipGetLastError();
}

__global__ void adam_guarded_kernel(float4* p, float4* m, float4* v, const float4* g, size_t n4, const OptCtrl* ctrl,
                                    float b1, float b2, float eps, float gscale, float* pt, float* mt, float* vt,
                                    const float* gt, size_t tail0, size_t n) {
    if (ctrl->found_inf) return;
    const float lr_t = ctrl->lr_t;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 pp = p[i], mm = m[i], vv = v[i], gg = g[i];
        float* P = &pp.x; float* M = &mm.x; float* V = &vv.x; const float* G = &gg.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) adam_update(P[k], M[k], V[k], G[k] * gscale, lr_t, b1, b2, eps);
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
    if (blockIdx.x == 0) {
        for (size_t i = tail0 + threadIdx.x; i < n; i += blockDim.x) {
            float pk = pt[i], mk = mt[i], vk = vt[i];
            adam_update(pk, mk, vk, gt[i] * gscale, lr_t, b1, b2, eps);
            mt[i] = mk; vt[i] = vk; pt[i] = pk;
        }
    }
}
hipError_t launch_adam_guarded(float* p, float* m, float* v, const float* g, size_t n, const void* ctrl, float b1,
                               float b2, float eps, float gscale, hipStream_t s) {
    const size_t n4 = n / 4;
    size_t nb = (n4 + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(adam_guarded_kernel, dim3((unsigned)nb), dim3(256), 0, s, (float4*)p, (float4*)m, (float4*)v,
                       (const float4*)g, n4, (const OptCtrl*)ctrl, b1, b2, eps, gscale, p, m, v, g, n4 * 4, n);
    return hipGetLastError();
}
__global__ void momentum_guarded_kernel(float* p, float* acc, const float* g, size_t n, const OptCtrl* ctrl, float lr,
                                        float mom, float gscale) {
    if (ctrl->found_inf) return;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float pk = p[i], a = acc[i];
        momentum_update(pk, a, g[i] * gscale, lr, mom);
        acc[i] = a;
        p[i] = pk;
    }
}
hipError_t launch_momentum_guarded(float* p, float* acc, const float* g, size_t n, const void* ctrl, float lr, float mom,
                                   float gscale, hipStream_t s) {
    size_t nb = (n + 255) / 256;
    if (nb > 8192) nb = 8192;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(momentum_guarded_kernel, dim3((unsigned)nb), dim3(256), 0, s, p, acc, g, n, (const OptCtrl*)ctrl,
                       lr, mom, gscale);
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
