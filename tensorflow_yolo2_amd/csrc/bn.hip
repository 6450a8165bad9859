// Batch-norm statistics merge, fused BN-apply + leaky(0.1) + 2x2 max-pool pass,
// and the two BN-backward passes.  All HBM-bound: 16-byte vector accesses,
// one read of every operand, outputs written straight into the next consumer's
// zero-bordered NHWC layout.
//
// Reference semantics: tf.layers.batch_normalization(center, scale, training)
// + tf.maximum(0.1*h, h) + tf.nn.max_pool(2,2,'SAME')
// (src/yolo2_nets/darknet.py:24-25,39-46); momentum 0.99, eps 1e-3 (TF defaults).
#include "common.h"
#include "kernels.h"

namespace y2 {

// ---------------------------------------------------------------------------
// merge per-block (count, mean, M2) partials -> batch mean / biased variance
// ---------------------------------------------------------------------------
// block = 8 channels x 128 slices of the partial list; loads are unrolled so the
// (latency-bound) walk over P partials keeps 4 independent loads in flight per thread
constexpr int kFinCh = 8, kFinSl = 128;

// sum over the 128 slices of one channel: 3 in-wave shuffle steps (a wave holds 8 channels x 8
// slices), then 16 per-wave partials through LDS -- two barriers instead of a 7-level tree
Y2_DEV double fin_block_sum(double v, double (*red)[kFinCh], int sl, int cl) {
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    const int wave = threadIdx.x >> 6;
    __syncthreads();   // previous use of `red` finished
    if ((threadIdx.x & 63) < kFinCh) red[wave][cl] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w][cl];
    return t;
}

__global__ __launch_bounds__(1024) void bn_finalize_kernel(BnFinalizeArgs a) {
    __shared__ double red[kFinSl][kFinCh];
    const int cl = threadIdx.x % kFinCh, sl = threadIdx.x / kFinCh;
    const int c = blockIdx.x * kFinCh + cl;
    const bool cv = c < a.C;
    const int cc = cv ? c : 0;
    // ONE pass over the partials, in double, about a per-channel shift s (the first partial's
    // mean): N = sum k, A = sum k (m - s), B = sum [M2 + k (m - s)^2]; then
    // mean = s + A/N and M2 = B - A^2/N.  With s within a few sigma of the mean the
    // subtraction loses nothing in double precision.
    const double sft = a.P > 0 ? (double)a.part_mean[cc] : 0.0;
    double n4[4] = {0, 0, 0, 0}, a4[4] = {0, 0, 0, 0}, b4[4] = {0, 0, 0, 0};
    int p = sl;
    for (; p + 3 * kFinSl < a.P; p += 4 * kFinSl) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t q = (size_t)(p + u * kFinSl) * a.ldp + cc;
            const double k = a.part_cnt[p + u * kFinSl];
            const double d = (double)a.part_mean[q] - sft;
            n4[u] += k;
            a4[u] += k * d;
            b4[u] += (double)a.part_m2[q] + k * d * d;
        }
    }
    for (; p < a.P; p += kFinSl) {
        const size_t q = (size_t)p * a.ldp + cc;
        const double k = a.part_cnt[p];
        const double d = (double)a.part_mean[q] - sft;
        n4[0] += k;
        a4[0] += k * d;
        b4[0] += (double)a.part_m2[q] + k * d * d;
    }
    const double ntot = fin_block_sum((n4[0] + n4[1]) + (n4[2] + n4[3]), red, sl, cl);
    const double atot = fin_block_sum((a4[0] + a4[1]) + (a4[2] + a4[3]), red, sl, cl);
    const double btot = fin_block_sum((b4[0] + b4[1]) + (b4[2] + b4[3]), red, sl, cl);
    const double mean = ntot > 0 ? sft + atot / ntot : 0.0;
    double mt = ntot > 0 ? btot - atot * atot / ntot : 0.0;
    if (mt < 0) mt = 0;
    if (sl == 0 && cv) {
        const float var = (float)(ntot > 0 ? mt / ntot : 0.0);
        const float meanf = (float)mean;
        const float inv = 1.0f / sqrtf(var + a.eps);
        const float sc = a.gamma[c] * inv;
        a.scale[c] = sc;
        a.shift[c] = a.beta[c] - meanf * sc;
        a.mean[c] = meanf;
        a.invstd[c] = inv;
        if (a.update_moving) {
            float vu = var;
            if (a.bessel && ntot > 1.0) vu = (float)(mt / (ntot - 1.0));
            const float dec = 1.0f - a.momentum;
            a.moving_mean[c] -= (a.moving_mean[c] - meanf) * dec;
            a.moving_var[c] -= (a.moving_var[c] - vu) * dec;
        }
    }
}

hipError_t launch_bn_finalize(const BnFinalizeArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((a.C + kFinCh - 1) / kFinCh), dim3(1024), 0, s, a);
    return hipGetLastError();
}

__global__ void bn_infer_prepare_kernel(const float* gamma, const float* beta, const float* mm, const float* mv,
                                        float* scale, float* shift, float* mean, float* invstd, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float inv = 1.0f / sqrtf(mv[c] + eps);
    const float sc = gamma[c] * inv;
    scale[c] = sc;
    shift[c] = beta[c] - mm[c] * sc;
    mean[c] = mm[c];
    invstd[c] = inv;
}
hipError_t launch_bn_infer_prepare(const float* gamma, const float* beta, const float* mm, const float* mv,
                                   float* scale, float* shift, float* mean, float* invstd, int C, float eps,
                                   hipStream_t s) {
    hipLaunchKernelGGL(bn_infer_prepare_kernel, dim3((C + 255) / 256), dim3(256), 0, s, gamma, beta, mm, mv, scale,
                       shift, mean, invstd, C, eps);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// forward: out = maxpool2x2?( leaky( y*scale + shift ) )
// ---------------------------------------------------------------------------
template <typename T, bool POOL, bool OUTF32>
__global__ __launch_bounds__(256) void bn_act_kernel(BnActArgs a) {
    constexpr int EPC = 16 / sizeof(T);
    const int cpr = a.ldy / EPC;
    const int Ho = POOL ? (a.H + 1) / 2 : a.H, Wo = POOL ? (a.W + 1) / 2 : a.W;
    const size_t total = (size_t)a.N * Ho * Wo * cpr;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % cpr);
        const size_t po = idx / cpr;
        const int wo = (int)(po % Wo);
        const int ho = (int)((po / Wo) % Ho);
        const int n = (int)(po / ((size_t)Wo * Ho));
        const int c0 = ch * EPC;
        if (!OUTF32 && c0 >= a.C) continue;
        float sc[EPC], sh[EPC], r[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            sc[e] = a.scale[c0 + e];
            sh[e] = a.shift[c0 + e];
        }
        if (POOL) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) r[e] = -INFINITY;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int hi = 2 * ho + (d >> 1), wi = 2 * wo + (d & 1);
                if (hi < a.H && wi < a.W) {
                    Chunk<T> v = ld_chunk<T>((const char*)a.y +
                                             (((size_t)(n * a.H + hi) * a.W + wi) * a.ldy + c0) * sizeof(T));
#pragma unroll
                    for (int e = 0; e < EPC; ++e)
                        r[e] = fmaxf(r[e], leaky01(Elem<T>::to_f32(v.v[e]) * sc[e] + sh[e]));
                }
            }
        } else {
            Chunk<T> v = ld_chunk<T>((const char*)a.y + ((po * a.ldy) + c0) * sizeof(T));
#pragma unroll
            for (int e = 0; e < EPC; ++e) r[e] = leaky01(Elem<T>::to_f32(v.v[e]) * sc[e] + sh[e]);
        }
        if (OUTF32) {
            float* o = (float*)a.out + po * a.C;
#pragma unroll
            for (int e = 0; e < EPC; ++e)
                if (c0 + e < a.C) o[c0 + e] = r[e];
        } else {
            Chunk<T> o;
#pragma unroll
            for (int e = 0; e < EPC; ++e) o.v[e] = Elem<T>::from_f32(r[e]);
            const size_t off = (bpix(n, ho, wo, Ho, Wo) * a.C + c0) * sizeof(T);
            st_chunk<T>((char*)a.out + off, o);
        }
    }
}

template <typename T>
static hipError_t bn_act_T(const BnActArgs& a, hipStream_t s) {
    constexpr int EPC = 16 / sizeof(T);
    const int Ho = a.pool ? (a.H + 1) / 2 : a.H, Wo = a.pool ? (a.W + 1) / 2 : a.W;
    const size_t total = (size_t)a.N * Ho * Wo * (a.ldy / EPC);
    size_t nb = (total + 255) / 256;
    if (nb > 256 * 16) nb = 256 * 16;
    if (nb == 0) nb = 1;
    dim3 g((unsigned)nb), b(256);
    if (a.pool) {
        if (a.out_f32) hipLaunchKernelGGL((bn_act_kernel<T, true, true>), g, b, 0, s, a);
        else hipLaunchKernelGGL((bn_act_kernel<T, true, false>), g, b, 0, s, a);
    } else {
        if (a.out_f32) hipLaunchKernelGGL((bn_act_kernel<T, false, true>), g, b, 0, s, a);
        else hipLaunchKernelGGL((bn_act_kernel<T, false, false>), g, b, 0, s, a);
    }
    return hipGetLastError();
}
hipError_t launch_bn_act(int dtype, const BnActArgs& a, hipStream_t s) {
    switch (dtype) {
        case 0: return bn_act_T<float>(a, s);
        case 1: return bn_act_T<half_t>(a, s);
        case 2: return bn_act_T<bf16_t>(a, s);
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------
// backward.  For one output pixel (pooled resolution when POOL) and EPC channels:
//   z_d   = y_d*scale + shift              (d = the 1 or 4 input pixels)
//   dz_d  = dA * slope(z_d) at the first arg-max d (row-major), 0 elsewhere
//   pass 1: S1 += dz, S2 += dz * xhat      (xhat = (y - mean)*invstd)
//   pass 2: dy_d = scale * (dz_d - c1 - xhat_d*c2)   c1 = S1/M, c2 = S2/M
// ---------------------------------------------------------------------------
struct BwdGeom {
    int cpr, CT, rows, Ho, Wo;
    size_t mout;
};
template <typename T>
__host__ __device__ inline BwdGeom bwd_geom(const BnBwdArgs& a) {
    constexpr int EPC = 16 / sizeof(T);
    BwdGeom g;
    g.cpr = a.ldy / EPC;
    g.CT = g.cpr;  // <= 256 for every layer of this network
    g.rows = 256 / g.CT;
    g.Ho = a.pool ? (a.H + 1) / 2 : a.H;
    g.Wo = a.pool ? (a.W + 1) / 2 : a.W;
    g.mout = (size_t)a.N * g.Ho * g.Wo;
    return g;
}

template <typename T, bool POOL, bool APPLY>
__global__ __launch_bounds__(256) void bn_bwd_kernel(BnBwdArgs a) {
    constexpr int EPC = 16 / sizeof(T);
    __shared__ float red[256 * 8 * 2 / 2 + 8];  // [rows][CT][EPC] * up to 2 sums (EPC<=8, CT*rows<=256)
    const BwdGeom g = bwd_geom<T>(a);
    const int tid = threadIdx.x;
    const bool active = tid < g.CT * g.rows;
    const int ch = tid % g.CT, row = tid / g.CT;
    const int c0 = ch * EPC;
    const size_t per_blk = (g.mout + gridDim.x - 1) / gridDim.x;
    const size_t p_begin = (size_t)blockIdx.x * per_blk;
    size_t p_end = p_begin + per_blk;
    if (p_end > g.mout) p_end = g.mout;

    float sc[EPC], sh[EPC], mu[EPC], is[EPC], c1[EPC], c2[EPC], s1[EPC], s2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        sc[e] = sh[e] = mu[e] = is[e] = c1[e] = c2[e] = 0.f;
        s1[e] = s2[e] = 0.f;
    }
    if (active) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            sc[e] = a.scale[c0 + e];
            sh[e] = a.shift[c0 + e];
            mu[e] = a.mean[c0 + e];
            is[e] = a.invstd[c0 + e];
            if (APPLY) {
                c1[e] = a.coef[c0 + e];
                c2[e] = a.coef[a.ldy + c0 + e];
            }
        }
        for (size_t po = p_begin + row; po < p_end; po += g.rows) {
            const int wo = (int)(po % g.Wo);
            const int ho = (int)((po / g.Wo) % g.Ho);
            const int n = (int)(po / ((size_t)g.Wo * g.Ho));
            Chunk<T> dav = ld_chunk<T>((const char*)a.dA + (po * a.ldd + c0) * sizeof(T));
            float da[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) da[e] = Elem<T>::to_f32(dav.v[e]);
            if (POOL) {
                float yv[4][EPC];
                bool valid[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int hi = 2 * ho + (d >> 1), wi = 2 * wo + (d & 1);
                    valid[d] = hi < a.H && wi < a.W;
                    if (valid[d]) {
                        Chunk<T> v = ld_chunk<T>((const char*)a.y +
                                                 (((size_t)(n * a.H + hi) * a.W + wi) * a.ldy + c0) * sizeof(T));
#pragma unroll
                        for (int e = 0; e < EPC; ++e) yv[d][e] = Elem<T>::to_f32(v.v[e]);
                    }
                }
                int arg[EPC];
                float zmax[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    arg[e] = 0;
                    zmax[e] = -INFINITY;
                }
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    if (valid[d]) {
#pragma unroll
                        for (int e = 0; e < EPC; ++e) {
                            // the pooled quantity is leaky(z); leaky is strictly increasing
                            const float act = leaky01(yv[d][e] * sc[e] + sh[e]);
                            if (act > zmax[e]) {
                                zmax[e] = act;
                                arg[e] = d;
                            }
                        }
                    }
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    if (!valid[d]) continue;
                    const int hi = 2 * ho + (d >> 1), wi = 2 * wo + (d & 1);
                    Chunk<T> o;
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        const float z = yv[d][e] * sc[e] + sh[e];
                        const float dz = (arg[e] == d) ? da[e] * leaky01_slope(z) : 0.f;
                        const float xh = (yv[d][e] - mu[e]) * is[e];
                        if (APPLY) {
                            const float dy = sc[e] * (dz - c1[e] - xh * c2[e]);
                            o.v[e] = Elem<T>::from_f32(dy);
                            s1[e] += dy;
                        } else {
                            s1[e] += dz;
                            s2[e] += dz * xh;
                        }
                    }
                    if (APPLY) {
                        const size_t off = (bpix(n, hi, wi, a.H, a.W) * a.ldy + c0) * sizeof(T);
                        st_chunk<T>((char*)a.dyp + off, o);
                    }
                }
            } else {
                Chunk<T> v = ld_chunk<T>((const char*)a.y + (po * a.ldy + c0) * sizeof(T));
                Chunk<T> o;
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float yv = Elem<T>::to_f32(v.v[e]);
                    const float z = yv * sc[e] + sh[e];
                    const float dz = da[e] * leaky01_slope(z);
                    const float xh = (yv - mu[e]) * is[e];
                    if (APPLY) {
                        const float dy = sc[e] * (dz - c1[e] - xh * c2[e]);
                        o.v[e] = Elem<T>::from_f32(dy);
                        s1[e] += dy;
                    } else {
                        s1[e] += dz;
                        s2[e] += dz * xh;
                    }
                }
                if (APPLY) {
                    const size_t off = (bpix(n, ho, wo, a.H, a.W) * a.ldy + c0) * sizeof(T);
                    st_chunk<T>((char*)a.dyp + off, o);
                }
            }
        }
    }
    // ---- block reduction over `rows`
    constexpr int NS = APPLY ? 1 : 2;
    for (int k = 0; k < NS; ++k) {
        __syncthreads();
        if (active) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) red[(row * g.CT + ch) * EPC + e] = (k == 0) ? s1[e] : s2[e];
        }
        __syncthreads();
        if (active && row == 0) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                float t = 0.f;
                for (int r = 0; r < g.rows; ++r) t += red[(r * g.CT + ch) * EPC + e];
                // APPLY: slot 0 holds the block's sum(dy) (conv-bias gradient partial)
                if (c0 + e < a.C) a.psum[((size_t)blockIdx.x * 2 + k) * a.ldy + c0 + e] = t;
            }
        }
    }
}

// mode 0: after the reduce pass -> dbeta, dgamma, coef.  mode 1: after the apply pass -> dbias.
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(BnBwdArgs a, int mode, int P) {
    __shared__ double red[kFinSl][kFinCh];
    const int cl = threadIdx.x % kFinCh, sl = threadIdx.x / kFinCh;
    const int c = blockIdx.x * kFinCh + cl;
    const bool cv = c < a.C;
    const int cc = cv ? c : 0;
    double t[2] = {0.0, 0.0};
    const int nk = mode == 0 ? 2 : 1;
    for (int k = 0; k < nk; ++k) {
        double v4[4] = {0, 0, 0, 0};
        int p = sl;
        for (; p + 3 * kFinSl < P; p += 4 * kFinSl) {
#pragma unroll
            for (int u = 0; u < 4; ++u) v4[u] += (double)a.psum[((size_t)(p + u * kFinSl) * 2 + k) * a.ldy + cc];
        }
        for (; p < P; p += kFinSl) v4[0] += (double)a.psum[((size_t)p * 2 + k) * a.ldy + cc];
        t[k] = fin_block_sum((v4[0] + v4[1]) + (v4[2] + v4[3]), red, sl, cl);
    }
    if (sl == 0 && cv) {
        if (mode == 0) {
            const double m = (double)a.N * a.H * a.W;
            a.dbeta[c] = (float)(t[0] * a.inv_grad_scale);
            a.dgamma[c] = (float)(t[1] * a.inv_grad_scale);
            a.coef[c] = a.training ? (float)(t[0] / m) : 0.f;
            a.coef[a.ldy + c] = a.training ? (float)(t[1] / m) : 0.f;
        } else {
            a.dbias[c] = (float)(t[0] * a.inv_grad_scale);
        }
    }
}

template <typename T>
static int bwd_blocks(const BnBwdArgs& a) {
    const BwdGeom g = bwd_geom<T>(a);
    size_t nb = (g.mout + (size_t)g.rows * 8 - 1) / ((size_t)g.rows * 8);  // >= 8 pixels per thread row
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    return (int)nb;
}
int bn_bwd_partials(const BnBwdArgs& a) { return 2048; }

template <typename T, bool APPLY>
static hipError_t bn_bwd_T(const BnBwdArgs& a, hipStream_t s) {
    dim3 g(APPLY ? bwd_blocks<T>(a) : a.P), b(256);
    if (a.pool) hipLaunchKernelGGL((bn_bwd_kernel<T, true, APPLY>), g, b, 0, s, a);
    else hipLaunchKernelGGL((bn_bwd_kernel<T, false, APPLY>), g, b, 0, s, a);
    return hipGetLastError();
}
hipError_t launch_bn_bwd_reduce(int dtype, BnBwdArgs& a, hipStream_t s) {
    switch (dtype) {
        case 0: a.P = bwd_blocks<float>(a); return bn_bwd_T<float, false>(a, s);
        case 1: a.P = bwd_blocks<half_t>(a); return bn_bwd_T<half_t, false>(a, s);
        case 2: a.P = bwd_blocks<bf16_t>(a); return bn_bwd_T<bf16_t, false>(a, s);
    }
    return hipErrorInvalidValue;
}
hipError_t launch_bn_bwd_finalize(const BnBwdArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((a.C + kFinCh - 1) / kFinCh), dim3(1024), 0, s, a, 0, a.P);
    return hipGetLastError();
}
hipError_t launch_bn_bwd_dbias(const BnBwdArgs& a, int P, hipStream_t s) {
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((a.C + kFinCh - 1) / kFinCh), dim3(1024), 0, s, a, 1, P);
    return hipGetLastError();
}
hipError_t launch_bn_bwd_apply(int dtype, const BnBwdArgs& a, hipStream_t s) {
    hipError_t e = hipErrorInvalidValue;
    int P = 0;
    switch (dtype) {
        case 0: P = bwd_blocks<float>(a); e = bn_bwd_T<float, true>(a, s); break;
        case 1: P = bwd_blocks<half_t>(a); e = bn_bwd_T<half_t, true>(a, s); break;
        case 2: P = bwd_blocks<bf16_t>(a); e = bn_bwd_T<bf16_t, true>(a, s); break;
    }
    if (e != hipSuccess) return e;
    if (a.dbias)   // conv-bias gradient = sum(dy): block partials -> one tiny reduction (no contended atomics)
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((a.C + kFinCh - 1) / kFinCh), dim3(1024), 0, s, a, 1, P);
    return hipGetLastError();
}

}  // namespace y2
