// Batch-norm statistics merge, fused BN-apply + leaky(0.1) + 2x2 max-pool pass,
// and the two BN-backward passes.  All HBM-bound: 16-byte vector accesses,
// one read of every operand, outputs written straight into the next consumer's
// zero-bordered NHWC layout.
//
// Reference semantics: tf.layers.batch_normalization(center, scale, training)
// + tf.maximum(0.1*h, h) + tf.nn.max_pool(2,2,'SAME')
// (src/yolo2_nets/darknet.py:24-25,39-46); momentum 0.99, eps 1e-3 (TF defaults).
#include <stdlib.h>
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
template <int NWV = kFinSl * kFinCh / 64>
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
    for (int w = 0; w < NWV; ++w) t += red[w][cl];
    return t;
}

template <int SLN>
__global__ __launch_bounds__(SLN * kFinCh) void bn_finalize_kernel(BnFinalizeArgs a) {
    constexpr int kFinSl = SLN;   // slices of the partial list per block
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
    const double ntot = fin_block_sum<kFinSl * kFinCh / 64>((n4[0] + n4[1]) + (n4[2] + n4[3]), red, sl, cl);
    const double atot = fin_block_sum<kFinSl * kFinCh / 64>((a4[0] + a4[1]) + (a4[2] + a4[3]), red, sl, cl);
    const double btot = fin_block_sum<kFinSl * kFinCh / 64>((b4[0] + b4[1]) + (b4[2] + b4[3]), red, sl, cl);
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
        float vu = var;
        if (a.bessel && ntot > 1.0) vu = (float)(mt / (ntot - 1.0));
        if (a.var) a.var[c] = vu;
        if (a.update_moving) {
            const float dec = 1.0f - a.momentum;
            a.moving_mean[c] -= (a.moving_mean[c] - meanf) * dec;
            a.moving_var[c] -= (a.moving_var[c] - vu) * dec;
        }
    }
}

// Long partial lists (the 208x208 / 104x104 layers: thousands of records, only C / 8 blocks in the merge above) are
// first compressed to kCompress records by a 2-D grid: thread = channel, block = (slice of the list, 64 channels),
// the same shifted-sum merge in double, emitted as (count, mean, M2) float records the merge above then takes.
constexpr int kCompress = 64;
__global__ __launch_bounds__(64) void bn_compress_kernel(const float* __restrict__ cnt, const float* __restrict__ mean,
                                                         const float* __restrict__ m2, int P, int C, int ldp, float* ocnt,
                                                         float* omean, float* om2) {
    const int c = blockIdx.y * 64 + threadIdx.x;
    const int sl = blockIdx.x;
    const int per = (P + kCompress - 1) / kCompress;
    const int p0 = sl * per;
    int p1 = p0 + per;
    if (p1 > P) p1 = P;
    if (c >= C) return;
    const double sft = p0 < p1 ? (double)mean[(size_t)p0 * ldp + c] : 0.0;
    double n4[4] = {0, 0, 0, 0}, a4[4] = {0, 0, 0, 0}, b4[4] = {0, 0, 0, 0};
    int p = p0;
    for (; p + 3 < p1; p += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double k = cnt[p + u];
            const double d = (double)mean[(size_t)(p + u) * ldp + c] - sft;
            n4[u] += k;
            a4[u] += k * d;
            b4[u] += (double)m2[(size_t)(p + u) * ldp + c] + k * d * d;
        }
    }
    for (; p < p1; ++p) {
        const double k = cnt[p];
        const double d = (double)mean[(size_t)p * ldp + c] - sft;
        n4[0] += k;
        a4[0] += k * d;
        b4[0] += (double)m2[(size_t)p * ldp + c] + k * d * d;
    }
    const double n = (n4[0] + n4[1]) + (n4[2] + n4[3]), A = (a4[0] + a4[1]) + (a4[2] + a4[3]),
                 B = (b4[0] + b4[1]) + (b4[2] + b4[3]);
    double mt = n > 0 ? B - A * A / n : 0.0;
    if (mt < 0) mt = 0;
    omean[(size_t)sl * ldp + c] = (float)(n > 0 ? sft + A / n : 0.0);
    om2[(size_t)sl * ldp + c] = (float)mt;
    if (c == 0 || threadIdx.x == 0) ocnt[sl] = (float)n;     // every channel block writes the same count
}

hipError_t launch_bn_finalize(const BnFinalizeArgs& a0, hipStream_t s) {
    BnFinalizeArgs a = a0;
    if (a.P > 4096 && a.scratch) {   // measured: pays only for the 10,816-record list of the 208x208 layer
        float* ocnt = a.scratch;
        float* omean = ocnt + kCompress;
        float* om2 = omean + (size_t)kCompress * a.ldp;
        hipLaunchKernelGGL(bn_compress_kernel, dim3(kCompress, (a.C + 63) / 64), dim3(64), 0, s, a.part_cnt, a.part_mean,
                           a.part_m2, a.P, a.C, a.ldp, ocnt, omean, om2);
        a.part_cnt = ocnt; a.part_mean = omean; a.part_m2 = om2; a.P = kCompress;
    }
    // short partial lists (the 13x13 / 26x26 layers: a few dozen records) finish sooner in 4-wave blocks
    if (a.P <= 512) hipLaunchKernelGGL(bn_finalize_kernel<32>, dim3((a.C + kFinCh - 1) / kFinCh), dim3(32 * kFinCh), 0, s, a);
    else hipLaunchKernelGGL(bn_finalize_kernel<128>, dim3((a.C + kFinCh - 1) / kFinCh), dim3(128 * kFinCh), 0, s, a);
    return hipGetLastError();
}

// deferred form of the update above (y2_update_moving_stats): same expressions on the saved batch statistics
__global__ void bn_update_moving_kernel(const float* mean, const float* var, float* mm, float* mv, int C, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float dec = 1.0f - momentum;
    mm[c] -= (mm[c] - mean[c]) * dec;
    mv[c] -= (mv[c] - var[c]) * dec;
}
hipError_t launch_bn_update_moving(const float* mean, const float* var, float* mm, float* mv, int C, float momentum,
                                   hipStream_t s) {
    hipLaunchKernelGGL(bn_update_moving_kernel, dim3((C + 255) / 256), dim3(256), 0, s, mean, var, mm, mv, C, momentum);
    return hipGetLastError();
}

// every inference-mode layer of a forward pass in ONE launch (blockIdx.y = layer): 18 launches of ~5 us each were 7 % of
// the batch-32 core forward
__global__ void bn_infer_prepare_all_kernel(const BnInferLayer* __restrict__ tab, int train_core, int train_head,
                                            float eps) {
    const BnInferLayer L = tab[blockIdx.y];
    if (L.is_core ? train_core : train_head) return;      // this layer normalises with its batch statistics
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= L.C) return;
    const float inv = 1.0f / sqrtf(L.mv[c] + eps);
    const float sc = L.gamma[c] * inv;
    L.scale[c] = sc;
    L.shift[c] = L.beta[c] - L.mm[c] * sc;
    L.mean[c] = L.mm[c];
    L.invstd[c] = inv;
}
hipError_t launch_bn_infer_prepare_all(const BnInferLayer* tab, int nlayers, int max_c, int train_core, int train_head,
                                       float eps, hipStream_t s) {
    hipLaunchKernelGGL(bn_infer_prepare_all_kernel, dim3((max_c + 255) / 256, nlayers), dim3(256), 0, s, tab, train_core,
                       train_head, eps);
    return hipGetLastError();
}

// one 16-byte chunk of channels c0.. of cell `cell` of a bordered tensor with C channels per cell.  SPLIT (f16x2 mode,
// T = float: four channels): the cell is [C halves hi][C halves lo] (common.h hsplit_t), 8 bytes go to each plane
// hi_only (f16x2f, backward passes: BnBwdArgs::hi_only): the consumers of this dY -- the dgrad and the weight gradient -- read
// its hi plane alone, so the lo plane is not written (2 of the 4 bytes per element)
template <typename T, bool SPLIT>
Y2_DEV void st_act(char* base, size_t cell, int C, int c0, const float* r, int hi_only = 0) {
    if constexpr (SPLIT) {
        static_assert(sizeof(T) == 4, "the split store takes fp32 chunks");
        if (hi_only) {
            *(u32x2*)(base + cell * (size_t)C * 4 + (size_t)c0 * 2) = u32x2{pack2<half_t>(r[0], r[1]), pack2<half_t>(r[2], r[3])};
            return;
        }
        st_split4(base + cell * (size_t)C * 4, C, c0, r);
    } else {
        constexpr int EPC = 16 / sizeof(T);
        Chunk<T> o;
#pragma unroll
        for (int e = 0; e < EPC; ++e) o.v[e] = Elem<T>::from_f32(r[e]);
        st_chunk<T>(base + (cell * (size_t)C + c0) * sizeof(T), o);
    }
}

// one chunk of dA (BnBwdArgs): EPC elements of T -- or, in the fp32-wide kernels of the f16x2f mode, four halves widened
// (dA_half: the dgrad above stored its output in f16, common.h hsplithh_t)
template <typename T>
Y2_DEV Chunk<T> ld_dA(const BnBwdArgs& a, size_t po, int c0) {
    if constexpr (sizeof(T) == 4) {
        if (a.dA_half) {
            half_t h[4];
            *(u32x2*)h = *(const u32x2*)((const char*)a.dA + (po * a.ldd + c0) * 2);
            Chunk<T> c;
#pragma unroll
            for (int e = 0; e < 4; ++e) c.v[e] = (float)h[e];
            return c;
        }
    }
    return ld_chunk<T>((const char*)a.dA + (po * a.ldd + c0) * sizeof(T));
}

// ---------------------------------------------------------------------------
// forward: out = maxpool2x2?( leaky( y*scale + shift ) )
// One 2x2 window of a pooled layer's apply pass: r = the activation's maximum, ys = the conv output at its first
// arg-max in row-major window order.  Whole windows (every even-sized map): the four loads are issued before the first
// is used, the maximum is taken on z = y * scale + shift (the activation max(slope z, z), slope in [0, 1], is
// non-decreasing: leaky(max z) = max leaky(z) bit for bit) and the position found by equality -- 6 vector operations
// per element for the search instead of 5 per position in a load-compare-select chain.
template <typename T>
Y2_DEV void pool_window(const BnActArgs& a, int n, int ho, int wo, int c0, const float* sc, const float* sh, float* r,
                        Chunk<T>& ys) {
    constexpr int EPC = 16 / sizeof(T);
    const int hi0 = 2 * ho, wi0 = 2 * wo;
    if (a.pool == 2) {       // subsample (kernels.h BnActArgs::pool): the window's position 0, nothing compared
        const Chunk<T> v0 = ld_chunk<T>((const char*)a.y + (((size_t)(n * a.H + hi0) * a.W + wi0) * a.ldy + c0) * sizeof(T));
#pragma unroll
        for (int e = 0; e < EPC; ++e) r[e] = leaky_s(Elem<T>::to_f32(v0.v[e]) * sc[e] + sh[e], a.slope);
        ys = v0;
        return;
    }
    if (hi0 + 1 < a.H && wi0 + 1 < a.W) {
        const char* p = (const char*)a.y + (((size_t)(n * a.H + hi0) * a.W + wi0) * a.ldy + c0) * sizeof(T);
        const size_t pxB = (size_t)a.ldy * sizeof(T), rowB = (size_t)a.W * pxB;
        const Chunk<T> v0 = ld_chunk<T>(p), v1 = ld_chunk<T>(p + pxB), v2 = ld_chunk<T>(p + rowB), v3 = ld_chunk<T>(p + rowB + pxB);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float z0 = Elem<T>::to_f32(v0.v[e]) * sc[e] + sh[e], z1 = Elem<T>::to_f32(v1.v[e]) * sc[e] + sh[e];
            const float z2 = Elem<T>::to_f32(v2.v[e]) * sc[e] + sh[e], z3 = Elem<T>::to_f32(v3.v[e]) * sc[e] + sh[e];
            const float zm = fmaxf(fmaxf(fmaxf(-INFINITY, z0), z1), fmaxf(z2, z3));     // NaN: skipped, as by `>`
            T y = z2 == zm ? v2.v[e] : v3.v[e];
            y = z1 == zm ? v1.v[e] : y;
            y = z0 == zm ? v0.v[e] : y;
            ys.v[e] = y;
            r[e] = leaky_s(zm, a.slope);
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) r[e] = -INFINITY;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int hi = hi0 + (d >> 1), wi = wi0 + (d & 1);
        if (hi < a.H && wi < a.W) {
            Chunk<T> v = ld_chunk<T>((const char*)a.y + (((size_t)(n * a.H + hi) * a.W + wi) * a.ldy + c0) * sizeof(T));
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float act = leaky_s(Elem<T>::to_f32(v.v[e]) * sc[e] + sh[e], a.slope);
                if (act > r[e]) { r[e] = act; ys.v[e] = v.v[e]; }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// block = (256 / cpr) pixel rows x cpr 16-byte channel chunks over a contiguous pixel range: the
// channel chunk of a thread is fixed (scale/shift live in registers) and (n, ho, wo) advance
// incrementally -- no per-element div/mod (64-bit ones cost more than the pass's arithmetic)
template <typename T, bool POOL, bool OUTF32, bool SPLIT = false>
__global__ __launch_bounds__(256) void bn_act_kernel(BnActArgs a) {
    constexpr int EPC = 16 / sizeof(T);
    const int cpr = a.ldy / EPC;           // <= 256 (checked by the launcher)
    const int rows = 256 / cpr;
    const int Ho = POOL ? (a.H + 1) / 2 : a.H, Wo = POOL ? (a.W + 1) / 2 : a.W;
    const uint32_t mout = (uint32_t)a.N * Ho * Wo;
    const int tid = threadIdx.x;
    const int ch = tid % cpr, row = tid / cpr;
    const int c0 = ch * EPC;
    if (row >= rows || (!OUTF32 && c0 >= a.C)) return;
    const uint32_t per_blk = (mout + gridDim.x - 1) / gridDim.x;
    const uint32_t p_begin = blockIdx.x * per_blk;
    uint32_t p_end = p_begin + per_blk;
    if (p_end > mout) p_end = mout;
    float sc[EPC], sh[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        sc[e] = a.scale[c0 + e];
        sh[e] = a.shift[c0 + e];
    }
    int wo, ho, n;
    {
        const uint32_t p0 = p_begin + row;
        wo = (int)(p0 % (uint32_t)Wo);
        const uint32_t q = p0 / (uint32_t)Wo;
        ho = (int)(q % (uint32_t)Ho);
        n = (int)(q / (uint32_t)Ho);
    }
    const int drow = rows / Wo, dcol = rows % Wo;
    for (uint32_t po = p_begin + row; po < p_end; po += rows, wo += dcol, ho += drow) {
        if (wo >= Wo) { wo -= Wo; ++ho; }
        while (ho >= Ho) { ho -= Ho; ++n; }
        float r[EPC];
        if (POOL) {
            Chunk<T> ys;   // conv output at the first arg-max of the window (backward: BnActArgs::ysel)
            pool_window<T>(a, n, ho, wo, c0, sc, sh, r, ys);
            if (a.ysel) st_chunk<T>((char*)a.ysel + ((size_t)po * a.ldy + c0) * sizeof(T), ys);
        } else {
            Chunk<T> v = ld_chunk<T>((const char*)a.y + (((size_t)po * a.ldy) + c0) * sizeof(T));
#pragma unroll
            for (int e = 0; e < EPC; ++e) r[e] = leaky_s(Elem<T>::to_f32(v.v[e]) * sc[e] + sh[e], a.slope);
        }
        if (a.join_t) {     // join from a bordered tensor of T (whole chunks: the launcher checks C % EPC == 0)
            if (c0 < a.C) {
                const Chunk<T> jv = ld_chunk<T>((const char*)a.join_t + (bpix(n, ho, wo, Ho, Wo) * a.C + c0) * sizeof(T));
#pragma unroll
                for (int e = 0; e < EPC; ++e) r[e] = fmaxf(r[e] + Elem<T>::to_f32(jv.v[e]), 0.f);
            }
        }
        if (OUTF32) {
            float* o = (float*)a.out + (size_t)po * a.C;
            if (a.join) {
                const float* j = a.join + (size_t)po * a.C;
#pragma unroll
                for (int e = 0; e < EPC; ++e)
                    if (c0 + e < a.C) o[c0 + e] = fmaxf(r[e] + j[c0 + e], 0.f);
            } else {
#pragma unroll
                for (int e = 0; e < EPC; ++e)
                    if (c0 + e < a.C) o[c0 + e] = r[e];
            }
        } else {
            st_act<T, SPLIT>((char*)a.out, bpix(n, ho, wo, Ho, Wo), a.C, c0, r);
        }
    }
}

// ---------------------------------------------------------------------------
// Short partial lists (the 26x26 / 13x13 layers: P <= 128 records): the merge rides in the apply pass.  A block owns
// a 64-channel slab (grid.y) and a pixel range (grid.x); it first merges the P records of ITS 64 channels (4 slices
// x 64 channels, the shifted-sum merge of bn_finalize_kernel in double, fixed order: every block of a slab gets the
// same bits), the blocks with blockIdx.x == 0 also write scale / shift / mean / invstd / variance and the moving
// statistics for the backward pass; then the slab's pixels as in bn_act_kernel.  One launch and ~6 us of critical
// path less per layer than merge + apply.
// ---------------------------------------------------------------------------
constexpr int kSlab = 64;
Y2_DEV void fin_slab_merge(const BnFinalizeArgs& f, int cbase, double (*red)[kSlab][3], float* s_sc, float* s_sh, bool writer) {
    const int c = threadIdx.x & (kSlab - 1), sl = threadIdx.x >> 6;   // 256 threads: 4 slices
    const int cc = cbase + c;
    const bool cv = cc < f.C;
    const int ci = cv ? cc : 0;
    const double sft = f.P > 0 ? (double)f.part_mean[ci] : 0.0;
    double n = 0.0, A = 0.0, B = 0.0;
    int p = sl;
    for (; p + 28 < f.P; p += 32) {           // eight list steps' loads up front (every block of the slab repeats this merge)
        float rk[8], rm[8], r2[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t q = (size_t)(p + 4 * u) * f.ldp + ci;
            rk[u] = f.part_cnt[p + 4 * u];
            rm[u] = f.part_mean[q];
            r2[u] = f.part_m2[q];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double k = rk[u];
            const double d = (double)rm[u] - sft;
            n += k;
            A += k * d;
            B += (double)r2[u] + k * d * d;
        }
    }
    for (; p < f.P; p += 4) {
        const size_t q = (size_t)p * f.ldp + ci;
        const double k = f.part_cnt[p];
        const double d = (double)f.part_mean[q] - sft;
        n += k;
        A += k * d;
        B += (double)f.part_m2[q] + k * d * d;
    }
    red[sl][c][0] = n; red[sl][c][1] = A; red[sl][c][2] = B;
    __syncthreads();
    if (sl == 0) {
        double nt = 0.0, at = 0.0, bt = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { nt += red[k][c][0]; at += red[k][c][1]; bt += red[k][c][2]; }
        const double mean = nt > 0 ? sft + at / nt : 0.0;
        double mt = nt > 0 ? bt - at * at / nt : 0.0;
        if (mt < 0) mt = 0;
        const float var = (float)(nt > 0 ? mt / nt : 0.0);
        const float meanf = (float)mean;
        const float inv = 1.0f / sqrtf(var + f.eps);
        const float sc = cv ? f.gamma[ci] * inv : 0.f;
        const float sh = cv ? f.beta[ci] - meanf * sc : 0.f;
        s_sc[c] = sc;
        s_sh[c] = sh;
        if (writer && cv) {
            f.scale[cc] = sc;
            f.shift[cc] = sh;
            f.mean[cc] = meanf;
            f.invstd[cc] = inv;
            float vu = var;
            if (f.bessel && nt > 1.0) vu = (float)(mt / (nt - 1.0));
            if (f.var) f.var[cc] = vu;
            if (f.update_moving) {
                const float dec = 1.0f - f.momentum;
                f.moving_mean[cc] -= (f.moving_mean[cc] - meanf) * dec;
                f.moving_var[cc] -= (f.moving_var[cc] - vu) * dec;
            }
        }
    }
    __syncthreads();
}

template <typename T, bool POOL, bool SPLIT = false>
__global__ __launch_bounds__(256) void bn_fin_act_kernel(BnActArgs a, BnFinalizeArgs f) {
    constexpr int EPC = 16 / sizeof(T);
    constexpr int CPB = kSlab / EPC;        // chunks per block row
    constexpr int rows = 256 / CPB;
    __shared__ double red[4][kSlab][3];
    __shared__ float s_sc[kSlab], s_sh[kSlab];
    const int cbase = blockIdx.y * kSlab;
    fin_slab_merge(f, cbase, red, s_sc, s_sh, blockIdx.x == 0);
    const int Ho = POOL ? (a.H + 1) / 2 : a.H, Wo = POOL ? (a.W + 1) / 2 : a.W;
    const uint32_t mout = (uint32_t)a.N * Ho * Wo;
    const int tid = threadIdx.x;
    const int ch = tid % CPB, row = tid / CPB;
    const int c0 = cbase + ch * EPC;
    if (c0 >= a.C) return;
    const uint32_t per_blk = (mout + gridDim.x - 1) / gridDim.x;
    const uint32_t p_begin = blockIdx.x * per_blk;
    uint32_t p_end = p_begin + per_blk;
    if (p_end > mout) p_end = mout;
    float sc[EPC], sh[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        sc[e] = s_sc[ch * EPC + e];
        sh[e] = s_sh[ch * EPC + e];
    }
    int wo, ho, n;
    {
        const uint32_t p0 = p_begin + row;
        wo = (int)(p0 % (uint32_t)Wo);
        const uint32_t q = p0 / (uint32_t)Wo;
        ho = (int)(q % (uint32_t)Ho);
        n = (int)(q / (uint32_t)Ho);
    }
    const int drow = rows / Wo, dcol = rows % Wo;
    for (uint32_t po = p_begin + row; po < p_end; po += rows, wo += dcol, ho += drow) {
        if (wo >= Wo) { wo -= Wo; ++ho; }
        while (ho >= Ho) { ho -= Ho; ++n; }
        float r[EPC];
        if (POOL) {
            Chunk<T> ys;
            pool_window<T>(a, n, ho, wo, c0, sc, sh, r, ys);
            if (a.ysel) st_chunk<T>((char*)a.ysel + ((size_t)po * a.ldy + c0) * sizeof(T), ys);
        } else {
            Chunk<T> v = ld_chunk<T>((const char*)a.y + (((size_t)po * a.ldy) + c0) * sizeof(T));
#pragma unroll
            for (int e = 0; e < EPC; ++e) r[e] = leaky_s(Elem<T>::to_f32(v.v[e]) * sc[e] + sh[e], a.slope);
        }
        if (a.join_t) {
            const Chunk<T> jv = ld_chunk<T>((const char*)a.join_t + (bpix(n, ho, wo, Ho, Wo) * a.C + c0) * sizeof(T));
#pragma unroll
            for (int e = 0; e < EPC; ++e) r[e] = fmaxf(r[e] + Elem<T>::to_f32(jv.v[e]), 0.f);
        }
        st_act<T, SPLIT>((char*)a.out, bpix(n, ho, wo, Ho, Wo), a.C, c0, r);
    }
}

bool bn_fin_act_ok(const BnActArgs& a, const BnFinalizeArgs& f) {
    static const int pmax = getenv("Y2DEV_FIN_PMAX") ? atoi(getenv("Y2DEV_FIN_PMAX")) : 128;
    return f.P > 0 && f.P <= pmax && !a.out_f32 && a.ldy % kSlab == 0 && a.C == a.ldy && f.ldp == a.ldy;
}
template <typename T, bool SPLIT = false>
static hipError_t bn_fin_act_T(const BnActArgs& a, const BnFinalizeArgs& f, hipStream_t s) {
    constexpr int EPC = 16 / sizeof(T);
    const int Ho = a.pool ? (a.H + 1) / 2 : a.H, Wo = a.pool ? (a.W + 1) / 2 : a.W;
    const size_t mout = (size_t)a.N * Ho * Wo;
    if (mout >= (1ull << 31)) return hipErrorInvalidValue;
    const int rows = 256 / (kSlab / EPC);
    const int ny = a.ldy / kSlab;
    size_t nx = (mout + (size_t)rows * 4 - 1) / ((size_t)rows * 4);
    const size_t cap = (size_t)(4096 / ny) > 0 ? (size_t)(4096 / ny) : 1;
    if (nx > cap) nx = cap;
    if (nx == 0) nx = 1;
    dim3 g((unsigned)nx, (unsigned)ny), b(256);
    if (a.pool) hipLaunchKernelGGL((bn_fin_act_kernel<T, true, SPLIT>), g, b, 0, s, a, f);
    else hipLaunchKernelGGL((bn_fin_act_kernel<T, false, SPLIT>), g, b, 0, s, a, f);
    return hipGetLastError();
}
hipError_t launch_bn_fin_act(int dtype, const BnActArgs& a, const BnFinalizeArgs& f, hipStream_t s) {
    if (!bn_fin_act_ok(a, f)) return hipErrorInvalidValue;
    switch (dtype) {
        case 0: return bn_fin_act_T<float>(a, f, s);
        case 1: return bn_fin_act_T<half_t>(a, f, s);
        case 2: return bn_fin_act_T<bf16_t>(a, f, s);
        case 3: return bn_fin_act_T<float, true>(a, f, s);
    }
    return hipErrorInvalidValue;
}

template <typename T, bool SPLIT = false>
static hipError_t bn_act_T(const BnActArgs& a, hipStream_t s) {
    constexpr int EPC = 16 / sizeof(T);
    const int Ho = a.pool ? (a.H + 1) / 2 : a.H, Wo = a.pool ? (a.W + 1) / 2 : a.W;
    const int cpr = a.ldy / EPC;
    if (cpr < 1 || cpr > 256) return hipErrorInvalidValue;
    const size_t mout = (size_t)a.N * Ho * Wo;
    if (mout >= (1ull << 31)) return hipErrorInvalidValue;
    const int rows = 256 / cpr;
    size_t nb = (mout + (size_t)rows * 4 - 1) / ((size_t)rows * 4);   // >= 4 pixels per thread row
    if (nb > 256 * 16) nb = 256 * 16;
    if (nb == 0) nb = 1;
    dim3 g((unsigned)nb), b(256);
    if (a.pool) {
        if (a.out_f32) hipLaunchKernelGGL((bn_act_kernel<T, true, true>), g, b, 0, s, a);
        else hipLaunchKernelGGL((bn_act_kernel<T, true, false, SPLIT>), g, b, 0, s, a);
    } else {
        if (a.out_f32) hipLaunchKernelGGL((bn_act_kernel<T, false, true>), g, b, 0, s, a);
        else hipLaunchKernelGGL((bn_act_kernel<T, false, false, SPLIT>), g, b, 0, s, a);
    }
    return hipGetLastError();
}
hipError_t launch_bn_act(int dtype, const BnActArgs& a, hipStream_t s) {
    switch (dtype) {
        case 0: return bn_act_T<float>(a, s);
        case 1: return bn_act_T<half_t>(a, s);
        case 2: return bn_act_T<bf16_t>(a, s);
        case 3: return bn_act_T<float, true>(a, s);
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------
// backward.  For one output pixel (pooled resolution when POOL) and EPC channels:
//   z_d  = y_d*scale + shift                 (d = the 1 or 4 input pixels)
//   g    = dA * leaky'(z) at the first arg-max d of leaky(z_d) (row-major); dz_d = g there, 0 elsewhere
//   pass 1 (reduce): S1 += g, S2 += g*y       -> dbeta = S1, dgamma = invstd*(S2 - mean*S1)
//   pass 2 (apply):  dy_d = scale*(dz_d - c1 - xhat_d*c2) = scale*dz_d - (ka + kb*y_d)
//                    with c1 = S1/M, c2 = dgamma/M, kb = scale*c2*invstd, ka = scale*c1 - kb*mean
// Only the arg-max position carries dz, so pass 1 touches one y per pooled pixel in its sums and
// pass 2 is one fma per element plus the arg-max search: both passes are VALU-lean enough to
// stay HBM-bound at three waves per SIMD.
// The conv-bias gradient sum(dy) is analytic: scale*(S1 - M*c1) (zero up to the rounding of c1
// with batch statistics, scale*S1 with moving statistics) -- no third reduction.
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

template <typename T, bool POOL, bool APPLY, bool SPLIT = false>
__global__ __launch_bounds__(256) void bn_bwd_kernel(BnBwdArgs a) {
    constexpr int EPC = 16 / sizeof(T);
    __shared__ float red[APPLY ? 1 : (256 * 8 + 8)];  // [rows][CT][EPC] (EPC<=8, CT*rows<=256)
    const BwdGeom g = bwd_geom<T>(a);
    const int tid = threadIdx.x;
    const bool active = tid < g.CT * g.rows;
    const int ch = tid % g.CT, row = tid / g.CT;
    const int c0 = ch * EPC;
    // 32-bit pixel indices (N*H*W < 2^31 for anything that fits HBM), (n, ho, wo) advance incrementally
    const uint32_t mout = (uint32_t)g.mout;
    const uint32_t per_blk = (mout + gridDim.x - 1) / gridDim.x;
    const uint32_t p_begin = blockIdx.x * per_blk;
    uint32_t p_end = p_begin + per_blk;
    if (p_end > mout) p_end = mout;

    float sc[EPC], sh[EPC], nka[EPC], nkb[EPC], s1[EPC], s2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) sc[e] = sh[e] = nka[e] = nkb[e] = s1[e] = s2[e] = 0.f;
    if (active) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            sc[e] = a.scale[c0 + e];
            sh[e] = a.shift[c0 + e];
            if (APPLY) {
                nka[e] = -a.coef[c0 + e];
                nkb[e] = -a.coef[a.ldy + c0 + e];
            }
        }
        int wo, ho, n;
        {
            const uint32_t p0 = p_begin + row;
            wo = (int)(p0 % (uint32_t)g.Wo);
            const uint32_t q = p0 / (uint32_t)g.Wo;
            ho = (int)(q % (uint32_t)g.Ho);
            n = (int)(q / (uint32_t)g.Ho);
        }
        const int drow = g.rows / g.Wo, dcol = g.rows % g.Wo;   // per-iteration advance
        for (uint32_t po = p_begin + row; po < p_end; po += g.rows, wo += dcol, ho += drow) {
            if (wo >= g.Wo) { wo -= g.Wo; ++ho; }
            while (ho >= g.Ho) { ho -= g.Ho; ++n; }
            Chunk<T> dav = ld_dA<T>(a, po, c0);
            if (POOL) {
                Chunk<T> yc[4];
                bool valid[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int hi = 2 * ho + (d >> 1), wi = 2 * wo + (d & 1);
                    valid[d] = hi < a.H && wi < a.W;
                    if (valid[d])
                        yc[d] = ld_chunk<T>((const char*)a.y + (((size_t)(n * a.H + hi) * a.W + wi) * a.ldy + c0) * sizeof(T));
                }
                int arg[EPC];
                float amax[EPC], yb[EPC], gz[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    arg[e] = 0;
                    amax[e] = -INFINITY;
                    yb[e] = 0.f;
                }
                if (valid[3]) {
                    // whole window (every even-sized map): the search of the forward pass (pool_window above) -- the
                    // maximum on z first, then the first position that holds it, by equality
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        const float y0 = Elem<T>::to_f32(yc[0].v[e]), y1 = Elem<T>::to_f32(yc[1].v[e]);
                        const float y2 = Elem<T>::to_f32(yc[2].v[e]), y3 = Elem<T>::to_f32(yc[3].v[e]);
                        const float z0 = y0 * sc[e] + sh[e], z1 = y1 * sc[e] + sh[e];
                        const float z2 = y2 * sc[e] + sh[e], z3 = y3 * sc[e] + sh[e];
                        const float zm = fmaxf(fmaxf(fmaxf(-INFINITY, z0), z1), fmaxf(z2, z3));
                        int d = z2 == zm ? 2 : 3;
                        float yv = z2 == zm ? y2 : y3;
                        d = z1 == zm ? 1 : d;
                        yv = z1 == zm ? y1 : yv;
                        d = z0 == zm ? 0 : d;
                        yv = z0 == zm ? y0 : yv;
                        arg[e] = d;
                        yb[e] = yv;
                    }
                } else {
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        if (valid[d]) {
#pragma unroll
                            for (int e = 0; e < EPC; ++e) {
                                // the pooled quantity is leaky(z): compare what the forward pass compared
                                const float yv = Elem<T>::to_f32(yc[d].v[e]);
                                const float act = leaky_s(fmaf(yv, sc[e], sh[e]), a.slope);
                                if (act > amax[e]) {
                                    amax[e] = act;
                                    yb[e] = yv;
                                    if (APPLY) arg[e] = d;
                                }
                            }
                        }
                }
                const bool sub = a.pool == 2;     // subsample: position 0 carries the gradient, the others are not in the batch norm
                if (sub) {
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        arg[e] = 0;
                        yb[e] = Elem<T>::to_f32(yc[0].v[e]);
                    }
                }
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    gz[e] = Elem<T>::to_f32(dav.v[e]) * leaky_slope_s(fmaf(yb[e], sc[e], sh[e]), a.slope);
                    if (!APPLY) {
                        s1[e] += gz[e];
                        s2[e] = fmaf(gz[e], yb[e], s2[e]);
                    }
                }
                if (APPLY) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        if (!valid[d]) continue;
                        const int hi = 2 * ho + (d >> 1), wi = 2 * wo + (d & 1);
                        float o[EPC];
#pragma unroll
                        for (int e = 0; e < EPC; ++e) {
                            const float t = (sub && d != 0) ? 0.f : fmaf(nkb[e], Elem<T>::to_f32(yc[d].v[e]), nka[e]);
                            o[e] = (arg[e] == d) ? fmaf(sc[e], gz[e], t) : t;
                        }
                        st_act<T, SPLIT>((char*)a.dyp, bpix(n, hi, wi, a.H, a.W), a.ldy, c0, o, a.hi_only);
                    }
                }
            } else {
                Chunk<T> v = ld_chunk<T>((const char*)a.y + ((size_t)po * a.ldy + c0) * sizeof(T));
                float o[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float yv = Elem<T>::to_f32(v.v[e]);
                    const float gz = Elem<T>::to_f32(dav.v[e]) * leaky_slope_s(fmaf(yv, sc[e], sh[e]), a.slope);
                    if (APPLY) {
                        o[e] = fmaf(sc[e], gz, fmaf(nkb[e], yv, nka[e]));
                    } else {
                        s1[e] += gz;
                        s2[e] = fmaf(gz, yv, s2[e]);
                    }
                }
                if (APPLY) st_act<T, SPLIT>((char*)a.dyp, bpix(n, ho, wo, a.H, a.W), a.ldy, c0, o, a.hi_only);
            }
        }
    }
    if (APPLY) return;
    // ---- block reduction over `rows`
    for (int k = 0; k < 2; ++k) {
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
                if (c0 + e < a.C) a.psum[((size_t)blockIdx.x * 2 + k) * a.ldy + c0 + e] = t;
            }
        }
    }
}

// after the reduce pass: dbeta, dgamma, dbias and the apply-pass constants coef = [ka][kb]
// 256-thread blocks (32 slices x 8 channels): this kernel sits on the backward critical path while the
// previous layer's weight gradient fills the CUs from the side stream -- a 16-wave block waits for a whole
// CU to drain (measured 27 us instead of 8), a 4-wave block is placed at once
constexpr int kBwdFinSl = 32;
template <int SLN>
__global__ __launch_bounds__(SLN * kFinCh) void bn_bwd_finalize_kernel(BnBwdArgs a, int P) {
    constexpr int kFinSl = SLN;   // shadows the forward kernel's slice count
    __shared__ double red[kFinSl][kFinCh];
    const int cl = threadIdx.x % kFinCh, sl = threadIdx.x / kFinCh;
    const int c = blockIdx.x * kFinCh + cl;
    const bool cv = c < a.C;
    const int cc = cv ? c : 0;
    // The kernel is a chain of dependent HBM round trips on the backward critical path, run beside the weight gradients
    // of the layer above (memory latency under that load is several times the idle one: 24 us per launch in the
    // production step against 8 serialised): both sums' records of EIGHT list steps are requested before the first is
    // added (16 loads in flight instead of 4).  The order of every accumulator's additions is that of the four-step
    // loop it replaces: the sums are bit-identical.
    double t[2] = {0.0, 0.0};
    {
        double v4[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        int p = sl;
        for (; p + 7 * kFinSl < P; p += 8 * kFinSl) {
            float r[2][8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const size_t q = (size_t)(p + u * kFinSl) * 2 * a.ldy + cc;
                r[0][u] = a.psum[q];
                r[1][u] = a.psum[q + a.ldy];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v4[0][u & 3] += (double)r[0][u];
                v4[1][u & 3] += (double)r[1][u];
            }
        }
        for (; p + 3 * kFinSl < P; p += 4 * kFinSl) {
            float r[2][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t q = (size_t)(p + u * kFinSl) * 2 * a.ldy + cc;
                r[0][u] = a.psum[q];
                r[1][u] = a.psum[q + a.ldy];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v4[0][u] += (double)r[0][u];
                v4[1][u] += (double)r[1][u];
            }
        }
        for (; p < P; p += kFinSl) {
            const size_t q = (size_t)p * 2 * a.ldy + cc;
            v4[0][0] += (double)a.psum[q];
            v4[1][0] += (double)a.psum[q + a.ldy];
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
            t[k] = fin_block_sum<kFinSl * kFinCh / 64>((v4[k][0] + v4[k][1]) + (v4[k][2] + v4[k][3]), red, sl, cl);
    }
    if (sl == 0 && cv) {
        const double m = a.pool == 2 ? (double)a.N * ((a.H + 1) / 2) * ((a.W + 1) / 2) : (double)a.N * a.H * a.W;   // (subsampling layers: the kept positions)
        const double mu = a.mean[c], is = a.invstd[c], sc = a.scale[c];
        const double dgam = is * (t[1] - mu * t[0]);          // sum(dz * xhat)
        a.dbeta[c] = (float)(t[0] * a.inv_grad_scale);
        a.dgamma[c] = (float)(dgam * a.inv_grad_scale);
        const float c1 = a.training ? (float)(t[0] / m) : 0.f;
        const float c2 = a.training ? (float)(dgam / m) : 0.f;
        const double kb = sc * (double)c2 * is;
        a.coef[c] = (float)(sc * (double)c1 - kb * mu);
        a.coef[a.ldy + c] = (float)kb;
        if (a.dbias) a.dbias[c] = (float)(sc * (t[0] - m * (double)c1) * a.inv_grad_scale);
    }
}

// ---------------------------------------------------------------------------
// Short partial lists (P <= 128: the 26x26 / 13x13 layers, whose sums come from the dgrad epilogue above): the
// finalize rides in the apply pass, as in bn_fin_act_kernel -- a block owns a 64-channel slab, sums the P records of
// its channels in double (4 slices, fixed order), forms ka / kb, and the blocks with blockIdx.x == 0 write dbeta,
// dgamma, dbias and coef.
// ---------------------------------------------------------------------------
template <typename T, bool POOL, bool SPLIT = false>
__global__ __launch_bounds__(256) void bn_bwd_fin_apply_kernel(BnBwdArgs a) {
    constexpr int EPC = 16 / sizeof(T);
    constexpr int CPB = kSlab / EPC;
    constexpr int rows = 256 / CPB;
    __shared__ double red[4][kSlab][2];
    __shared__ float s_nka[kSlab], s_nkb[kSlab];
    const int cbase = blockIdx.y * kSlab;
    {
        const int c = threadIdx.x & (kSlab - 1), sl = threadIdx.x >> 6;
        const int cc = cbase + c;
        const bool cv = cc < a.C;
        const int ci = cv ? cc : 0;
        double t0 = 0.0, t1 = 0.0;
        int p = sl;
        for (; p + 28 < a.P; p += 32) {       // eight list steps' loads up front (see bn_bwd_finalize_kernel); same order
            float r0[8], r1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const size_t q = (size_t)(p + 4 * u) * 2 * a.ldy + ci;
                r0[u] = a.psum[q];
                r1[u] = a.psum[q + a.ldy];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { t0 += (double)r0[u]; t1 += (double)r1[u]; }
        }
        for (; p < a.P; p += 4) {
            t0 += (double)a.psum[((size_t)p * 2 + 0) * a.ldy + ci];
            t1 += (double)a.psum[((size_t)p * 2 + 1) * a.ldy + ci];
        }
        red[sl][c][0] = t0; red[sl][c][1] = t1;
        __syncthreads();
        if (sl == 0) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { s0 += red[k][c][0]; s1 += red[k][c][1]; }
            const double m = a.pool == 2 ? (double)a.N * ((a.H + 1) / 2) * ((a.W + 1) / 2) : (double)a.N * a.H * a.W;   // (subsampling layers: the kept positions)
            const double mu = a.mean[ci], is = a.invstd[ci], sc = a.scale[ci];
            const double dgam = is * (s1 - mu * s0);
            const float c1 = a.training ? (float)(s0 / m) : 0.f;
            const float c2 = a.training ? (float)(dgam / m) : 0.f;
            const double kb = sc * (double)c2 * is;
            const float ka = (float)(sc * (double)c1 - kb * mu);
            s_nka[c] = cv ? -ka : 0.f;
            s_nkb[c] = cv ? -(float)kb : 0.f;
            if (blockIdx.x == 0 && cv) {
                a.dbeta[cc] = (float)(s0 * a.inv_grad_scale);
                a.dgamma[cc] = (float)(dgam * a.inv_grad_scale);
                a.coef[cc] = ka;
                a.coef[a.ldy + cc] = (float)kb;
                if (a.dbias) a.dbias[cc] = (float)(sc * (s0 - m * (double)c1) * a.inv_grad_scale);
            }
        }
        __syncthreads();
    }
    const int Ho = POOL ? (a.H + 1) / 2 : a.H, Wo = POOL ? (a.W + 1) / 2 : a.W;
    const uint32_t mout = (uint32_t)a.N * Ho * Wo;
    const int tid = threadIdx.x;
    const int ch = tid % CPB, row = tid / CPB;
    const int c0 = cbase + ch * EPC;
    if (c0 >= a.ldy) return;
    const uint32_t per_blk = (mout + gridDim.x - 1) / gridDim.x;
    const uint32_t p_begin = blockIdx.x * per_blk;
    uint32_t p_end = p_begin + per_blk;
    if (p_end > mout) p_end = mout;
    float sc[EPC], sh[EPC], nka[EPC], nkb[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        sc[e] = a.scale[c0 + e];
        sh[e] = a.shift[c0 + e];
        nka[e] = s_nka[ch * EPC + e];
        nkb[e] = s_nkb[ch * EPC + e];
    }
    int wo, ho, n;
    {
        const uint32_t p0 = p_begin + row;
        wo = (int)(p0 % (uint32_t)Wo);
        const uint32_t q = p0 / (uint32_t)Wo;
        ho = (int)(q % (uint32_t)Ho);
        n = (int)(q / (uint32_t)Ho);
    }
    const int drow = rows / Wo, dcol = rows % Wo;
    for (uint32_t po = p_begin + row; po < p_end; po += rows, wo += dcol, ho += drow) {
        if (wo >= Wo) { wo -= Wo; ++ho; }
        while (ho >= Ho) { ho -= Ho; ++n; }
        Chunk<T> dav = ld_dA<T>(a, po, c0);
        if (POOL) {
            Chunk<T> yc[4];
            bool valid[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int hi = 2 * ho + (d >> 1), wi = 2 * wo + (d & 1);
                valid[d] = hi < a.H && wi < a.W;
                if (valid[d])
                    yc[d] = ld_chunk<T>((const char*)a.y + (((size_t)(n * a.H + hi) * a.W + wi) * a.ldy + c0) * sizeof(T));
            }
            int arg[EPC];
            float amax[EPC], yb[EPC], gz[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                arg[e] = 0;
                amax[e] = -INFINITY;
                yb[e] = 0.f;
            }
            if (valid[3]) {      // whole window: the forward pass's search (pool_window; bn_bwd_kernel above)
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float y0 = Elem<T>::to_f32(yc[0].v[e]), y1 = Elem<T>::to_f32(yc[1].v[e]);
                    const float y2 = Elem<T>::to_f32(yc[2].v[e]), y3 = Elem<T>::to_f32(yc[3].v[e]);
                    const float z0 = y0 * sc[e] + sh[e], z1 = y1 * sc[e] + sh[e];
                    const float z2 = y2 * sc[e] + sh[e], z3 = y3 * sc[e] + sh[e];
                    const float zm = fmaxf(fmaxf(fmaxf(-INFINITY, z0), z1), fmaxf(z2, z3));
                    int d = z2 == zm ? 2 : 3;
                    float yv = z2 == zm ? y2 : y3;
                    d = z1 == zm ? 1 : d;
                    yv = z1 == zm ? y1 : yv;
                    d = z0 == zm ? 0 : d;
                    yv = z0 == zm ? y0 : yv;
                    arg[e] = d;
                    yb[e] = yv;
                }
            } else {
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    if (valid[d]) {
#pragma unroll
                        for (int e = 0; e < EPC; ++e) {
                            const float yv = Elem<T>::to_f32(yc[d].v[e]);
                            const float act = leaky_s(fmaf(yv, sc[e], sh[e]), a.slope);
                            if (act > amax[e]) {
                                amax[e] = act;
                                yb[e] = yv;
                                arg[e] = d;
                            }
                        }
                    }
            }
            const bool sub = a.pool == 2;     // subsample: position 0 carries the gradient, the others are not in the batch norm
            if (sub) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    arg[e] = 0;
                    yb[e] = Elem<T>::to_f32(yc[0].v[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < EPC; ++e) gz[e] = Elem<T>::to_f32(dav.v[e]) * leaky_slope_s(fmaf(yb[e], sc[e], sh[e]), a.slope);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                if (!valid[d]) continue;
                const int hi = 2 * ho + (d >> 1), wi = 2 * wo + (d & 1);
                float o[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float t = (sub && d != 0) ? 0.f : fmaf(nkb[e], Elem<T>::to_f32(yc[d].v[e]), nka[e]);
                    o[e] = (arg[e] == d) ? fmaf(sc[e], gz[e], t) : t;
                }
                st_act<T, SPLIT>((char*)a.dyp, bpix(n, hi, wi, a.H, a.W), a.ldy, c0, o, a.hi_only);
            }
        } else {
            Chunk<T> v = ld_chunk<T>((const char*)a.y + ((size_t)po * a.ldy + c0) * sizeof(T));
            float o[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float yv = Elem<T>::to_f32(v.v[e]);
                const float gz = Elem<T>::to_f32(dav.v[e]) * leaky_slope_s(fmaf(yv, sc[e], sh[e]), a.slope);
                o[e] = fmaf(sc[e], gz, fmaf(nkb[e], yv, nka[e]));
            }
            st_act<T, SPLIT>((char*)a.dyp, bpix(n, ho, wo, a.H, a.W), a.ldy, c0, o, a.hi_only);
        }
    }
}

bool bn_bwd_fin_apply_ok(const BnBwdArgs& a) {
    static const int pmax = getenv("Y2DEV_FIN_PMAX") ? atoi(getenv("Y2DEV_FIN_PMAX")) : 128;
    return a.P > 0 && a.P <= pmax && a.ldy % kSlab == 0 && a.C == a.ldy; }
template <typename T, bool SPLIT = false>
static hipError_t bn_bwd_fin_apply_T(const BnBwdArgs& a, hipStream_t s) {
    constexpr int EPC = 16 / sizeof(T);
    const int Ho = a.pool ? (a.H + 1) / 2 : a.H, Wo = a.pool ? (a.W + 1) / 2 : a.W;
    const size_t mout = (size_t)a.N * Ho * Wo;
    const int rows = 256 / (kSlab / EPC);
    const int ny = a.ldy / kSlab;
    size_t nx = (mout + (size_t)rows * 8 - 1) / ((size_t)rows * 8);
    const size_t cap = (size_t)(2048 / ny) > 0 ? (size_t)(2048 / ny) : 1;
    if (nx > cap) nx = cap;
    if (nx == 0) nx = 1;
    dim3 g((unsigned)nx, (unsigned)ny), b(256);
    if (a.pool) hipLaunchKernelGGL((bn_bwd_fin_apply_kernel<T, true, SPLIT>), g, b, 0, s, a);
    else hipLaunchKernelGGL((bn_bwd_fin_apply_kernel<T, false, SPLIT>), g, b, 0, s, a);
    return hipGetLastError();
}
hipError_t launch_bn_bwd_fin_apply(int dtype, const BnBwdArgs& a, hipStream_t s) {
    if (!bn_bwd_fin_apply_ok(a)) return hipErrorInvalidValue;
    switch (dtype) {
        case 0: return bn_bwd_fin_apply_T<float>(a, s);
        case 1: return bn_bwd_fin_apply_T<half_t>(a, s);
        case 2: return bn_bwd_fin_apply_T<bf16_t>(a, s);
        case 3: return bn_bwd_fin_apply_T<float, true>(a, s);
    }
    return hipErrorInvalidValue;
}

template <typename T>
static int bwd_blocks(const BnBwdArgs& a) {
    const BwdGeom g = bwd_geom<T>(a);
    size_t nb = (g.mout + (size_t)g.rows * 8 - 1) / ((size_t)g.rows * 8);  // >= 8 pixels per thread row
    if (nb < 256) {      // small tensors (the 13x13 output layer: 21 blocks took 21 us): one pixel row group per block
        nb = (g.mout + g.rows - 1) / g.rows;
        if (nb > 256) nb = 256;
    }
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    return (int)nb;
}
int bn_bwd_partials(const BnBwdArgs& a) { return 2048; }

template <typename T, bool APPLY, bool SPLIT = false>
static hipError_t bn_bwd_T(const BnBwdArgs& a, hipStream_t s) {
    dim3 g(APPLY ? bwd_blocks<T>(a) : a.P), b(256);
    if (a.pool) hipLaunchKernelGGL((bn_bwd_kernel<T, true, APPLY, SPLIT>), g, b, 0, s, a);
    else hipLaunchKernelGGL((bn_bwd_kernel<T, false, APPLY, SPLIT>), g, b, 0, s, a);
    return hipGetLastError();
}
hipError_t launch_bn_bwd_reduce(int dtype, BnBwdArgs& a, hipStream_t s) {
    switch (dtype) {
        case 0: a.P = bwd_blocks<float>(a); return bn_bwd_T<float, false>(a, s);
        case 1: a.P = bwd_blocks<half_t>(a); return bn_bwd_T<half_t, false>(a, s);
        case 2: a.P = bwd_blocks<bf16_t>(a); return bn_bwd_T<bf16_t, false>(a, s);
        case 3: a.P = bwd_blocks<float>(a); return bn_bwd_T<float, false>(a, s);     // f16x2: every input of the pass is fp32
    }
    return hipErrorInvalidValue;
}
hipError_t launch_bn_bwd_finalize(const BnBwdArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(bn_bwd_finalize_kernel<kBwdFinSl>, dim3((a.C + kFinCh - 1) / kFinCh), dim3(kBwdFinSl * kFinCh), 0, s, a, a.P);
    return hipGetLastError();
}
// ---------------------------------------------------------------------------
// Subsampling layers (pool == 2): batch-norm partial records over the kept positions of y.  Block = 64 channels x 4
// pixel slices of one record's kBnSubRec kept pixels; sums about the record's first kept value (shifted sums), the four
// slices added in double in a fixed order.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_sub_kernel(const T* __restrict__ y, int N, int H, int W, int ldy,
                                                           float* part_cnt, float* part_mean, float* part_m2) {
    __shared__ float red[4][64][2];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
    const int Ho = H / 2, Wo = W / 2, mout = N * Ho * Wo;
    const int q0 = blockIdx.y * kBnSubRec;
    const int q1 = q0 + kBnSubRec < mout ? q0 + kBnSubRec : mout;
    const bool cv = c < ldy;
    auto pixel = [&](int q) -> size_t {
        const int wo = q % Wo, t = q / Wo, ho = t % Ho, n = t / Ho;
        return ((size_t)(n * H + 2 * ho) * W + 2 * wo) * ldy;
    };
    const float piv = cv ? Elem<T>::to_f32(y[pixel(q0) + c]) : 0.f;
    float s1 = 0.f, s2 = 0.f;
    if (cv) {
        for (int q = q0 + sl; q < q1; q += 4) {
            const float d = Elem<T>::to_f32(y[pixel(q) + c]) - piv;
            s1 += d;
            s2 = fmaf(d, d, s2);
        }
    }
    red[sl][threadIdx.x & 63][0] = s1;
    red[sl][threadIdx.x & 63][1] = s2;
    __syncthreads();
    if (sl == 0 && cv) {
        const int k = threadIdx.x & 63;
        const double S1 = ((double)red[0][k][0] + red[1][k][0]) + ((double)red[2][k][0] + red[3][k][0]);
        const double S2 = ((double)red[0][k][1] + red[1][k][1]) + ((double)red[2][k][1] + red[3][k][1]);
        const double n = (double)(q1 - q0), md = S1 / n, m2 = S2 - S1 * md;
        part_mean[(size_t)blockIdx.y * ldy + c] = (float)((double)piv + md);
        part_m2[(size_t)blockIdx.y * ldy + c] = (float)(m2 > 0.0 ? m2 : 0.0);
        if (c == 0) part_cnt[blockIdx.y] = (float)(q1 - q0);
    }
}
hipError_t launch_bn_stats_sub(int dtype, const void* y, int N, int H, int W, int ldy, float* part_cnt, float* part_mean,
                               float* part_m2, int* records, hipStream_t s) {
    if ((H & 1) || (W & 1)) return hipErrorInvalidValue;
    const int mout = N * (H / 2) * (W / 2);
    const int rec = (mout + kBnSubRec - 1) / kBnSubRec;
    const dim3 g((ldy + 63) / 64, rec), b(256);
    switch (dtype) {
        case 0: case 3: hipLaunchKernelGGL(bn_stats_sub_kernel<float>, g, b, 0, s, (const float*)y, N, H, W, ldy, part_cnt, part_mean, part_m2); break;
        case 1: hipLaunchKernelGGL(bn_stats_sub_kernel<half_t>, g, b, 0, s, (const half_t*)y, N, H, W, ldy, part_cnt, part_mean, part_m2); break;
        case 2: hipLaunchKernelGGL(bn_stats_sub_kernel<bf16_t>, g, b, 0, s, (const bf16_t*)y, N, H, W, ldy, part_cnt, part_mean, part_m2); break;
        default: return hipErrorInvalidValue;
    }
    if (records) *records = rec;
    return hipGetLastError();
}

hipError_t launch_bn_bwd_apply(int dtype, const BnBwdArgs& a, hipStream_t s) {
    switch (dtype) {
        case 0: return bn_bwd_T<float, true>(a, s);
        case 1: return bn_bwd_T<half_t, true>(a, s);
        case 2: return bn_bwd_T<bf16_t, true>(a, s);
        case 3: return bn_bwd_T<float, true, true>(a, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace y2
