// First layer: 3x3 SAME conv 3 -> 32 (reference darknet.py:150, Cin = 3).
// K = 27 is too small for the LDS-tiled implicit GEMM and the layer is
// HBM-bound (12 B in, 64 B out per pixel at fp16), so it gets its own kernel:
//   * input stored as zero-bordered NHWC with 4 channels: the three taps of one
//     filter row are 12 contiguous elements -> each MFMA B-fragment is ONE
//     16-byte global load straight into registers (no LDS staging)
//   * K padded 27 -> 3 x 16 (zero weights), D[cout][pixel] on 32x32 MFMA
//   * persistent waves; epilogue repacks through a wave-private LDS patch so the
//     stores are whole 64-byte pixel rows, 1 KiB contiguous per wave-instruction
//   * BN statistics as bias-shifted sums per wave, Chan-merged per block.
#include <stdlib.h>
#include "common.h"
#include "kernels.h"

namespace y2 {

template <typename T, bool STORE>
__global__ __launch_bounds__(256) void conv1_fwd_kernel(Conv1Args a) {
    typedef typename Elem<T>::frag frag_t;
    // a pixel is 4*SZ bytes, so a fragment load is only 8-byte aligned at f16/bf16
    struct __attribute__((packed, aligned(8))) UFrag { frag_t v; };
    constexpr int SZ = sizeof(T);
    constexpr int KGC = 16 * SZ / 32;        // k-groups per filter row (1: f16/bf16, 2: f32)
    constexpr int EROW = 32 * SZ + 16;
    constexpr int EPC = 16 / SZ;
    constexpr int CPR = 32 / EPC;            // chunks per pixel row
    constexpr int RPIe = 64 / CPR;
    constexpr int NIT = 32 / RPIe;
    __shared__ __attribute__((aligned(16))) char smem[4 * 32 * EROW + 4 * 32 * 3 * 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, hh = lane >> 5;
    char* ew = smem + w * 32 * EROW;

    // weights: A operand rows = cout r32
    frag_t fw[3][KGC];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int g = 0; g < KGC; ++g)
            fw[kh][g] = *(const frag_t*)((const char*)a.w + ((r32 * 3 + kh) * 16) * SZ + 32 * g + 16 * hh);

    const int ch = lane % CPR, prow0 = lane / CPR;
    float bsh[EPC], s1[EPC], s2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        bsh[e] = a.bias[ch * EPC + e];
        s1[e] = s2[e] = 0.f;
    }
    float b4[4][4];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
        for (int j = 0; j < 4; ++j) b4[q4][j] = a.bias[8 * q4 + 4 * hh + j];

    const int ntiles = (a.M + 31) / 32;
    const int nwaves = gridDim.x * 4;
    const int rowpitch = (a.W + 1) * 4 * SZ;
    int my_cnt = 0;
    for (int tile = blockIdx.x * 4 + w; tile < ntiles; tile += nwaves) {
        const int p = tile * 32 + r32;
        uint32_t base = 0;
        if (p < a.M) {
            const int hw = a.H * a.W;
            const int n = p / hw, rem = p - n * hw;
            const int h = rem / a.W, ww = rem - h * a.W;
            base = (uint32_t)(bpix(n, h, ww, a.H, a.W) - (size_t)(a.W + 2)) * (uint32_t)(4 * SZ);
        }
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        frag_t fx[3][KGC];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int g = 0; g < KGC; ++g)
                fx[kh][g] = ((const UFrag*)((const char*)a.x4 + base + kh * rowpitch + 32 * g + 16 * hh))->v;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int g = 0; g < KGC; ++g) mma32(acc, fw[kh][g], fx[kh][g]);
        // acc[q]: cout = acc_row(q, hh), pixel = r32
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            T o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = Elem<T>::from_f32(acc[4 * q4 + j] + b4[q4][j]);
            char* dst = ew + r32 * EROW + (8 * q4 + 4 * hh) * SZ;
            if (SZ == 2) *(u32x2*)dst = *(const u32x2*)o;
            else *(u32x4*)dst = *(const u32x4*)o;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int prow = it * RPIe + prow0;
            Chunk<T> c = ld_chunk<T>(ew + prow * EROW + ch * 16);
            const int pp = tile * 32 + prow;
            if (pp < a.M) {
                if (STORE) st_chunk<T>((char*)a.y + ((size_t)pp * 32 + ch * EPC) * SZ, c);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float d = Elem<T>::to_f32(c.v[e]) - bsh[e];
                    s1[e] += d;
                    s2[e] += d * d;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        int tc = a.M - tile * 32;
        my_cnt += tc > 32 ? 32 : tc;
    }
    // ---- per-wave (count, mean, M2), then Chan-merge the 4 waves
    float* st = (float*)(smem + 4 * 32 * EROW);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
#pragma unroll
        for (int msk = CPR; msk < 64; msk <<= 1) {
            s1[e] = wave_sum_xor(s1[e], msk);
            s2[e] = wave_sum_xor(s2[e], msk);
        }
    }
    if (prow0 == 0) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float n = (float)my_cnt;
            const float md = n > 0 ? s1[e] / n : 0.f;
            st[(w * 32 + ch * EPC + e) * 3 + 0] = n;
            st[(w * 32 + ch * EPC + e) * 3 + 1] = bsh[e] + md;
            st[(w * 32 + ch * EPC + e) * 3 + 2] = n > 0 ? fmaxf(s2[e] - s1[e] * md, 0.f) : 0.f;
        }
    }
    __syncthreads();
    if (tid < 32) {
        float n_acc = 0.f, mean_acc = 0.f, m2_acc = 0.f;
        for (int k = 0; k < 4; ++k) {
            const float nk = st[(k * 32 + tid) * 3 + 0];
            if (nk == 0.f) continue;
            const float mk = st[(k * 32 + tid) * 3 + 1], vk = st[(k * 32 + tid) * 3 + 2];
            const float nn = n_acc + nk, dlt = mk - mean_acc;
            mean_acc += dlt * (nk / nn);
            m2_acc += vk + dlt * dlt * (n_acc * nk / nn);
            n_acc = nn;
        }
        a.part_mean[blockIdx.x * 32 + tid] = mean_acc;
        a.part_m2[blockIdx.x * 32 + tid] = m2_acc;
        if (tid == 0) a.part_cnt[blockIdx.x] = n_acc;
    }
}

// Statistics-only pass (first pass of the pooled first layer): no output, so no transpose either --
// every lane keeps bias-shifted sums of its 16 couts (values rounded to T as they WILL be stored by
// the second pass) over all its tiles; lanes are reduced once, at the end.
// XS (f16x2 mode, T = float: round 5): the fp32 operands are split into (hi, lo) halves in registers and a filter row is
// three v_mfma_f32_32x32x16_f16 (common.h mma32_split; filters times kSplitWScale) instead of eight
// v_mfma_f32_32x32x2_f32 -- the exact-fp32 form is bound by its matrix instructions (16x the f16 cycles per product).
template <typename T, bool XS = false>
__global__ __launch_bounds__(256) void conv1_stats_kernel(Conv1Args a) {
    typedef typename Elem<T>::frag frag_t;
    struct __attribute__((packed, aligned(8))) UFrag { frag_t v; };
    constexpr int SZ = sizeof(T);
    constexpr int KGC = 16 * SZ / 32;
    static_assert(!XS || SZ == 4, "the in-register split reads fp32 operands");
    __shared__ float st[4 * 32 * 3];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, hh = lane >> 5;
    frag_t fw[3][KGC];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int g = 0; g < KGC; ++g)
            fw[kh][g] = *(const frag_t*)((const char*)a.w + ((r32 * 3 + kh) * 16) * SZ + 32 * g + 16 * hh);
    f16x8 fwh[3], fwl[3];
    if constexpr (XS) {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) split_frag8(fw[kh][0], fw[kh][KGC - 1], kSplitWScale, fwh[kh], fwl[kh]);
    }
    float bq[16], s1[16], s2[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        bq[q] = a.bias[acc_row(q, hh)];
        s1[q] = s2[q] = 0.f;
    }
    // tiles of 2 image rows x 32 columns (as the second pass): the tile index is wave-uniform, so the
    // (n, h, w) split costs scalar work only -- per-lane div/mod of a linear pixel index cost more than
    // the three MFMAs of a tile -- and two accumulators share four input-row loads
    const int nseg = (a.W + 31) / 32;
    const int Hp = (a.H + 1) / 2;
    const int ntiles = a.N * Hp * nseg;
    const int nwaves = gridDim.x * 4;
    const int rowpitch = (a.W + 1) * 4 * SZ;
    int my_cnt = 0;
    for (int tile = blockIdx.x * 4 + w; tile < ntiles; tile += nwaves) {
        const int sg = tile % nseg, pr = tile / nseg;
        const int n = pr / Hp, hp = pr - n * Hp, h0 = 2 * hp, w0 = sg * 32;
        const int wc = (w0 + r32 < a.W) ? w0 + r32 : a.W - 1;
        const uint32_t base = (uint32_t)(bpix(n, h0, wc, a.H, a.W) - (size_t)(a.W + 2)) * (uint32_t)(4 * SZ);
        const bool row1 = h0 + 1 < a.H;               // odd H: the last pair has one row (wave-uniform)
        frag_t fx[4][KGC];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int g = 0; g < KGC; ++g)
                fx[r][g] = ((const UFrag*)((const char*)a.x4 + base + (r < 3 || row1 ? r : 2) * rowpitch + 32 * g + 16 * hh))->v;
        const float vm = (w0 + r32 < a.W) ? 1.f : 0.f;
        const int cols = a.W - w0 < 32 ? a.W - w0 : 32;
        f16x8 fxh[4], fxl[4];
        if constexpr (XS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) split_frag8(fx[r][0], fx[r][KGC - 1], 1.0f, fxh[r], fxl[r]);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (r == 1 && !row1) break;
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
            if constexpr (XS) {
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) mma32_split(acc, fwh[kh], fwl[kh], fxh[r + kh], fxl[r + kh]);
                acc *= kSplitWScaleInv;
            } else {
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int g = 0; g < KGC; ++g) mma32(acc, fw[kh][g], fx[r + kh][g]);
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float d = (Elem<T>::to_f32(Elem<T>::from_f32(acc[q] + bq[q])) - bq[q]) * vm;
                s1[q] += d;
                s2[q] = fmaf(d, d, s2[q]);
            }
            my_cnt += cols;
        }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int msk = 1; msk < 32; msk <<= 1) {
            s1[q] = wave_sum_xor(s1[q], msk);
            s2[q] = wave_sum_xor(s2[q], msk);
        }
    if (r32 == 0) {
        const float n = (float)my_cnt;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int co = acc_row(q, hh);
            const float md = n > 0 ? s1[q] / n : 0.f;
            st[(w * 32 + co) * 3 + 0] = n;
            st[(w * 32 + co) * 3 + 1] = bq[q] + md;
            st[(w * 32 + co) * 3 + 2] = n > 0 ? fmaxf(s2[q] - s1[q] * md, 0.f) : 0.f;
        }
    }
    __syncthreads();
    if (tid < 32) {
        float n_acc = 0.f, mean_acc = 0.f, m2_acc = 0.f;
        for (int k = 0; k < 4; ++k) {
            const float nk = st[(k * 32 + tid) * 3 + 0];
            if (nk == 0.f) continue;
            const float mk = st[(k * 32 + tid) * 3 + 1], vk = st[(k * 32 + tid) * 3 + 2];
            const float nn = n_acc + nk, dlt = mk - mean_acc;
            mean_acc += dlt * (nk / nn);
            m2_acc += vk + dlt * dlt * (n_acc * nk / nn);
            n_acc = nn;
        }
        a.part_mean[blockIdx.x * 32 + tid] = mean_acc;
        a.part_m2[blockIdx.x * 32 + tid] = m2_acc;
        if (tid == 0) a.part_cnt[blockIdx.x] = n_acc;
    }
}

hipError_t launch_conv1_fwd(int dtype, const Conv1Args& a, hipStream_t s) {
    dtype = dtype_plain(dtype);      // f16x2: the 3-channel layer reads fp32 operands (exact fp32, or Conv1Args::xs)
    dim3 g(a.nblocks), b(256);
    if (a.stats_only) {
        switch (dtype) {
            case 0:
                if (a.xs) hipLaunchKernelGGL((conv1_stats_kernel<float, true>), g, b, 0, s, a);
                else hipLaunchKernelGGL(conv1_stats_kernel<float>, g, b, 0, s, a);
                break;
            case 1: hipLaunchKernelGGL(conv1_stats_kernel<half_t>, g, b, 0, s, a); break;
            case 2: hipLaunchKernelGGL(conv1_stats_kernel<bf16_t>, g, b, 0, s, a); break;
            default: return hipErrorInvalidValue;
        }
    } else {
        switch (dtype) {
            case 0: hipLaunchKernelGGL((conv1_fwd_kernel<float, true>), g, b, 0, s, a); break;
            case 1: hipLaunchKernelGGL((conv1_fwd_kernel<half_t, true>), g, b, 0, s, a); break;
            case 2: hipLaunchKernelGGL((conv1_fwd_kernel<bf16_t, true>), g, b, 0, s, a); break;
            default: return hipErrorInvalidValue;
        }
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Second pass of the pooled first layer: conv AGAIN (K = 27: recomputing is cheaper than
// re-reading 64 B/pixel) + batch norm + leaky + 2x2 max pool, writing the conv output (kept
// for the backward pass) and the pooled activation in one sweep.  Together with the
// statistics-only first pass this replaces  conv1 (write y) -> finalize -> bn_act (read y):
// the 709 MB read of y at 416x416x64 disappears.
// A wave tile is 2 image rows x 32 columns: two accumulators share the four input rows; the
// wave-private LDS patch [64 px][32 co] feeds both the 1-KiB row stores of y and the 2x2 windows.
// ---------------------------------------------------------------------------
// TRACK 1: the window's arg-max and its conv output are kept for the backward pass (a.ysel / a.idx; f32 parity mode and
// Y2_CONV1_YSEL=1), found by a per-position compare / select chain on the activation.
// TRACK 0 (inference, or a training binding that stores y): the window maximum is taken BEFORE the activation -- leaky is
// non-decreasing, so leaky(max z) = max leaky(z) bit for bit -- and that chain (88 v_cndmask + 32 v_bfi + 31 compares of
// the ~400 vector instructions a tile cost; the kernel is bound by vector issue, not by HBM) drops out.
// TRACK 2 (Conv1PoolArgs::idx3; training, 16-bit types): the same maximum first, then the first position that holds it by
// equality, and 3 bits per element (position, activation branch) instead of any conv output.
// XS: as conv1_stats_kernel (f16x2 mode: the SAME three-product sequence, so both passes see the same conv output).
template <typename T, bool STOREY, int TRACK, bool XS = false>
__global__ __launch_bounds__(256) void conv1_pool_kernel(Conv1PoolArgs a) {
    typedef typename Elem<T>::frag frag_t;
    struct __attribute__((packed, aligned(8))) UFrag { frag_t v; };
    constexpr int SZ = sizeof(T);
    constexpr int KGC = 16 * SZ / 32;
    static_assert(!XS || SZ == 4, "the in-register split reads fp32 operands");
    constexpr int EROW = 32 * SZ + 16;
    constexpr int EPC = 16 / SZ;
    constexpr int CPR = 32 / EPC;            // chunks per pixel row (4: f16/bf16, 8: f32)
    constexpr int RPIe = 64 / CPR;           // pixels per wave instruction
    __shared__ __attribute__((aligned(16))) char smem[4 * 64 * EROW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, hh = lane >> 5;
    char* ew = smem + w * 64 * EROW;

    frag_t fw[3][KGC];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int g = 0; g < KGC; ++g)
            fw[kh][g] = *(const frag_t*)((const char*)a.w + ((r32 * 3 + kh) * 16) * SZ + 32 * g + 16 * hh);
    f16x8 fwh[3], fwl[3];
    if constexpr (XS) {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) split_frag8(fw[kh][0], fw[kh][KGC - 1], kSplitWScale, fwh[kh], fwl[kh]);
    }
    float b4[4][4];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
        for (int j = 0; j < 4; ++j) b4[q4][j] = a.bias[8 * q4 + 4 * hh + j];
    const int ch = lane % CPR, pl = lane / CPR;
    float sc[EPC], sh[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        sc[e] = a.scale[ch * EPC + e];
        sh[e] = a.shift[ch * EPC + e];
    }

    const int Ho = a.H / 2, Wo = a.W / 2;
    const int nseg = (a.W + 31) / 32;
    const int ntiles = a.N * Ho * nseg;
    const int nwaves = gridDim.x * 4;
    const int rowpitch = (a.W + 1) * 4 * SZ;
    auto load_tile = [&](int tile, frag_t (&fx)[4][KGC]) {
        const int sg = tile % nseg, pr = tile / nseg;
        const int n = pr / Ho, ho = pr - n * Ho, h0 = 2 * ho, w0 = sg * 32;
        const int wc = (w0 + r32 < a.W) ? w0 + r32 : a.W - 1;      // clamp (masked at the stores)
        const uint32_t base = (uint32_t)(bpix(n, h0, wc, a.H, a.W) - (size_t)(a.W + 2)) * (uint32_t)(4 * SZ);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int g = 0; g < KGC; ++g)
                fx[r][g] = ((const UFrag*)((const char*)a.x4 + base + r * rowpitch + 32 * g + 16 * hh))->v;
    };
    auto body = [&](int tile, frag_t (&fx)[4][KGC]) {
        const int sg = tile % nseg, pr = tile / nseg;
        const int n = pr / Ho, ho = pr - n * Ho, h0 = 2 * ho, w0 = sg * 32;
        f32x16 acc[2];
        f16x8 fxh[4], fxl[4];
        if constexpr (XS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) split_frag8(fx[r][0], fx[r][KGC - 1], 1.0f, fxh[r], fxl[r]);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[r][q] = 0.f;
            if constexpr (XS) {
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) mma32_split(acc[r], fwh[kh], fwl[kh], fxh[r + kh], fxl[r + kh]);
                acc[r] *= kSplitWScaleInv;
            } else {
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int g = 0; g < KGC; ++g) mma32(acc[r], fw[kh][g], fx[r + kh][g]);
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                T o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = Elem<T>::from_f32(acc[r][4 * q4 + j] + b4[q4][j]);
                char* dst = ew + (r * 32 + r32) * EROW + (8 * q4 + 4 * hh) * SZ;
                if (SZ == 2) *(u32x2*)dst = *(const u32x2*)o;
                else *(u32x4*)dst = *(const u32x4*)o;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (STOREY) {
#pragma unroll
            for (int it = 0; it < 64 / RPIe; ++it) {
                const int prow = it * RPIe + pl;            // patch row = r*32 + column
                const int r = prow >> 5, col = prow & 31;
                if (w0 + col < a.W) {
                    Chunk<T> c = ld_chunk<T>(ew + prow * EROW + ch * 16);
                    st_chunk<T>((char*)a.y + (((size_t)(n * a.H + h0 + r) * a.W + w0 + col) * 32 + ch * EPC) * SZ, c);
                }
            }
        }
#pragma unroll
        for (int ps = 0; ps < 16 / RPIe; ++ps) {
            const int j = ps * RPIe + pl;                   // pooled pixel of the tile, 0..15
            float m[EPC];
            Chunk<T> ys;
            unsigned arg = 0;
#pragma unroll
            for (int e = 0; e < EPC; ++e) m[e] = -INFINITY;
            if constexpr (TRACK == 2) {
                // the maximum first (as the untracked form: on z, leaky is non-decreasing), then WHICH position holds
                // it by equality, first in window order: 6 vector operations per element instead of 5 per position
                float z[4][EPC];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    Chunk<T> c = ld_chunk<T>(ew + ((d >> 1) * 32 + 2 * j + (d & 1)) * EROW + ch * 16);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) z[d][e] = fmaf(Elem<T>::to_f32(c.v[e]), sc[e], sh[e]);
                }
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float zm = fmaxf(fmaxf(fmaxf(m[e], z[0][e]), z[1][e]), fmaxf(z[2][e], z[3][e]));   // NaN: skipped
                    unsigned d = z[2][e] == zm ? 2u : 3u;      // three selects (a nested conditional became branches)
                    d = z[1][e] == zm ? 1u : d;
                    d = z[0][e] == zm ? 0u : d;
                    m[e] = leaky01(zm);
                    // leaky01_slope(z) = 0.1 exactly where 0.1 * z >= z, i.e. where max(0.1 z, z) <= 0
                    const unsigned neg = m[e] <= 0.f ? 4u : 0u;
                    arg |= (d | neg) << (3 * e);
                }
            } else {
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    Chunk<T> c = ld_chunk<T>(ew + ((d >> 1) * 32 + 2 * j + (d & 1)) * EROW + ch * 16);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        if constexpr (TRACK == 1) {
                            const float act = leaky01(fmaf(Elem<T>::to_f32(c.v[e]), sc[e], sh[e]));
                            if (act > m[e]) {          // first maximum in row-major window order
                                m[e] = act;
                                ys.v[e] = c.v[e];
                                arg = (arg & ~(3u << (2 * e))) | ((unsigned)d << (2 * e));
                            }
                        } else {
                            m[e] = fmaxf(m[e], fmaf(Elem<T>::to_f32(c.v[e]), sc[e], sh[e]));   // NaN: skipped, as by `>`
                        }
                    }
                }
            }
            if constexpr (TRACK == 0) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) m[e] = leaky01(m[e]);
            }

            const int wo = w0 / 2 + j;
            if (wo < Wo) {
                Chunk<T> o;
#pragma unroll
                for (int e = 0; e < EPC; ++e) o.v[e] = Elem<T>::from_f32(m[e]);
                if constexpr (SZ == 4) {
                    // f16x2 mode: the consumer's tensor is split, [32 halves hi][32 halves lo] per cell (common.h)
                    if (a.out_split) st_split4((char*)a.out + bpix(n, ho, wo, Ho, Wo) * 128, 32, ch * EPC, m);
                    else st_chunk<T>((char*)a.out + (bpix(n, ho, wo, Ho, Wo) * 32 + ch * EPC) * SZ, o);
                } else {
                    st_chunk<T>((char*)a.out + (bpix(n, ho, wo, Ho, Wo) * 32 + ch * EPC) * SZ, o);
                }
                if (TRACK == 1 && a.ysel) {
                    const size_t pix = ((size_t)n * Ho + ho) * Wo + wo;
                    st_chunk<T>((char*)a.ysel + (pix * 32 + ch * EPC) * SZ, ys);
                    a.idx[pix * CPR + ch] = (unsigned short)arg;
                }
                if (TRACK == 2) a.idx3[(((size_t)n * Ho + ho) * Wo + wo) * CPR + ch] = arg;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    };
    int tile = blockIdx.x * 4 + w;
    if constexpr (TRACK == 0) {
        for (; tile < ntiles; tile += nwaves) {
            frag_t fx[4][KGC];
            load_tile(tile, fx);
            body(tile, fx);
        }
    } else {
        // training form (more stores per tile): the input rows of a tile are requested one tile ahead, in two register
        // sets with the loop unrolled over them (a rotating copy costs 16 v_mov per tile).  Measured on one box, layer 0
        // forward at 416x416x64: 207 us against 212-216 with the loads in place; the inference form gains nothing
        frag_t fxa[4][KGC], fxb[4][KGC];
        if (tile < ntiles) load_tile(tile, fxa);
        while (tile < ntiles) {
            const int t2 = tile + nwaves, t3 = t2 + nwaves;
            if (t2 < ntiles) load_tile(t2, fxb);
            body(tile, fxa);
            if (t2 >= ntiles) break;
            if (t3 < ntiles) load_tile(t3, fxa);
            body(t2, fxb);
            tile = t3;
        }
    }
}

bool conv1_pool_ok(int H, int W, int pool, int cout) { return pool && (H % 2) == 0 && (W % 2) == 0 && cout == 32; }

template <typename T, bool XS = false>
static void conv1_pool_T(const Conv1PoolArgs& a, hipStream_t s) {
    dim3 g(a.nblocks), b(256);
    if (a.store_y) {
        if (a.idx3) hipLaunchKernelGGL((conv1_pool_kernel<T, true, 2, XS>), g, b, 0, s, a);
        else if (a.ysel) hipLaunchKernelGGL((conv1_pool_kernel<T, true, 1, XS>), g, b, 0, s, a);
        else hipLaunchKernelGGL((conv1_pool_kernel<T, true, 0, XS>), g, b, 0, s, a);
    } else {
        if (a.idx3) hipLaunchKernelGGL((conv1_pool_kernel<T, false, 2, XS>), g, b, 0, s, a);
        else if (a.ysel) hipLaunchKernelGGL((conv1_pool_kernel<T, false, 1, XS>), g, b, 0, s, a);
        else hipLaunchKernelGGL((conv1_pool_kernel<T, false, 0, XS>), g, b, 0, s, a);
    }
}

hipError_t launch_conv1_pool(int dtype, const Conv1PoolArgs& a, hipStream_t s) {
    dtype = dtype_plain(dtype);      // f16x2: fp32 operands (exact fp32, or Conv1PoolArgs::xs)
    switch (dtype) {
        case 0:
            if (a.xs) conv1_pool_T<float, true>(a, s);
            else conv1_pool_T<float>(a, s);
            break;
        case 1: conv1_pool_T<half_t>(a, s); break;
        case 2: conv1_pool_T<bf16_t>(a, s); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Backward reduce pass of the pooled first layer with the conv output RECOMPUTED from the input
// (same tiles and rounding as conv1_pool_kernel, so the values are bit-identical to the stored y):
//   g = dA * leaky'(z) at the first arg-max of the 2x2 window;  S1 += g, S2 += g*y   (bn.hip)
// reads x4 (8 B/pixel) + dA (16 B/pixel at f16) instead of y + dA (80 B/pixel).
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv1_bnbwd_reduce_kernel(Conv1BnBwdArgs a) {
    typedef typename Elem<T>::frag frag_t;
    struct __attribute__((packed, aligned(8))) UFrag { frag_t v; };
    constexpr int SZ = sizeof(T);
    constexpr int KGC = 16 * SZ / 32;
    constexpr int EROW = 32 * SZ + 16;
    constexpr int EPC = 16 / SZ;
    constexpr int CPR = 32 / EPC;
    constexpr int RPIe = 64 / CPR;
    __shared__ __attribute__((aligned(16))) char smem[4 * 64 * EROW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, hh = lane >> 5;
    char* ew = smem + w * 64 * EROW;

    frag_t fw[3][KGC];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int g = 0; g < KGC; ++g)
            fw[kh][g] = *(const frag_t*)((const char*)a.w + ((r32 * 3 + kh) * 16) * SZ + 32 * g + 16 * hh);
    float b4[4][4];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
        for (int j = 0; j < 4; ++j) b4[q4][j] = a.bias[8 * q4 + 4 * hh + j];
    const int ch = lane % CPR, pl = lane / CPR;
    float sc[EPC], sh[EPC], s1[EPC], s2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        sc[e] = a.scale[ch * EPC + e];
        sh[e] = a.shift[ch * EPC + e];
        s1[e] = s2[e] = 0.f;
    }
    const int Ho = a.H / 2, Wo = a.W / 2;
    const int nseg = (a.W + 31) / 32;
    const int ntiles = a.N * Ho * nseg;
    const int nwaves = gridDim.x * 4;
    const int rowpitch = (a.W + 1) * 4 * SZ;
    for (int tile = blockIdx.x * 4 + w; tile < ntiles; tile += nwaves) {
        const int sg = tile % nseg, pr = tile / nseg;
        const int n = pr / Ho, ho = pr - n * Ho, h0 = 2 * ho, w0 = sg * 32;
        const int wc = (w0 + r32 < a.W) ? w0 + r32 : a.W - 1;
        const uint32_t base = (uint32_t)(bpix(n, h0, wc, a.H, a.W) - (size_t)(a.W + 2)) * (uint32_t)(4 * SZ);
        frag_t fx[4][KGC];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int g = 0; g < KGC; ++g)
                fx[r][g] = ((const UFrag*)((const char*)a.x4 + base + r * rowpitch + 32 * g + 16 * hh))->v;
        // this lane's dA chunks (issued early: they are consumed after the MFMAs and the transpose)
        Chunk<T> dav[16 / RPIe];
#pragma unroll
        for (int ps = 0; ps < 16 / RPIe; ++ps) {
            const int wo = w0 / 2 + ps * RPIe + pl;
            const int woc = wo < Wo ? wo : Wo - 1;
            dav[ps] = ld_chunk<T>((const char*)a.dA + (((size_t)(n * Ho + ho) * Wo + woc) * 32 + ch * EPC) * SZ);
        }
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[r][q] = 0.f;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int g = 0; g < KGC; ++g) mma32(acc[r], fw[kh][g], fx[r + kh][g]);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                T o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = Elem<T>::from_f32(acc[r][4 * q4 + j] + b4[q4][j]);
                char* dst = ew + (r * 32 + r32) * EROW + (8 * q4 + 4 * hh) * SZ;
                if (SZ == 2) *(u32x2*)dst = *(const u32x2*)o;
                else *(u32x4*)dst = *(const u32x4*)o;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
        for (int ps = 0; ps < 16 / RPIe; ++ps) {
            const int j = ps * RPIe + pl;
            float amax[EPC], yb[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                amax[e] = -INFINITY;
                yb[e] = 0.f;
            }
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                Chunk<T> c = ld_chunk<T>(ew + ((d >> 1) * 32 + 2 * j + (d & 1)) * EROW + ch * 16);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float yv = Elem<T>::to_f32(c.v[e]);
                    const float act = leaky01(fmaf(yv, sc[e], sh[e]));
                    if (act > amax[e]) {
                        amax[e] = act;
                        yb[e] = yv;
                    }
                }
            }
            const float vm = (w0 / 2 + j < Wo) ? 1.f : 0.f;
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float gz = Elem<T>::to_f32(dav[ps].v[e]) * leaky01_slope(fmaf(yb[e], sc[e], sh[e])) * vm;
                s1[e] += gz;
                s2[e] = fmaf(gz, yb[e], s2[e]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    // ---- lanes with the same channel chunk, then the 4 waves -> psum[block][2][32]
#pragma unroll
    for (int e = 0; e < EPC; ++e)
#pragma unroll
        for (int msk = CPR; msk < 64; msk <<= 1) {
            s1[e] = wave_sum_xor(s1[e], msk);
            s2[e] = wave_sum_xor(s2[e], msk);
        }
    __syncthreads();
    float* red = (float*)smem;   // [4 waves][2][32]
    if (pl == 0) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            red[(w * 2 + 0) * 32 + ch * EPC + e] = s1[e];
            red[(w * 2 + 1) * 32 + ch * EPC + e] = s2[e];
        }
    }
    __syncthreads();
    if (tid < 64) {
        const int k = tid >> 5, c = tid & 31;
        a.psum[((size_t)blockIdx.x * 2 + k) * 32 + c] =
            red[(0 * 2 + k) * 32 + c] + red[(1 * 2 + k) * 32 + c] + red[(2 * 2 + k) * 32 + c] + red[(3 * 2 + k) * 32 + c];
    }
}

hipError_t launch_conv1_bnbwd_reduce(int dtype, const Conv1BnBwdArgs& a, hipStream_t s) {
    dtype = dtype_plain(dtype);      // f16x2: the 3-channel layer computes in exact fp32
    dim3 g(a.nblocks), b(256);
    switch (dtype) {
        case 0: hipLaunchKernelGGL(conv1_bnbwd_reduce_kernel<float>, g, b, 0, s, a); break;
        case 1: hipLaunchKernelGGL(conv1_bnbwd_reduce_kernel<half_t>, g, b, 0, s, a); break;
        case 2: hipLaunchKernelGGL(conv1_bnbwd_reduce_kernel<bf16_t>, g, b, 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace y2
