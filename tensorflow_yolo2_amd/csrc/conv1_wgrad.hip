// First-layer weight gradient: dW[kh][kw][c][co] = sum_p x4[p (+) (kh,kw)][c] * dy[p][co]
// (TF Conv2DBackpropFilter of the 3 -> 32 layer, darknet.py:150).
// Output is tiny (27 x 32) and the reduction is over all N*H*W pixels, so the kernel is a
// pure HBM stream of dy (64 B/pixel at f16) + x4 (8 B/pixel).  MI355X design:
//   * persistent blocks walk whole image rows; per row the dy row and the three x4 rows
//     are staged with global_load_lds (dy padded to 16 pixels with reads of its zero border)
//   * D[(kh,kw,c)][co] on 32x32 MFMA with the PIXEL as k: both operands are read
//     k-strided out of the row images with ds_read_b64_tr_b16; the 16 elements
//     (kw*4+c) of one filter row are contiguous in x4, so the transposed read takes
//     overlapping 32-byte windows of the x4 row image directly (no im2col)
//   * accumulators live in registers across all rows of the block; one LDS reduction
//     over the 4 waves and 864 float atomics per block at the end.
#include <stdlib.h>
#include "common.h"
#include "kernels.h"

#ifndef Y2_C1F_MAXI
#define Y2_C1F_MAXI 0   // register-prefetched items per thread in the fused kernel (f16/bf16)
#endif

namespace y2 {

template <typename T>
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(Conv1WgradArgs a) {
    constexpr int SZ = sizeof(T);
    constexpr int DYP = 32 * SZ;   // bytes per dy pixel
    constexpr int XP = 4 * SZ;     // bytes per x4 pixel
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Wp = (a.W + 15) & ~15;                  // pixels per row, padded to the MFMA k step
    const int dy_bytes = Wp * DYP;
    const int x_bytes = ((Wp + 4) * XP + 15) & ~15;   // one x4 row image (window of the last pixel included)
    char* dy_l = smem;
    char* x_l = smem + dy_bytes;
    const int dy_chunks = dy_bytes / 16, x_chunks = x_bytes / 16;
    const int rows = a.N * a.H;
    const int r32 = lane & 31, hh = lane >> 5;
    const int qq = (lane & 15) >> 2, pp = lane & 3, g1 = (lane >> 4) & 1;

    f32x16 acc1, acc2;   // acc1: filter rows kh = 0 (MFMA rows 0..15) and 1 (16..31); acc2: kh = 2 (rows 0..15)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc1[q] = acc2[q] = 0.f;

    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int n = row / a.H, h = row - n * a.H;
        const char* dyrow = (const char*)a.dy + bpix(n, h, 0, a.H, a.W) * DYP;
        const char* dyzero = dyrow - DYP;   // left border pixel: 32 zeros
        const char* xrow = (const char*)a.x4 + (bpix(n, h, 0, a.H, a.W) - (size_t)(a.W + 2)) * XP;
        const size_t xpitch = (size_t)(a.W + 1) * XP;
        __syncthreads();   // previous row fully consumed
        for (int i0 = w * 64; i0 < dy_chunks; i0 += 256) {
            const int i = i0 + lane;
            if (i < dy_chunks) {
                const int pix = (i * 16) / DYP;
                const char* src = pix < a.W ? dyrow + (size_t)i * 16 : dyzero + (i * 16) % DYP;
                glds16(src, dy_l + i0 * 16);
            }
        }
        for (int kh = 0; kh < 3; ++kh)
            for (int i0 = w * 64; i0 < x_chunks; i0 += 256) {
                const int i = i0 + lane;
                if (i < x_chunks) glds16(xrow + kh * xpitch + (size_t)i * 16, x_l + kh * x_bytes + i0 * 16);
            }
        __syncthreads();   // hipcc drains the LDS-DMA (vmcnt(0)) before the barrier
        for (int s = w; s * 16 < a.W; s += 4) {
            const int w0 = s * 16;
            if constexpr (SZ == 2) {
                const int pix = w0 + 8 * hh + qq;
                const char* pb = dy_l + pix * DYP + (16 * g1 + 4 * pp) * 2;
                typename Elem<T>::frag fb = tr_frag<T>(pb, pb + 4 * DYP);
                const char* pa1 = x_l + g1 * x_bytes + (pix + pp) * XP;
                typename Elem<T>::frag fa1 = tr_frag<T>(pa1, pa1 + 4 * XP);
                const char* pa2 = x_l + 2 * x_bytes + (pix + pp) * XP;
                typename Elem<T>::frag fa2 = tr_frag<T>(pa2, pa2 + 4 * XP);
                mma32(acc1, fa1, fb);
                mma32(acc2, fa2, fb);
            } else {
#pragma unroll
                for (int k2 = 0; k2 < 8; ++k2) {
                    const int pix = w0 + 2 * k2 + hh;
                    const float b = *(const float*)(dy_l + pix * DYP + r32 * 4);
                    const float a1 = *(const float*)(x_l + (r32 >> 4) * x_bytes + pix * XP + (r32 & 15) * 4);
                    const float a2 = *(const float*)(x_l + 2 * x_bytes + pix * XP + (r32 & 15) * 4);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc1, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b, acc2, 0, 0, 0);
                }
            }
        }
    }
    // ---- reduce the 4 waves through LDS, then one atomic per (tap, c, co)
    __syncthreads();
    float* red = (float*)smem;   // [4 waves][48 rows][32 co]
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int r = acc_row(q, hh);
        red[(w * 48 + r) * 32 + r32] = acc1[q];
        if (r < 16) red[(w * 48 + 32 + r) * 32 + r32] = acc2[q];
    }
    __syncthreads();
    for (int i = tid; i < 48 * 32; i += 256) {
        const int r = i >> 5, co = i & 31;
        const int kh = r >> 4, e = r & 15, kw = e >> 2, c = e & 3;
        if (kw < 3 && c < 3) {
            const float v = red[i] + red[48 * 32 + i] + red[2 * 48 * 32 + i] + red[3 * 48 * 32 + i];
            atomicAdd(a.dW + ((kh * 3 + kw) * 3 + c) * 32 + co, v * a.scale);
        }
    }
}

template <typename T>
static hipError_t c1wg_T(const Conv1WgradArgs& a, hipStream_t s) {
    constexpr int SZ = sizeof(T);
    const int Wp = (a.W + 15) & ~15;
    size_t lds = (size_t)Wp * 32 * SZ + 3 * (size_t)((((Wp + 4) * 4 * SZ) + 15) & ~15);
    const size_t red = 4 * 48 * 32 * sizeof(float);
    if (lds < red) lds = red;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto kern = conv1_wgrad_kernel<T>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    int rows = a.N * a.H;
    int nb = rows < 512 ? rows : 512;
    hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_conv1_wgrad(int dtype, const Conv1WgradArgs& a, hipStream_t s) {
    dtype = dtype_plain(dtype);      // f16x2: the 3-channel layer computes in exact fp32
    switch (dtype) {
        case 0: return c1wg_T<float>(a, s);
        case 1: return c1wg_T<half_t>(a, s);
        case 2: return c1wg_T<bf16_t>(a, s);
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------
// Fused form for the pooled first layer: the batch-norm backward APPLY pass
//   dy_d = scale*dz_d - (ka + kb*y_d),  dz = dA * leaky'(z) at the 2x2 arg-max   (bn.hip)
// is computed here, row pair by row pair, straight into the LDS dy images the MFMAs read.
// dy of the first layer has exactly one consumer (this kernel: there is no dgrad below
// conv1), so at 416x416x64 the 709 MB tensor is never written nor re-read: the apply pass
// (886 MB in, 709 MB out) disappears and this kernel streams y + dA (886 MB) instead of dy.
// ---------------------------------------------------------------------------
template <typename T, int MAXI>
__global__ __launch_bounds__(256, 2) void conv1_wgrad_fused_kernel(Conv1WgradFusedArgs a) {
    constexpr int SZ = sizeof(T), EPC = 16 / SZ, CPP = 32 / EPC;
    constexpr int DYP = 32 * SZ, XP = 4 * SZ;
    constexpr int NTH = 256, NW = NTH / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Wp = (a.W + 15) & ~15;
    const int dy_bytes = Wp * DYP;
    const int x_bytes = ((Wp + 4) * XP + 15) & ~15;
    char* dy_l = smem;                   // [2 rows][Wp][32]
    char* x_l = smem + 2 * dy_bytes;     // [4 rows][x_bytes]
    const int x_chunks = x_bytes / 16;
    const int Ho = a.H / 2, Wo = a.W / 2;
    const int prs = a.N * Ho;
    const int r32 = lane & 31, hh = lane >> 5;
    const int qq = (lane & 15) >> 2, pp = lane & 3, g1 = (lane >> 4) & 1;

    // k-padding pixels of the two dy rows: zero, once
    for (int i = tid; i < 2 * (Wp - a.W) * CPP; i += NTH) {
        const int r = i / ((Wp - a.W) * CPP), j = i % ((Wp - a.W) * CPP);
        *(u32x4*)(dy_l + r * dy_bytes + a.W * DYP + j * 16) = u32x4{0, 0, 0, 0};
    }
    const int ch = tid % CPP, c0 = ch * EPC;
    float sc[EPC], sh[EPC], nka[EPC], nkb[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        sc[e] = a.scale[c0 + e];
        sh[e] = a.shift[c0 + e];
        nka[e] = -a.coef[c0 + e];
        nkb[e] = -a.coef[32 + c0 + e];
    }

    f32x16 acc1, acc2;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc1[q] = acc2[q] = 0.f;

    // one work item = one pooled pixel x EPC channels: 1 dA chunk + 4 y chunks in, 4 dy chunks out.
    // The first MAXI items per thread of the NEXT row pair are loaded into registers before the
    // MFMA phase of the current one (HBM latency hides behind it); the rest load in place.
    const int nitems = Wo * CPP;
    u32x4 pda[MAXI ? MAXI : 1], py[MAXI ? MAXI : 1][4];   // raw 16-byte vectors (kept packed)
    auto item_ptrs = [&](int pr, const char*& yrow, const char*& darow) {
        const int n = pr / Ho, ho = pr - n * Ho;
        yrow = (const char*)a.y + ((size_t)(n * a.H + 2 * ho) * a.W) * DYP;
        darow = (const char*)a.dA + ((size_t)(n * Ho + ho) * Wo) * DYP;
    };
    auto prefetch = [&](int pr) {
        const char *yrow, *darow;
        item_ptrs(pr, yrow, darow);
#pragma unroll
        for (int k = 0; k < MAXI; ++k) {
            const int item = tid + k * NTH;
            if (item < nitems) {
                const int wo = item / CPP;
                pda[k] = *(const u32x4*)(darow + (size_t)item * 16);
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    py[k][d] = *(const u32x4*)(yrow + ((size_t)(d >> 1) * a.W + 2 * wo + (d & 1)) * DYP + ch * 16);
            }
        }
    };
    auto process = [&](const u32x4& dar, const u32x4 (&yr)[4], int item) {
        const int wo = item / CPP;
        Chunk<T> dav, yv[4];
        *(u32x4*)dav.v = dar;
#pragma unroll
        for (int d = 0; d < 4; ++d) *(u32x4*)yv[d].v = yr[d];
        int arg[EPC];
        float amax[EPC], yb[EPC], gz[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            arg[e] = 0;
            amax[e] = -INFINITY;
            yb[e] = 0.f;
        }
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float yy = Elem<T>::to_f32(yv[d].v[e]);
                const float act = leaky01(fmaf(yy, sc[e], sh[e]));
                if (act > amax[e]) {
                    amax[e] = act;
                    yb[e] = yy;
                    arg[e] = d;
                }
            }
#pragma unroll
        for (int e = 0; e < EPC; ++e)
            gz[e] = Elem<T>::to_f32(dav.v[e]) * leaky01_slope(fmaf(yb[e], sc[e], sh[e]));
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            Chunk<T> o;
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float t = fmaf(nkb[e], Elem<T>::to_f32(yv[d].v[e]), nka[e]);
                o.v[e] = Elem<T>::from_f32((arg[e] == d) ? fmaf(sc[e], gz[e], t) : t);
            }
            st_chunk<T>(dy_l + (d >> 1) * dy_bytes + (size_t)(2 * wo + (d & 1)) * DYP + ch * 16, o);
        }
    };
    if (MAXI && blockIdx.x < prs) prefetch(blockIdx.x);

    for (int pr = blockIdx.x; pr < prs; pr += gridDim.x) {
        const int n = pr / Ho, ho = pr - n * Ho, h0 = 2 * ho;
        const char* xrow = (const char*)a.x4 + (bpix(n, h0, 0, a.H, a.W) - (size_t)(a.W + 2)) * XP;
        const size_t xpitch = (size_t)(a.W + 1) * XP;
        __syncthreads();   // previous row pair fully consumed
        for (int kh = 0; kh < 4; ++kh)
            for (int i0 = w * 64; i0 < x_chunks; i0 += NTH) {
                const int i = i0 + lane;
                if (i < x_chunks) glds16(xrow + kh * xpitch + (size_t)i * 16, x_l + kh * x_bytes + i0 * 16);
            }
#pragma unroll
        for (int k = 0; k < MAXI; ++k) {
            const int item = tid + k * NTH;
            if (item < nitems) process(pda[k], py[k], item);
        }
        {
            const char *yrow, *darow;
            item_ptrs(pr, yrow, darow);
            for (int item = tid + MAXI * NTH; item < nitems; item += NTH) {
                const int wo = item / CPP;
                const u32x4 dav = *(const u32x4*)(darow + (size_t)item * 16);
                u32x4 yv[4];
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    yv[d] = *(const u32x4*)(yrow + ((size_t)(d >> 1) * a.W + 2 * wo + (d & 1)) * DYP + ch * 16);
                process(dav, yv, item);
            }
        }
        if (MAXI && pr + (int)gridDim.x < prs) prefetch(pr + gridDim.x);
        __syncthreads();   // LDS-DMA drained (vmcnt(0)) and the dy images complete
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const char* dyr = dy_l + r * dy_bytes;
            for (int s = w; s * 16 < a.W; s += NW) {
                const int w0 = s * 16;
                if constexpr (SZ == 2) {
                    const int pix = w0 + 8 * hh + qq;
                    const char* pb = dyr + pix * DYP + (16 * g1 + 4 * pp) * 2;
                    typename Elem<T>::frag fb = tr_frag<T>(pb, pb + 4 * DYP);
                    const char* pa1 = x_l + (r + g1) * x_bytes + (pix + pp) * XP;
                    typename Elem<T>::frag fa1 = tr_frag<T>(pa1, pa1 + 4 * XP);
                    const char* pa2 = x_l + (r + 2) * x_bytes + (pix + pp) * XP;
                    typename Elem<T>::frag fa2 = tr_frag<T>(pa2, pa2 + 4 * XP);
                    mma32(acc1, fa1, fb);
                    mma32(acc2, fa2, fb);
                } else {
#pragma unroll
                    for (int k2 = 0; k2 < 8; ++k2) {
                        const int pix = w0 + 2 * k2 + hh;
                        const float b = *(const float*)(dyr + pix * DYP + r32 * 4);
                        const float a1 = *(const float*)(x_l + (r + (r32 >> 4)) * x_bytes + pix * XP + (r32 & 15) * 4);
                        const float a2 = *(const float*)(x_l + (r + 2) * x_bytes + pix * XP + (r32 & 15) * 4);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc1, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b, acc2, 0, 0, 0);
                    }
                }
            }
        }
    }
    // ---- reduce the waves through LDS, then one atomic per (tap, c, co)
    __syncthreads();
    float* red = (float*)smem;   // [NW waves][48 rows][32 co]
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int r = acc_row(q, hh);
        red[(w * 48 + r) * 32 + r32] = acc1[q];
        if (r < 16) red[(w * 48 + 32 + r) * 32 + r32] = acc2[q];
    }
    __syncthreads();
    for (int i = tid; i < 48 * 32; i += NTH) {
        const int r = i >> 5, co = i & 31;
        const int kh = r >> 4, e = r & 15, kw = e >> 2, c = e & 3;
        if (kw < 3 && c < 3) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < NW; ++k) v += red[k * 48 * 32 + i];
            atomicAdd(a.dW + ((kh * 3 + kw) * 3 + c) * 32 + co, v * a.inv_grad_scale);
        }
    }
}

template <typename T, int MAXI>
static hipError_t c1wgf_M(const Conv1WgradFusedArgs& a, hipStream_t s) {
    constexpr int SZ = sizeof(T);
    const int Wp = (a.W + 15) & ~15;
    size_t lds = 2 * (size_t)Wp * 32 * SZ + 4 * (size_t)((((Wp + 4) * 4 * SZ) + 15) & ~15);
    const size_t red = 4 * 48 * 32 * sizeof(float);
    if (lds < red) lds = red;
    if (lds > 160 * 1024) return hipErrorOutOfMemory;
    auto kern = conv1_wgrad_fused_kernel<T, MAXI>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int prs = a.N * (a.H / 2);
    const int nb = prs < 512 ? prs : 512;
    hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, s, a);
    return hipGetLastError();
}

template <typename T>
static hipError_t c1wgf_T(const Conv1WgradFusedArgs& a, hipStream_t s) {
    // measured at 416x416x64: prefetching the next row pair into registers (MAXI 2) is no faster
    // than loading in place with two blocks per CU (249 vs 245 us), so the simple form ships
    if (sizeof(T) == 2 && Y2_C1F_MAXI > 0) return c1wgf_M<T, Y2_C1F_MAXI>(a, s);
    return c1wgf_M<T, 0>(a, s);
}

bool conv1_wgrad_fused_ok(int H, int W, int pool, int ldy, int elem_size) {
    const int Wp = (W + 15) & ~15;
    const size_t lds = 2 * (size_t)Wp * 32 * elem_size + 4 * (size_t)((((Wp + 4) * 4 * elem_size) + 15) & ~15);
    return pool && (H % 2) == 0 && (W % 2) == 0 && ldy == 32 && lds <= 160 * 1024;   // two dy rows + four x rows in LDS
}

hipError_t launch_conv1_wgrad_fused(int dtype, const Conv1WgradFusedArgs& a, hipStream_t s) {
    dtype = dtype_plain(dtype);      // f16x2: the 3-channel layer computes in exact fp32
    switch (dtype) {
        case 0: return c1wgf_T<float>(a, s);
        case 1: return c1wgf_T<half_t>(a, s);
        case 2: return c1wgf_T<bf16_t>(a, s);
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------
// LINEAR form for the pooled first layer (kernels.h: Conv1WgradLinArgs): nothing of the first layer's conv
// output is read -- it is never stored.  Per row pair: the four input rows go to LDS, dz (= dA * leaky'(z) at the
// window's arg-max position, zero at the other three: both from the forward pass's ysel / idx) is scattered into
// two LDS row images, and the tr-read MFMAs accumulate X(dz) and the Gram matrix G of the input patches.
// Streams x4 + dA + ysel + idx = 42 B per pixel pair-of-rows-quarter instead of y + dA + x4 = 88 B/pixel.
// Each block leaves its partial [48*32 + 48*48] in `part`; conv1_lin_reduce adds the blocks (fixed order: no
// atomics, run-to-run identical), conv1_dw_finalize combines with the BN-backward constants.
// ---------------------------------------------------------------------------
constexpr int kLinAcc = 48 * 32 + 48 * 48;

constexpr int kLinThreads = 256;   // measured: 512 threads + register prefetch of the next row pair 297 us, this form 252
constexpr int kLinItems = 0;     // items (pooled pixel x chunk) per thread held in registers one row pair ahead

// GRAM: the Gram matrix is accumulated here (rounds 2-3); false: the forward pass of this step left it in a.gram
// (conv1_gram_kernel below: the batch-norm statistics of the layer come from it too) -- three of five MFMAs and the
// tail masks drop out, the partial record shrinks from 15 KB to 6 KB
// NOSEL (kernels.h Conv1PoolArgs::idx3): no conv output is read; the slope of the activation at the arg-max comes with
// the position (3 bits per channel), and this kernel's S2 records stay zero -- conv1_lin_s2_kernel adds sum g * y as one
// record from the reduced X(dz)
// XS (f16x2 mode, T = float: round 5): the fp32 row images are read as in the exact-fp32 form, split into (hi, lo) halves
// in registers, and X(dz) / G are three v_mfma_f32_32x32x16_f16 per 16-pixel group and product (common.h mma32_split)
// instead of eight v_mfma_f32_32x32x2_f32: 15 matrix instructions of 32 cycles per group instead of 40 of 64.
// XS == 2 (f16x2f, round 6): the same with the HI planes alone -- one product per accumulator, half the LDS row images (no
// column segments at 416), a third of the matrix instructions: the backward contractions of that mode read f16(dz) and f16(x)
template <typename T, bool GRAM, bool NOSEL, int XS = 0>
__global__ __launch_bounds__(kLinThreads, 2) void conv1_wgrad_lin_kernel(Conv1WgradLinArgs a, float* part) {
    // also the BN-backward REDUCE of this layer, for free: dA and ysel pass through here anyway, and nothing in this
    // kernel needs the sums (S1 = sum g, S2 = sum g * ysel -> psum[block][2][32]; the finalize runs after it)
    constexpr int SZ = sizeof(T), EPC = 16 / SZ, CPP = 32 / EPC;
    // XS: the LDS row images are TWO half planes (hi, lo) in the 16-bit types' layout -- dz is split once where it is
    // scattered, x once where it is staged -- and the matrix section is the 16-bit one on plane pairs
    constexpr int LSZ = XS ? 2 : SZ;
    constexpr int NPLN = XS == 1 ? 2 : 1;          // LDS planes per row image
    constexpr int DYP = 32 * LSZ, XP = 4 * LSZ;
    static_assert(!XS || SZ == 4, "the in-register split reads fp32 operands");
    constexpr int NTH = kLinThreads, NW = NTH / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nseg = a.nseg > 1 ? a.nseg : 1;          // column segments per row pair (Conv1WgradLinArgs::nseg)
    const int Wp = nseg > 1 ? a.ws : ((a.W + 15) & ~15);
    const int dz_bytes = Wp * DYP;
    const int x_bytes = ((Wp + 4) * XP + 15) & ~15;
    char* dz_l = smem;                   // [2 rows][Wp][32]            (XS: [2 planes][2 rows][Wp][32] halves)
    char* x_l = smem + NPLN * 2 * dz_bytes;         // [4 rows][x_bytes]  (XS 1: [2 planes][4 rows][x_bytes])
    const int x_chunks = x_bytes / 16;
    const int dz_plane = 2 * dz_bytes, x_plane = 4 * x_bytes;      // XS: byte distance of the lo plane
    const int Ho = a.H / 2, Wo = a.W / 2;
    const int prs = a.N * Ho;
    const int r32 = lane & 31, hh = lane >> 5;
    const int qq = (lane & 15) >> 2, pp = lane & 3, g1 = (lane >> 4) & 1;

    {   // k-padding pixels of the dz rows: zero, once
        constexpr int LCPP = DYP / 16;      // 16-byte chunks per pixel of an LDS row image
        const int padw = nseg > 1 ? 0 : Wp - a.W;       // (segments: zeroed per unit, below)
        for (int i = tid; i < NPLN * 2 * padw * LCPP; i += NTH) {
            const int r = i / (padw * LCPP), j = i % (padw * LCPP);
            *(u32x4*)(dz_l + r * dz_bytes + a.W * DYP + j * 16) = u32x4{0, 0, 0, 0};
        }
    }
    const int ch = tid % CPP, c0 = ch * EPC;
    float sc[EPC], sh[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        sc[e] = a.scale[c0 + e];
        sh[e] = a.shift[c0 + e];
    }
    f32x16 acc1, acc2, g11, g12, g22;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc1[q] = acc2[q] = g11[q] = g12[q] = g22[q] = 0.f;
    float s1[EPC], s2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s1[e] = s2[e] = 0.f;

    int nitems = Wo * CPP;
    // the dA / ysel / idx of a row pair are requested while the MFMAs of the previous one run (two blocks of eight
    // waves per CU: in-place loads leave the HBM latency of every row pair exposed)
    u32x4 pda[kLinItems ? kLinItems : 1], pys[kLinItems ? kLinItems : 1];
    unsigned pix_[kLinItems ? kLinItems : 1];
    auto prefetch = [&](int pr) {
        const int n = pr / Ho, ho = pr - n * Ho;
        const size_t prow = ((size_t)n * Ho + ho) * Wo;
#pragma unroll
        for (int k = 0; k < kLinItems; ++k) {
            const int item = tid + k * NTH;
            if (item < nitems) {
                pda[k] = *(const u32x4*)((const char*)a.dA + (prow * CPP + item) * 16);
                if constexpr (NOSEL) {
                    pix_[k] = a.idx3[prow * CPP + item];
                } else {
                    pys[k] = *(const u32x4*)((const char*)a.ysel + (prow * CPP + item) * 16);
                    pix_[k] = a.idx[prow * CPP + item];
                }
            }
        }
    };
    auto process = [&](const u32x4& dar, const u32x4& ysr, unsigned ix, int item) {
        const int wo = item / CPP;
        Chunk<T> dav, ysv;
        *(u32x4*)dav.v = dar;
        *(u32x4*)ysv.v = ysr;
        T gz[EPC];
        constexpr int IB = NOSEL ? 3 : 2;      // index bits per channel
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            float g;
            if constexpr (NOSEL) {
                g = Elem<T>::to_f32(dav.v[e]) * (((ix >> (3 * e + 2)) & 1u) ? 0.1f : 1.0f);
            } else {
                const float yf = Elem<T>::to_f32(ysv.v[e]);
                g = Elem<T>::to_f32(dav.v[e]) * leaky01_slope(fmaf(yf, sc[e], sh[e]));
                s2[e] = fmaf(g, yf, s2[e]);
            }
            s1[e] += g;
            gz[e] = Elem<T>::from_f32(g);
        }
        if constexpr (XS) {
            half_t gh[EPC], gl[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) split_f16(gz[e], gh[e], gl[e]);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                half_t oh[EPC], ol[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const bool here = ((ix >> (IB * e)) & 3u) == (unsigned)d;
                    oh[e] = here ? gh[e] : (half_t)0.f;
                    ol[e] = here ? gl[e] : (half_t)0.f;
                }
                char* dst = dz_l + (d >> 1) * dz_bytes + (size_t)(2 * wo + (d & 1)) * DYP + ch * 8;
                *(u32x2*)dst = *(const u32x2*)oh;
                if constexpr (XS == 1) *(u32x2*)(dst + dz_plane) = *(const u32x2*)ol;
            }
        } else {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            Chunk<T> o;
#pragma unroll
            for (int e = 0; e < EPC; ++e) o.v[e] = (((ix >> (IB * e)) & 3u) == (unsigned)d) ? gz[e] : Elem<T>::from_f32(0.f);
            st_chunk<T>(dz_l + (d >> 1) * dz_bytes + (size_t)(2 * wo + (d & 1)) * DYP + ch * 16, o);
        }
        }
    };
    for (int un = blockIdx.x; un < prs * nseg; un += gridDim.x) {
        const int pr = un / nseg, seg = un - pr * nseg;
        const int cbeg = seg * Wp;                                  // first pixel column of this unit
        const int Wv = a.W - cbeg < Wp ? a.W - cbeg : Wp;           // its width (nseg == 1: the whole row, a.W)
        nitems = (Wv / 2) * CPP;
        const int n = pr / Ho, ho = pr - n * Ho, h0 = 2 * ho;
        const char* xrow = (const char*)a.x4 + (bpix(n, h0, 0, a.H, a.W) - (size_t)(a.W + 2) + (size_t)cbeg) * (4 * SZ);
        const size_t xpitch = (size_t)(a.W + 1) * (4 * SZ);
        const size_t prow = ((size_t)n * Ho + ho) * Wo + cbeg / 2;
        // this row pair's first kBatch items per thread: every load issued before the first one is used (left to the
        // compiler the loop below loads, waits and processes item by item: three to four exposed latencies per row pair)
        constexpr int kBatch = kLinItems ? 0 : 4;
        u32x4 bda[kBatch ? kBatch : 1], bys[kBatch ? kBatch : 1];
        unsigned bix[kBatch ? kBatch : 1];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const int item = tid + k * NTH;
            if (item < nitems) {
                bda[k] = *(const u32x4*)((const char*)a.dA + (prow * CPP + item) * 16);
                if constexpr (NOSEL) {
                    bix[k] = a.idx3[prow * CPP + item];
                } else {
                    bys[k] = *(const u32x4*)((const char*)a.ysel + (prow * CPP + item) * 16);
                    bix[k] = a.idx[prow * CPP + item];
                }
            }
        }
        // XS: the four fp32 input rows come through registers (every load issued before the barrier), are split once and
        // land as two half planes in the 16-bit types' image layout (4 halves per pixel)
        constexpr int kXB = XS ? 4 : 1;      // (register budget: 256 with two blocks per CU)
        const int xpix = Wp + 4;           // pixels per staged row (x_bytes / XP)
        u32x4 xv[kXB];
        if constexpr (XS) {
#pragma unroll
            for (int k = 0; k < kXB; ++k) {
                const int i = tid + k * NTH;
                if (i < 4 * xpix) xv[k] = *(const u32x4*)(xrow + (i / xpix) * xpitch + (size_t)(i % xpix) * 16);
            }
        }
        __syncthreads();   // previous row pair fully consumed
        if (nseg > 1 && Wv < Wp) {      // a narrower last segment: its k-padding pixels held the previous unit's dz
            constexpr int LCPP = DYP / 16;
            for (int i = tid; i < NPLN * 2 * (Wp - Wv) * LCPP; i += NTH) {
                const int r = i / ((Wp - Wv) * LCPP), j = i % ((Wp - Wv) * LCPP);
                *(u32x4*)(dz_l + r * dz_bytes + Wv * DYP + j * 16) = u32x4{0, 0, 0, 0};
            }
        }
        // XS (fp32 dA: 6.5 items per thread at W = 416): a second batch of items is requested here, behind the barrier, and
        // lands while the input rows are split and the first batch is scattered (the tail loop below pays one exposed
        // latency per item)
        constexpr int kBatch2 = (XS && NOSEL) ? 3 : 0;
        u32x4 bda2[kBatch2 ? kBatch2 : 1], bys2[kBatch2 ? kBatch2 : 1];
        unsigned bix2[kBatch2 ? kBatch2 : 1];
#pragma unroll
        for (int k = 0; k < kBatch2; ++k) {
            const int item = tid + (kBatch + k) * NTH;
            if (item < nitems) {
                bda2[k] = *(const u32x4*)((const char*)a.dA + (prow * CPP + item) * 16);
                if constexpr (NOSEL) {
                    bix2[k] = a.idx3[prow * CPP + item];
                } else {
                    bys2[k] = *(const u32x4*)((const char*)a.ysel + (prow * CPP + item) * 16);
                    bix2[k] = a.idx[prow * CPP + item];
                }
            }
        }
        if constexpr (XS) {
            auto put = [&](int i, const u32x4& v) {
                const f32x4 f = __builtin_bit_cast(f32x4, v);
                half_t h[4], l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) split_f16(f[e], h[e], l[e]);
                char* dst = x_l + (i / xpix) * x_bytes + (i % xpix) * XP;
                *(u32x2*)dst = *(const u32x2*)h;
                if constexpr (XS == 1) *(u32x2*)(dst + x_plane) = *(const u32x2*)l;
            };
#pragma unroll
            for (int k = 0; k < kXB; ++k) {
                const int i = tid + k * NTH;
                if (i < 4 * xpix) put(i, xv[k]);
            }
            for (int i = tid + kXB * NTH; i < 4 * xpix; i += NTH)      // wider images: the rest in place
                put(i, *(const u32x4*)(xrow + (i / xpix) * xpitch + (size_t)(i % xpix) * 16));
        } else {
        for (int kh = 0; kh < 4; ++kh)
            for (int i0 = w * 64; i0 < x_chunks; i0 += NTH) {
                const int i = i0 + lane;
                if (i < x_chunks) glds16(xrow + kh * xpitch + (size_t)i * 16, x_l + kh * x_bytes + i0 * 16);
            }
        }
#pragma unroll
        for (int k = 0; k < kLinItems; ++k) {
            const int item = tid + k * NTH;
            if (item < nitems) process(pda[k], pys[k], pix_[k], item);
        }
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const int item = tid + k * NTH;
            if (item < nitems) process(bda[k], NOSEL ? bda[k] : bys[k], bix[k], item);
        }
#pragma unroll
        for (int k = 0; k < kBatch2; ++k) {
            const int item = tid + (kBatch + k) * NTH;
            if (item < nitems) process(bda2[k], NOSEL ? bda2[k] : bys2[k], bix2[k], item);
        }
        for (int item = tid + (kLinItems + kBatch + kBatch2) * NTH; item < nitems; item += NTH) {    // wider images: the rest in place
            const u32x4 dar = *(const u32x4*)((const char*)a.dA + (prow * CPP + item) * 16);
            if constexpr (NOSEL) process(dar, dar, a.idx3[prow * CPP + item], item);
            else process(dar, *(const u32x4*)((const char*)a.ysel + (prow * CPP + item) * 16), a.idx[prow * CPP + item], item);
        }
        __syncthreads();   // LDS-DMA drained (vmcnt(0)) and the dz images complete
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const char* dzr = dz_l + r * dz_bytes;
            for (int s = w; s * 16 < Wv; s += NW) {
                const int w0 = s * 16;
                if constexpr (SZ == 2) {
                    const int pix = w0 + 8 * hh + qq;
                    const char* pb = dzr + pix * DYP + (16 * g1 + 4 * pp) * 2;
                    typename Elem<T>::frag fb = tr_frag<T>(pb, pb + 4 * DYP);
                    const char* pa1 = x_l + (r + g1) * x_bytes + (pix + pp) * XP;
                    typename Elem<T>::frag fa1 = tr_frag<T>(pa1, pa1 + 4 * XP);
                    const char* pa2 = x_l + (r + 2) * x_bytes + (pix + pp) * XP;
                    typename Elem<T>::frag fa2 = tr_frag<T>(pa2, pa2 + 4 * XP);
                    // Gram operands: the SAME registers serve as B (k = pixel on both sides); k-padding pixels of
                    // the last group hold the next row's data in the x image: masked out of one side
                    mma32(acc1, fa1, fb);
                    mma32(acc2, fa2, fb);
                    if constexpr (GRAM) {
                        typename Elem<T>::frag fg1 = fa1, fg2 = fa2;
                        if (w0 + 16 > Wv) {
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                if (w0 + 8 * hh + j >= Wv) { fg1[j] = (T)0.f; fg2[j] = (T)0.f; }
                        }
                        mma32(g11, fa1, fg1);
                        mma32(g12, fa1, fg2);
                        mma32(g22, fa2, fg2);
                    }
                } else if constexpr (XS == 2) {
                    // hi planes only: the 16-bit section as it stands, one product per accumulator
                    const int pix = w0 + 8 * hh + qq;
                    const char* pb = dzr + pix * DYP + (16 * g1 + 4 * pp) * 2;
                    const f16x8 bh = tr_frag<half_t>(pb, pb + 4 * DYP);
                    const char* pa1 = x_l + (r + g1) * x_bytes + (pix + pp) * XP;
                    const f16x8 a1h = tr_frag<half_t>(pa1, pa1 + 4 * XP);
                    const char* pa2 = x_l + (r + 2) * x_bytes + (pix + pp) * XP;
                    const f16x8 a2h = tr_frag<half_t>(pa2, pa2 + 4 * XP);
                    mma32(acc1, a1h, bh);
                    mma32(acc2, a2h, bh);
                    if constexpr (GRAM) {
                        f16x8 m1h = a1h, m2h = a2h;
                        if (w0 + 16 > Wv) {
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                if (w0 + 8 * hh + j >= Wv) { m1h[j] = m2h[j] = (half_t)0.f; }
                        }
                        mma32(g11, a1h, m1h);
                        mma32(g12, a1h, m2h);
                        mma32(g22, a2h, m2h);
                    }
                } else if constexpr (XS == 1) {
                    // the 16-bit section on plane pairs: three products per accumulator (common.h mma32_split)
                    const int pix = w0 + 8 * hh + qq;
                    const char* pb = dzr + pix * DYP + (16 * g1 + 4 * pp) * 2;
                    const f16x8 bh = tr_frag<half_t>(pb, pb + 4 * DYP);
                    const f16x8 bl = tr_frag<half_t>(pb + dz_plane, pb + dz_plane + 4 * DYP);
                    const char* pa1 = x_l + (r + g1) * x_bytes + (pix + pp) * XP;
                    const f16x8 a1h = tr_frag<half_t>(pa1, pa1 + 4 * XP);
                    const f16x8 a1l = tr_frag<half_t>(pa1 + x_plane, pa1 + x_plane + 4 * XP);
                    const char* pa2 = x_l + (r + 2) * x_bytes + (pix + pp) * XP;
                    const f16x8 a2h = tr_frag<half_t>(pa2, pa2 + 4 * XP);
                    const f16x8 a2l = tr_frag<half_t>(pa2 + x_plane, pa2 + x_plane + 4 * XP);
                    mma32_split(acc1, a1h, a1l, bh, bl);
                    mma32_split(acc2, a2h, a2l, bh, bl);
                    if constexpr (GRAM) {
                        f16x8 m1h = a1h, m1l = a1l, m2h = a2h, m2l = a2l;
                        if (w0 + 16 > Wv) {       // k-padding pixels hold the next row's data: masked out of one side
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                if (w0 + 8 * hh + j >= Wv) { m1h[j] = m1l[j] = m2h[j] = m2l[j] = (half_t)0.f; }
                        }
                        mma32_split(g11, a1h, a1l, m1h, m1l);
                        mma32_split(g12, a1h, a1l, m2h, m2l);
                        mma32_split(g22, a2h, a2l, m2h, m2l);
                    }
                } else {
#pragma unroll
                    for (int k2 = 0; k2 < 8; ++k2) {
                        const int pix = w0 + 2 * k2 + hh;
                        const float vm = pix < Wv ? 1.f : 0.f;
                        const float b = *(const float*)(dzr + pix * DYP + r32 * 4);
                        const float a1 = *(const float*)(x_l + (r + (r32 >> 4)) * x_bytes + pix * XP + (r32 & 15) * 4);
                        const float a2 = *(const float*)(x_l + (r + 2) * x_bytes + pix * XP + (r32 & 15) * 4);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc1, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b, acc2, 0, 0, 0);
                        if constexpr (GRAM) {
                            g11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, a1 * vm, g11, 0, 0, 0);
                            g12 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, a2 * vm, g12, 0, 0, 0);
                            g22 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, a2 * vm, g22, 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    // ---- the waves through LDS, four at a time (fixed order) -> this block's partial: Xdz [48][32] then G [48][48]
    float* red = (float*)smem;   // [4][kLinAcc] (G's lower-left block is filled by symmetry in the finalize)
    constexpr int kAcc = GRAM ? kLinAcc : 48 * 32;      // entries of this block's record that carry sums
    constexpr int PER = (kAcc + NTH - 1) / NTH;
    float tot[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) tot[k] = 0.f;
    for (int round = 0; round < NW / 4; ++round) {
        __syncthreads();
        if ((w >> 2) == round) {
            float* o = red + (w & 3) * kLinAcc;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int r = acc_row(q, hh);
                o[r * 32 + r32] = acc1[q];
                if (r < 16) o[(32 + r) * 32 + r32] = acc2[q];
                if constexpr (GRAM) {
                    float* g = o + 48 * 32;
                    g[r * 48 + r32] = g11[q];
                    if (r32 < 16) g[r * 48 + 32 + r32] = g12[q];
                    if (r < 16 && r32 < 16) g[(32 + r) * 48 + 32 + r32] = g22[q];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = tid + k * NTH;
            if (i < kAcc) {
                bool used = true;
                if (i >= 48 * 32) {
                    const int j = i - 48 * 32, r = j / 48, c = j % 48;
                    used = !(r >= 32 && c < 32);
                }
                if (used) tot[k] += (red[i] + red[kLinAcc + i]) + (red[2 * kLinAcc + i] + red[3 * kLinAcc + i]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = tid + k * NTH;
        if (i < kAcc) part[(size_t)blockIdx.x * kLinAcc + i] = tot[k];
    }
    // ---- BN-backward partial sums of this block: threads with the same chunk (tid % CPP) are CPP apart
    __syncthreads();
    float* sr = (float*)smem;    // [2][NTH][EPC]
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        sr[tid * EPC + e] = s1[e];
        sr[(NTH + tid) * EPC + e] = s2[e];
    }
    __syncthreads();
    if (tid < 64) {
        const int k = tid >> 5, cc = tid & 31;
        const int chunk = cc / EPC, e = cc % EPC;
        float t = 0.f;
        for (int j = chunk; j < NTH; j += CPP) t += sr[(k * NTH + j) * EPC + e];
        a.psum[((size_t)blockIdx.x * 2 + k) * 32 + cc] = t;
    }
}

// block partials -> kLinMid slice sums (fixed order; a 2-D grid so that 512 partial records of 15 KB do not queue
// behind one another in 15 blocks); the finalize adds the kLinMid slices
constexpr int kLinMid = 16;
__global__ __launch_bounds__(256) void conv1_lin_reduce_kernel(const float* part, int nblocks, float* mid, int first,
                                                               int count) {
    const int i = first + blockIdx.x * 256 + threadIdx.x;      // entries [first, first + count) of the kLinAcc-float records
    if (i >= first + count) return;
    const int per = (nblocks + kLinMid - 1) / kLinMid;
    const int b0 = blockIdx.y * per;
    int b1 = b0 + per;
    if (b1 > nblocks) b1 = nblocks;
    float v4[4] = {0.f, 0.f, 0.f, 0.f};
    int b = b0;
    for (; b + 7 < b1; b += 8) {       // eight records' loads up front; the accumulators add in the four-record loop's order
        float r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = part[(size_t)(b + u) * kLinAcc + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) v4[u & 3] += r[u];
    }
    for (; b + 3 < b1; b += 4) {
        float r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = part[(size_t)(b + u) * kLinAcc + i];
#pragma unroll
        for (int u = 0; u < 4; ++u) v4[u] += r[u];
    }
    for (; b < b1; ++b) v4[0] += part[(size_t)b * kLinAcc + i];
    mid[(size_t)blockIdx.y * kLinAcc + i] = (v4[0] + v4[1]) + (v4[2] + v4[3]);
}

// NOSEL form: S2 = sum_p dz y of the first layer from the reduced X(dz) (rows r = kh*16 + kw*4 + c, the 16 slice sums of
// conv1_lin_reduce_kernel): y = Wq^T patch + b per pixel with Wq the filter as the forward pass reads it (rounded to T),
// so sum_p dz[co] y[co] = sum_r Wq[r][co] X(dz)[r][co] + b[co] sum_p dz[co].  Written as record P (S1 = 0) behind the P
// block records of conv1_wgrad_lin_kernel: bn_bwd_finalize adds it like any other.  One block, 32 channels x 32 slices.
template <typename T>
__global__ __launch_bounds__(1024) void conv1_lin_s2_kernel(const float* mid, const float* W, const float* bias, float* psum,
                                                            int P) {
    __shared__ double red[32][33], red1[32][33];
    const int co = threadIdx.x & 31, sl = threadIdx.x >> 5;
    {                                                // S1 of the block records: slice sl takes records sl, sl + 32, ...
        double s1 = 0.0;
        int p = sl;
        for (; p + 7 * 32 < P; p += 8 * 32) {      // eight records' loads up front (this one-block kernel is a chain of
            float r[8];                            // dependent loads at the tail of the backward pass); same order
#pragma unroll
            for (int u = 0; u < 8; ++u) r[u] = psum[((size_t)(p + 32 * u) * 2 + 0) * 32 + co];
#pragma unroll
            for (int u = 0; u < 8; ++u) s1 += (double)r[u];
        }
        for (; p < P; p += 32) s1 += (double)psum[((size_t)p * 2 + 0) * 32 + co];
        red1[sl][co] = s1;
    }
    __syncthreads();
    double t = 0.0;
    if (sl < 27) {                                   // slice = tap-channel tc = (kh*3 + kw)*3 + c
        const int c = sl % 3, kw = (sl / 3) % 3, kh = sl / 9;
        const int r = kh * 16 + kw * 4 + c;
        float xdz = 0.f;
#pragma unroll
        for (int k = 0; k < kLinMid; ++k) xdz += mid[(size_t)k * kLinAcc + r * 32 + co];
        t = (double)Elem<T>::to_f32(Elem<T>::from_f32(W[sl * 32 + co])) * (double)xdz;
    } else if (sl == 27) {                           // b * S1
        double s1 = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) s1 += red1[k][co];
        t = (double)bias[co] * s1;
    }
    red[sl][co] = t;
    __syncthreads();
    if (sl == 0) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 28; ++k) v += red[k][co];
        psum[((size_t)P * 2 + 0) * 32 + co] = 0.f;
        psum[((size_t)P * 2 + 1) * 32 + co] = (float)v;
    }
}

// dW[t][c][co] = inv_gs * (scale X(dz) - ka X(1) - kb (G W + b X(1)))   (rows r = kh*16 + kw*4 + c)
__global__ __launch_bounds__(1024) void conv1_dw_finalize_kernel(Conv1DwFinalizeArgs a) {
    __shared__ float G[48][49];
    __shared__ float Wp[48][32];
    const int tid = threadIdx.x;
    auto total = [&](int i) {     // slice sums of conv1_lin_reduce_kernel, fixed order
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < kLinMid; ++k) v += a.acc[(size_t)k * kLinAcc + i];
        return v;
    };
    for (int i = tid; i < 48 * 48; i += 1024) {
        const int r = i / 48, c = i % 48;
        if (a.gram) G[r][c] = a.gram[i];                      // complete [48][48] totals of the forward pass
        else G[r][c] = (r >= 32 && c < 32) ? total(48 * 32 + c * 48 + r) : total(48 * 32 + i);   // symmetry
    }
    for (int i = tid; i < 48 * 32; i += 1024) {
        const int r = i >> 5, co = i & 31;
        const int kh = r >> 4, e = r & 15, kw = e >> 2, c = e & 3;
        Wp[r][co] = (kw < 3 && c < 3) ? a.W[((kh * 3 + kw) * 3 + c) * 32 + co] : 0.f;
    }
    __syncthreads();
    if (tid >= 27 * 32) return;
    const int co = tid & 31, tc = tid >> 5;           // tc = (kh*3 + kw)*3 + c
    const int c = tc % 3, kw = (tc / 3) % 3, kh = tc / 9;
    const int r = kh * 16 + kw * 4 + c;
    constexpr int kOnes = 1 * 16 + 1 * 4 + 3;          // centre tap, channel 3 (= 1 inside the image)
    const float x1 = G[kOnes][r];
    float gw = 0.f;
    for (int k = 0; k < 48; ++k) gw = fmaf(G[r][k], Wp[k][co], gw);
    const float xy = gw + a.bias[co] * x1;
    const float xdz = total(r * 32 + co);
    const float ka = a.coef[co], kb = a.coef[32 + co];
    a.dW[tc * 32 + co] = (a.scale[co] * xdz - ka * x1 - kb * xy) * a.inv_grad_scale;
}

// ---------------------------------------------------------------------------
// Round 4: the Gram matrix in the FORWARD pass.  G depends on the input alone, and with y = W^T p + b per pixel (p = the
// 27-element input patch, its centre-tap channel 3 = 1) the batch-norm statistics of the layer follow from it:
//     sum_p y = W^T s + M b,   sum_p (y - mean)^2 = W^T (G - s s^T / M) W,   s = G[ones][.],  M = G[ones][ones]
// so the statistics-only convolution pass of the pooled first layer (conv1_stats_kernel: 97 us at 416x416x64, VALU- and
// latency-bound on 7 vector operations per output element) becomes this MFMA-only sweep of the input (8 B per pixel),
// and the backward pass (conv1_wgrad_lin_kernel<.., false>) no longer builds G.  The statistics are those of the
// UN-ROUNDED conv output (the stored-value statistics of the other layers differ from them by the half-precision
// rounding noise, ~2^-12 relative and unbiased); the centred form is evaluated in double.
// ---------------------------------------------------------------------------
constexpr int kGram = 48 * 48;
constexpr int kGramMaxBlocks = 768;
// Tile = RT output rows (4 where H % 4 == 0, else 2) x the whole image row: RT + 2 input rows in LDS (each input row is
// staged (RT + 2) / RT times: 1.5x at RT = 4), double-buffered.  Work item = (32-pixel group, pair of output rows): four
// input-row fragments X_q [16 patch elements (kw, c) x 32 pixels] by transposed LDS reads -- the 16 elements of one filter
// row are contiguous in the 4-channel image, so a fragment is two ds_read_b64_tr_b16 of overlapping 32-byte windows --
// and G_block(kh, kh') += X_{r+kh} X_{r+kh'}^T on 16x16x32 MFMAs with the SAME registers as both operands (six of the
// nine 16x16 blocks: G is symmetric): 12 MFMAs of 16 cycles per 64 output pixels.
template <typename T>
__global__ __launch_bounds__(256, 2) void conv1_gram_kernel(const void* x4, int N, int H, int W, int RT, float* part) {
    static_assert(sizeof(T) == 2, "half-precision modes only (the f32 parity mode keeps the statistics-only conv pass)");
    constexpr int SZ = 2, XP = 4 * SZ;
    constexpr int NTH = 256, NW = 4;
    typedef typename Elem<T>::frag frag_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Wp = (W + 31) & ~31;
    const int x_bytes = ((Wp + 4) * XP + 15) & ~15;
    const int x_chunks = x_bytes / 16;
    const int nrow = RT + 2;
    char* x_l = smem;                    // [2 buffers][RT + 2 rows][x_bytes]
    const int Ht = H / RT;
    const int ntile = N * Ht;
    const int g4 = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    f32x4 gb[6];                         // (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
#pragma unroll
    for (int k = 0; k < 6; ++k) gb[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto stage = [&](int t, int buf) {
        const int n = t / Ht, h0 = (t - n * Ht) * RT;
        const char* xrow = (const char*)x4 + (bpix(n, h0, 0, H, W) - (size_t)(W + 2)) * XP;
        const size_t xpitch = (size_t)(W + 1) * XP;
        char* dst = x_l + buf * nrow * x_bytes;
        for (int kh = 0; kh < nrow; ++kh)
            for (int i0 = w * 64; i0 < x_chunks; i0 += NTH) {
                const int i = i0 + lane;
                if (i < x_chunks) glds16(xrow + kh * xpitch + (size_t)i * 16, dst + kh * x_bytes + i0 * 16);
            }
    };
    const int ngrp = (W + 31) / 32, nitem = ngrp * (RT / 2);
    int buf = 0;
    if ((int)blockIdx.x < ntile) stage(blockIdx.x, 0);
    for (int t = blockIdx.x; t < ntile; t += gridDim.x, buf ^= 1) {
        __syncthreads();   // this tile's rows have landed (vmcnt(0) before the barrier); the other buffer is free
        if (t + (int)gridDim.x < ntile) stage(t + gridDim.x, buf ^ 1);
        const char* xb = x_l + buf * nrow * x_bytes;
        for (int it = w; it < nitem; it += NW) {
            const int s = it % ngrp, r0 = 2 * (it / ngrp);      // output rows r0, r0 + 1: input rows r0 .. r0 + 3
            const int w0 = s * 32;
            const int pix = w0 + 8 * g4 + qq;
            frag_t X[4], B[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const char* pa = xb + (r0 + q) * x_bytes + (pix + pp) * XP;
                X[q] = tr_frag<T>(pa, pa + 4 * XP);
                B[q] = X[q];
            }
            if (w0 + 32 > W) {      // k-padding pixels of the last group hold neighbouring data: masked out of one side
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (w0 + 8 * g4 + j >= W) B[q][j] = (T)0.f;
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                mma16(gb[0], X[r], B[r]);
                mma16(gb[1], X[r], B[r + 1]);
                mma16(gb[2], X[r], B[r + 2]);
                mma16(gb[3], X[r + 1], B[r + 1]);
                mma16(gb[4], X[r + 1], B[r + 2]);
                mma16(gb[5], X[r + 2], B[r + 2]);
            }
        }
    }
    // ---- the four waves through LDS (fixed order) -> this block's partial G [48][48] (lower blocks by symmetry)
    __syncthreads();
    float* red = (float*)smem;   // [4][kGram]
    {
        float* g = red + w * kGram;
        const int bi[6] = {0, 0, 0, 1, 1, 2}, bj[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) g[(bi[k] * 16 + 4 * g4 + q) * 48 + bj[k] * 16 + (lane & 15)] = gb[k][q];
    }
    __syncthreads();
    for (int i = tid; i < kGram; i += NTH) {
        const int r = i / 48, c = i % 48;
        const int j = (r / 16 > c / 16) ? c * 48 + r : i;       // blocks below the diagonal: the transposed entry
        part[(size_t)blockIdx.x * kGram + i] = (red[j] + red[kGram + j]) + (red[2 * kGram + j] + red[3 * kGram + j]);
    }
}

// slices of the block partials (fixed order), then one block: totals in double, batch-norm statistics of the layer
__global__ __launch_bounds__(256) void conv1_gram_reduce_kernel(const float* part, int nblocks, float* mid) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= kGram) return;
    const int per = (nblocks + kLinMid - 1) / kLinMid;
    const int b0 = blockIdx.y * per;
    int b1 = b0 + per;
    if (b1 > nblocks) b1 = nblocks;
    float v4[4] = {0.f, 0.f, 0.f, 0.f};
    int b = b0;
    for (; b + 3 < b1; b += 4)
#pragma unroll
        for (int u = 0; u < 4; ++u) v4[u] += part[(size_t)(b + u) * kGram + i];
    for (; b < b1; ++b) v4[0] += part[(size_t)b * kGram + i];
    mid[(size_t)blockIdx.y * kGram + i] = (v4[0] + v4[1]) + (v4[2] + v4[3]);
}

template <typename T>
__global__ __launch_bounds__(1024) void conv1_gram_stats_kernel(Conv1GramStatsArgs a) {
    __shared__ double G[48][49];
    __shared__ float Wq[48][32];
    __shared__ double t[48][33];
    const int tid = threadIdx.x;
    for (int i = tid; i < kGram; i += 1024) {
        double v = 0.0;
#pragma unroll 4      // (fully unrolled, the 1024-thread block's 128-register budget spilled; the order of the sum is the same)
        for (int k = 0; k < kLinMid; ++k) v += (double)a.mid[(size_t)k * kGram + i];
        G[i / 48][i % 48] = v;
        a.gram[i] = (float)v;                               // the backward pass of this step reads the totals
    }
    for (int i = tid; i < 48 * 32; i += 1024) {
        const int r = i >> 5, co = i & 31;
        const int kh = r >> 4, e = r & 15, kw = e >> 2, c = e & 3;
        // the convolution multiplies the filter as PACKED (rounded to T): the statistics must see the same values
        Wq[r][co] = (kw < 3 && c < 3) ? Elem<T>::to_f32(Elem<T>::from_f32(a.W[((kh * 3 + kw) * 3 + c) * 32 + co])) : 0.f;
    }
    __syncthreads();
    constexpr int kOnes = 1 * 16 + 1 * 4 + 3;               // centre tap, channel 3 (= 1 inside the image)
    const double M = G[kOnes][kOnes];
    const double invM = M > 0 ? 1.0 / M : 0.0;
    // centred Gram matrix in place: C = G - s s^T / M (the row of ones is read before anything is overwritten)
    __shared__ double sv[48];
    if (tid < 48) sv[tid] = G[kOnes][tid];
    __syncthreads();
    for (int i = tid; i < kGram; i += 1024) {
        const int r = i / 48, c = i % 48;
        G[r][c] -= sv[r] * sv[c] * invM;
    }
    __syncthreads();
    for (int i = tid; i < 48 * 32; i += 1024) {
        const int r = i >> 5, co = i & 31;
        double v = 0.0;
#pragma unroll 4
        for (int k = 0; k < 48; ++k) v += G[r][k] * (double)Wq[k][co];
        t[r][co] = v * (double)Wq[r][co];
    }
    __syncthreads();
    if (tid < 32) {
        const int c = tid;
        double m2 = 0.0, s = 0.0;
#pragma unroll 4
        for (int r = 0; r < 48; ++r) {
            m2 += t[r][c];
            s += sv[r] * (double)Wq[r][c];
        }
        if (m2 < 0) m2 = 0;
        const double mean = (double)a.bias[c] + s / M;
        // the tail of bn_finalize_kernel (bn.hip), expression for expression
        const float var = (float)(M > 0 ? m2 / M : 0.0);
        const float meanf = (float)mean;
        const float inv = 1.0f / sqrtf(var + a.eps);
        const float sc = a.gamma[c] * inv;
        a.scale[c] = sc;
        a.shift[c] = a.beta[c] - meanf * sc;
        a.mean[c] = meanf;
        a.invstd[c] = inv;
        float vu = var;
        if (a.bessel && M > 1.0) vu = (float)(m2 / (M - 1.0));
        if (a.var) a.var[c] = vu;
        if (a.update_moving) {
            const float dec = 1.0f - a.momentum;
            a.moving_mean[c] -= (a.moving_mean[c] - meanf) * dec;
            a.moving_var[c] -= (a.moving_var[c] - vu) * dec;
        }
    }
}

size_t conv1_gram_scratch_floats() { return (size_t)kGram * (1 + kLinMid + kGramMaxBlocks); }
bool conv1_gram_ok(int H, int W, int elem_size) {
    const int Wp = (W + 31) & ~31;
    const size_t lds = 2 * 6 * (size_t)((((Wp + 4) * 4 * elem_size) + 15) & ~15);
    return elem_size == 2 && (H % 2) == 0 && lds <= 72 * 1024;
}
template <typename T>
static hipError_t c1gram_T(const Conv1GramStatsArgs& a, hipStream_t s) {
    const int Wp = (a.Wd + 31) & ~31;
    const int RT = (a.H % 4) == 0 ? 4 : 2;
    size_t lds = 2 * (size_t)(RT + 2) * (size_t)((((Wp + 4) * 4 * sizeof(T)) + 15) & ~15);
    const size_t red = 4 * (size_t)kGram * sizeof(float);
    if (lds < red) lds = red;
    auto kern = conv1_gram_kernel<T>;
    // the attribute is set once per size class (a call per training forward was host work on the hot path and
    // is not allowed inside a stream capture)
    static size_t attr = 0;
    if (lds > attr) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr = lds;
    }
    const int ntile = a.N * (a.H / RT);
    const int nb = ntile < kGramMaxBlocks ? ntile : kGramMaxBlocks;
    float* part = a.mid + (size_t)kLinMid * kGram;
    hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, s, a.x4, a.N, a.H, a.Wd, RT, part);
    hipLaunchKernelGGL(conv1_gram_reduce_kernel, dim3((kGram + 255) / 256, kLinMid), dim3(256), 0, s, part, nb, a.mid);
    hipLaunchKernelGGL(conv1_gram_stats_kernel<T>, dim3(1), dim3(1024), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_conv1_gram_stats(int dtype, const Conv1GramStatsArgs& a, hipStream_t s) {
    dtype = dtype_plain(dtype);      // f16x2: the 3-channel layer computes in exact fp32
    switch (dtype) {
        case 1: return c1gram_T<half_t>(a, s);
        case 2: return c1gram_T<bf16_t>(a, s);
    }
    return hipErrorInvalidValue;
}

template <typename T, int XS = 0>
static hipError_t c1lin_T(const Conv1WgradLinArgs& a0, hipStream_t s) {
    constexpr int SZ = sizeof(T);
    constexpr int ISZ = XS == 2 ? 2 : SZ;        // bytes per element of the LDS row images (XS 1: two half planes = 4)
    Conv1WgradLinArgs a = a0;
    int Wp = (a.W + 15) & ~15;
    auto images = [](int wp) { return 2 * (size_t)wp * 32 * ISZ + 4 * (size_t)((((wp + 4) * 4 * ISZ) + 15) & ~15); };
    size_t lds = images(Wp);
    if (XS && lds > 80 * 1024) {     // fp32-wide row images: column segments, so that two workgroups share a CU
        static const int want = getenv("Y2_CONV1_LIN_NSEG") ? atoi(getenv("Y2_CONV1_LIN_NSEG")) : 2;    // (A/B knob)
        a.nseg = want < 1 ? 1 : (want > 8 ? 8 : want);
        a.ws = ((a.W + a.nseg - 1) / a.nseg + 15) & ~15;
        while (a.nseg > 1 && (a.nseg - 1) * a.ws >= a.W) --a.nseg;     // no empty segment
        if (a.nseg == 1) a.ws = 0;
        else lds = images(Wp = a.ws);
    }
    size_t red = 4 * (size_t)kLinAcc * sizeof(float);
    const size_t red2 = 2 * (size_t)kLinThreads * (16 / SZ) * sizeof(float);
    if (red < red2) red = red2;
    if (lds < red) lds = red;
    if (lds > 160 * 1024) return hipErrorOutOfMemory;
    const bool nosel = a.idx3 != nullptr;
    if (nosel && !(a.Wf && a.bias)) return hipErrorInvalidValue;
    auto kern = nosel ? (a.gram ? conv1_wgrad_lin_kernel<T, false, true, XS> : conv1_wgrad_lin_kernel<T, true, true, XS>)
                      : (a.gram ? conv1_wgrad_lin_kernel<T, false, false, XS> : conv1_wgrad_lin_kernel<T, true, false, XS>);
    static size_t attr[4] = {0, 0, 0, 0};        // per kernel form of this T
    const int form = (nosel ? 2 : 0) + (a.gram ? 1 : 0);
    if (lds > attr[form]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr[form] = lds;
    }
    const int prs = a.N * (a.H / 2) * (a.nseg > 1 ? a.nseg : 1);
    const int nb = prs < 512 ? prs : 512;
    float* part = a.acc + (size_t)kLinMid * kLinAcc;        // [nb][kLinAcc] behind the slice sums
    if (a.nblocks_out) *a.nblocks_out = nb;
    const int count = a.gram ? 48 * 32 : kLinAcc;
    hipLaunchKernelGGL(kern, dim3(nb), dim3(kLinThreads), lds, s, a, part);
    hipLaunchKernelGGL(conv1_lin_reduce_kernel, dim3((count + 255) / 256, kLinMid), dim3(256), 0, s, part, nb, a.acc, 0, count);
    if (nosel) {
        hipLaunchKernelGGL(conv1_lin_s2_kernel<T>, dim3(1), dim3(1024), 0, s, a.acc, a.Wf, a.bias, a.psum, nb);
        if (a.nblocks_out) *a.nblocks_out = nb + 1;
    }
    return hipGetLastError();
}

bool conv1_wgrad_lin_ok(int H, int W, int pool, int ldy, int elem_size) {
    const int Wp = (W + 15) & ~15;
    size_t lds = 2 * (size_t)Wp * 32 * elem_size + 4 * (size_t)((((Wp + 4) * 4 * elem_size) + 15) & ~15);
    const size_t red = 4 * (size_t)kLinAcc * sizeof(float);
    if (lds < red) lds = red;
    return pool && (H % 2) == 0 && (W % 2) == 0 && ldy == 32 && lds <= 160 * 1024;
}
size_t conv1_wgrad_lin_scratch_floats() { return (size_t)kLinAcc * (kLinMid + 512); }

hipError_t launch_conv1_wgrad_lin(int dtype, const Conv1WgradLinArgs& a, hipStream_t s) {
    dtype = dtype_plain(dtype);      // f16x2: fp32 operands (exact fp32, or Conv1WgradLinArgs::xs)
    switch (dtype) {
        case 0: return a.xs == 2 ? c1lin_T<float, 2>(a, s) : (a.xs ? c1lin_T<float, 1>(a, s) : c1lin_T<float>(a, s));
        case 1: return c1lin_T<half_t>(a, s);
        case 2: return c1lin_T<bf16_t>(a, s);
    }
    return hipErrorInvalidValue;
}
hipError_t launch_conv1_dw_finalize(const Conv1DwFinalizeArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(conv1_dw_finalize_kernel, dim3(1), dim3(1024), 0, s, a);
    return hipGetLastError();
}

}  // namespace y2
