// 3x3 stride-1 SAME convolution, halo image in LDS + FILTER FRAGMENTS STRAIGHT TO REGISTERS
// (gfx950).  Second form of conv_halo.hip (same op: tf.nn.conv2d(..., 'SAME') + bias for
// filter_size 3, reference src/yolo2_nets/darknet.py:20-21,32-36, and its dgrad).
//
// conv_halo.hip streams the filter tile of every (tap, chunk) step through an LDS ring, which
// costs one workgroup barrier per step (72 per block on the 1024-channel layers), a second LDS
// read per MFMA pair and LDS-DMA writes that compete with the fragment reads.  Measured on the
// head layers: MFMA + LDS reads alone 1.24 PFLOP/s, whole kernel 0.93.  Here the filters are
// packed in MFMA-FRAGMENT ORDER (pack.hip: [cout tile of 32][tap][k-group][lane][16 B]), so a
// wave's B fragment is ONE contiguous 1-KiB global load straight into registers (L2/TCP
// resident: the WP waves that share a cout tile fetch the same lines), prefetched one step
// ahead.  Only the halo image lives in LDS, so the workgroup barrier falls once per K-chunk
// (nine tap steps), and the LDS serves a single fragment stream.
#include "common.h"
#include "conv_epilogue.h"
#include "kernels.h"

namespace y2 {

template <int V> struct IntC { static constexpr int value = V; };

template <typename T, int WP, int WC, int TP, int TC, int BKB, bool ADB>
__global__ __launch_bounds__(WP* WC * 64) void conv_haloq_kernel(ConvArgs a, int arows) {
    typedef typename Elem<T>::frag frag_t;
    constexpr int NW = WP * WC, BP = WP * TP * 32, BC = WC * TC * 32, SZ = sizeof(T);
    constexpr int LPR = BKB / 16, RPI = 64 / LPR, RPB = 256 / BKB, KG = BKB / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = w / WC, wc = w % WC;
    const int nCT = (a.Cout + BC - 1) / BC;
    const int bx = xcd_block(blockIdx.x, gridDim.x, a.xcd);
    const int ct = bx % nCT, pt = bx / nCT;
    const int m0 = pt * BP, n0 = ct * BC;
    const int pitch = a.W + 1, hw = a.H * a.W;
    const char* __restrict__ xg = (const char*)a.x;

    auto bpos = [&](int p) -> long {
        const int n = p / hw, rem = p - n * hw;
        const int h = rem / a.W, ww = rem - h * a.W;
        return (long)bpix(n, h, ww, a.H, a.W);
    };
    const int p_last = (m0 + BP - 1 < a.M) ? m0 + BP - 1 : a.M - 1;
    const long lo = bpos(m0) - pitch - 1;
    const int nrows = (int)(bpos(p_last) + pitch + 1 - lo) + 1;
    const int npieces = (nrows + RPI - 1) / RPI;
    const int abytes = arows * BKB;

    const int lrow = lane / LPR, lslot = lane % LPR;
    const int rowbytes = a.C * SZ;
    auto issueA = [&](int c, int ab) {
        const char* xs = xg + lo * (long)rowbytes + (long)c * BKB;
        char* dst = smem + ab * abytes;
        for (int i = w; i < npieces; i += NW) {
            const int row = i * RPI + lrow;
            const uint32_t off = (uint32_t)row * (uint32_t)rowbytes + (uint32_t)((lslot ^ ((row / RPB) % LPR)) * 16);
            glds16(xs + off, dst + i * 1024);
        }
    };
    // filter fragments: [cout tile of 32][tap][k-group of 32 bytes][lane][16 B]
    const int kgrow = rowbytes / 32;                     // k-groups per tap
    const char* wbase[TC];
#pragma unroll
    for (int i = 0; i < TC; ++i)
        wbase[i] = (const char*)a.w + ((size_t)(n0 / 32 + wc * TC + i) * 9 * kgrow * 64 + lane) * 16;
    auto loadB = [&](int c, int t, frag_t (&fb)[TC][KG]) {
        const size_t off = (size_t)(t * kgrow + c * KG) * 1024;
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int g = 0; g < KG; ++g) fb[i][g] = *(const frag_t*)(wbase[i] + off + g * 1024);
    };

    const int r32 = lane & 31, hh = lane >> 5;
    int arow_tl[TP];
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        int p = m0 + (wp * TP + j) * 32 + r32;
        if (p > a.M - 1) p = a.M - 1;
        arow_tl[j] = (int)(bpos(p) - pitch - 1 - lo);
    }

    f32x16 acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    const int nchunks = rowbytes / BKB;
    const int steps = nchunks * 9;
    frag_t fbq[2][TC][KG];
    issueA(0, 0);
    loadB(0, 0, fbq[0]);
    int c = 0, t = 0, kh = 0, kw = 0;
    auto step = [&](auto par, int s) {
        constexpr int P = decltype(par)::value;
        if (t == 0) {
            if (!ADB && c > 0) {
                __builtin_amdgcn_s_barrier();      // everyone is done with the single image buffer
                asm volatile("" ::: "memory");
                issueA(c, 0);
            }
            wait_vmcnt<0>();                       // this chunk's image pieces (and all older loads) have landed
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        // next step's filter fragments first (so they never queue behind an image), then the next image
        {
            int tn = t + 1, cn = c;
            if (tn == 9) { tn = 0; ++cn; }
            if (s + 1 < steps) loadB(cn, tn, fbq[P ^ 1]);
        }
        if (ADB && t == 0 && c + 1 < nchunks) issueA(c + 1, (c + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);

        const char* ab = smem + (ADB ? (c & 1) : 0) * abytes;
        const int shift = kh * pitch + kw;
        int aoff[TP], asw[TP];
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int row = arow_tl[j] + shift;
            aoff[j] = row * BKB;
            asw[j] = (row / RPB) % LPR;
        }
        auto load_frags = [&](int g, frag_t (&fp)[TP]) {
#pragma unroll
            for (int j = 0; j < TP; ++j) fp[j] = *(const frag_t*)(ab + aoff[j] + (((2 * g + hh) ^ asw[j]) * 16));
        };
        frag_t fp0[TP], fp1[TP];
        load_frags(0, fp0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < KG; g += 2) {
            if (g + 1 < KG) load_frags(g + 1, fp1);
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) mma32(acc[i][j], fbq[P][i][g], fp0[j]);
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < KG) {
                if (g + 2 < KG) load_frags(g + 2, fp0);
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int j = 0; j < TP; ++j) mma32(acc[i][j], fbq[P][i][g + 1], fp1[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (++kw == 3) { kw = 0; ++kh; }
        if (++t == 9) { t = 0; kh = 0; ++c; }
    };
    for (int s = 0; s < steps; s += 2) {
        step(IntC<0>{}, s);
        if (s + 1 < steps) step(IntC<1>{}, s + 1);
    }
    __syncthreads();
    conv_epilogue<T, WP, WC, TP, TC>(a, acc, smem, w, lane, m0, n0, pt, ct);
}

// ---------------------------------------------------------------------------
// Same kernel on 16x16x32 MFMA tiles (v_mfma_f32_16x16x32_f16/bf16, 16x16x4 f32): identical FLOPs, cycles
// and operand traffic per wave tile, but the chip holds a higher clock on this shape under load
// (MI355X_MICROARCH.md "DVFS give-back" (7); measured here with the results discarded: +7 % on the
// 1024-channel layers).  Filters in the 16-row fragment order (pack.hip frag_chunk16):
//   [cout tile of 16][tap][k-group of 64 bytes][lane = (16-byte chunk)*16 + cout%16][16 B].
// TP / TC still count 32-wide units, so tiles, LDS image and epilogue patch are those of the kernel above.
// ---------------------------------------------------------------------------
template <typename T, int WP, int WC, int TP, int TC, int BKB, bool ADB>
__global__ __launch_bounds__(WP* WC * 64) void conv_haloq16_kernel(ConvArgs a, int arows) {
    typedef typename Elem<T>::frag frag_t;
    constexpr int NW = WP * WC, BP = WP * TP * 32, BC = WC * TC * 32, SZ = sizeof(T);
    constexpr int LPR = BKB / 16, RPI = 64 / LPR, RPB = 256 / BKB, KG = BKB / 64;   // k-groups of 64 bytes
    constexpr int TP16 = 2 * TP, TC16 = 2 * TC;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = w / WC, wc = w % WC;
    const int nCT = (a.Cout + BC - 1) / BC;
    const int bx = xcd_block(blockIdx.x, gridDim.x, a.xcd);
    const int ct = bx % nCT, pt = bx / nCT;
    const int m0 = pt * BP, n0 = ct * BC;
    const int pitch = a.W + 1, hw = a.H * a.W;
    const char* __restrict__ xg = (const char*)a.x;

    auto bpos = [&](int p) -> long {
        const int n = p / hw, rem = p - n * hw;
        const int h = rem / a.W, ww = rem - h * a.W;
        return (long)bpix(n, h, ww, a.H, a.W);
    };
    const int p_last = (m0 + BP - 1 < a.M) ? m0 + BP - 1 : a.M - 1;
    const long lo = bpos(m0) - pitch - 1;
    const int nrows = (int)(bpos(p_last) + pitch + 1 - lo) + 1;
    const int npieces = (nrows + RPI - 1) / RPI;
    const int abytes = arows * BKB;

    const int lrow = lane / LPR, lslot = lane % LPR;
    const int rowbytes = a.C * SZ;
    auto issueA = [&](int c, int ab) {
        const char* xs = xg + lo * (long)rowbytes + (long)c * BKB;
        char* dst = smem + ab * abytes;
        for (int i = w; i < npieces; i += NW) {
            const int row = i * RPI + lrow;
            const uint32_t off = (uint32_t)row * (uint32_t)rowbytes + (uint32_t)((lslot ^ ((row / RPB) % LPR)) * 16);
            glds16(xs + off, dst + i * 1024);
        }
    };
    const int kgrow = rowbytes / 64;                     // 64-byte k-groups per tap
    const char* wbase[TC16];
#pragma unroll
    for (int i = 0; i < TC16; ++i)
        wbase[i] = (const char*)a.w + ((size_t)(n0 / 16 + wc * TC16 + i) * 9 * kgrow * 64 + lane) * 16;
    auto loadB = [&](int c, int t, frag_t (&fb)[TC16][KG]) {
        const size_t off = (size_t)(t * kgrow + c * KG) * 1024;
#pragma unroll
        for (int i = 0; i < TC16; ++i)
#pragma unroll
            for (int g = 0; g < KG; ++g) fb[i][g] = *(const frag_t*)(wbase[i] + off + g * 1024);
    };

    const int r16 = lane & 15, kc = lane >> 4;
    int arow_tl[TP16];
#pragma unroll
    for (int j = 0; j < TP16; ++j) {
        int p = m0 + (wp * TP16 + j) * 16 + r16;
        if (p > a.M - 1) p = a.M - 1;
        arow_tl[j] = (int)(bpos(p) - pitch - 1 - lo);
    }

    f32x4 acc[TC16][TP16];
#pragma unroll
    for (int i = 0; i < TC16; ++i)
#pragma unroll
        for (int j = 0; j < TP16; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = rowbytes / BKB;
    const int steps = nchunks * 9;
    frag_t fbq[2][TC16][KG];
    issueA(0, 0);
    loadB(0, 0, fbq[0]);
    int c = 0, t = 0, kh = 0, kw = 0;
    auto step = [&](auto par, int s) {
        constexpr int P = decltype(par)::value;
        if (t == 0) {
            if (!ADB && c > 0) {
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                issueA(c, 0);
            }
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        {
            int tn = t + 1, cn = c;
            if (tn == 9) { tn = 0; ++cn; }
            if (s + 1 < steps) loadB(cn, tn, fbq[P ^ 1]);
        }
        if (ADB && t == 0 && c + 1 < nchunks) issueA(c + 1, (c + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);

        const char* ab = smem + (ADB ? (c & 1) : 0) * abytes;
        const int shift = kh * pitch + kw;
        int aoff[TP16], asw[TP16];
#pragma unroll
        for (int j = 0; j < TP16; ++j) {
            const int row = arow_tl[j] + shift;
            aoff[j] = row * BKB;
            asw[j] = (row / RPB) % LPR;
        }
        auto load_frags = [&](int g, frag_t (&fp)[TP16]) {
#pragma unroll
            for (int j = 0; j < TP16; ++j) fp[j] = *(const frag_t*)(ab + aoff[j] + (((4 * g + kc) ^ asw[j]) * 16));
        };
        frag_t fp0[TP16], fp1[TP16];
        load_frags(0, fp0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < KG; g += 2) {
            if (g + 1 < KG) load_frags(g + 1, fp1);
#pragma unroll
            for (int i = 0; i < TC16; ++i)
#pragma unroll
                for (int j = 0; j < TP16; ++j) mma16(acc[i][j], fbq[P][i][g], fp0[j]);
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < KG) {
                if (g + 2 < KG) load_frags(g + 2, fp0);
#pragma unroll
                for (int i = 0; i < TC16; ++i)
#pragma unroll
                    for (int j = 0; j < TP16; ++j) mma16(acc[i][j], fbq[P][i][g + 1], fp1[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (++kw == 3) { kw = 0; ++kh; }
        if (++t == 9) { t = 0; kh = 0; ++c; }
    };
    for (int s = 0; s < steps; s += 2) {
        step(IntC<0>{}, s);
        if (s + 1 < steps) step(IntC<1>{}, s + 1);
    }
    __syncthreads();
    conv_epilogue16<T, WP, WC, TP, TC>(a, acc, smem, w, lane, m0, n0, pt, ct);
}

static int haloq_rows(int H, int W, int BP, int RPI) {
    const int pitch = W + 1;
    const int rows_cross = (BP - 1) / W + 1;
    const int img_cross = (BP - 1) / (H * W) + 1;
    const int span = (BP - 1) + rows_cross + img_cross * pitch;
    const int nrows = span + 2 * (pitch + 1) + 1;
    return (nrows + RPI - 1) / RPI * RPI;
}

template <typename T, int WP, int WC, int TP, int TC, int BKB, bool ADB, bool M16 = false>
static hipError_t haloq_launch(const ConvArgs& a, hipStream_t s) {
    typedef EpiCfg<T, WP, WC, TP, TC> Epi;
    constexpr int BP = WP * TP * 32, BC = WC * TC * 32, RPI = 64 / (BKB / 16);
    if ((a.C * (int)sizeof(T)) % BKB != 0) return hipErrorInvalidValue;
    const int arows = haloq_rows(a.H, a.W, BP, RPI);
    size_t lds = (size_t)(ADB ? 2 : 1) * arows * BKB;
    if (lds < (size_t)Epi::LDS) lds = Epi::LDS;
    if (lds > 160 * 1024) return hipErrorOutOfMemory;
    auto kern = M16 ? conv_haloq16_kernel<T, WP, WC, TP, TC, BKB, ADB> : conv_haloq_kernel<T, WP, WC, TP, TC, BKB, ADB>;
    static size_t attr = 0;
    if (lds > attr) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr = lds;
    }
    const int nPT = (a.M + BP - 1) / BP;
    const int nCT = (a.Cout + BC - 1) / BC;
    hipLaunchKernelGGL(kern, dim3(nPT * nCT), dim3(WP * WC * 64), lds, s, a, arows);
    return hipGetLastError();
}

template <typename T, int WP, int WC, int TP, int TC, int BKB, bool M16 = false>
static hipError_t haloq_pick(const ConvArgs& a, hipStream_t s) {
    constexpr int BP = WP * TP * 32, RPI = 64 / (BKB / 16);
    const int nchunks = a.C * (int)sizeof(T) / BKB;
    const size_t arows = haloq_rows(a.H, a.W, BP, RPI);
    if (nchunks > 1 && 2 * arows * BKB <= 150 * 1024) return haloq_launch<T, WP, WC, TP, TC, BKB, true, M16>(a, s);
    return haloq_launch<T, WP, WC, TP, TC, BKB, false, M16>(a, s);
}

template <typename T>
static hipError_t haloq_T(const ConvArgs& a, hipStream_t s, int* bp) {
    const int kb = a.C * (int)sizeof(T);
    const bool k128 = (kb % 128) == 0;
    if (!k128 && (kb % 64) != 0) return hipErrorInvalidValue;
    if (a.W > 52) {
        // long rows (104, 208): 512-pixel tiles amortise the two-row halo; one K-chunk per tile where it fits
        hipError_t e = hipErrorOutOfMemory;
        if (a.Cout > 64) {
            *bp = 256;
            e = haloq_pick<T, 4, 2, 2, 2, 64>(a, s);
        } else if (a.Cout > 32) {
            *bp = 512;
            e = k128 ? haloq_pick<T, 4, 2, 4, 1, 128>(a, s) : haloq_pick<T, 4, 2, 4, 1, 64>(a, s);
        } else {
            *bp = 512;
            e = k128 ? haloq_pick<T, 8, 1, 2, 1, 128>(a, s) : haloq_pick<T, 8, 1, 2, 1, 64>(a, s);
        }
        if (e != hipErrorOutOfMemory) return e;
        (void)hipGetLastError();
    }
    if (a.Cout > 64) {
        hipError_t e = hipErrorOutOfMemory;
        if (a.M >= 384 * 8) {   // every fragment-filter layer (W <= 26): 384-pixel tiles measured best
            *bp = 384;
            const bool narrow = ((a.M + 383) / 384) * ((a.Cout + 127) / 128) < 160 && k128;
            const bool m16 = conv_filter_layout(9, a.W, kb, a.Cout, a.M) == 2;   // filters packed for 16x16 tiles
            if (narrow) e = haloq_pick<T, 4, 2, 3, 1, 128>(a, s);
            else if (m16) e = haloq_pick<T, 4, 2, 3, 2, 128, true>(a, s);
            else e = k128 ? haloq_pick<T, 4, 2, 3, 2, 128>(a, s) : haloq_pick<T, 4, 2, 3, 2, 64>(a, s);
        } else if (a.M >= 256 * 8) {
            *bp = 256;
            e = k128 ? haloq_pick<T, 4, 2, 2, 2, 128>(a, s) : haloq_pick<T, 4, 2, 2, 2, 64>(a, s);
        }
        if (e != hipErrorOutOfMemory) return e;
        (void)hipGetLastError();
        *bp = 128;
        return haloq_pick<T, 2, 2, 2, 2, 64>(a, s);
    } else if (a.Cout > 32) {
        *bp = 256;
        return k128 ? haloq_pick<T, 4, 1, 2, 2, 128>(a, s) : haloq_pick<T, 4, 1, 2, 2, 64>(a, s);
    } else {
        *bp = 256;
        return k128 ? haloq_pick<T, 4, 1, 2, 1, 128>(a, s) : haloq_pick<T, 4, 1, 2, 1, 64>(a, s);
    }
}

// filters must be packed in fragment order (pack.hip, PackLayer::wf_frag / wd_frag)
hipError_t launch_conv_haloq(int dtype, const ConvArgs& a, hipStream_t s, int* bp) {
    if (a.taps != 9) return hipErrorInvalidValue;
    switch (dtype) {
        case 0: return haloq_T<float>(a, s, bp);
        case 1: return haloq_T<half_t>(a, s, bp);
        case 2: return haloq_T<bf16_t>(a, s, bp);
    }
    return hipErrorInvalidValue;
}

#ifdef Y2_DEV
// development variants (f16) for scripts/bench_conv.py (timing only: the bench does not care about the
// filter layout)
hipError_t launch_conv_haloq_variant(int variant, const ConvArgs& a, hipStream_t s, int* bp) {
    typedef half_t T;
#define HQ(id, WP, WC, TP, TC, BKB) \
    case id: if (bp) *bp = WP * TP * 32; return haloq_pick<T, WP, WC, TP, TC, BKB>(a, s);
    switch (variant) {
        HQ(120, 4, 2, 3, 2, 128)
        case 119: if (bp) *bp = 384; return haloq_pick<T, 4, 2, 3, 2, 128, true>(a, s);    // 16x16x32 MFMA tiles
        case 118: if (bp) *bp = 384; return haloq_pick<T, 4, 2, 3, 1, 128, true>(a, s);
        HQ(121, 4, 2, 2, 2, 128)
        HQ(122, 4, 2, 2, 2, 64)
        HQ(123, 4, 2, 3, 1, 128)
        HQ(124, 2, 2, 2, 2, 128)
        HQ(125, 2, 2, 3, 2, 128)
        HQ(126, 4, 2, 4, 2, 128)
        HQ(127, 4, 2, 3, 2, 64)
        HQ(128, 2, 4, 3, 2, 128)      // 192 x 256, 8 waves
        HQ(129, 2, 4, 2, 2, 128)      // 128 x 256
        HQ(130, 4, 1, 2, 2, 128)
        HQ(131, 4, 1, 3, 2, 128)      // 384 x 64, 4 waves
        HQ(132, 2, 2, 4, 2, 128)      // 256 x 128, 4 waves
        HQ(133, 4, 2, 4, 1, 128)      // 512 x 64
        HQ(134, 4, 2, 2, 1, 128)      // 256 x 64
        HQ(135, 4, 2, 4, 1, 64)
        HQ(136, 4, 2, 2, 1, 64)
        HQ(137, 8, 1, 2, 1, 128)      // 512 x 32
        HQ(138, 8, 1, 2, 1, 64)
        HQ(139, 4, 2, 4, 2, 64)       // 512 x 128, 64-byte chunks
    }
#undef HQ
    return hipErrorInvalidValue;
}
#endif  // Y2_DEV

}  // namespace y2
