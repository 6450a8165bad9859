// Implicit-GEMM stride-1 SAME convolution (3x3 or 1x1) on MFMA, gfx950.
//
// Replaces tf.nn.conv2d(x, W, [1,1,1,1], 'SAME') + bias add
// (reference src/yolo2_nets/darknet.py:20-21,32-36) and, with tap-flipped
// transposed weights, its Conv2DBackpropInput (dgrad).
//
// Layout (MI355X-first, not TF's):
//   x   : zero-bordered NHWC  [N][H+2][W+2][C]   -> no bounds checks, every tap
//         of every pixel row is one contiguous BKB-byte read
//   w   : packed [Cout_pad][taps][C] (K contiguous per output channel)
//   y   : [M = N*H*W][ldy]
// GEMM view: D[cout][pixel] += Wp[cout][k] * X[pixel][k], k = (tap, c).
// The accumulator tile keeps the pixel on the lane and 4 consecutive couts in
// 4 consecutive registers, so the epilogue packs 4 couts per LDS store, then
// re-reads whole 16-byte chunks per pixel row and stores full lines to HBM.
// Batch-norm sufficient statistics (per block: count, mean, M2 about that
// mean -- Chan/Welford form, never E[x^2]-E[x]^2) are produced by the same
// epilogue from the values as stored.
//
// Staging: global_load_lds 16 B/lane, double-buffered LDS, XOR swizzle applied
// on the per-lane SOURCE address and on the ds_read (LDS image stays linear,
// cdna guide rule 21), conflict-free for ds_read_b128.
#include <stdlib.h>
#include "common.h"
#include "conv_epilogue.h"
#include "kernels.h"

#ifndef Y2_CONV_STAGES
#define Y2_CONV_STAGES 3
#endif

namespace y2 {

// NS = LDS stages of the global_load_lds pipeline (NS-1 K-steps in flight)
template <typename T, int WP, int WC, int TP, int TC, int BKB, int NS_>
struct ConvCfg {
    static constexpr int NS = NS_;
    static constexpr int NW = WP * WC;
    static constexpr int NT = NW * 64;
    static constexpr int BP = WP * TP * 32;  // pixels per block
    static constexpr int BC = WC * TC * 32;  // output channels per block
    static constexpr int SZ = sizeof(T);
    static constexpr int LPR = BKB / 16;     // lanes per staged row
    static constexpr int RPI = 64 / LPR;     // rows per glds wave-instruction
    static constexpr int RPB = 256 / BKB;    // rows per 256-B LDS bank row
    static constexpr int NI_P = BP / RPI;
    static constexpr int NI_C = BC / RPI;
    static constexpr int NI = NI_P + NI_C;
    static constexpr int IPW = (NI + NW - 1) / NW;
    static constexpr int STAGE = (BP + BC) * BKB;
    static constexpr int KG = BKB / 32;
    // epilogue staging: per wave [TP*32 pixels][TC*32 couts] + 16 B row pad
    static constexpr int EROW = TC * 32 * SZ + 16;
    static constexpr int EPW = TP * 32 * EROW;
    static constexpr int ESTAT = NW * TC * 32 * 2 * 4;
    static constexpr int IPW_MIN = NI / NW;
    static constexpr int LDS_MAIN = NS * STAGE;
    static constexpr int LDS_EPI = NW * EPW + ESTAT;
    static constexpr int LDS = LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI;
    static_assert(NI_P % NW == 0, "pixel rows must split evenly over waves");
};

// ABL: timing-only ablation bits (dev): 1 skip pixel-tile loads, 2 skip filter-tile loads,
// 4 skip MFMAs, 8 skip LDS fragment reads.  0 in every product launch.
// PL2 (f16x2 mode, round 5; as conv_haloq.hip): a staged row of either operand is [BKB/2 bytes of the hi plane | BKB/2 of the
// lo plane] of the K chunk, and a K step runs the three plane products on it instead of the K loop running three plane passes
// KS (round 6; 1x1 launches of fewer than 3072 pixels -- single images, the reference's batch 24 at 7x7, the ResNet swap's 7x7
// units): the K steps are split over a.ks_splits workgroups per tile, each leaves its fp32 partial tile in a.ks_scratch
// [split][M][ldy] and conv_ks_finish (conv_haloq.hip) adds them in split order -- as the 3x3 kernels' haloq_ks: such a launch
// is a handful of workgroups walking a serial K loop at one memory latency per step
template <typename T, int WP, int WC, int TP, int TC, int BKB, int NS, int ABL = 0, bool PL2 = false, bool KS = false>
__global__ __launch_bounds__(WP* WC * 64) void conv_igemm_kernel(ConvArgs a) {
    typedef ConvCfg<T, WP, WC, TP, TC, BKB, NS> Cfg;
    typedef typename Elem<T>::frag frag_t;
    typedef typename Types<T>::op_t OT;      // operand type in LDS / the fragments (f16x2 mode: half planes)
    typedef typename Types<T>::out_t YT;     // what the epilogue stores
    constexpr bool SPLIT = Types<T>::kSplit;
    constexpr int NW = Cfg::NW, BP = Cfg::BP, BC = Cfg::BC, SZ = Cfg::SZ;
    constexpr int LPR = Cfg::LPR, RPI = Cfg::RPI, RPB = Cfg::RPB, IPW = Cfg::IPW, KG = Cfg::KG;
    static_assert(!PL2 || (SPLIT && Types<T>::kPasses == 3 && KG >= 2 && KG % 2 == 0), "the two-plane form: split operands, an even number of k-groups");
    constexpr int KGH = KG / 2;
    // 16-byte chunk `src` of a staged row -> byte offset in the operand's row (PL2: first half hi plane, second half lo plane)
    auto chunk_off = [&](uint32_t src) -> uint32_t {
        return PL2 ? (src % (LPR / 2)) * 16u + (src / (LPR / 2)) * (uint32_t)(a.C * 2) : src * 16u;
    };
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = w / WC, wc = w % WC;
    const int nCT = (a.Cout + BC - 1) / BC;
    int bx = xcd_block(blockIdx.x, gridDim.x, a.xcd);
    int split = 0;
    if constexpr (KS) {
        const int tiles = ((a.M + BP - 1) / BP) * nCT;
        split = bx / tiles;
        bx -= split * tiles;
    }
    const int ct = bx % nCT, pt = bx / nCT;
    const int m0 = pt * BP, n0 = ct * BC;
    const int Ktot = a.taps * a.C;  // elements per packed weight row
    const char* __restrict__ xg = (const char*)a.x;
    const char* __restrict__ wg = (const char*)a.w;

    // ---- per-lane source offsets for the staging loads
    uint32_t voff[IPW];
    const int lrow = lane / LPR, lslot = lane % LPR;
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int ii = i * NW + w;
        const int row = ii * RPI + lrow;
        if (i * NW < Cfg::NI_P) {  // pixel rows (compile-time per i)
            const int p = m0 + row;
            uint32_t base = 0;
            if (p < a.M) {
                const int hw = a.H * a.W;
                const int n = p / hw, rem = p - n * hw;
                const int h = rem / a.W, ww = rem - h * a.W;
                // top-left tap of the 3x3 window (bordered layout, common.h)
                base = (uint32_t)(bpix(n, h, ww, a.H, a.W) - (size_t)(a.W + 2)) * (uint32_t)(a.C * SZ);
            }
            voff[i] = base + chunk_off((uint32_t)(lslot ^ ((row / RPB) % LPR)));
        } else {
            const int r = row - BP;
            voff[i] = (uint32_t)(n0 + r) * (uint32_t)(Ktot * SZ) + chunk_off((uint32_t)(lslot ^ ((r / RPB) % LPR)));
        }
    }
    // k-chunks per tap; f16x2: three passes over the planes, [x hi | x lo | x hi] against [w hi | w hi | w lo]
    const int npl = (a.C * (int)sizeof(OT)) / BKB;
    const int cpt = PL2 ? (a.C * (int)sizeof(OT)) / (BKB / 2) : Types<T>::kPasses * npl;
    int k_begin = 0, nK = a.taps * cpt;          // this workgroup's K steps [k_begin, nK)
    if constexpr (KS) {
        const int per = (nK + a.ks_splits - 1) / a.ks_splits;
        k_begin = split * per;
        nK = k_begin + per < nK ? k_begin + per : nK;
        if (k_begin > nK) k_begin = nK;
    }
    const int rowpitch = (a.W + 1) * a.C * SZ;

    auto stage = [&](int kk, int buf) {
        const int t = kk / cpt, c = kk - t * cpt;
        int tapoff;
        if (a.taps == 9) {
            const int kh = t / 3, kw = t - kh * 3;
            tapoff = kh * rowpitch + kw * a.C * SZ;
        } else {
            tapoff = rowpitch + a.C * SZ;
        }
        const char* xs = xg + tapoff + (PL2 ? c * (BKB / 2) : split_act_chunk<SPLIT>(c, npl) * BKB);
        const char* ws = wg + (size_t)(t * a.C * SZ + (PL2 ? c * (BKB / 2) : split_flt_chunk<SPLIT>(c, npl) * BKB));
        char* lbase = smem + buf * Cfg::STAGE;
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int ii = i * NW + w;
            if (i * NW < Cfg::NI_P) {
                if (!(ABL & 1)) glds16(xs + voff[i], lbase + ii * 1024);
            } else if ((i + 1) * NW <= Cfg::NI || ii < Cfg::NI) {
                if (!(ABL & 2)) glds16(ws + voff[i], lbase + ii * 1024);
            }
        }
    };

    // ---- fragment read offsets (swizzle term is tile-independent: rows differ by 32)
    const int r32 = lane & 31, hh = lane >> 5;
    int foff[KG];
#pragma unroll
    for (int g = 0; g < KG; ++g) foff[g] = r32 * BKB + (((2 * g + hh) ^ ((r32 / RPB) % LPR)) * 16);
    const int pbase = (wp * TP) * 32 * BKB;
    const int cbase = BP * BKB + (wc * TC) * 32 * BKB;

    f32x16 acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    // ---- software pipeline: NS-1 K-steps of LDS-DMA stay in flight ACROSS the barrier.
    // Raw s_barrier + counted vmcnt (a __syncthreads() would drain vmcnt(0) every step).
    //   iteration kk:  wait until stage kk has landed (this wave's part)  -> barrier (everyone's
    //   part landed AND everyone finished reading stage kk-1) -> refill the buffer stage kk-1
    //   used with stage kk+NS-1 -> MFMAs on stage kk.
#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0)
        if (k_begin + s0 < nK) stage(k_begin + s0, s0);
    int cbuf = 0, ibuf = NS - 1;
    for (int kk = k_begin; kk < nK; ++kk) {
        if (kk + NS - 2 < nK) wait_vmcnt<(NS - 2) * Cfg::IPW_MIN>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kk + NS - 1 < nK) stage(kk + NS - 1, ibuf);
        const char* lb = smem + cbuf * Cfg::STAGE;
        if constexpr (PL2) {
#pragma unroll
            for (int q = 0; q < KGH; ++q) {
                frag_t wh[TC], wl[TC], xh[TP], xl[TP];
#pragma unroll
                for (int i = 0; i < TC; ++i) {
                    wh[i] = *(const frag_t*)(lb + cbase + i * 32 * BKB + foff[q]);
                    wl[i] = *(const frag_t*)(lb + cbase + i * 32 * BKB + foff[KGH + q]);
                }
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    xh[j] = *(const frag_t*)(lb + pbase + j * 32 * BKB + foff[q]);
                    xl[j] = *(const frag_t*)(lb + pbase + j * 32 * BKB + foff[KGH + q]);
                }
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int j = 0; j < TP; ++j) {
                        mma32(acc[i][j], wh[i], xh[j]);
                        mma32(acc[i][j], wh[i], xl[j]);
                        mma32(acc[i][j], wl[i], xh[j]);
                    }
            }
        } else
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            frag_t fc[TC], fp[TP];
            if (ABL & 8) {
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int e = 0; e < Elem<T>::kPerFrag; ++e) fc[i][e] = (OT)(float)(kk + e);
#pragma unroll
                for (int j = 0; j < TP; ++j)
#pragma unroll
                    for (int e = 0; e < Elem<T>::kPerFrag; ++e) fp[j][e] = (OT)(float)(kk - e);
            } else {
#pragma unroll
                for (int i = 0; i < TC; ++i) fc[i] = *(const frag_t*)(lb + cbase + i * 32 * BKB + foff[g]);
#pragma unroll
                for (int j = 0; j < TP; ++j) fp[j] = *(const frag_t*)(lb + pbase + j * 32 * BKB + foff[g]);
            }
            if (ABL & 4) {
#pragma unroll
                for (int i = 0; i < TC; ++i) asm volatile("" ::"v"(fc[i]));
#pragma unroll
                for (int j = 0; j < TP; ++j) asm volatile("" ::"v"(fp[j]));
            } else {
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int j = 0; j < TP; ++j) mma32(acc[i][j], fc[i], fp[j]);
            }
        }
        cbuf = (cbuf + 1 == NS) ? 0 : cbuf + 1;
        ibuf = (ibuf + 1 == NS) ? 0 : ibuf + 1;
    }
    __syncthreads();
    if constexpr (KS) {
        // fp32 partial tile: 4 consecutive couts (registers 4 q4 .. 4 q4 + 3) per 16-byte store
        float* const part = a.ks_scratch + (size_t)split * a.M * a.ldy;
        const int r32_ = lane & 31, hh_ = lane >> 5;
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const int p = m0 + (wp * TP + j) * 32 + r32_;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int co = n0 + (wc * TC + i) * 32 + 8 * q4 + 4 * hh_;
                    if (p < a.M && co < a.ldy)
                        *(f32x4*)(part + (size_t)p * a.ldy + co) =
                            f32x4{acc[i][j][4 * q4], acc[i][j][4 * q4 + 1], acc[i][j][4 * q4 + 2], acc[i][j][4 * q4 + 3]};
                }
            }
        return;
    }
    if (ABL & 16) {   // dev: skip the epilogue (keep the accumulators alive)
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j) t += acc[i][j][0] + acc[i][j][7];
        if (t == 123.456f) ((float*)a.y)[0] = t;
        return;
    }
    if constexpr (SPLIT) {      // the filters were packed times kSplitWScale (a power of two)
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j) acc[i][j] *= kSplitWScaleInv;
    }
    conv_epilogue<YT, WP, WC, TP, TC, (ABL >> 6), Types<T>::kBwF32>(a, acc, smem, w, lane, m0, n0, pt, ct);
}

template <typename T, int WP, int WC, int TP, int TC, int BKB, int NS, int ABL = 0>
static hipError_t launch_cfg(const ConvArgs& a, hipStream_t s) {
    typedef ConvCfg<T, WP, WC, TP, TC, BKB, NS> Cfg;
    static_assert(Cfg::LDS <= 160 * 1024, "LDS budget");
    void (*kern)(ConvArgs) = conv_igemm_kernel<T, WP, WC, TP, TC, BKB, NS, ABL>;
    // f16x2: both operand planes per K chunk (PL2); Y2_NO_CONV_PL2=1: three plane passes
    static const bool no_pl2 = getenv("Y2_NO_CONV_PL2") != nullptr;
    int pl2 = 0;
    if constexpr (Types<T>::kPasses == 3 && ABL == 0 && (BKB == 128 || BKB == 64)) {
        if (!no_pl2) { kern = conv_igemm_kernel<T, WP, WC, TP, TC, BKB, NS, ABL, true>; pl2 = 1; }
    }
    static bool attr_set[2] = {false, false};
    if (!attr_set[pl2]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
        if (e != hipSuccess) return e;
        attr_set[pl2] = true;
    }
    const int nPT = (a.M + Cfg::BP - 1) / Cfg::BP;
    const int nCT = (a.Cout + Cfg::BC - 1) / Cfg::BC;
    hipLaunchKernelGGL(kern, dim3(nPT * nCT), dim3(Cfg::NT), Cfg::LDS, s, a);
    return hipGetLastError();
}

// K split of a small 1x1 launch (conv_igemm_kernel<.., KS>): fewer than 48 workgroups of the 128 x 128 tile (single images,
// the reference's batch 24 at 7x7: 40) and at least four K steps; the depth fills about one round of the chip.  < 2: not this
// form.  (Measured: at 52 .. 208 workgroups -- the ResNet swap's 7x7 units at batch 32 -- the split is neutral to slightly
// negative: its partial tiles and the two extra launches cost what the shorter K loops save; configs[0] gains 2.5 %.)
static int igemm_ks_depth(int M, int Cout, int nK) {
    static const bool off = getenv("Y2_NO_KSPLIT") != nullptr;
    const int wgs = ((M + 127) / 128) * ((Cout + 127) / 128);
    if (off || wgs >= 48 || nK < 4) return 1;
    int d = 256 / wgs;
    d = d > 8 ? 8 : d;
    d = d > nK / 2 ? nK / 2 : d;
    return d < 2 ? 1 : d;
}
int conv_igemm_ks_depth(int M, int Cout, int row_bytes) {
    if (M >= 384 * 8 || (row_bytes % 128) != 0 || Cout <= 64) return 1;
    return igemm_ks_depth(M, Cout, row_bytes / 128);
}
hipError_t launch_conv_ks_finish(int dtype, const ConvArgs& a, int depth, hipStream_t s);      // conv_haloq.hip

template <typename T>
static hipError_t launch_ks(const ConvArgs& a0, hipStream_t s, int dtype) {
    typedef ConvCfg<T, 2, 4, 2, 1, 128, 2> Cfg;
    const int nK = a0.C * (int)sizeof(T) / 128;
    int depth = igemm_ks_depth(a0.M, a0.Cout, nK);
    if (depth < 2 || !a0.ks_scratch || a0.bw_psum || a0.nonfinite || (a0.ldy % 4) != 0) return hipErrorNotSupported;
    while (depth > 1 && (size_t)depth * a0.M * a0.ldy > a0.ks_floats) --depth;
    if (depth < 2) return hipErrorNotSupported;
    ConvArgs a = a0;
    a.ks_splits = depth;
    void (*kern)(ConvArgs) = conv_igemm_kernel<T, 2, 4, 2, 1, 128, 2, 0, false, true>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int tiles = ((a.M + Cfg::BP - 1) / Cfg::BP) * ((a.Cout + Cfg::BC - 1) / Cfg::BC);
    hipLaunchKernelGGL(kern, dim3(tiles * depth), dim3(Cfg::NT), Cfg::LDS, s, a);
    hipError_t e = hipGetLastError();
    return e != hipSuccess ? e : launch_conv_ks_finish(dtype, a, depth, s);
}

// tile choice by output-channel count; rows-per-partial (BP) is reported back
template <typename T>
static hipError_t launch_T(const ConvArgs& a, hipStream_t s) {
    const int kb = a.C * (int)sizeof(typename Types<T>::op_t);  // bytes per tap per pixel (of one operand plane)
    bool k128 = (kb % 128) == 0;
    if (!k128 && (kb % 64) != 0) return hipErrorInvalidValue;
    // round 6: a two-plane (PL2) chunk of 128 bytes is [64 B hi | 64 B lo] = 32 channels of BOTH planes, so the 32-channel
    // layer of the split-operand forward (208x208 32 -> 64) takes ONE K step per tap instead of two of half the depth -- half
    // the barriers and stage waits per matrix instruction.  Y2_IGEMM_PL2_64=1 restores the 64-byte chunks (A/B).
    if constexpr (Types<T>::kPasses == 3) {
        static const bool no_pl2 = getenv("Y2_NO_CONV_PL2") != nullptr, keep64 = getenv("Y2_IGEMM_PL2_64") != nullptr;
        if (!no_pl2 && !keep64 && kb == 64) k128 = true;
    }
    // 2 LDS stages and two blocks per CU beat deeper rings here (global->LDS fill rate, not
    // latency, bounds this kernel); 8 waves of 64x32 beat 4 waves of 64x64 by ~5-8 %
    if (a.Cout > 64) {
        // round 6: launches of at most one workgroup per CU have nobody to hide a stage's latency behind -- a ring of three
        // stages keeps two K steps in flight (LDS is free at one workgroup per CU).  Same box, alternating: the ResNet swap's
        // step 9.90 / 10.00 -> 9.66 / 9.82 ms (its 14x14 / 7x7 units and the 28x28 ones with 128 couts), configs[2] 4.47 -> 4.45;
        // four stages 9.89 / 9.83.  No configs[3] launch has so few workgroups.  Y2_IGEMM_DEEP=2 | 4: A/B.
        if constexpr (!Types<T>::kSplit) {
            static const int deep = getenv("Y2_IGEMM_DEEP") ? atoi(getenv("Y2_IGEMM_DEEP")) : 3;
            const long wgs = (long)((a.M + 127) / 128) * ((a.Cout + 127) / 128);
            if (k128 && deep >= 3 && wgs <= 256)
                return deep >= 4 ? launch_cfg<T, 2, 4, 2, 1, 128, 4>(a, s) : launch_cfg<T, 2, 4, 2, 1, 128, 3>(a, s);
        }
        return k128 ? launch_cfg<T, 2, 4, 2, 1, 128, 2>(a, s) : launch_cfg<T, 2, 4, 2, 1, 64, 2>(a, s);
    } else if (a.Cout > 32) {
        return k128 ? launch_cfg<T, 4, 1, 2, 2, 128, 2>(a, s) : launch_cfg<T, 4, 1, 2, 2, 64, 2>(a, s);
    } else {
        return k128 ? launch_cfg<T, 4, 1, 2, 1, 128, 2>(a, s) : launch_cfg<T, 4, 1, 2, 1, 64, 2>(a, s);
    }
}

int conv_block_pixels(int Cout) { return Cout > 64 ? 128 : 256; }
int conv_block_couts(int Cout) { return Cout > 64 ? 128 : (Cout > 32 ? 64 : 32); }

hipError_t launch_conv_igemm(int dtype, const ConvArgs& a, hipStream_t s) {
    switch (dtype) {
        case 0: case 1: case 2:
            // small 1x1 launches: K split over workgroups (the batch-norm records of that form cover 128 pixels each, the
            // default record size of this kernel's 128-cout tiles: launch_conv's record count does not change)
            if (a.taps == 1 && conv_igemm_ks_depth(a.M, a.Cout, a.C * (int)dtype_size(dtype)) >= 2) {
                const hipError_t e = dtype == 0 ? launch_ks<float>(a, s, dtype)
                                                : (dtype == 1 ? launch_ks<half_t>(a, s, dtype) : launch_ks<bf16_t>(a, s, dtype));
                if (e != hipErrorNotSupported) return e;
            }
            return dtype == 0 ? launch_T<float>(a, s) : (dtype == 1 ? launch_T<half_t>(a, s) : launch_T<bf16_t>(a, s));
        case 3: return launch_T<hsplit_t>(a, s);
        case 4: return launch_T<hsplith_t>(a, s);        // f16x2f backward launches: the hi planes of split tensors
        case 5: return launch_T<hsplithh_t>(a, s);       // ... with dA stored in f16 (common.h hsplithh_t)
    }
    return hipErrorInvalidValue;
}

}  // namespace y2

#ifdef Y2_DEVBUILD
// ---------------------------------------------------------------------------
// development library only: explicit tile / pipeline variants (f16 only) for A/B timing
// ---------------------------------------------------------------------------
namespace y2 {
hipError_t launch_conv_igemm_variant(int variant, const ConvArgs& a, hipStream_t s, int* bp, int* bc) {
    typedef half_t T;
    switch (variant) {
        case 0: *bp = 128; *bc = 128; return launch_cfg<T, 2, 2, 2, 2, 128, 2>(a, s);
        case 1: *bp = 128; *bc = 128; return launch_cfg<T, 2, 2, 2, 2, 128, 3>(a, s);
        case 2: *bp = 256; *bc = 128; return launch_cfg<T, 4, 2, 2, 2, 128, 3>(a, s);
        case 3: *bp = 256; *bc = 128; return launch_cfg<T, 4, 2, 2, 2, 128, 2>(a, s);
        case 4: *bp = 128; *bc = 128; return launch_cfg<T, 2, 4, 2, 1, 128, 2>(a, s);
        case 5: *bp = 128; *bc = 128; return launch_cfg<T, 2, 2, 2, 2, 128, 4>(a, s);
        case 6: *bp = 128; *bc = 256; return launch_cfg<T, 2, 4, 2, 2, 128, 3>(a, s);
        case 7: *bp = 128; *bc = 128; return launch_cfg<T, 2, 4, 2, 1, 128, 3>(a, s);
        case 8: *bp = 256; *bc = 256; return launch_cfg<T, 4, 2, 2, 4, 128, 2>(a, s);
        // the Cout <= 64 product tile (64-byte chunks) and its ablations
        case 17: *bp = 256; *bc = 64; return launch_cfg<T, 4, 1, 2, 2, 64, 2>(a, s);
        case 18: *bp = 256; *bc = 64; return launch_cfg<T, 4, 1, 2, 2, 64, 2, 3>(a, s);    // no loads
        case 19: *bp = 256; *bc = 64; return launch_cfg<T, 4, 1, 2, 2, 64, 2, 16>(a, s);   // no epilogue
        case 20: *bp = 256; *bc = 64; return launch_cfg<T, 4, 1, 2, 2, 64, 2, 64>(a, s);   // no global stores
        case 21: *bp = 256; *bc = 64; return launch_cfg<T, 4, 1, 2, 2, 64, 2, 128>(a, s);  // wave-level sync in the epilogue
        case 22: *bp = 256; *bc = 64; return launch_cfg<T, 4, 1, 2, 2, 64, 2, 67>(a, s);   // no loads, no stores
        // smaller / more numerous tiles for the small-K, large-M layers
        case 200: *bp = 128; *bc = 64; return launch_cfg<T, 2, 2, 2, 1, 64, 2>(a, s);
        case 201: *bp = 256; *bc = 64; return launch_cfg<T, 4, 2, 2, 1, 64, 2>(a, s);
        case 202: *bp = 128; *bc = 64; return launch_cfg<T, 2, 2, 2, 1, 128, 2>(a, s);
        case 203: *bp = 256; *bc = 64; return launch_cfg<T, 4, 2, 2, 1, 128, 2>(a, s);
        case 204: *bp = 128; *bc = 64; return launch_cfg<T, 4, 1, 1, 2, 64, 2>(a, s);
        case 205: *bp = 128; *bc = 64; return launch_cfg<T, 4, 1, 1, 2, 128, 2>(a, s);
        case 206: *bp = 256; *bc = 64; return launch_cfg<T, 4, 1, 2, 2, 128, 2>(a, s);   // product, Cout <= 64
        case 207: *bp = 128; *bc = 128; return launch_cfg<T, 2, 4, 2, 1, 64, 2>(a, s);  // product, Cout > 64
        case 208: *bp = 128; *bc = 128; return launch_cfg<T, 2, 4, 2, 1, 128, 2>(a, s);
        case 209: *bp = 256; *bc = 32; return launch_cfg<T, 4, 1, 2, 1, 128, 2>(a, s);   // product, Cout <= 32
        case 210: *bp = 512; *bc = 32; return launch_cfg<T, 8, 1, 2, 1, 128, 2>(a, s);
        // ablations of variant 0
        case 10: *bp = 128; *bc = 128; return launch_cfg<T, 2, 2, 2, 2, 128, 2, 1>(a, s);
        case 11: *bp = 128; *bc = 128; return launch_cfg<T, 2, 2, 2, 2, 128, 2, 2>(a, s);
        case 12: *bp = 128; *bc = 128; return launch_cfg<T, 2, 2, 2, 2, 128, 2, 3>(a, s);
        case 13: *bp = 128; *bc = 128; return launch_cfg<T, 2, 2, 2, 2, 128, 2, 4>(a, s);
        case 14: *bp = 128; *bc = 128; return launch_cfg<T, 2, 2, 2, 2, 128, 2, 8>(a, s);
        case 15: *bp = 128; *bc = 128; return launch_cfg<T, 2, 2, 2, 2, 128, 2, 11>(a, s);
        case 16: *bp = 128; *bc = 128; return launch_cfg<T, 2, 2, 2, 2, 128, 2, 12>(a, s);
    }
    return hipErrorInvalidValue;
}
}  // namespace y2
#endif  // Y2_DEVBUILD
