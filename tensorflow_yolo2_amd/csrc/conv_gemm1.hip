// 1x1 stride-1 convolution (a plain GEMM  D[pixel][cout] = X[pixel][k] W[k][cout]) with the FILTER FRAGMENTS STRAIGHT FROM L2 and
// the pixel tile alone in a DEEP LDS ring (gfx950).  Fourth form of the op of conv_igemm.hip for filter_size 1
// (tf.nn.conv2d(x, W, [1,1,1,1], 'SAME') + bias, reference src/yolo2_nets/darknet.py:20-21 at the 1x1 layers :162,167,169,174,176;
// slim's bottleneck 1x1s src/slim_dir/nets/resnet_v1.py:99-112) and of its dgrad.
//
// Why another form (round 6).  conv_igemm stages BOTH operands of a K step through LDS in a two-stage
// pipeline: every 64-deep K step of a 128 x 128 tile waits for its own 32 KB to arrive (one memory latency per step, two
// workgroups per CU to hide it) and every wave reads 12 KB of fragments for 8 MFMAs -- LDS-read bound on top.  The 3x3 kernels
// of conv_haloq.hip do not have either problem: their filter fragments come pre-packed in MFMA-fragment order straight from
// L2 into registers (pack.hip layout 1), and one staged pixel chunk feeds nine tap steps.  A 1x1 layer has ONE step per chunk,
// so conv_haloq's double buffer ("Y2_HALOQ_1X1", round 3) measured a wash: every step still waited for its own chunk.
// Here the ring is D chunks deep for BOTH streams: at the top of step s the pixel chunks AND the filter-fragment register sets
// of steps s+1 .. s+D-1 are in flight (vmcnt retires in order, so the fragment prefetch has to be as deep as the ring: a wait
// for this step's fragments drains every older pixel piece), each step costs one workgroup barrier, and a pixel fragment read
// from LDS feeds TC MFMAs.
//   x : zero-bordered NHWC [N][H+1][W+1][C] (common.h bpix): the rows of a tile are gathered by cell index (no halo)
//   w : pack.hip layout 1  [cout tile of 32][k-group of 32 B][lane][16 B]
//   y : [M][ldy] through the shared epilogue (conv_epilogue.h: bias, rounding, batch-norm partials, the fused BN-backward
//       reduce of dgrad launches, the folded inference batch norm)
#include <stdlib.h>
#include "common.h"
#include "conv_epilogue.h"
#include "kernels.h"

namespace y2 {

namespace {

template <int V> struct GInt { static constexpr int value = V; };

// filter-fragment load the compiler does not see (conv_haloq.hip frag_load): counted by hand with the LDS-DMA pieces
Y2_DEV void g1_frag_load(u32x4& dst, const char* lane_ptr) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(lane_ptr) : "memory");
}
Y2_DEV void g1_frag_ready(u32x4& v) { asm volatile("" : "+v"(v)); }
Y2_DEV u32x4 g1_lds_read16(uint32_t lds_addr) {
    return *(const __attribute__((address_space(3))) u32x4*)(uintptr_t)lds_addr;
}

template <typename T, int WP, int WC, int TP, int TC, int D>
struct G1Cfg {
    static constexpr int NW = WP * WC, NT = NW * 64;
    static constexpr int BP = WP * TP * 32, BC = WC * TC * 32;
    static constexpr int BKB = 128, KG = BKB / 32, LPR = BKB / 16, RPI = 64 / LPR, RPB = 256 / BKB;
    static constexpr int NSTG = D + 1;
    static constexpr int STAGE = BP * BKB;
    static constexpr int PIECES = BP / RPI;                 // 1-KiB LDS-DMA pieces per stage
    static constexpr int KA = PIECES / NW;                  // ... per wave
    static constexpr int NBL = TC * KG;                     // fragment loads per wave and step
    static constexpr int RING = NSTG * STAGE;
    static_assert(PIECES % NW == 0, "the pieces of a stage split evenly over the waves");
    static_assert(D >= 1 && D <= 3, "ring depth");
};

// One K step = one 128-byte chunk of every pixel row (64 elements of a 16-bit type): KG = 4 k-groups of TC x TP MFMAs.
template <typename T, int WP, int WC, int TP, int TC, int D>
__global__ __launch_bounds__(WP* WC * 64) void conv_gemm1_kernel(ConvArgs a) {
    typedef G1Cfg<T, WP, WC, TP, TC, D> Cfg;
    typedef typename Elem<T>::frag frag_t;
    typedef typename Types<T>::op_t OT;
    typedef typename Types<T>::out_t YT;
    constexpr bool SPLIT = Types<T>::kSplit;
    constexpr int NW = Cfg::NW, BP = Cfg::BP, BC = Cfg::BC, BKB = Cfg::BKB, KG = Cfg::KG, LPR = Cfg::LPR, RPI = Cfg::RPI,
                  RPB = Cfg::RPB, NSTG = Cfg::NSTG, KA = Cfg::KA, NBL = Cfg::NBL, SZ = sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = w / WC, wc = w % WC;
    const int nCT = (a.Cout + BC - 1) / BC;
    const int bx = xcd_block(blockIdx.x, gridDim.x, a.xcd);
    const int ct = bx % nCT, pt = bx / nCT;
    const int m0 = pt * BP, n0 = ct * BC;
    const int hw = a.H * a.W;
    const int rowbytes = a.C * SZ;                       // bytes of one cell (f16x2 modes: both planes)
    const int npl = a.C * (int)sizeof(OT) / BKB;         // chunks per operand plane
    const int nch = Types<T>::kPasses * npl;             // K steps of this launch
    const char* __restrict__ xg = (const char*)a.x;

    // ---- LDS-DMA sources: the cell of every row this lane stages (KA pieces per wave and stage, 8 rows each)
    const int lrow = lane / LPR, lslot = lane % LPR;
    uint32_t voff[KA];
#pragma unroll
    for (int k = 0; k < KA; ++k) {
        const int row = (k * NW + w) * RPI + lrow;
        int p = m0 + row;
        if (p > a.M - 1) p = a.M - 1;                    // tail rows re-read the last pixel: their outputs are masked
        const int n = p / hw, rem = p - n * hw;
        const int h = rem / a.W, ww = rem - h * a.W;
        voff[k] = (uint32_t)bpix(n, h, ww, a.H, a.W) * (uint32_t)rowbytes + (uint32_t)((lslot ^ ((row / RPB) % LPR)) * 16);
    }
    auto issueA = [&](int c) {
        const char* xs = xg + (size_t)split_act_chunk<SPLIT>(c, npl) * BKB;
        char* dst = smem + (c % NSTG) * Cfg::STAGE;
#pragma unroll
        for (int k = 0; k < KA; ++k) glds16(xs + voff[k], dst + (k * NW + w) * 1024);
    };
    // ---- filter fragments: [cout tile of 32][k-group of 32 bytes][lane][16 B]
    const int kgrow = rowbytes / 32;
    const char* wbase[TC];
#pragma unroll
    for (int i = 0; i < TC; ++i) wbase[i] = (const char*)a.w + ((size_t)(n0 / 32 + wc * TC + i) * kgrow * 64 + lane) * 16;
    auto loadB = [&](int c, u32x4 (&fb)[TC][KG]) {
        const size_t off = (size_t)(split_flt_chunk<SPLIT>(c, npl) * KG) * 1024;
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int g = 0; g < KG; ++g) g1_frag_load(fb[i][g], wbase[i] + off + (size_t)g * 1024);
    };

    // ---- fragment read addresses (row of the stage, swizzle key as staged)
    const int smem_lds = (int)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int r32 = lane & 31, hh = lane >> 5;
    int arow[TP];
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const int row = (wp * TP + j) * 32 + r32;
        arow[j] = smem_lds + row * BKB + ((hh ^ ((row / RPB) % LPR)) << 4);      // k-group g: ^ (g * 32)
    }

    f32x16 acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    u32x4 fbq[NSTG][TC][KG];
    // groups 0 .. D-1 in flight before the first step: fragments first, then the pixel pieces, per group
    auto issue_group = [&](auto set_tag, int c) {
        constexpr int S = decltype(set_tag)::value;
        loadB(c, fbq[S]);
        issueA(c);
    };
    if (0 < nch) issue_group(GInt<0>{}, 0);
    if constexpr (D >= 2) { if (1 < nch) issue_group(GInt<1 % NSTG>{}, 1); }
    if constexpr (D >= 3) { if (2 < nch) issue_group(GInt<2 % NSTG>{}, 2); }

    auto step = [&](auto set_tag, int s) {
        constexpr int S = decltype(set_tag)::value;            // s % NSTG: register set and ring stage of this step
        // group s has landed when only the younger groups (s+1 .. s+D-1, as far as they exist) are outstanding
        int young = nch - 1 - s;
        young = young > D - 1 ? D - 1 : young;
        wait_vmcnt_dyn(young * (NBL + KA));
        __builtin_amdgcn_s_barrier();      // everyone's pieces of stage s are in LDS, everyone has left stage s-1
        asm volatile("" ::: "memory");
        if (s + D < nch) issue_group(GInt<(S + D) % NSTG>{}, s + D);      // into the set / stage step s-1 used
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int g = 0; g < KG; ++g) g1_frag_ready(fbq[S][i][g]);
        const int stageB = S * Cfg::STAGE;
        frag_t fp[2][TP];
#pragma unroll
        for (int j = 0; j < TP; ++j) fp[0][j] = __builtin_bit_cast(frag_t, g1_lds_read16((uint32_t)(arow[j] + stageB)));
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            if (g + 1 < KG) {
#pragma unroll
                for (int j = 0; j < TP; ++j)
                    fp[(g + 1) & 1][j] = __builtin_bit_cast(frag_t, g1_lds_read16((uint32_t)((arow[j] + stageB) ^ ((g + 1) * 32))));
            }
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) mma32(acc[i][j], __builtin_bit_cast(frag_t, fbq[S][i][g]), fp[g & 1][j]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int s = 0; s < nch; s += NSTG) {
        step(GInt<0>{}, s);
        if constexpr (NSTG > 1) { if (s + 1 < nch) step(GInt<1 % NSTG>{}, s + 1); }
        if constexpr (NSTG > 2) { if (s + 2 < nch) step(GInt<2 % NSTG>{}, s + 2); }
        if constexpr (NSTG > 3) { if (s + 3 < nch) step(GInt<3 % NSTG>{}, s + 3); }
    }
    wait_vmcnt<0>();
    __syncthreads();
    if constexpr (SPLIT) {      // the filters were packed times kSplitWScale (a power of two)
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j) acc[i][j] *= kSplitWScaleInv;
    }
    conv_epilogue<YT, WP, WC, TP, TC, 0, Types<T>::kBwF32>(a, acc, smem, w, lane, m0, n0, pt, ct);
}

template <typename T, int WP, int WC, int TP, int TC, int D>
hipError_t g1_launch(const ConvArgs& a, hipStream_t s) {
    typedef G1Cfg<T, WP, WC, TP, TC, D> Cfg;
    typedef EpiCfg<typename Types<T>::out_t, WP, WC, TP, TC> Epi;
    constexpr int LDS = Cfg::RING > Epi::LDS ? Cfg::RING : Epi::LDS;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    void (*kern)(ConvArgs) = conv_gemm1_kernel<T, WP, WC, TP, TC, D>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int nPT = (a.M + Cfg::BP - 1) / Cfg::BP, nCT = (a.Cout + Cfg::BC - 1) / Cfg::BC;
    hipLaunchKernelGGL(kern, dim3(nPT * nCT), dim3(Cfg::NT), LDS, s, a);
    return hipGetLastError();
}

// tile: 256 x 128 (8 waves of 64 px x 64 co, ring 3 x 32 KB) where that leaves at least ~1.5 workgroups per CU, else 128 x 128
// (4 waves of 64 x 64, ring 3 x 16 KB: two to three workgroups share a CU)
template <typename T>
hipError_t g1_T(const ConvArgs& a, hipStream_t s, int* bp) {
    static const int force = getenv("Y2_GEMM1_TILE") ? atoi(getenv("Y2_GEMM1_TILE")) : 0;      // 256 / 128: A/B
    static const int depth = getenv("Y2_GEMM1_DEPTH") ? atoi(getenv("Y2_GEMM1_DEPTH")) : 2;
    const long big = (long)((a.M + 255) / 256) * ((a.Cout + 127) / 128);
    const bool t256 = force ? force == 256 : big >= 384;
    if (t256) {
        *bp = 256;
        if (depth >= 3) return g1_launch<T, 4, 2, 2, 2, 3>(a, s);
        return g1_launch<T, 4, 2, 2, 2, 2>(a, s);
    }
    *bp = 128;
    if (depth >= 3) return g1_launch<T, 2, 2, 2, 2, 3>(a, s);
    return g1_launch<T, 2, 2, 2, 2, 2>(a, s);
}

}  // namespace

// MEASURED (round 6, one MI355X, configs[3], per-layer HIP events inside the train step; profiles/r06_ab_gemm1_1x1.txt):
// parity-green on every 1x1 test -- and SLOWER than conv_igemm in every configuration tried (ring depth 2 / 3, 128 / 256
// pixel tiles): forward 52x52 256->128 48.9 vs 40.7 us, 26x26 512->256 35.5 vs 32.2, 13x13 1024->512 31.0 vs 28.0; dgrad
// 79.6 vs 64.4, 51.1 vs 44.1, 34.1 vs 33.6 (the 128-pixel tile ties the dgrads and loses 2-5 us on the forwards).  So the
// per-step wait of the two-stage loop is NOT what bounds this class: with two to three groups in flight and the fragment
// reads halved the layers take the same 30 us.  What the two kernels share is the traffic of the decomposition -- 340
// tiles of 128 x 128 re-read the pixel panel four times and the filter panel 85 times: 178 MB through L2 for 11 GFLOP --
// and the one-to-1.3-round grids of the 13x13 / 26x26 layers.  Kept as an opt-in (Y2_GEMM1=1), NOT the default.
//
// which launches take this form: 1x1, 16-bit operand planes in whole 128-byte chunks, more than 64 output channels (the
// 30-channel head output and the few-channel shapes stay on conv_igemm).
bool conv_gemm1_ok(int taps, int row_bytes, int Cout, int M) {
    static const bool on = getenv("Y2_GEMM1") && atoi(getenv("Y2_GEMM1")) != 0;
    return on && taps == 1 && (row_bytes % 128) == 0 && Cout > 64 && M >= 128;
}

hipError_t launch_conv_gemm1(int dtype, const ConvArgs& a, hipStream_t s, int* bp) {
    switch (dtype) {
        case 1: return g1_T<half_t>(a, s, bp);
        case 2: return g1_T<bf16_t>(a, s, bp);
    }
    return hipErrorInvalidValue;
}

}  // namespace y2
