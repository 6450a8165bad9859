// Weight-gradient GEMM (TF Conv2DBackpropFilter of darknet.py:20-21):
//     dW[t][ci][co] = sum_p X[p (+) t][ci] * dY[p][co]
// Both operands are pixel-major in HBM (NHWC) and the reduction runs over
// pixels, so the MFMA k index is the slow memory index of BOTH operands.
// gfx950 answer: stage [pixels][channels] tiles with global_load_lds and read
// the fragments with ds_read_b64_tr_b16 (hardware transpose) -- no transposed
// copies of the activations are ever written to HBM.
//
// K runs LINEARLY over the zero-bordered pixel space [0, N*(H+2)*(W+2)):
// dY's border is zero, so border positions contribute nothing and no
// per-pixel index arithmetic is needed (row addresses are affine in k).  The
// tap shift is a constant offset on X; guard bands around the tensors keep the
// shifted reads in finite memory.
// Split-K over blocks; partial tiles are accumulated into the fp32 HWIO
// gradient with float atomics (256-byte row segments per wave-instruction).
#include <stdlib.h>
#include "common.h"
#include "conv_epilogue.h"
#include "kernels.h"
#include "wgrad_finish.h"
// The in-kernel split-K sum (wgrad_finish.h) measured slower in every form (profiles/r05_ab_wgrad_finish.txt) and its mere
// presence cost the weight gradients 1.8 % (2.53 vs 2.49 ms per step, same box): it is compiled into the DEVELOPMENT
// library only (make dev, Y2_WGRAD_FINISH=<max partials>); the product kernels store their partials plainly.
#ifdef Y2_DEVBUILD
#define Y2_FIN_STORE(p, v) do { if (a.cnt_stride) slab_store((p), (v)); else *(p) = (v); } while (0)
#else
#define Y2_FIN_STORE(p, v) (*(p) = (v))
#endif

namespace y2 {

// NS = LDS stages: NS - 1 K steps of LDS-DMA stay in flight across the raw barrier (counted vmcnt).  The 1x1
// layers run one workgroup per CU (split-K atomics cost as much as the reads): there a step's 16 MFMAs per wave
// cannot cover the latency of the next step's fetch, and two stages leave the kernel latency-bound.
template <typename T, int WI, int WO, int TI, int TO, int NS = 2>
struct WgCfg {
    static constexpr int NW = WI * WO, NT = NW * 64;
    static constexpr int SZ = sizeof(T);
    static constexpr int BI = WI * TI * 32, BO = WO * TO * 32;
    static constexpr int BKP = (SZ == 2) ? 64 : 32;  // pixels per K step
    static constexpr int ROWX = BI * SZ, ROWY = BO * SZ;
    static constexpr int LPRX = ROWX / 16, LPRY = ROWY / 16;
    static constexpr int RPIX = 64 / LPRX, RPIY = 64 / LPRY;
    static constexpr int NIX = BKP / RPIX, NIY = BKP / RPIY;
    static constexpr int IPWX = NIX / NW, IPWY = NIY / NW;
    static constexpr int XS = BKP * ROWX, YS = BKP * ROWY;
    static constexpr int STAGE = XS + YS;
    static constexpr int LDS = NS * STAGE;
    static_assert(NIX % NW == 0 && NIY % NW == 0, "staging must split evenly over waves");
};

// swizzle: rows q = row&3 of a transposed 4x16 read must land in distinct
// 64-byte bank segments of the 256-byte LDS bank row
template <int ROWB, int SZ>
Y2_DEV int wg_swz(int row) {
    if (SZ != 2) return 0;
    if (ROWB >= 256) return (row & 3) << 2;
    if (ROWB == 128) return ((row & 3) >> 1) << 2;
    return 0;
}

// PL2 (f16x2 mode): both planes of x and dy staged per K step, the three plane products in one block (wgrad9.hip wg9_body)
template <typename T, int WI, int WO, int TI, int TO, int NS, bool PL2 = false>
__global__ __launch_bounds__(WI* WO * 64) void wgrad_kernel(WgradArgs a) {
    typedef WgCfg<T, WI, WO, TI, TO, NS> Cfg;
    constexpr int NW = Cfg::NW, SZ = Cfg::SZ, BI = Cfg::BI, BO = Cfg::BO, BKP = Cfg::BKP;
    constexpr int ROWX = Cfg::ROWX, ROWY = Cfg::ROWY;
    constexpr int NPL = PL2 ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* const s_fin = (int*)(smem + NPL * Cfg::LDS);      // the finish flag sits behind the staging buffers (wgrad_finish.h)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = w / WO, wo = w % WO;

    const int nIT = (a.Cin + BI - 1) / BI, nOT = (a.Cout + BO - 1) / BO;
    int b = xcd_block(blockIdx.x, gridDim.x, a.xcd);
    const int ot = b % nOT; b /= nOT;
    const int it = b % nIT; b /= nIT;
    const int tap = b % a.taps;
    const int split = b / a.taps;
    const int ci0 = it * BI, co0 = ot * BO;

    const long Mp = (long)bbody_pixels(a.N, a.H, a.W);
    const long ksteps = (Mp + BKP - 1) / BKP;
    const long spb = (ksteps + a.splitk - 1) / a.splitk;
    const long s_begin = (long)split * spb;
    long s_end = s_begin + spb;
    if (s_end > ksteps) s_end = ksteps;
    const int nsteps = (int)(s_end > s_begin ? s_end - s_begin : 0);

    int toff;  // tap shift in pixels relative to the centre
    if (a.taps == 9) {
        const int kh = tap / 3, kw = tap - kh * 3;
        toff = (kh - 1) * (a.W + 1) + (kw - 1);
    } else {
        toff = 0;
    }
    const long kb = s_begin * BKP;
    const char* xg = (const char*)a.x + ((kb + toff) * a.xpitch + ci0) * SZ;
    const char* yg = (const char*)a.dy + (kb * a.ypitch + co0) * SZ;
    const long xstep = (long)BKP * a.xpitch * SZ, ystep = (long)BKP * a.ypitch * SZ;

    uint32_t voffx[Cfg::IPWX], voffy[Cfg::IPWY];
#pragma unroll
    for (int i = 0; i < Cfg::IPWX; ++i) {
        const int row = (i * NW + w) * Cfg::RPIX + lane / Cfg::LPRX;
        const int sl = (lane % Cfg::LPRX) ^ wg_swz<ROWX, SZ>(row);
        voffx[i] = (uint32_t)row * (uint32_t)(a.xpitch * SZ) + sl * 16;
    }
#pragma unroll
    for (int i = 0; i < Cfg::IPWY; ++i) {
        const int row = (i * NW + w) * Cfg::RPIY + lane / Cfg::LPRY;
        const int sl = (lane % Cfg::LPRY) ^ wg_swz<ROWY, SZ>(row);
        voffy[i] = (uint32_t)row * (uint32_t)(a.ypitch * SZ) + sl * 16;
    }
    const int xplaneB = a.Cin * SZ, yplaneB = a.Cdy * SZ;      // PL2: byte distance of the lo plane inside a cell
    auto stage = [&](int st, int buf) {
        const char* xs = xg + (long)st * xstep;
        const char* ys = yg + (long)st * ystep;
        char* lb = smem + buf * NPL * Cfg::STAGE;         // [X hi][X lo][dY hi][dY lo]
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
            for (int i = 0; i < Cfg::IPWX; ++i) glds16(xs + pl * xplaneB + voffx[i], lb + pl * Cfg::XS + (i * NW + w) * 1024);
#pragma unroll
            for (int i = 0; i < Cfg::IPWY; ++i)
                glds16(ys + pl * yplaneB + voffy[i], lb + NPL * Cfg::XS + pl * Cfg::YS + (i * NW + w) * 1024);
        }
    };

    f32x16 acc[TI][TO];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TO; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    const int r32 = lane & 31, hh = lane >> 5;
    // transposed-read lane constants (f16/bf16)
    const int qq = (lane & 15) >> 2, pp = lane & 3, g1 = (lane >> 4) & 1;

    constexpr int lps = NPL * (Cfg::IPWX + Cfg::IPWY);     // LDS-DMA pieces per wave and stage
#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0)
        if (s0 < nsteps) stage(s0, s0);
    int cbuf = 0, ibuf = NS - 1;
    for (int st = 0; st < nsteps; ++st) {
        if (st + NS - 2 < nsteps) wait_vmcnt_dyn((NS - 2) * lps);
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();      // step st has landed for every wave; everyone is done with step st - 1's buffer
        asm volatile("" ::: "memory");
        if (st + NS - 1 < nsteps) stage(st + NS - 1, ibuf);
        const int buf = cbuf;
        cbuf = (cbuf + 1 == NS) ? 0 : cbuf + 1;
        ibuf = (ibuf + 1 == NS) ? 0 : ibuf + 1;
        const char* xs = smem + buf * NPL * Cfg::STAGE;
        const char* ys = xs + NPL * Cfg::XS;
        if constexpr (SZ == 2) {
            const int fx = wg_swz<ROWX, SZ>(qq), fy = wg_swz<ROWY, SZ>(qq);
#pragma unroll
            for (int kg = 0; kg < BKP / 16; ++kg) {
                typename Elem<T>::frag fa[NPL][TI], fb[NPL][TO];
                const int row0 = kg * 16 + 8 * hh + qq;
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
                    for (int i = 0; i < TI; ++i) {
                        const int slot = (((wi * TI + i) * 4 + 2 * g1 + (pp >> 1)) ^ fx);
                        const char* p = xs + pl * Cfg::XS + row0 * ROWX + slot * 16 + (pp & 1) * 8;
                        fa[pl][i] = tr_frag<T>(p, p + 4 * ROWX);
                    }
#pragma unroll
                    for (int j = 0; j < TO; ++j) {
                        const int slot = (((wo * TO + j) * 4 + 2 * g1 + (pp >> 1)) ^ fy);
                        const char* p = ys + pl * Cfg::YS + row0 * ROWY + slot * 16 + (pp & 1) * 8;
                        fb[pl][j] = tr_frag<T>(p, p + 4 * ROWY);
                    }
                }
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TO; ++j) {
                        mma32(acc[i][j], fa[0][i], fb[0][j]);
                        if constexpr (PL2) {
                            mma32(acc[i][j], fa[1][i], fb[0][j]);      // x lo * dy hi
                            mma32(acc[i][j], fa[0][i], fb[1][j]);      // x hi * dy lo
                        }
                    }
            }
        } else {
#pragma unroll
            for (int s2 = 0; s2 < BKP / 2; ++s2) {
                const int row = 2 * s2 + hh;
                float fa[TI], fb[TO];
#pragma unroll
                for (int i = 0; i < TI; ++i) fa[i] = *(const float*)(xs + row * ROWX + ((wi * TI + i) * 32 + r32) * 4);
#pragma unroll
                for (int j = 0; j < TO; ++j) fb[j] = *(const float*)(ys + row * ROWY + ((wo * TO + j) * 32 + r32) * 4);
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TO; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    // ---- accumulate into fp32 HWIO gradient
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TO; ++j) {
            const int co = co0 + (wo * TO + j) * 32 + r32;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int ci = ci0 + (wi * TI + i) * 32 + acc_row(q, hh);
                if (ci < a.Cin && co < a.Cout) {
                    const size_t o = ((size_t)tap * a.Cin + ci) * a.Cout + co;
                    if (a.splitk == 1 && a.quads == 1) a.dW[o] = acc[i][j][q] * a.scale;
                    else if (a.slab) { float* sp = a.slab + (size_t)(a.part0 + split) * a.taps * a.Cin * a.Cout + o; Y2_FIN_STORE(sp, acc[i][j][q]); }
                    else atomicAdd(a.dW + o, acc[i][j][q] * a.scale);
                }
            }
        }
#ifdef Y2_DEVBUILD
    if (a.slab && a.cnt_stride) {     // the split-K sum rides in this kernel (wgrad_finish.h; development library only)
        const size_t nn = (size_t)a.taps * a.Cin * a.Cout;
        splitk_finish(s_fin, a.tile_cnt + (size_t)((tap * nIT + it) * nOT + ot) * a.cnt_stride, a.part0 + split, a.splitk * a.quads,
                      [&](int first, int stride, int count, bool final) __attribute__((always_inline)) {
#pragma unroll 1
            for (int i = 0; i < TI; ++i)
#pragma unroll 1
                for (int j = 0; j < TO; ++j) {
                    const int co = co0 + (wo * TO + j) * 32 + r32;
#pragma unroll 1
                    for (int q = 0; q < 16; ++q) {
                        const int ci = ci0 + (wi * TI + i) * 32 + acc_row(q, hh);
                        if (ci < a.Cin && co < a.Cout) {
                            const size_t o = ((size_t)tap * a.Cin + ci) * a.Cout + co;
                            const float v = splitk_sum_slots(a.slab + (size_t)first * nn + o, (size_t)stride * nn, count);
                            if (final) a.dW[o] = v * a.scale;
                            else slab_store(a.slab + (size_t)first * nn + o, v);
                        }
                    }
                }
        });
    }
#endif
}

template <typename T, int WI, int WO, int TI, int TO, int NS = 2, bool PL2 = false>
static hipError_t wg_launch(WgradArgs a, hipStream_t s, int target1 = 256) {
    typedef WgCfg<T, WI, WO, TI, TO, NS> Cfg;
    static_assert(Cfg::LDS <= 160 * 1024, "LDS");
    if constexpr (sizeof(T) == 2 && !PL2 && 2 * Cfg::LDS <= 160 * 1024) {
        // f16x2 mode: both planes staged, the three plane products in one block (one partial per split instead of three)
        static const bool quads_form = getenv("Y2_SPLIT_WGRAD_QUADS") != nullptr;
        if (a.quads == 3 && !quads_form) {
            a.quads = 1;
            return wg_launch<T, WI, WO, TI, TO, NS, true>(a, s, 256);      // twice the LDS: one block per CU
        }
    }
    constexpr int LDSB = (PL2 ? 2 : 1) * Cfg::LDS;
    // (the finish flag of the opt-in in-kernel sum sits behind the staging buffers: 16 more bytes only when it is on --
    //  a request of exactly 1/2 or 1/3 of the CU's LDS must stay that)
    const int fin16 = (wgrad_finish_max_parts() > 0 && LDSB + 16 <= 160 * 1024) ? 16 : 0;
    if (!fin16) a.tile_cnt = nullptr;      // no room (or no wish) for the finish flag: the separate sum kernel
    auto kern = wgrad_kernel<T, WI, WO, TI, TO, NS, PL2>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           LDSB + 16 <= 160 * 1024 ? LDSB + 16 : LDSB);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int nIT = (a.Cin + Cfg::BI - 1) / Cfg::BI, nOT = (a.Cout + Cfg::BO - 1) / Cfg::BO;
    const long Mp = (long)bbody_pixels(a.N, a.H, a.W);
    const long ksteps = (Mp + Cfg::BKP - 1) / Cfg::BKP;
    const int tiles = a.taps * nIT * nOT;
    if (a.splitk <= 0) {
        // 1x1: with float atomics the partial tiles cost as much as the streaming reads and one block per CU was
        // the optimum; through the slab two per CU are (scripts/bench_wgrad.py: 38 -> 33.5 us at 26x26 / 13x13);
        // 3x3 fallback: ~6 per CU
        const int target = a.taps == 1 ? (a.slab && target1 == 256 && !PL2 ? 512 : target1) : 1536;
        long sk = (target + tiles - 1) / tiles;
        const long maxsk = (ksteps + 7) / 8;         // at least 8 K steps per block
        if (sk > maxsk) sk = maxsk;
        if (sk < 1) sk = 1;
        a.splitk = (int)sk;
    }
    hipError_t e = wgrad_split_prepare(a, s);
    if (e != hipSuccess) return e;
    e = wgrad_launch_quads(kern, dim3(tiles * a.splitk), dim3(Cfg::NT), LDSB + fin16, s, a);
    return e != hipSuccess ? e : wgrad_split_finish(a, s);
}

template <typename T>
static hipError_t wg_T(const WgradArgs& a, hipStream_t s) {
    const int bi = a.Cin >= 128 ? 128 : a.Cin;       // Cin is 32, 64 or a multiple of 128
    const int bo = a.Cdy >= 128 ? 128 : (a.Cdy >= 64 ? 64 : 32);
    if (bi == 128 && bo == 128) return wg_launch<T, 2, 2, 2, 2>(a, s);
    if (bi == 128 && bo == 64) return wg_launch<T, 2, 2, 2, 1>(a, s);
    if (bi == 128 && bo == 32) return wg_launch<T, 4, 1, 1, 1>(a, s);
    if (bi == 64 && bo == 128) return wg_launch<T, 2, 2, 1, 2>(a, s);
    if (bi == 64 && bo == 64) return wg_launch<T, 2, 2, 1, 1>(a, s);
    if (bi == 64 && bo == 32) return wg_launch<T, 2, 1, 1, 1>(a, s);
    if (bi == 32 && bo == 128) return wg_launch<T, 1, 4, 1, 1>(a, s);
    if (bi == 32 && bo == 64) return wg_launch<T, 1, 2, 1, 1>(a, s);
    if (bi == 32 && bo == 32) return wg_launch<T, 1, 1, 1, 1>(a, s);
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------
// Split-K partials: slab[split][taps*Cin*Cout] -> dW = scale * sum over the splits, in split order (deterministic).
// SL = 1: a thread owns four consecutive elements and walks the splits (few splits, large dW);
// SL = 16: 16 lanes share a column group and take every 16th split (hundreds of splits of a tiny dW), LDS add.
// ---------------------------------------------------------------------------
template <int SL>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dW, size_t n4,
                                                           int sk, size_t stride4, float scale) {
    constexpr int COLS = 256 / SL;
    __shared__ f32x4 red[SL > 1 ? 256 : 1];
    const int col = threadIdx.x % COLS, sl = threadIdx.x / COLS;
    const size_t i = (size_t)blockIdx.x * COLS + col;
    const f32x4* p = (const f32x4*)slab;
    f32x4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (i < n4) {
        int s = sl;
        // eight splits' loads up front (the kernel runs beside the dgrads: memory latency under that load is what it
        // waits for); every accumulator adds in the order of the four-split loop it replaces: the same bits
        for (; s + 7 * SL < sk; s += 8 * SL) {
            f32x4 r[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) r[u] = p[(size_t)(s + u * SL) * stride4 + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u & 3] += r[u];
        }
        for (; s + 3 * SL < sk; s += 4 * SL) {
            f32x4 r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) r[u] = p[(size_t)(s + u * SL) * stride4 + i];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] += r[u];
        }
        for (; s < sk; s += SL) acc[0] += p[(size_t)s * stride4 + i];
    }
    f32x4 t = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    if (SL > 1) {
        red[threadIdx.x] = t;
        __syncthreads();
        if (sl == 0) {
#pragma unroll
            for (int k = 1; k < SL; ++k) t += red[k * COLS + col];
        }
    }
    if (sl == 0 && i < n4) ((f32x4*)dW)[i] = t * scale;
}

int wgrad_finish_max_parts() {
#ifdef Y2_DEVBUILD
    static const int v = getenv("Y2_WGRAD_FINISH") ? atoi(getenv("Y2_WGRAD_FINISH")) : 0;
    return v;
#else
    return 0;      // the in-kernel sum exists in the development library only (see the top of this file)
#endif
}
hipError_t wgrad_split_prepare(WgradArgs& a, hipStream_t s) {
    a.cnt_stride = 0;
    if (a.splitk * a.quads <= 1) return hipSuccess;
    const size_t n = (size_t)a.taps * a.Cin * a.Cout;
    static const bool no_slab = getenv("Y2_NO_WGRAD_SLAB") != nullptr;      // A/B switch: float atomics instead
    // In-kernel sum (wgrad_finish.h): built and measured in round 5, NOT the default.  Y2_WGRAD_FINISH=<max parts> turns it on
    // for launches of up to that many partials per tile.  Same box, configs[3] step: separate sum launches 8.78 ms; in-kernel
    // with agent-scope fences 10.9 ms (every fence writes back / invalidates a whole L2 under the dgrad kernels); with
    // sc1 atomics instead of fences 9.71 ms (the ONE block that completes a group of 16 partials of a 64-KB tile reads
    // 1 MB through 4-byte sc1 loads, a few in flight per lane: a serial tail per tile where the separate kernel spreads
    // the same reads over the whole chip in 6 us)
    const bool sum_kernel = a.splitk * a.quads > wgrad_finish_max_parts();
    if (!no_slab && a.slab && (n & 3) == 0 && (size_t)a.splitk * a.quads * n <= a.slab_floats) {
        // in-kernel sum (wgrad_finish.h) where the caller lent counters: one set per dW tile.  The tile count is bounded by
        // the smallest tiles any kernel form uses (32 x 32 per tap)
        const int cps = wgrad_cnt_per_tile(a.splitk * a.quads);
        const size_t tiles_max = (size_t)a.taps * ((a.Cin + 31) / 32) * ((a.Cout + 31) / 32);
        if (a.tile_cnt && !sum_kernel && tiles_max * cps <= a.cnt_ints) a.cnt_stride = cps;
        return hipSuccess;
    }
    a.slab = nullptr;      // atomics into a zeroed dW
    return hipMemsetAsync(a.dW, 0, n * sizeof(float), s);
}
hipError_t wgrad_split_finish(const WgradArgs& a, hipStream_t s) {
    const int parts = a.splitk * a.quads;       // f16x2: the three operand-plane pairs are summed like splits
    if (parts <= 1 || !a.slab || a.cnt_stride) return hipSuccess;     // (cnt_stride: summed inside the launch)
    const size_t n4 = (size_t)a.taps * a.Cin * a.Cout / 4;
    if (parts <= 8) {
        hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, a.slab, a.dW, n4,
                           parts, n4, a.scale);
    } else {
        hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, s, a.slab, a.dW, n4,
                           parts, n4, a.scale);
    }
    return hipGetLastError();
}

#ifdef Y2_DEVBUILD
// development variants (f16, 128 x 128 tiles): stages x blocks target of the 1x1 form
hipError_t launch_wgrad_variant(int variant, const WgradArgs& a0, hipStream_t s) {
    WgradArgs a = a0;
    wgrad_split_args(1, a);
    typedef half_t T;
    switch (variant) {
        case 100: return wg_launch<T, 2, 2, 2, 2, 2>(a, s, 256);
        case 101: return wg_launch<T, 2, 2, 2, 2, 3>(a, s, 256);
        case 102: return wg_launch<T, 2, 2, 2, 2, 4>(a, s, 256);
        case 103: return wg_launch<T, 2, 2, 2, 2, 2>(a, s, 512);
        case 104: return wg_launch<T, 2, 2, 2, 2, 3>(a, s, 512);
        case 105: return wg_launch<T, 2, 2, 2, 2, 5>(a, s, 256);
        case 106: return wg_launch<T, 2, 2, 2, 2, 4>(a, s, 128);
        case 107: return wg_launch<T, 2, 2, 2, 2, 2>(a, s, 768);
    }
    return hipErrorInvalidValue;
}
#endif

hipError_t launch_wgrad(int dtype, const WgradArgs& a0, hipStream_t s) {
    WgradArgs a = a0;
    if (a.Cin % 32 != 0 || (a.Cin > 128 && a.Cin % 128 != 0)) return hipErrorInvalidValue;
    dtype = wgrad_split_args(dtype, a);
    switch (dtype) {
        case 0: return wg_T<float>(a, s);
        case 1: return wg_T<half_t>(a, s);
        case 2: return wg_T<bf16_t>(a, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace y2
