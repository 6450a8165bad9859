// Weight gradient of a 3x3 convolution with ALL NINE TAPS in one block
// (TF Conv2DBackpropFilter, reference darknet.py:20-21):
//     dW[t][ci][co] = sum_p X[p (+) t][ci] * dY[p][co],   t = (kh, kw)
// The per-tap kernel (wgrad.hip) re-stages X and dY for every tap and is bound by the
// global->LDS fill rate (65 FLOP per staged byte).  Here a block owns a (BI x BO) slice
// of ci x co for all nine taps: per K step it stages ONE dY tile [64 px][BO] and ONE
// X window [64 + 2*pitch + 2 px][BI] of the bordered pixel space; the nine taps read
// row-shifted views of that window.  ~235 FLOP per staged byte at 13x13, 36 MFMAs per
// wave between barriers, and the dY fragment is shared by the nine MFMAs of a k-group.
// K runs linearly over the bordered pixel space (dY border = 0); fragments come out of
// the [pixel][channel] images through ds_read_b64_tr_b16.
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

// TG = tap groups: with TG = 2 two waves share one (ci, co) tile, taps 0-4 and 5-8 -- twice the
// waves per SIMD at the same accumulator footprint per wave (LDS reads and latency bubbles of
// one wave hide behind the other's MFMAs) and no cross-wave reduction.
// CW = 32-wide co sub-tiles per wave: with CW = 2 every X fragment feeds two MFMAs (1.4 instead of 2.4
// transposed LDS reads per MFMA) and the block stages a dY tile twice as wide per X window (337 instead of
// 144 FLOP per staged byte at 13x13 with 64 ci).
template <typename T, int WI, int WO, int TG = 1, int KS = 1, int CW = 1>
struct Wg9Cfg {
    static constexpr int NW = WI * WO * TG, NT = NW * 64;
    static constexpr int SZ = sizeof(T);
    static constexpr int BI = 32 * WI, BO = 32 * WO * CW;
    static constexpr int BKP = ((SZ == 2) ? 64 : 32) * KS;   // pixels per K step
    static constexpr int ROWX = BI * SZ, ROWY = BO * SZ;
    static constexpr int LPRX = ROWX / 16, LPRY = ROWY / 16;
    static constexpr int RPIX = 64 / LPRX, RPIY = 64 / LPRY;
    static constexpr int NIY = BKP / RPIY;
    static constexpr int IPWY = (NIY + NW - 1) / NW;
    static constexpr int YS = BKP * ROWY;
};

// bank swizzle of the [pixel][channel] images for the transposed 4x16 reads (see wgrad.hip)
template <int ROWB, int SZ>
Y2_DEV int wg9_swz(int row) {
    if (SZ != 2) return 0;
    if (ROWB >= 256) return (row & 3) << 2;
    if (ROWB == 128) return ((row & 3) >> 1) << 2;
    return 0;
}

// NS LDS stages; NS-1 K steps of LDS-DMA stay in flight across the raw barrier (counted vmcnt):
// with one wave per SIMD (as many waves as the dW tiling yields) this is what hides HBM latency.
// PL2 (f16x2 mode, round 5): x and dy are SPLIT tensors -- a cell is [C halves hi][C halves lo] -- and the block forms all
// three plane products itself: both planes of the X window and of the dY tile are staged per K step (2x the LDS of the
// one-plane form) and every fragment pair feeds three MFMAs, hi hi + lo hi + hi lo, into ONE accumulator.  Against
// three launches on plane pairs (WgradArgs::quads, the first form of the mode): 2/3 of the staging and of the LDS
// fragment reads per MFMA, one partial tile per split instead of three.
template <typename T, int WI, int WO, int NS, int TG, int T0, int NTAP, int KS = 1, int CW = 1, bool PL2 = false>
Y2_DEV void wg9_body(const WgradArgs& a, int wrows, char* smem, int* s_fin) {
    typedef Wg9Cfg<T, WI, WO, TG, KS, CW> Cfg;
    constexpr int NW = Cfg::NW, SZ = Cfg::SZ, BI = Cfg::BI, BO = Cfg::BO, BKP = Cfg::BKP;
    constexpr int ROWX = Cfg::ROWX, ROWY = Cfg::ROWY;
    constexpr int NPL = PL2 ? 2 : 1;
    static_assert(!PL2 || SZ == 2, "two planes: 16-bit operands");
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = w % (WI * WO);
    const int wi = wq / WO, wo = wq % WO;
    const int pitch = a.W + 1;

    const int nIT = (a.Cin + BI - 1) / BI, nOT = (a.Cout + BO - 1) / BO;
    // split-K launches: the slices of one K range meet in one XCD's L2 (measured +2..7 %); without a split the plain
    // order already gives every XCD a fixed set of co tiles
    int b = xcd_block(blockIdx.x, gridDim.x, a.xcd && a.splitk > 1);
    const int ot = b % nOT; b /= nOT;
    const int it = b % nIT;
    const int split = b / nIT;
    const int ci0 = it * BI, co0 = ot * BO;

    const long Mp = (long)bbody_pixels(a.N, a.H, a.W);
    const long ksteps = (Mp + BKP - 1) / BKP;
    const long spb = (ksteps + a.splitk - 1) / a.splitk;
    const long s_begin = (long)split * spb;
    long s_end = s_begin + spb;
    if (s_end > ksteps) s_end = ksteps;
    const int nsteps = (int)(s_end > s_begin ? s_end - s_begin : 0);

    const int xs_bytes = wrows * ROWX;              // one X window (of one plane)
    const int stage_bytes = NPL * (xs_bytes + Cfg::YS);      // [X hi][X lo][dY hi][dY lo]
    const int xpieces = wrows / Cfg::RPIX;
    const int xplaneB = a.Cin * SZ, yplaneB = a.Cdy * SZ;    // PL2: byte distance of the lo plane inside a cell
    const long kb = s_begin * BKP;
    // window of step s starts at bordered position kb + s*BKP - pitch - 1 (top-left tap)
    const char* xg = (const char*)a.x + ((kb - pitch - 1) * a.xpitch + ci0) * SZ;
    const char* yg = (const char*)a.dy + (kb * a.ypitch + co0) * SZ;
    const long xstep = (long)BKP * a.xpitch * SZ, ystep = (long)BKP * a.ypitch * SZ;

    const int lrx = lane / Cfg::LPRX, lsx = lane % Cfg::LPRX;
    uint32_t voffy[Cfg::IPWY];
#pragma unroll
    for (int i = 0; i < Cfg::IPWY; ++i) {
        const int row = (i * NW + w) * Cfg::RPIY + lane / Cfg::LPRY;
        const int sl = (lane % Cfg::LPRY) ^ wg9_swz<ROWY, SZ>(row);
        voffy[i] = (uint32_t)row * (uint32_t)(a.ypitch * SZ) + sl * 16;
    }
    auto stage = [&](int st, int buf) {
        const char* xs = xg + (long)st * xstep;
        const char* ys = yg + (long)st * ystep;
        char* lb = smem + buf * stage_bytes;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            for (int i = w; i < xpieces; i += NW) {
                const int row = i * Cfg::RPIX + lrx;
                const uint32_t off = (uint32_t)row * (uint32_t)(a.xpitch * SZ) + ((lsx ^ wg9_swz<ROWX, SZ>(row)) * 16);
                glds16(xs + pl * xplaneB + off, lb + pl * xs_bytes + i * 1024);
            }
#pragma unroll
            for (int i = 0; i < Cfg::IPWY; ++i) {
                const int ii = i * NW + w;
                if ((i + 1) * NW <= Cfg::NIY || ii < Cfg::NIY)
                    glds16(ys + pl * yplaneB + voffy[i], lb + NPL * xs_bytes + pl * Cfg::YS + ii * 1024);
            }
        }
    };

    f32x16 acc[NTAP][CW];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int j = 0; j < CW; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][j][q] = 0.f;

    const int r32 = lane & 31, hh = lane >> 5;
    const int qq = (lane & 15) >> 2, pp = lane & 3, g1 = (lane >> 4) & 1;
    int shift[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t) shift[t] = ((T0 + t) / 3) * pitch + ((T0 + t) % 3);

    // loads per wave per stage (the launcher makes the X window a multiple of RPIX*NW rows)
    const int lps = NPL * (xpieces / NW + Cfg::NIY / NW);
#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0)
        if (s0 < nsteps) stage(s0, s0);
    int cbuf = 0, ibuf = NS - 1;
    for (int st = 0; st < nsteps; ++st) {
        if (st + NS - 2 < nsteps) wait_vmcnt_dyn((NS - 2) * lps);
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (st + NS - 1 < nsteps) stage(st + NS - 1, ibuf);
        const int buf = cbuf;
        cbuf = (cbuf + 1 == NS) ? 0 : cbuf + 1;
        ibuf = (ibuf + 1 == NS) ? 0 : ibuf + 1;
        const char* xs = smem + buf * stage_bytes;
        const char* ys = xs + NPL * xs_bytes;
        if constexpr (SZ == 2) {
            // dY fragment address (no tap shift): rows kg*16 + 8*hh + qq (+4)
            const int fy = wg9_swz<ROWY, SZ>(qq);
            const char* pyb[CW];
#pragma unroll
            for (int j = 0; j < CW; ++j)
                pyb[j] = ys + (8 * hh + qq) * ROWY + ((((wo * CW + j) * 4 + 2 * g1 + (pp >> 1)) ^ fy) * 16) + (pp & 1) * 8;
            // X fragment per tap: row = 8*hh + qq + shift[t]; the swizzle follows the LDS row
            const char* pxb[NTAP];
#pragma unroll
            for (int t = 0; t < NTAP; ++t) {
                const int row = 8 * hh + qq + shift[t];
                const int fx = wg9_swz<ROWX, SZ>(row);
                pxb[t] = xs + row * ROWX + (((wi * 4 + 2 * g1 + (pp >> 1)) ^ fx) * 16) + (pp & 1) * 8;
            }
            // Software pipeline over the k-groups: the fragments of group kg+1 are requested
            // while the nine MFMAs of group kg issue (one wave per SIMD: nothing else hides the
            // ~100-cycle LDS latency; the compiler's own schedule keeps only one fragment ahead).
            typedef typename Elem<T>::frag frag_t;
            frag_t fa0[NPL][NTAP], fa1[NPL][NTAP], fb0[NPL][CW], fb1[NPL][CW];
            auto load_group = [&](int kg, frag_t (&fa)[NPL][NTAP], frag_t (&fb)[NPL][CW]) {
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
                    for (int j = 0; j < CW; ++j) {
                        const char* py = pyb[j] + pl * Cfg::YS + kg * 16 * ROWY;
                        fb[pl][j] = tr_frag<T>(py, py + 4 * ROWY);
                    }
#pragma unroll
                    for (int t = 0; t < NTAP; ++t) {
                        const char* px = pxb[t] + pl * xs_bytes + kg * 16 * ROWX;
                        fa[pl][t] = tr_frag<T>(px, px + 4 * ROWX);
                    }
                }
            };
            auto mma_group = [&](frag_t (&fa)[NPL][NTAP], frag_t (&fb)[NPL][CW]) {
#pragma unroll
                for (int t = 0; t < NTAP; ++t)
#pragma unroll
                    for (int j = 0; j < CW; ++j) {
                        mma32(acc[t][j], fa[0][t], fb[0][j]);
                        if constexpr (PL2) {
                            mma32(acc[t][j], fa[1][t], fb[0][j]);      // x lo * dy hi
                            mma32(acc[t][j], fa[0][t], fb[1][j]);      // x hi * dy lo
                        }
                    }
            };
            load_group(0, fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kg = 0; kg < BKP / 16; kg += 2) {
                load_group(kg + 1, fa1, fb1);
                mma_group(fa0, fb0);
                __builtin_amdgcn_sched_barrier(0);
                if (kg + 2 < BKP / 16) load_group(kg + 2, fa0, fb0);
                mma_group(fa1, fb1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll 4
            for (int s2 = 0; s2 < BKP / 2; ++s2) {
                const int row = 2 * s2 + hh;
                float fb[CW];
#pragma unroll
                for (int j = 0; j < CW; ++j) fb[j] = *(const float*)(ys + row * ROWY + ((wo * CW + j) * 32 + r32) * 4);
#pragma unroll
                for (int t = 0; t < NTAP; ++t) {
                    const float fa = *(const float*)(xs + (row + shift[t]) * ROWX + (wi * 32 + r32) * 4);
#pragma unroll
                    for (int j = 0; j < CW; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb[j], acc[t][j], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < CW; ++j) {
        const int co = co0 + (wo * CW + j) * 32 + r32;
        if (co >= a.Cout) continue;
#pragma unroll
        for (int t = 0; t < NTAP; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int ci = ci0 + wi * 32 + acc_row(q, hh);
                if (ci < a.Cin) {
                    const size_t o = ((size_t)(T0 + t) * a.Cin + ci) * a.Cout + co;
                    if (a.splitk == 1 && a.quads == 1) a.dW[o] = acc[t][j][q] * a.scale;
                    else if (a.slab) { float* sp = a.slab + (size_t)(a.part0 + split) * 9 * a.Cin * a.Cout + o; Y2_FIN_STORE(sp, acc[t][j][q]); }
                    else atomicAdd(a.dW + o, acc[t][j][q] * a.scale);
                }
            }
    }
#ifdef Y2_DEVBUILD
    if (a.slab && a.cnt_stride) {     // the split-K sum rides in this kernel (wgrad_finish.h; development library only)
        const size_t n9 = (size_t)9 * a.Cin * a.Cout;
        splitk_finish(s_fin, a.tile_cnt + (size_t)(it * nOT + ot) * a.cnt_stride, a.part0 + split, a.splitk * a.quads,
                      [&](int first, int stride, int count, bool final) __attribute__((always_inline)) {
#pragma unroll 1
            for (int j = 0; j < CW; ++j) {
                const int co = co0 + (wo * CW + j) * 32 + r32;
                if (co >= a.Cout) continue;
#pragma unroll 1
                for (int t = 0; t < NTAP; ++t)
#pragma unroll 1
                    for (int q = 0; q < 16; ++q) {
                        const int ci = ci0 + wi * 32 + acc_row(q, hh);
                        if (ci < a.Cin) {
                            const size_t o = ((size_t)(T0 + t) * a.Cin + ci) * a.Cout + co;
                            const float v = splitk_sum_slots(a.slab + (size_t)first * n9 + o, (size_t)stride * n9, count);
                            if (final) a.dW[o] = v * a.scale;
                            else slab_store(a.slab + (size_t)first * n9 + o, v);
                        }
                    }
            }
        });
    }
#endif
}

template <typename T, int WI, int WO, int NS, int TG, int KS = 1, int CW = 1, bool PL2 = false>
__global__ __launch_bounds__(WI* WO* TG * 64) void wgrad9_kernel(WgradArgs a, int wrows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* const s_fin = (int*)(smem + a.fin_lds_off);
    if constexpr (TG == 1) {
        wg9_body<T, WI, WO, NS, 1, 0, 9, KS, CW, PL2>(a, wrows, smem, s_fin);
    } else {
        const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        if (w < WI * WO) wg9_body<T, WI, WO, NS, 2, 0, 5, KS, CW, PL2>(a, wrows, smem, s_fin);   // same barrier count in both arms
        else wg9_body<T, WI, WO, NS, 2, 5, 4, KS, CW, PL2>(a, wrows, smem, s_fin);
    }
}


// ---------------------------------------------------------------------------
// RING form for long image rows (f16/bf16).  The X window of a K step is
// 64 + 2*pitch + 2 bordered pixels; consecutive steps overlap in all but 64 of them, and
// at W = 104 / 208 re-staging the whole window per step costs 4.3x / 7.6x the traffic of the
// step itself (measured: those layers were bound by the global->LDS fill).  Here the X rows
// live in an LDS ring of R = 2^k rows: every step stages only its 64 NEW rows (one aligned
// group), the nine taps read row-shifted views of the ring, and the wrap is one AND on the
// byte offset (v_add + v_and_or per address, hidden in the MFMA shadow).
//   group g = window rows [BKP*g, BKP*(g+1)) (BKP = 64 or 128 pixels per K step); step st reads groups
//   st .. st+G-1, G = ceil(wrows/BKP); group st+G is in flight during step st, so R >= BKP*(G+1).
// ---------------------------------------------------------------------------
template <typename T>
Y2_DEV typename Elem<T>::frag tr_frag_off(uint32_t o0, uint32_t o1) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)o0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)o1);
    s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(typename Elem<T>::frag, both);
}

template <typename T, int WI, int WO, int TG, int T0, int NTAP, int KS = 1, bool PL2 = false>
Y2_DEV void wg9r_body(const WgradArgs& a, int lgR, int G, char* smem, int* s_fin) {
    typedef Wg9Cfg<T, WI, WO, TG, KS> Cfg;
    constexpr int NW = Cfg::NW, SZ = Cfg::SZ, BI = Cfg::BI, BO = Cfg::BO, BKP = Cfg::BKP;
    constexpr int ROWX = Cfg::ROWX, ROWY = Cfg::ROWY;
    constexpr int NPL = PL2 ? 2 : 1;          // PL2: both operand planes of the split-operand mode (see wg9_body)
    static_assert(SZ == 2, "ring form: 16-bit elements");
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = w % (WI * WO);
    const int wi = wq / WO, wo = wq % WO;
    const int pitch = a.W + 1;
    const uint32_t R = 1u << lgR;
    const uint32_t ringB = R * ROWX, maskB = ringB - 1;
    const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    if (lbase & maskB) __builtin_trap();   // the wrap ORs the column bits into the masked row offset

    const int nIT = (a.Cin + BI - 1) / BI, nOT = (a.Cout + BO - 1) / BO;
    // split-K launches: the slices of one K range meet in one XCD's L2 (measured +2..7 %); without a split the plain
    // order already gives every XCD a fixed set of co tiles
    int b = xcd_block(blockIdx.x, gridDim.x, a.xcd && a.splitk > 1);
    const int ot = b % nOT; b /= nOT;
    const int it = b % nIT;
    const int split = b / nIT;
    const int ci0 = it * BI, co0 = ot * BO;

    const long Mp = (long)bbody_pixels(a.N, a.H, a.W);
    const long ksteps = (Mp + BKP - 1) / BKP;
    const long spb = (ksteps + a.splitk - 1) / a.splitk;
    const long s_begin = (long)split * spb;
    long s_end = s_begin + spb;
    if (s_end > ksteps) s_end = ksteps;
    const int nsteps = (int)(s_end > s_begin ? s_end - s_begin : 0);

    const long kb = s_begin * BKP;
    const char* xg = (const char*)a.x + ((kb - pitch - 1) * a.xpitch + ci0) * SZ;   // window row 0 of step 0
    const char* yg = (const char*)a.dy + (kb * a.ypitch + co0) * SZ;
    const long xstep = (long)BKP * a.xpitch * SZ, ystep = (long)BKP * a.ypitch * SZ;
    char* const ybase = smem + NPL * ringB;      // [ring hi][ring lo][dY: buffer x plane]
    const int xplaneB = a.Cin * SZ, yplaneB = a.Cdy * SZ;

    const int lrx = lane / Cfg::LPRX, lsx = lane % Cfg::LPRX;
    uint32_t voffy[Cfg::IPWY];
#pragma unroll
    for (int i = 0; i < Cfg::IPWY; ++i) {
        const int row = (i * NW + w) * Cfg::RPIY + lane / Cfg::LPRY;
        const int sl = (lane % Cfg::LPRY) ^ wg9_swz<ROWY, SZ>(row);
        voffy[i] = (uint32_t)row * (uint32_t)(a.ypitch * SZ) + sl * 16;
    }
    auto stage_x = [&](int g) {   // BKP rows: BKP / RPIX pieces of 1 KiB (aligned: a piece never wraps)
        const char* xs = xg + (long)g * xstep;
        char* dst = smem + (((uint32_t)g * (uint32_t)BKP * ROWX) & maskB);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
            for (int i = w; i < BKP / Cfg::RPIX; i += NW) {
                const int row = i * Cfg::RPIX + lrx;   // row & 3 == LDS row & 3 (groups are 64-row aligned)
                const uint32_t off = (uint32_t)row * (uint32_t)(a.xpitch * SZ) + ((lsx ^ wg9_swz<ROWX, SZ>(row)) * 16);
                glds16(xs + pl * xplaneB + off, dst + pl * ringB + i * 1024);
            }
    };
    auto stage_y = [&](int st, int buf) {
        const char* ys = yg + (long)st * ystep;
        char* lb = ybase + buf * NPL * Cfg::YS;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int i = 0; i < Cfg::IPWY; ++i) {
                const int ii = i * NW + w;
                if ((i + 1) * NW <= Cfg::NIY || ii < Cfg::NIY) glds16(ys + pl * yplaneB + voffy[i], lb + pl * Cfg::YS + ii * 1024);
            }
    };

    f32x16 acc[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

    const int r32 = lane & 31, hh = lane >> 5;
    const int qq = (lane & 15) >> 2, pp = lane & 3, g1 = (lane >> 4) & 1;
    // per tap: byte offset of this lane's first row in the ring (advances 64 rows per step) and the
    // constant column part (swizzle follows row & 3, which neither the step nor the wrap changes)
    uint32_t rbB[NTAP], cb[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
        const int row = 8 * hh + qq + ((T0 + t) / 3) * pitch + ((T0 + t) % 3);
        const int fx = wg9_swz<ROWX, SZ>(row);
        rbB[t] = ((uint32_t)row * ROWX) & maskB;
        cb[t] = lbase | (uint32_t)((((wi * 4 + 2 * g1 + (pp >> 1)) ^ fx) * 16) + (pp & 1) * 8);
    }

    if (nsteps > 0) {
        for (int g = 0; g < G; ++g) stage_x(g);
        stage_y(0, 0);
    }
    for (int st = 0; st < nsteps; ++st) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (st + 1 < nsteps) {
            stage_x(st + G);
            stage_y(st + 1, (st + 1) & 1);
        }
        const char* ys = ybase + (st & 1) * NPL * Cfg::YS;
        const int fy = wg9_swz<ROWY, SZ>(qq);
        const char* pyb = ys + (8 * hh + qq) * ROWY + (((wo * 4 + 2 * g1 + (pp >> 1)) ^ fy) * 16) + (pp & 1) * 8;
        typedef typename Elem<T>::frag frag_t;
        frag_t fa0[NPL][NTAP], fa1[NPL][NTAP], fb0[NPL], fb1[NPL];
        auto load_group = [&](int kg, frag_t (&fa)[NPL][NTAP], frag_t (&fb)[NPL]) {
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                const char* py = pyb + pl * Cfg::YS + kg * 16 * ROWY;
                fb[pl] = tr_frag<T>(py, py + 4 * ROWY);
#pragma unroll
                for (int t = 0; t < NTAP; ++t) {
                    // (the second ring starts ringB further: a multiple of the ring size, so the OR of the column bits holds)
                    const uint32_t o0 = (((rbB[t] + (uint32_t)(kg * 16 * ROWX)) & maskB) | cb[t]) + (uint32_t)pl * ringB;
                    const uint32_t o1 = (((rbB[t] + (uint32_t)((kg * 16 + 4) * ROWX)) & maskB) | cb[t]) + (uint32_t)pl * ringB;
                    fa[pl][t] = tr_frag_off<T>(o0, o1);
                }
            }
        };
        auto mma_group = [&](frag_t (&fa)[NPL][NTAP], frag_t (&fb)[NPL]) {
#pragma unroll
            for (int t = 0; t < NTAP; ++t) {
                mma32(acc[t], fa[0][t], fb[0]);
                if constexpr (PL2) {
                    mma32(acc[t], fa[1][t], fb[0]);      // x lo * dy hi
                    mma32(acc[t], fa[0][t], fb[1]);      // x hi * dy lo
                }
            }
        };
        load_group(0, fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kg = 0; kg < BKP / 16; kg += 2) {
            load_group(kg + 1, fa1, fb1);
            mma_group(fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            if (kg + 2 < BKP / 16) load_group(kg + 2, fa0, fb0);
            mma_group(fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < NTAP; ++t) rbB[t] = (rbB[t] + (uint32_t)BKP * ROWX) & maskB;
    }
    const int co = co0 + wo * 32 + r32;
    if (co < a.Cout) {
#pragma unroll
        for (int t = 0; t < NTAP; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int ci = ci0 + wi * 32 + acc_row(q, hh);
                if (ci < a.Cin) {
                    const size_t o = ((size_t)(T0 + t) * a.Cin + ci) * a.Cout + co;
                    if (a.splitk == 1 && a.quads == 1) a.dW[o] = acc[t][q] * a.scale;
                    else if (a.slab) { float* sp = a.slab + (size_t)(a.part0 + split) * 9 * a.Cin * a.Cout + o; Y2_FIN_STORE(sp, acc[t][q]); }
                    else atomicAdd(a.dW + o, acc[t][q] * a.scale);
                }
            }
    }
#ifdef Y2_DEVBUILD
    if (a.slab && a.cnt_stride) {     // the split-K sum rides in this kernel (wgrad_finish.h; development library only)
        const size_t n9 = (size_t)9 * a.Cin * a.Cout;
        splitk_finish(s_fin, a.tile_cnt + (size_t)(it * nOT + ot) * a.cnt_stride, a.part0 + split, a.splitk * a.quads,
                      [&](int first, int stride, int count, bool final) __attribute__((always_inline)) {
            if (co >= a.Cout) return;
#pragma unroll 1
            for (int t = 0; t < NTAP; ++t)
#pragma unroll 1
                for (int q = 0; q < 16; ++q) {
                    const int ci = ci0 + wi * 32 + acc_row(q, hh);
                    if (ci < a.Cin) {
                        const size_t o = ((size_t)(T0 + t) * a.Cin + ci) * a.Cout + co;
                        const float v = splitk_sum_slots(a.slab + (size_t)first * n9 + o, (size_t)stride * n9, count);
                        if (final) a.dW[o] = v * a.scale;
                        else slab_store(a.slab + (size_t)first * n9 + o, v);
                    }
                }
        });
    }
#endif
}

template <typename T, int WI, int WO, int TG, int KS = 1, bool PL2 = false>
__global__ __launch_bounds__(WI* WO* TG * 64) void wgrad9r_kernel(WgradArgs a, int lgR, int G) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* const s_fin = (int*)(smem + a.fin_lds_off);
    if constexpr (TG == 1) {
        wg9r_body<T, WI, WO, 1, 0, 9, KS, PL2>(a, lgR, G, smem, s_fin);
    } else {
        const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        if (w < WI * WO) wg9r_body<T, WI, WO, 2, 0, 5, KS, PL2>(a, lgR, G, smem, s_fin);
        else wg9r_body<T, WI, WO, 2, 5, 4, KS, PL2>(a, lgR, G, smem, s_fin);
    }
}

// blocks_target: split-K so that tiles * splitk ~ blocks_target; 0 = one full wave of resident
// blocks (256 CUs x blocks per CU by LDS, at most 3: measured best on every long-row shape --
// 1.5 waves of blocks cost 30 % at 104x104)
template <typename T, int WI, int WO, int TG, int KS = 1, bool PL2 = false>
static hipError_t wg9r_launch(WgradArgs a, hipStream_t s, int blocks_target = 0) {
    typedef Wg9Cfg<T, WI, WO, TG, KS> Cfg;
    static_assert(Cfg::NIY % Cfg::NW == 0, "dY pieces must split evenly over waves");
    const int pitch = a.W + 1;
    const int wrows = Cfg::BKP + 2 * pitch + 2;
    const int G = (wrows + Cfg::BKP - 1) / Cfg::BKP;
    int lgR = 7;
    while ((1 << lgR) < Cfg::BKP * (G + 1)) ++lgR;
    // (+ 16 bytes for the finish flag of the opt-in in-kernel sum, wgrad_finish.h, only when it is on: a request of
    //  exactly 1/2 or 1/3 of the CU's LDS must stay that -- the blocks-per-CU count below depends on it)
    const size_t fin16 = wgrad_finish_max_parts() > 0 ? 16 : 0;
    const size_t lds = (PL2 ? 2 : 1) * (((size_t)Cfg::ROWX << lgR) + 2 * (size_t)Cfg::YS) + fin16;
    if (lds > 160 * 1024) return hipErrorOutOfMemory;
    a.fin_lds_off = (int)(lds - fin16);
    if (PL2) a.quads = 1;       // the three plane products are formed inside the block
    auto kern = wgrad9r_kernel<T, WI, WO, TG, KS, PL2>;
    const int nIT = (a.Cin + Cfg::BI - 1) / Cfg::BI, nOT = (a.Cout + Cfg::BO - 1) / Cfg::BO;
    const long Mp = (long)bbody_pixels(a.N, a.H, a.W);
    const long ksteps = (Mp + Cfg::BKP - 1) / Cfg::BKP;
    const int tiles = nIT * nOT;
    if (blocks_target <= 0) {
        int bpc = (int)((160 * 1024) / lds);
        bpc = bpc > 3 ? 3 : (bpc < 1 ? 1 : bpc);
        blocks_target = 256 * bpc;
    }
    if (a.splitk <= 0) {
        long sk = blocks_target / tiles;
        const long maxsk = (ksteps + 4 * G - 1) / (4 * G);   // the G-group prologue must stay a small part of a block
        if (sk > maxsk) sk = maxsk;
        if (sk < 1) sk = 1;
        a.splitk = (int)sk;
    }
    static size_t attr = 0;
    if (lds > attr) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr = lds;
    }
    hipError_t e = wgrad_split_prepare(a, s);
    if (e != hipSuccess) return e;
    e = wgrad_launch_quads(kern, dim3(tiles * a.splitk), dim3(Cfg::NT), lds, s, a, lgR, G);
    return e != hipSuccess ? e : wgrad_split_finish(a, s);
}

template <typename T, int WI, int WO, int NS, int TG = 1, int KS = 1, int CW = 1, bool PL2 = false>
static hipError_t wg9_launch_ns(WgradArgs a, hipStream_t s, int blocks_target = 0) {
    typedef Wg9Cfg<T, WI, WO, TG, KS, CW> Cfg;
    static_assert(Cfg::NIY % Cfg::NW == 0, "dY pieces must split evenly over waves");
    const int pitch = a.W + 1;
    int wrows = Cfg::BKP + 2 * pitch + 2;
    // deeper rings count the loads in flight, so every wave must issue the same number of pieces; two
    // stages drain to zero and take whole pieces only
    const int gran = NS > 2 ? Cfg::RPIX * Cfg::NW : Cfg::RPIX;
    wrows = (wrows + gran - 1) / gran * gran;
    const size_t fin16 = wgrad_finish_max_parts() > 0 ? 16 : 0;     // the finish flag (wgrad_finish.h), when that form is on
    size_t lds = NS * (PL2 ? 2 : 1) * ((size_t)wrows * Cfg::ROWX + Cfg::YS) + fin16;
    if (lds > 160 * 1024) return hipErrorOutOfMemory;
    a.fin_lds_off = (int)(lds - fin16);
    if (PL2) a.quads = 1;       // the three plane products are formed inside the block
    auto kern = wgrad9_kernel<T, WI, WO, NS, TG, KS, CW, PL2>;
    const int nIT = (a.Cin + Cfg::BI - 1) / Cfg::BI, nOT = (a.Cout + Cfg::BO - 1) / Cfg::BO;
    const long Mp = (long)bbody_pixels(a.N, a.H, a.W);
    const long ksteps = (Mp + Cfg::BKP - 1) / Cfg::BKP;
    const int tiles = nIT * nOT;
    if (a.splitk <= 0) {
        // short image rows (big dW, K = a few thousand steps): two blocks per CU measured best;
        // long rows (tiny dW, K = 10^5 steps): ~3 blocks per CU to cover the HBM stream -- but never more blocks than
        // LDS lets a CU hold at once (round 4: 28x28 128 -> 256 at batch 128 asked for 768 blocks of 64 KB, 1.5
        // rounds of the chip: 91 us against 82 with 512)
        int bpc = (int)((160 * 1024) / lds);
        bpc = bpc > 3 ? 3 : (bpc < 1 ? 1 : bpc);
        const long long_rows = 256L * bpc;
        long sk = a.W <= 26 ? (512 + tiles / 2) / tiles : (long_rows + tiles - 1) / tiles;
        if (blocks_target > 0) sk = (blocks_target + tiles / 2) / tiles;
        const long maxsk = (ksteps + 7) / 8;
        if (sk > maxsk) sk = maxsk;
        if (sk < 1) sk = 1;
        a.splitk = (int)sk;
    }
    // With no more blocks than CUs, ask for > half of a CU's LDS: the dispatcher then cannot
    // co-locate two blocks on one CU while another CU idles (measured: it does otherwise).
    if (tiles * a.splitk <= 256 && lds < 84 * 1024) lds = 84 * 1024;
    static size_t attr = 0;
    if (lds > attr) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr = lds;
    }
    hipError_t e = wgrad_split_prepare(a, s);
    if (e != hipSuccess) return e;
    e = wgrad_launch_quads(kern, dim3(tiles * a.splitk), dim3(Cfg::NT), lds, s, a, wrows);
    return e != hipSuccess ? e : wgrad_split_finish(a, s);
}
// deepest ring that fits (4 stages where the window is small)
template <typename T, int WI, int WO>
static hipError_t wg9_launch(const WgradArgs& a, hipStream_t s, int ns = 0) {
    typedef Wg9Cfg<T, WI, WO> Cfg;
    const int gran = Cfg::RPIX * Cfg::NW;
    const int wrows = (Cfg::BKP + 2 * (a.W + 1) + 2 + gran - 1) / gran * gran;
    const size_t stage = (size_t)wrows * Cfg::ROWX + Cfg::YS;
    (void)stage;
    if (ns == 0) ns = 2;   // deeper rings measured slower on every Darknet-19 shape (bench_wgrad.py)
    if (ns >= 4) return wg9_launch_ns<T, WI, WO, 4>(a, s);
    if (ns == 3) return wg9_launch_ns<T, WI, WO, 3>(a, s);
    return wg9_launch_ns<T, WI, WO, 2>(a, s);
}

template <typename T>
static hipError_t wg9_T(const WgradArgs& a, hipStream_t s) {
    // measured (scripts/bench_wgrad.py): narrow co tiles with the taps split over two waves --
    // many small blocks, two waves per SIMD -- beat 64x64 tiles on every Darknet-19 shape
    if constexpr (sizeof(T) == 2) {
        // long rows: the ring form stages 64 new rows per step instead of the whole window
#ifdef Y2_DEVBUILD
        static const int minw = getenv("Y2DEV_WG9R_MINW") ? atoi(getenv("Y2DEV_WG9R_MINW")) : 52;
#else
        constexpr int minw = 52;
#endif
        if (a.W >= minw) {
            hipError_t e;
            // f16x2 mode: both planes in the ring (PL2, see wg9_body), 64-pixel K steps (the two rings of the 128-pixel form
            // leave no room at 104 / 208)
            static const bool quads_form_r = getenv("Y2_SPLIT_WGRAD_QUADS") != nullptr;
            if (a.quads == 3 && !quads_form_r) {
                if (a.Cin >= 64) e = wg9r_launch<T, 2, 1, 2, 1, true>(a, s);
                else if (a.Cdy >= 64) e = wg9r_launch<T, 1, 2, 2, 1, true>(a, s);
                else e = wg9r_launch<T, 1, 1, 2, 1, true>(a, s);
                if (e != hipErrorOutOfMemory) return e;
                (void)hipGetLastError();
            }
            // 64 ci x 32 co tiles with 128-pixel K steps measured best at 52 and 104 (3-7 % over 64-pixel
            // steps); 32-channel inputs (208x208): 32 ci x 64 co tiles, 64-pixel steps
            if (a.Cin >= 64) e = wg9r_launch<T, 2, 1, 2, 2>(a, s);
            else if (a.Cdy >= 64) {
                // 128-pixel K steps where the ring of that form still leaves two blocks per CU (112x112 at batch 128:
                // 90 us against 129; at 208x208 its 96 KB allow one block and the 64-pixel form wins, 160 against 184)
                typedef Wg9Cfg<T, 1, 2, 2, 2> C2;
                const int wrows2 = C2::BKP + 2 * (a.W + 1) + 2;
                const int G2 = (wrows2 + C2::BKP - 1) / C2::BKP;
                int lg = 7;
                while ((1 << lg) < C2::BKP * (G2 + 1)) ++lg;
                const size_t lds2 = ((size_t)C2::ROWX << lg) + 2 * (size_t)C2::YS;
                e = lds2 <= 80 * 1024 ? wg9r_launch<T, 1, 2, 2, 2>(a, s) : wg9r_launch<T, 1, 2, 2>(a, s);
            }
            else e = wg9r_launch<T, 1, 1, 2>(a, s);
            if (e != hipErrorOutOfMemory) return e;
            (void)hipGetLastError();
        }
    }
    if constexpr (sizeof(T) == 2) {
        // f16x2 mode (a.quads == 3): both operand planes staged per K step, the three plane products in one block (PL2) --
        // 64-pixel K steps so that two blocks still share a CU's LDS.  Y2_SPLIT_WGRAD_QUADS=1: three launches on plane pairs
        static const bool quads_form = getenv("Y2_SPLIT_WGRAD_QUADS") != nullptr;
        if (a.quads == 3 && a.Cin >= 64 && !quads_form) {
            hipError_t e = wg9_launch_ns<T, 2, 1, 2, 2, 1, 1, true>(a, s);
            if (e != hipErrorOutOfMemory) return e;
            (void)hipGetLastError();
        }
    }
    if (a.Cin >= 64) {
        // split-K shapes (fewer than 512 dW tiles): K steps of 128 pixels -- half the barriers and a
        // 1.2x instead of 1.4x window at 13x13 (6-8 % faster at 26x26 and on the 512-channel 13x13 layers;
        // the 1024 x 1024 layers, one block per tile, are 2 % faster with 64)
        const int tiles = ((a.Cin + 63) / 64) * ((a.Cout + 31) / 32);
        if (sizeof(T) == 2 && tiles < 512) {
            hipError_t e = wg9_launch_ns<T, 2, 1, 2, 2, 2>(a, s);
            if (e != hipErrorOutOfMemory) return e;
            (void)hipGetLastError();
        }
        return wg9_launch_ns<T, 2, 1, 2, 2>(a, s);     // 64 ci x 32 co, 4 waves
    }
    if (a.Cdy >= 64) return wg9_launch_ns<T, 1, 2, 2, 2>(a, s);     // 32 ci x 64 co, 4 waves
    return wg9_launch_ns<T, 1, 1, 2, 2>(a, s);
}

// 3x3 only; the window grows with the image row, so this form is for short rows
hipError_t launch_wgrad9(int dtype, const WgradArgs& a0, hipStream_t s) {
    WgradArgs a = a0;
    if (a.taps != 9 || a.Cin % 32 != 0) return hipErrorInvalidValue;
    dtype = wgrad_split_args(dtype, a);
    switch (dtype) {
        case 0: return wg9_T<float>(a, s);
        case 1: return wg9_T<half_t>(a, s);
        case 2: return wg9_T<bf16_t>(a, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace y2

namespace y2 {
// policy: all-taps kernel where the image rows are short (window ~1.5-2x the K step),
// per-tap kernel on the large feature maps and for 1x1 filters
#ifdef Y2_DEVBUILD
// development variants (f16): explicit block shapes
hipError_t launch_wgrad9_variant(int variant, const WgradArgs& a0, hipStream_t s) {
    WgradArgs a = a0;
    wgrad_split_args(1, a);
    switch (variant) {
        case 2: return wg9_launch<half_t, 2, 1>(a, s);
        case 3: return wg9_launch<half_t, 1, 2>(a, s);
        case 4: return wg9_launch<half_t, 1, 1>(a, s);
        case 5: return wg9_launch<half_t, 2, 2>(a, s);
        case 6: return wg9_launch<half_t, 2, 2>(a, s, 2);
        case 7: return wg9_launch<half_t, 2, 2>(a, s, 3);
        case 8: return wg9_launch<half_t, 2, 2>(a, s, 4);
        case 9: return wg9_launch_ns<half_t, 2, 2, 2, 2>(a, s);      // two tap groups, 8 waves
        case 10: return wg9_launch_ns<half_t, 2, 2, 3, 2>(a, s);
        case 11: return wg9_launch_ns<half_t, 2, 1, 2, 2>(a, s);     // 64 x 32 tiles, 4 waves
        case 12: return wg9_launch_ns<half_t, 1, 1, 2, 2>(a, s);     // 32 x 32 tiles, 2 waves
        case 13: return wg9_launch_ns<half_t, 1, 2, 2, 2>(a, s);     // 32 x 64 tiles, 4 waves
        case 14: return wg9_launch_ns<half_t, 2, 1, 3, 2>(a, s);     // 64 x 32, 3 stages
        case 16: return wg9_launch_ns<half_t, 2, 1, 2, 1>(a, s);     // 64 x 32 tiles, 2 waves (no tap split)
        case 50: return wg9_launch_ns<half_t, 2, 1, 2, 2, 2>(a, s);   // 64 x 32, K step of 128 pixels
        // two co sub-tiles per wave
        case 60: return wg9_launch_ns<half_t, 2, 2, 2, 2, 1, 2>(a, s, 256);   // 64 ci x 128 co, 8 waves, one block per CU
        case 61: return wg9_launch_ns<half_t, 2, 2, 2, 2, 2, 2>(a, s, 256);   // same, K step 128
        case 62: return wg9_launch_ns<half_t, 2, 1, 2, 2, 1, 2>(a, s, 512);   // 64 ci x 64 co, 4 waves, two blocks per CU
        case 63: return wg9_launch_ns<half_t, 2, 1, 2, 2, 2, 2>(a, s, 512);
        case 64: return wg9_launch_ns<half_t, 4, 1, 2, 2, 1, 2>(a, s, 256);   // 128 ci x 64 co, 8 waves
        case 65: return wg9_launch_ns<half_t, 4, 1, 2, 2, 2, 2>(a, s, 256);
        case 66: return wg9_launch_ns<half_t, 2, 2, 3, 2, 1, 2>(a, s, 256);   // 3 stages
        case 67: return wg9_launch_ns<half_t, 4, 2, 2, 2, 1, 2>(a, s, 256);   // 128 ci x 128 co, 16 waves
        case 68: return wg9_launch_ns<half_t, 2, 1, 3, 2, 1, 2>(a, s, 512);
        case 69: return wg9_launch_ns<half_t, 2, 2, 2, 2, 1, 2>(a, s, 512);   // 64 x 128, two blocks per CU
        case 51: return wg9_launch_ns<half_t, 2, 2, 2, 2, 2>(a, s);   // 64 x 64, 8 waves, K step 128
        case 52: return wg9_launch_ns<half_t, 1, 2, 2, 2, 2>(a, s);
        // ring form (long rows)
        case 30: return wg9r_launch<half_t, 2, 1, 2>(a, s, 768);     // 64 x 32, 4 waves
        case 31: return wg9r_launch<half_t, 1, 2, 2>(a, s, 768);     // 32 x 64, 4 waves
        case 32: return wg9r_launch<half_t, 1, 1, 2>(a, s, 768);     // 32 x 32, 2 waves
        case 33: return wg9r_launch<half_t, 2, 2, 2>(a, s, 768);     // 64 x 64, 8 waves
        case 34: return wg9r_launch<half_t, 2, 1, 2>(a, s, 512);
        case 35: return wg9r_launch<half_t, 1, 2, 2>(a, s, 512);
        case 36: return wg9r_launch<half_t, 2, 2, 2>(a, s, 512);
        case 37: return wg9r_launch<half_t, 2, 1, 2>(a, s, 1024);
        case 38: return wg9r_launch<half_t, 1, 2, 2>(a, s, 1024);
        case 39: return wg9r_launch<half_t, 2, 2, 2>(a, s, 1024);
        case 44: return wg9r_launch<half_t, 1, 2, 2, 2>(a, s);       // ring, K step 128
        case 45: return wg9r_launch<half_t, 2, 1, 2, 2>(a, s);
        case 46: return wg9r_launch<half_t, 1, 2, 2, 2>(a, s, 512);
        case 47: return wg9r_launch<half_t, 2, 1, 2, 2>(a, s, 512);
        case 40: return wg9r_launch<half_t, 1, 2, 2>(a, s, 384);
        case 41: return wg9r_launch<half_t, 1, 2, 2>(a, s, 256);
        case 42: return wg9r_launch<half_t, 2, 1, 2>(a, s, 384);
        case 43: return wg9r_launch<half_t, 2, 1, 2>(a, s, 256);
    }
    return hipErrorInvalidValue;
}
#endif  // Y2_DEVBUILD

hipError_t launch_wgrad_auto(int dtype, const WgradArgs& a0, hipStream_t s) {
    static const int xcd_mode = getenv("Y2_XCD_WGRAD") ? atoi(getenv("Y2_XCD_WGRAD")) : 1;
    WgradArgs a = a0;
    a.xcd = xcd_mode;
    // measured per shape (scripts/bench_wgrad.py): nine-tap blocks win on every 3x3 layer
    if (a.taps == 9) {
        hipError_t e = launch_wgrad9(dtype, a, s);
        if (e != hipErrorOutOfMemory) return e;   // window too large for LDS: fall through
        (void)hipGetLastError();
    }
    return launch_wgrad(dtype, a, s);
}
}  // namespace y2
