// Shared epilogue of the implicit-GEMM convolution kernels.
// Accumulators hold D[cout (4 consecutive registers)][pixel (lane)]; the epilogue
//   1. adds the bias, rounds to T and packs 4 couts per LDS store into a wave-private
//      [pixel][cout] patch (16-byte row pad: conflict-light ds_write_b64/b128),
//   2. re-reads whole 16-byte chunks per pixel row and stores full lines to y[M][ldy],
//   3. (training) reduces per-channel batch-norm partials of the values AS STORED: per lane
//      sums of (y - bias) and (y - bias)^2 over its rows, added over the block through LDS
//      -> one (count, mean, M2) record per block and channel.
#pragma once
#include <type_traits>
#include "common.h"
#include "kernels.h"

namespace y2 {

template <typename T, int WP, int WC, int TP, int TC>
struct EpiCfg {
    static constexpr int NW = WP * WC;
    static constexpr int SZ = sizeof(T);
    static constexpr int BP = WP * TP * 32, BC = WC * TC * 32;
    static constexpr int EROW = TC * 32 * SZ + 16;
    static constexpr int EPW = TP * 32 * EROW;
    static constexpr int ESTAT = NW * TC * 32 * 2 * 4;
    static constexpr int LDS = NW * EPW + ESTAT;
};

// phases 2 and 3 (after the wave's patch has been written)
// cstride / coff (conv_epilogue16 callers that run the epilogue in TWO passes over halves of a wave's couts, because the
// fp32 patch of the whole tile does not fit LDS): the couts of wave column wc start at n0 + wc * cstride + coff
// BWF32 (16-bit T only; f16x2f dgrad launches, common.h hsplithh_t): ConvArgs::bw_y is fp32 [M][ldy]
template <typename T, int WP, int WC, int TP, int TC, int EABL = 0, bool BWF32 = false>
Y2_DEV void conv_epilogue_finish(const ConvArgs& a, char* smem, int w, int lane, int m0, int n0, int pt, int ct,
                                 int cstride = TC * 32, int coff = 0);

// must be entered by ALL threads of the block, after a barrier that retires every read
// of the staging buffers (the patch aliases them)
template <typename T, int WP, int WC, int TP, int TC, int EABL = 0, bool BWF32 = false>
Y2_DEV void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[TC][TP], char* smem, int w, int lane, int m0, int n0,
                          int pt, int ct) {
    typedef EpiCfg<T, WP, WC, TP, TC> Cfg;
    constexpr int SZ = Cfg::SZ, EROW = Cfg::EROW;
    const int wc = w % WC;
    const int r32 = lane & 31, hh = lane >> 5;
    char* ew = smem + w * Cfg::EPW;
    const int cw0 = n0 + wc * TC * 32;  // first cout of this wave
#pragma unroll
    for (int i = 0; i < TC; ++i) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int cl = i * 32 + 8 * q4 + 4 * hh;  // local cout of register 4*q4
            float b4[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.bias) {
#pragma unroll
                for (int j = 0; j < 4; ++j) b4[j] = (cw0 + cl + j < a.Cout) ? a.bias[cw0 + cl + j] : 0.f;
            }
#pragma unroll
            for (int j2 = 0; j2 < TP; ++j2) {
                T o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = Elem<T>::from_f32(acc[i][j2][4 * q4 + j] + b4[j]);
                char* dst = ew + (j2 * 32 + r32) * EROW + cl * SZ;
                if (SZ == 2) *(u32x2*)dst = *(const u32x2*)o;
                else *(u32x4*)dst = *(const u32x4*)o;
            }
        }
    }
    conv_epilogue_finish<T, WP, WC, TP, TC, EABL, BWF32>(a, smem, w, lane, m0, n0, pt, ct);
}

// same, accumulators of 16x16 MFMA tiles: acc[i16][j16][r] = D[cout i16*16 + 4*(lane>>4) + r][pixel j16*16 + (lane&15)]
// (TP, TC still count 32-wide units: the wave tile and the patch are the same as above)
// 16x16 tiles over the compact halo image (conv_haloq.hip): MFMA column c of a 16-pixel fragment holds pixel offset
// perm16(c) -- lanes {0-3, 12-15} the even pixels, lanes {4-11} the odd ones
Y2_DEV int perm16(int c) { return c < 4 ? 2 * c : (c >= 12 ? 2 * (c - 8) : 2 * (c - 4) + 1); }

// PERM: the accumulator columns are dealt by perm16 (the patch is written in pixel order either way)
template <typename T, int WP, int WC, int TP, int TC, bool PERM = false, bool BWF32 = false>
Y2_DEV void conv_epilogue16(const ConvArgs& a, f32x4 (&acc)[2 * TC][2 * TP], char* smem, int w, int lane, int m0,
                            int n0, int pt, int ct, int cstride = TC * 32, int coff = 0) {
    typedef EpiCfg<T, WP, WC, TP, TC> Cfg;
    constexpr int SZ = Cfg::SZ, EROW = Cfg::EROW;
    const int wc = w % WC;
    const int r16 = PERM ? perm16(lane & 15) : (lane & 15), g4 = lane >> 4;
    char* ew = smem + w * Cfg::EPW;
    const int cw0 = n0 + wc * cstride + coff;
#pragma unroll
    for (int i = 0; i < 2 * TC; ++i) {
        const int cl = i * 16 + 4 * g4;
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
#pragma unroll
            for (int j = 0; j < 4; ++j) b4[j] = (cw0 + cl + j < a.Cout) ? a.bias[cw0 + cl + j] : 0.f;
        }
#pragma unroll
        for (int j2 = 0; j2 < 2 * TP; ++j2) {
            T o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = Elem<T>::from_f32(acc[i][j2][j] + b4[j]);
            char* dst = ew + (j2 * 16 + r16) * EROW + cl * SZ;
            if (SZ == 2) *(u32x2*)dst = *(const u32x2*)o;
            else *(u32x4*)dst = *(const u32x4*)o;
        }
    }
    conv_epilogue_finish<T, WP, WC, TP, TC, 0, BWF32>(a, smem, w, lane, m0, n0, pt, ct, cstride, coff);
}

template <typename T, int WP, int WC, int TP, int TC, int EABL, bool BWF32>
Y2_DEV void conv_epilogue_finish(const ConvArgs& a, char* smem, int w, int lane, int m0, int n0, int pt, int ct,
                                 int cstride, int coff) {
    static_assert(!BWF32 || sizeof(T) == 2, "BWF32: a 16-bit output beside an fp32 conv output of the layer below");
    typedef EpiCfg<T, WP, WC, TP, TC> Cfg;
    constexpr int NW = Cfg::NW, SZ = Cfg::SZ, BP = Cfg::BP, EROW = Cfg::EROW;
    const int wp = w / WC, wc = w % WC;
    char* ew = smem + w * Cfg::EPW;
    const int cw0 = n0 + wc * cstride + coff;  // first cout of this wave
    if (EABL & 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else __syncthreads();

    constexpr int EPC = 16 / SZ;        // elements per chunk
    constexpr int CPR = TC * 32 / EPC;  // chunks per pixel row
    constexpr int RPIe = 64 / CPR;      // pixel rows per read instruction
    constexpr int NIT = TP * 32 / RPIe;
    static_assert(Cfg::EPW >= 2 * 64 * EPC * 4, "statistics scratch must fit the wave's patch");
    const int ch = lane % CPR, prow0 = lane / CPR;
    const int mw0 = m0 + wp * TP * 32;  // first pixel of this wave
    const int cch = cw0 + ch * EPC;     // first cout of this lane's chunk
    const bool stats = a.part_mean != nullptr;
    const bool bw = a.bw_psum != nullptr;    // dgrad launch: BN-backward reduce of the layer below (kernels.h)
    // Statistics of the values AS STORED, as sums of d = y - bias and of d^2: one pass, no value
    // buffer; the bias is a free pivot (it removes the offset the filter response rides on).
    float piv[EPC], s1[EPC], s2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        piv[e] = (stats && a.bias && cch + e < a.Cout) ? a.bias[cch + e] : 0.f;
        s1[e] = 0.f;
        s2[e] = 0.f;
    }
    // 16-bit types, whole tiles: the batch-norm sums come from the MATRIX pipe (idle in the epilogue) instead of
    // ~4 VALU operations per stored element.  The wave's patch [pixel][cout] is read back transposed
    // (ds_read_b64_tr_b16: lane = cout, 8 pixels per fragment); S1 = ones x y and S2 = diag(y y^T) on 32x32x16 MFMAs:
    // exact products of the stored values, fp32 accumulation; the block's (mean, M2) record is formed in double.
    bool mfma_stats = false;
    if constexpr (SZ == 2) mfma_stats = stats && !bw && (m0 + BP <= a.M);   // block-uniform
    if (!bw && a.aff_out) {
        // ---- inference batch norm folded in (kernels.h ConvArgs::aff_*): out = leaky(T(conv + b) * scale + shift)
        // into the consumer's bordered tensor.  Bordered position of pixel m = r*W + w, r = n*H + h:
        //   bpix = (r + n + 1) * (W + 1) + w + 1 = m + r + (n + 1) * (W + 1) + 1
        const bool cv = cch < a.ldy;
        float sc[EPC], sh[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            sc[e] = cv ? a.aff_scale[cch + e] : 0.f;
            sh[e] = cv ? a.aff_shift[cch + e] : 0.f;
        }
        if (a.aff_pool) {
            // pooled layer (ConvArgs::aff_pool): the patch rows are in window-major order, four consecutive rows = one
            // 2x2 window; the maximum on z = y * scale + shift, then the activation (bn.hip pool_window, bit for bit)
            constexpr int NWIN = TP * 32 / 4;              // windows of this wave
            const int Ho = a.H >> 1, Wo = a.W >> 1;
#pragma unroll
            for (int it = 0; it * RPIe < NWIN; ++it) {
                const int wrow = it * RPIe + prow0;
                const uint32_t u = (uint32_t)(mw0 >> 2) + (uint32_t)wrow;          // window index (n, ho, wo)
                if (wrow < NWIN && (int)(4 * u) < a.M && cv) {
                    const char* pr = ew + 4 * wrow * EROW + ch * 16;
                    const Chunk<T> v0 = ld_chunk<T>(pr), v1 = ld_chunk<T>(pr + EROW), v2 = ld_chunk<T>(pr + 2 * EROW),
                                   v3 = ld_chunk<T>(pr + 3 * EROW);
                    const uint32_t t = u / (uint32_t)Wo, wo = u - t * (uint32_t)Wo;
                    const uint32_t n = t / (uint32_t)Ho, ho = t - n * (uint32_t)Ho;
                    Chunk<T> o;
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        const float z0 = Elem<T>::to_f32(v0.v[e]) * sc[e] + sh[e], z1 = Elem<T>::to_f32(v1.v[e]) * sc[e] + sh[e];
                        const float z2 = Elem<T>::to_f32(v2.v[e]) * sc[e] + sh[e], z3 = Elem<T>::to_f32(v3.v[e]) * sc[e] + sh[e];
                        const float zm = fmaxf(fmaxf(fmaxf(-INFINITY, z0), z1), fmaxf(z2, z3));
                        o.v[e] = Elem<T>::from_f32(leaky_s(zm, a.aff_slope));
                    }
                    st_chunk<T>((char*)a.aff_out + (bpix((int)n, (int)ho, (int)wo, Ho, Wo) * a.ldy + cch) * SZ, o);
                }
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int prow = it * RPIe + prow0;
            Chunk<T> c = ld_chunk<T>(ew + prow * EROW + ch * 16);
            const uint32_t m = (uint32_t)(mw0 + prow);
            if ((int)m < a.M && cv) {
                const uint32_t r = (uint32_t)(((uint64_t)m * a.aff_magW) >> a.aff_shW);
                const uint32_t n = (uint32_t)(((uint64_t)r * a.aff_magH) >> a.aff_shH);
                const size_t bp = (size_t)m + r + (size_t)(n + 1) * (uint32_t)(a.W + 1) + 1;
                Chunk<T> o;
#pragma unroll
                for (int e = 0; e < EPC; ++e) o.v[e] = Elem<T>::from_f32(leaky_s(Elem<T>::to_f32(c.v[e]) * sc[e] + sh[e], a.aff_slope));
                st_chunk<T>((char*)a.aff_out + (bp * a.ldy + cch) * SZ, o);
            }
        }
        return;
    }
    if (!bw) {
        const bool chk = a.nonfinite != nullptr;
        bool bad = false;
        auto sweep = [&](auto full_tag, auto stat_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            constexpr bool VSTAT = decltype(stat_tag)::value;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int prow = it * RPIe + prow0;
                Chunk<T> c = ld_chunk<T>(ew + prow * EROW + ch * 16);
                const bool pv = FULL || (mw0 + prow) < a.M;
                if (!(EABL & 1) && pv && cch < a.ldy) st_chunk<T>((char*)a.y + ((size_t)(mw0 + prow) * a.ldy + cch) * SZ, c);
                if (chk && pv) {
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        const float f = Elem<T>::to_f32(c.v[e]);
                        bad |= (__float_as_uint(f) & 0x7F800000u) == 0x7F800000u;
                    }
                }
                if (VSTAT && stats) {
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        float d = Elem<T>::to_f32(c.v[e]) - piv[e];
                        if (!FULL) d = pv ? d : 0.f;
                        s1[e] += d;
                        s2[e] = fmaf(d, d, s2[e]);
                    }
                }
            }
        };
        if (mfma_stats) sweep(std::true_type{}, std::false_type{});
        else if (mw0 + TP * 32 <= a.M) sweep(std::true_type{}, std::true_type{});   // wave-uniform
        else sweep(std::false_type{}, std::true_type{});
        if (chk && __any(bad) && lane == 0) atomicOr(a.nonfinite, 1u);
    } else {
        // ---- dgrad: store dA and reduce S1 = sum g, S2 = sum g * y_sel of the layer below on the fly
        const bool cv = cch < a.ldy;
        float sc[EPC], sh[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            sc[e] = cv ? a.bw_scale[cch + e] : 0.f;
            sh[e] = cv ? a.bw_shift[cch + e] : 0.f;
        }
        constexpr int YSZ = BWF32 ? 4 : SZ;          // element size of the conv output the reduce reads
        const char* yb = (const char*)a.bw_y + (size_t)cch * YSZ;
        const size_t rowB = (size_t)a.ldy * YSZ;
        // every y chunk of the sweep is requested before the first one is used (the accumulators are dead: the
        // registers are free, and one HBM latency is paid per tile instead of one per row group)
        struct YChunk { typename std::conditional<BWF32, float, T>::type v[EPC]; };
        YChunk yv[NIT];
        if (cv) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                int p = mw0 + it * RPIe + prow0;
                if (p > a.M - 1) p = a.M - 1;
#pragma unroll
                for (int k = 0; k < (int)sizeof(YChunk) / 16; ++k)
                    *((u32x4*)yv[it].v + k) = *((const u32x4*)(yb + (size_t)p * rowB) + k);
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int prow = it * RPIe + prow0;
            Chunk<T> c = ld_chunk<T>(ew + prow * EROW + ch * 16);
            const bool pv = (mw0 + prow) < a.M;
            if (pv && cv) {
                st_chunk<T>((char*)a.y + ((size_t)(mw0 + prow) * a.ldy + cch) * SZ, c);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float yf = (float)yv[it].v[e];
                    const float g = Elem<T>::to_f32(c.v[e]) * leaky_slope_s(fmaf(yf, sc[e], sh[e]), a.bw_slope);
                    s1[e] += g;
                    s2[e] = fmaf(g, yf, s2[e]);
                }
            }
        }
    }
    if constexpr (SZ == 2) {
        // Block-uniform choice: every wave of the block must take the same reduction path (tail blocks: the VALU one)
        const bool blk_mfma = stats && !bw && (m0 + BP <= a.M);
        if (blk_mfma) {
            typedef typename Elem<T>::frag frag_t;
            const int r32 = lane & 31, hh = lane >> 5;
            const int qq = (lane & 15) >> 2, pp = lane & 3, g1 = (lane >> 4) & 1;
            frag_t ones;
#pragma unroll
            for (int j = 0; j < 8; ++j) ones[j] = (T)1.0f;
            float S1w[TC], S2w[TC];
#pragma unroll
            for (int i = 0; i < TC; ++i) {
                f32x16 q1, q2;
#pragma unroll
                for (int q = 0; q < 16; ++q) q1[q] = q2[q] = 0.f;
                const char* pb = ew + (8 * hh + qq) * EROW + (i * 32 + 16 * g1 + 4 * pp) * 2;
#pragma unroll
                for (int kg = 0; kg < TP * 2; ++kg) {
                    const char* p0 = pb + kg * 16 * EROW;
                    const frag_t f = tr_frag<T>(p0, p0 + 4 * EROW);
                    mma32(q1, ones, f);
                    mma32(q2, f, f);
                }
                float dg = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) dg += (acc_row(q, hh) == r32) ? q2[q] : 0.f;
                S1w[i] = q1[0];
                S2w[i] = dg + __shfl_xor(dg, 32, 64);
            }
            __syncthreads();     // every wave is done reading its patch: the scratch below aliases it
            float* sw = (float*)ew;   // [2][TC*32]
            if (hh == 0) {
#pragma unroll
                for (int i = 0; i < TC; ++i) {
                    sw[i * 32 + r32] = S1w[i];
                    sw[TC * 32 + i * 32 + r32] = S2w[i];
                }
            }
            __syncthreads();
            for (int c = threadIdx.x; c < Cfg::BC; c += NW * 64) {
                const int wcs = c / (TC * 32), cl = c % (TC * 32);
                double S1 = 0.0, S2 = 0.0;
#pragma unroll
                for (int k = 0; k < WP; ++k) {
                    const float* q = (const float*)(smem + (k * WC + wcs) * Cfg::EPW);
                    S1 += (double)q[cl];
                    S2 += (double)q[TC * 32 + cl];
                }
                const int co = n0 + wcs * cstride + coff + cl;
                if (co < a.ldy) {
                    const double md = S1 / (double)BP;
                    const double m2 = S2 - S1 * md;
                    a.part_mean[(size_t)pt * a.ldy + co] = (float)md;
                    a.part_m2[(size_t)pt * a.ldy + co] = (float)(m2 > 0.0 ? m2 : 0.0);
                }
            }
            if (threadIdx.x == 0 && ct == 0) a.part_cnt[pt] = (float)BP;
            return;
        }
    }
    if (stats || bw) {
        // lane partials -> the wave's own patch (dead now: a wave's LDS operations retire in order),
        // then one thread per block channel adds the RPIe row groups of the WP waves
        float* sw = (float*)ew;   // [2][64][EPC]
#pragma unroll
        for (int e = 0; e < EPC; e += 4) {
            *(f32x4*)(sw + lane * EPC + e) = f32x4{s1[e], s1[e + 1], s1[e + 2], s1[e + 3]};
            *(f32x4*)(sw + 64 * EPC + lane * EPC + e) = f32x4{s2[e], s2[e + 1], s2[e + 2], s2[e + 3]};
        }
        __syncthreads();
        int cb = a.M - m0;
        cb = cb > BP ? BP : cb;
        const float inv = 1.0f / (float)cb;
        for (int c = threadIdx.x; c < Cfg::BC; c += NW * 64) {
            const int wcs = c / (TC * 32), cl = c % (TC * 32);
            const int idx = (cl / EPC) * EPC + (cl % EPC);   // = cl: [chunk][element]
            float S1 = 0.f, S2 = 0.f;
#pragma unroll
            for (int k = 0; k < WP; ++k) {
                const float* q = (const float*)(smem + (k * WC + wcs) * Cfg::EPW);
#pragma unroll
                for (int g = 0; g < RPIe; ++g) {
                    S1 += q[g * CPR * EPC + idx];
                    S2 += q[64 * EPC + g * CPR * EPC + idx];
                }
            }
            const int co = n0 + wcs * cstride + coff + cl;
            if (co < a.ldy) {
                if (bw) {
                    a.bw_psum[((size_t)pt * 2 + 0) * a.ldy + co] = S1;
                    a.bw_psum[((size_t)pt * 2 + 1) * a.ldy + co] = S2;
                } else {
                    const float md = S1 * inv;
                    const float pb = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
                    a.part_mean[(size_t)pt * a.ldy + co] = pb + md;
                    a.part_m2[(size_t)pt * a.ldy + co] = fmaxf(S2 - S1 * md, 0.f);
                }
            }
        }
        if (stats && threadIdx.x == 0 && ct == 0) a.part_cnt[pt] = (float)cb;
    }
}

template <int N>
Y2_DEV void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// run-time count (wave-uniform): the immediate must be a literal, so dispatch over 0..31
// (larger counts wait for 31: conservative)
Y2_DEV void wait_vmcnt_dyn(int n) {
    switch (n) {
#define Y2_W(k) case k: wait_vmcnt<k>(); break;
        Y2_W(0) Y2_W(1) Y2_W(2) Y2_W(3) Y2_W(4) Y2_W(5) Y2_W(6) Y2_W(7) Y2_W(8) Y2_W(9) Y2_W(10) Y2_W(11)
        Y2_W(12) Y2_W(13) Y2_W(14) Y2_W(15) Y2_W(16) Y2_W(17) Y2_W(18) Y2_W(19) Y2_W(20) Y2_W(21) Y2_W(22)
        Y2_W(23) Y2_W(24) Y2_W(25) Y2_W(26) Y2_W(27) Y2_W(28) Y2_W(29) Y2_W(30)
#undef Y2_W
        default: wait_vmcnt<31>(); break;
    }
}

}  // namespace y2
