// Shared epilogue of the implicit-GEMM convolution kernels.
// Accumulators hold D[cout (4 consecutive registers)][pixel (lane)]; the epilogue
//   1. adds the bias, rounds to T and packs 4 couts per LDS store into a wave-private
//      [pixel][cout] patch (16-byte row pad: conflict-light ds_write_b64/b128),
//   2. re-reads whole 16-byte chunks per pixel row and stores full lines to y[M][ldy],
//   3. (training) reduces per-channel batch-norm partials of the values AS STORED:
//      per-wave two-pass (sum -> mean, squared deviations), Chan-merged over the WP
//      waves of the block -> one (count, mean, M2) record per block and channel.
#pragma once
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

// must be entered by ALL threads of the block, after a barrier that retires every read
// of the staging buffers (the patch aliases them)
template <typename T, int WP, int WC, int TP, int TC>
Y2_DEV void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[TC][TP], char* smem, int w, int lane, int m0, int n0,
                          int pt, int ct) {
    typedef EpiCfg<T, WP, WC, TP, TC> Cfg;
    constexpr int NW = Cfg::NW, SZ = Cfg::SZ, BP = Cfg::BP, EROW = Cfg::EROW;
    const int wp = w / WC, wc = w % WC;
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
    __syncthreads();

    constexpr int EPC = 16 / SZ;        // elements per chunk
    constexpr int CPR = TC * 32 / EPC;  // chunks per pixel row
    constexpr int RPIe = 64 / CPR;      // pixel rows per read instruction
    constexpr int NIT = TP * 32 / RPIe;
    const int ch = lane % CPR, prow0 = lane / CPR;
    const int mw0 = m0 + wp * TP * 32;  // first pixel of this wave
    const int cch = cw0 + ch * EPC;     // first cout of this lane's chunk
    float vals[NIT][EPC];
    float s[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s[e] = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int prow = it * RPIe + prow0;
        Chunk<T> c = ld_chunk<T>(ew + prow * EROW + ch * 16);
        const bool pv = (mw0 + prow) < a.M;
        if (pv && cch < a.ldy) st_chunk<T>((char*)a.y + ((size_t)(mw0 + prow) * a.ldy + cch) * SZ, c);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            vals[it][e] = pv ? Elem<T>::to_f32(c.v[e]) : 0.f;
            s[e] += vals[it][e];
        }
    }
    if (a.part_mean) {
        int cntw = a.M - mw0;
        cntw = cntw < 0 ? 0 : (cntw > TP * 32 ? TP * 32 : cntw);
        const float inv = cntw > 0 ? 1.0f / (float)cntw : 0.f;
        float mean[EPC], m2[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
#pragma unroll
            for (int msk = CPR; msk < 64; msk <<= 1) s[e] = wave_sum_xor(s[e], msk);
            mean[e] = s[e] * inv;
            m2[e] = 0.f;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const bool pv = (mw0 + it * RPIe + prow0) < a.M;
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float d = pv ? vals[it][e] - mean[e] : 0.f;
                m2[e] += d * d;
            }
        }
#pragma unroll
        for (int e = 0; e < EPC; ++e)
#pragma unroll
            for (int msk = CPR; msk < 64; msk <<= 1) m2[e] = wave_sum_xor(m2[e], msk);
        // combine the WP waves that share these channels (Chan's parallel update)
        float* st = (float*)(smem + NW * Cfg::EPW);
        if (prow0 == 0) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                st[(w * TC * 32 + ch * EPC + e) * 2 + 0] = mean[e];
                st[(w * TC * 32 + ch * EPC + e) * 2 + 1] = m2[e];
            }
        }
        __syncthreads();
        if (wp == 0 && prow0 == 0) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                float n_acc = 0.f, mean_acc = 0.f, m2_acc = 0.f;
                for (int k = 0; k < WP; ++k) {
                    int cntk = a.M - (m0 + k * TP * 32);
                    cntk = cntk < 0 ? 0 : (cntk > TP * 32 ? TP * 32 : cntk);
                    if (cntk == 0) continue;
                    const int wk = k * WC + wc;
                    const float mk = st[(wk * TC * 32 + ch * EPC + e) * 2 + 0];
                    const float vk = st[(wk * TC * 32 + ch * EPC + e) * 2 + 1];
                    const float nn = n_acc + (float)cntk;
                    const float dlt = mk - mean_acc;
                    mean_acc += dlt * ((float)cntk / nn);
                    m2_acc += vk + dlt * dlt * (n_acc * (float)cntk / nn);
                    n_acc = nn;
                }
                const int co = cch + e;
                if (co < a.ldy) {
                    a.part_mean[(size_t)pt * a.ldy + co] = mean_acc;
                    a.part_m2[(size_t)pt * a.ldy + co] = m2_acc;
                }
            }
        }
        if (threadIdx.x == 0 && ct == 0) {
            int cb = a.M - m0;
            a.part_cnt[pt] = (float)(cb > BP ? BP : cb);
        }
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
