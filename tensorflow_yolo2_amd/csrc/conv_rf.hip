// 3x3 stride-1 SAME convolution with the FILTERS RESIDENT IN REGISTERS and PERSISTENT workgroups (gfx950).
// Third form of the op of conv_halo.hip / conv_haloq.hip (tf.nn.conv2d(..., 'SAME') + bias for filter_size 3,
// reference src/yolo2_nets/darknet.py:20-21,32-36, and its dgrad), for the few-channel layers on the large
// feature maps (208x208 x 32 -> 64: K = 288 in all).
//
// There a 256-pixel tile carries only 72 MFMAs per wave: the per-tile prologue (filter fetch, first image
// fetch at full HBM latency) and the epilogue outweigh the matrix work, and the one-tile-per-workgroup kernels
// reach 0.38 PFLOP/s on a layer whose HBM floor is 2.3x shorter.  Here
//   * a wave keeps its whole filter slice (NCT x 32 couts x 9 taps x C channels = 144 VGPRs) for the life of the
//     workgroup, which walks a contiguous run of pixel tiles: no filter traffic and no prologue per tile;
//   * the tiles run LINEARLY over the bordered pixel space (common.h: pitch W + 1, shared zero borders), like the
//     K loop of wgrad9.hip: output q reads inputs q + (kh-1)*pitch + (kw-1), so the input of a run is ONE
//     contiguous stream, staged by LDS-DMA into a three-slot ring of BP-row groups, each input pixel once per
//     workgroup; border positions are computed and dropped (pitch / W - 1 = 0.5 % more work at 208);
//   * the group for the next tile is in flight while the epilogue of the current one runs, and two workgroups
//     per CU in different phases keep the matrix pipe busy through each other's epilogues.
// Epilogue = conv_epilogue.h's: bias, rounding, wave-private [pixel][cout] patch, full-line stores, batch-norm
// partials of the values as stored from the matrix pipe (S1 = ones x y, S2 = diag(y y^T)); here a record covers
// the valid pixels of one BP-position tile (border rows of the patch are written as zeros), and the
// bordered -> NHWC index of a patch row is carried incrementally (no division per tile).
#include <stdlib.h>
#include "common.h"
#include "conv_epilogue.h"
#include "kernels.h"

namespace y2 {

struct RfGeom {
    int pitch, rows_img;     // W + 1, H + 1
    int ntiles, tiles_per_block;
    int qmax;                // readable bordered rows (bbody_pixels)
};

template <typename T, int C, int NCT, int WP, int TP>
struct RfCfg {
    static constexpr int SZ = sizeof(T);
    static constexpr int NW = WP, NT = NW * 64;
    static constexpr int ROWB = C * SZ, LPR = ROWB / 16, RPI = 64 / LPR, RPB = 256 / ROWB;
    static constexpr int G = C * SZ / 32, KGT = 9 * G;       // 32-byte k-groups per tap / in all
    static constexpr int BP = WP * TP * 32, BC = NCT * 32;
    static constexpr int R = 3 * BP, RINGB = R * ROWB;
    static constexpr int EROW = BC * SZ + 16, PATCHB = 32 * EROW;
    static constexpr int SCR = (2 * BC + 4) * 4;             // per wave: S1[BC], S2[BC], count
    static constexpr int LDS = RINGB + NW * PATCHB + NW * SCR + BC * 4;   // + the bias slice
};

template <typename T, int C, int NCT, int WP, int TP>
__global__ __launch_bounds__(WP * 64, 2) void conv_rf_kernel(ConvArgs a, RfGeom gm) {
    typedef RfCfg<T, C, NCT, WP, TP> Cfg;
    typedef typename Elem<T>::frag frag_t;
    constexpr int SZ = Cfg::SZ, NW = Cfg::NW, ROWB = Cfg::ROWB, LPR = Cfg::LPR, RPI = Cfg::RPI, RPB = Cfg::RPB;
    constexpr int G = Cfg::G, KGT = Cfg::KGT, BP = Cfg::BP, BC = Cfg::BC, R = Cfg::R, EROW = Cfg::EROW;
    static_assert(SZ == 2, "16-bit element types");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ring = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* const patch = smem + Cfg::RINGB + w * Cfg::PATCHB;
    float* const scr_all = (float*)(smem + Cfg::RINGB + NW * Cfg::PATCHB);
    float* const scr = scr_all + w * (Cfg::SCR / 4);
    const int r32 = lane & 31, hh = lane >> 5;
    const int pitch = gm.pitch;

    const int T0 = blockIdx.x * gm.tiles_per_block;
    int T1 = T0 + gm.tiles_per_block;
    if (T1 > gm.ntiles) T1 = gm.ntiles;
    if (T0 >= T1) return;

    // ---- the wave's filter slice, once: A operand rows = couts, 16 B = 8 consecutive k per lane
    const int n0 = blockIdx.y * BC;
    frag_t wreg[NCT][KGT];
#pragma unroll
    for (int i = 0; i < NCT; ++i) {
        const char* wr = (const char*)a.w + (size_t)(n0 + i * 32 + r32) * 9 * ROWB + hh * 16;
#pragma unroll
        for (int s = 0; s < KGT; ++s) wreg[i][s] = *(const frag_t*)(wr + s * 32);
    }
    float* const biasl = scr_all + NW * (Cfg::SCR / 4);   // registers are for the filters: the bias slice waits in LDS
    if (tid < BC) biasl[tid] = (a.bias && n0 + tid < a.Cout) ? a.bias[n0 + tid] : 0.f;

    // ---- LDS-DMA of one BP-row group into its ring slot (rows below 0 / past the tensor only feed dropped outputs)
    const int lrow = lane / LPR, lslot = lane % LPR;
    auto stage = [&](int gi) {
        if (gi < 0) return;
        const int slot = gi % 3;
        const char* xs = (const char*)a.x + (size_t)gi * BP * ROWB;
        char* dst = ring + slot * BP * ROWB;
        for (int i = w; i < BP / RPI; i += NW) {
            if (gi * BP + i * RPI >= gm.qmax) break;
            const int row = i * RPI + lrow;
            const int rr = slot * BP + row;
            glds16(xs + (size_t)row * ROWB + ((lslot ^ ((rr / RPB) % LPR)) * 16), dst + i * 1024);
        }
    };

    // ---- per lane and pixel sub-tile: ring row of the pixel, and its (col, row-in-image, image) for the epilogue
    int qm[TP], pcol[TP], prow_[TP], pimg[TP];
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const int q = T0 * BP + (w * TP + j) * 32 + r32;
        qm[j] = q % R;
        const int rowi = q / pitch;
        pcol[j] = q - rowi * pitch;
        pimg[j] = rowi / gm.rows_img;
        prow_[j] = rowi - pimg[j] * gm.rows_img;
    }
    int sh[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) sh[t] = (t / 3 - 1) * pitch + (t % 3 - 1);

    const bool stats = a.part_mean != nullptr;
    const bool chk = a.nonfinite != nullptr;
    const int qq = (lane & 15) >> 2, pp = lane & 3, g1 = (lane >> 4) & 1;   // transposed patch reads (statistics)

    stage(T0 - 1);
    stage(T0);
    stage(T0 + 1);
    for (int tile = T0; tile < T1; ++tile) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // the record of the previous tile, from the scratch its epilogue filled before this barrier
        if (stats && tile > T0 && tid < BC) {
            double S1 = 0.0, S2 = 0.0;
            float cnt = 0.f;
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const float* q = scr_all + k * (Cfg::SCR / 4);
                S1 += (double)q[tid];
                S2 += (double)q[BC + tid];
                cnt += q[2 * BC];
            }
            const int co = n0 + tid;
            if (co < a.ldy) {
                const double md = cnt > 0.f ? S1 / (double)cnt : 0.0;
                const double m2 = S2 - S1 * md;
                a.part_mean[(size_t)(tile - 1) * a.ldy + co] = (float)md;
                a.part_m2[(size_t)(tile - 1) * a.ldy + co] = (float)(m2 > 0.0 ? m2 : 0.0);
            }
            if (tid == 0 && blockIdx.y == 0) a.part_cnt[tile - 1] = cnt;
        }

        f32x16 acc[NCT][TP];
#pragma unroll
        for (int i = 0; i < NCT; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

        frag_t fa[2][TP];
        auto load_frags = [&](int s, frag_t (&f)[TP]) {
            const int t = s / G, g = s % G;
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                int rr = qm[j] + sh[t];
                rr = rr < 0 ? rr + R : rr;
                rr = rr >= R ? rr - R : rr;
                f[j] = *(const frag_t*)(ring + rr * ROWB + (((2 * g + hh) ^ ((rr / RPB) % LPR)) * 16));
            }
        };
        load_frags(0, fa[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KGT; ++s) {
            if (s + 1 < KGT) load_frags(s + 1, fa[(s + 1) & 1]);
#pragma unroll
            for (int i = 0; i < NCT; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) mma32(acc[i][j], wreg[i][s], fa[s & 1][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_barrier();          // every wave is done with the oldest group: its slot takes the group of tile + 2
        asm volatile("" ::: "memory");
        if (tile + 1 < T1) stage(tile + 2);

        // ---- epilogue, one 32-pixel sub-tile at a time through the wave's own patch
        float S1w[NCT], S2w[NCT];
#pragma unroll
        for (int i = 0; i < NCT; ++i) S1w[i] = S2w[i] = 0.f;
        int cntw = 0;
        bool bad = false;
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const bool valid = pcol[j] >= 1 && prow_[j] >= 1 && pimg[j] < a.N;
            const int p = valid ? (pimg[j] * a.H + prow_[j] - 1) * a.W + pcol[j] - 1 : -1;
#pragma unroll
            for (int i = 0; i < NCT; ++i)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    T o[4];
                    const f32x4 b4 = *(const f32x4*)(biasl + i * 32 + 8 * q4 + 4 * hh);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        o[k] = valid ? Elem<T>::from_f32(acc[i][j][4 * q4 + k] + b4[k]) : (T)0.f;
                    *(u32x2*)(patch + r32 * EROW + (i * 32 + 8 * q4 + 4 * hh) * SZ) = *(const u32x2*)o;
                }
            cntw += __popcll(__ballot(valid && hh == 0));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            constexpr int EPC = 8, CPR = BC / EPC, RPIe = 64 / CPR, NIT = 32 / RPIe;
            const int ch = lane % CPR, prow0 = lane / CPR;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int prow = it * RPIe + prow0;
                const int pr = __shfl(p, prow, 64);
                Chunk<T> c = ld_chunk<T>(patch + prow * EROW + ch * 16);
                const int cch = n0 + ch * EPC;
                if (pr >= 0 && cch < a.ldy) {
                    st_chunk<T>((char*)a.y + ((size_t)pr * a.ldy + cch) * SZ, c);
                    if (chk) {
#pragma unroll
                        for (int e = 0; e < EPC; ++e)
                            bad |= (__float_as_uint(Elem<T>::to_f32(c.v[e])) & 0x7F800000u) == 0x7F800000u;
                    }
                }
            }
            if (stats) {
                frag_t ones;
#pragma unroll
                for (int k = 0; k < 8; ++k) ones[k] = (T)1.0f;
#pragma unroll
                for (int i = 0; i < NCT; ++i) {
                    f32x16 q1, q2;
#pragma unroll
                    for (int q = 0; q < 16; ++q) q1[q] = q2[q] = 0.f;
                    const char* pb = patch + (8 * hh + qq) * EROW + (i * 32 + 16 * g1 + 4 * pp) * 2;
#pragma unroll
                    for (int kg = 0; kg < 2; ++kg) {
                        const char* p0 = pb + kg * 16 * EROW;
                        const frag_t f = tr_frag<T>(p0, p0 + 4 * EROW);
                        mma32(q1, ones, f);
                        mma32(q2, f, f);
                    }
                    float dg = 0.f;
#pragma unroll
                    for (int q = 0; q < 16; ++q) dg += (acc_row(q, hh) == r32) ? q2[q] : 0.f;
                    S1w[i] += q1[0];
                    S2w[i] += dg + __shfl_xor(dg, 32, 64);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the patch is rewritten by the next sub-tile
        }
        if (chk && __any(bad) && lane == 0) atomicOr(a.nonfinite, 1u);
        if (stats) {
            if (hh == 0) {
#pragma unroll
                for (int i = 0; i < NCT; ++i) {
                    scr[i * 32 + r32] = S1w[i];
                    scr[BC + i * 32 + r32] = S2w[i];
                }
            }
            if (lane == 0) scr[2 * BC] = (float)cntw;
        }
        // ---- next tile: BP positions on
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            qm[j] += BP;
            qm[j] = qm[j] >= R ? qm[j] - R : qm[j];
            pcol[j] += BP;
            while (pcol[j] >= pitch) { pcol[j] -= pitch; ++prow_[j]; }
            while (prow_[j] >= gm.rows_img) { prow_[j] -= gm.rows_img; ++pimg[j]; }
        }
    }
    if (stats) {
        __syncthreads();
        if (tid < BC) {
            double S1 = 0.0, S2 = 0.0;
            float cnt = 0.f;
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const float* q = scr_all + k * (Cfg::SCR / 4);
                S1 += (double)q[tid];
                S2 += (double)q[BC + tid];
                cnt += q[2 * BC];
            }
            const int co = n0 + tid;
            if (co < a.ldy) {
                const double md = cnt > 0.f ? S1 / (double)cnt : 0.0;
                const double m2 = S2 - S1 * md;
                a.part_mean[(size_t)(T1 - 1) * a.ldy + co] = (float)md;
                a.part_m2[(size_t)(T1 - 1) * a.ldy + co] = (float)(m2 > 0.0 ? m2 : 0.0);
            }
            if (tid == 0 && blockIdx.y == 0) a.part_cnt[T1 - 1] = cnt;
        }
    }
}

template <typename T, int C, int NCT, int WP, int TP>
static hipError_t rf_launch(const ConvArgs& a, hipStream_t s, int* bp, int* records) {
    typedef RfCfg<T, C, NCT, WP, TP> Cfg;
    RfGeom g{};
    g.pitch = a.W + 1;
    g.rows_img = a.H + 1;
    const long qtot = (long)a.N * g.rows_img * g.pitch;
    g.ntiles = (int)((qtot + Cfg::BP - 1) / Cfg::BP);
    g.qmax = (int)bbody_pixels(a.N, a.H, a.W);
    const int nct = (a.Cout + Cfg::BC - 1) / Cfg::BC;
    int nblk = (Cfg::LDS <= 80 * 1024 ? 512 : 256) / nct;    // two workgroups per CU where the LDS allows
    if (nblk > g.ntiles) nblk = g.ntiles;
    g.tiles_per_block = (g.ntiles + nblk - 1) / nblk;
    nblk = (g.ntiles + g.tiles_per_block - 1) / g.tiles_per_block;
    auto kern = conv_rf_kernel<T, C, NCT, WP, TP>;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
        if (e != hipSuccess) return e;
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3(nblk, nct), dim3(Cfg::NT), Cfg::LDS, s, a, g);
    if (bp) *bp = Cfg::BP;
    if (records) *records = g.ntiles;
    return hipGetLastError();
}

// 0: not this form (16-bit launches only: row_bytes = input channels * 2).
//   1: 32 -> 64 on 208-wide maps (forward of the second layer)      4 waves x 64 pixels x 64 couts, two workgroups per CU
//   2: 64 -> 32 on 208-wide maps (its dgrad)                        8 waves x 32 pixels x 32 couts, one workgroup per CU
int conv_rf_config(int taps, int W, int row_bytes, int Cout, int M) {
    static const bool off = getenv("Y2_NO_CONV_RF") != nullptr;
    if (off || taps != 9 || W <= 104 || W + 2 >= 256 || M < 256 * 1024) return 0;
    if (row_bytes == 64 && Cout > 32 && Cout <= 64) return 1;
    if (row_bytes == 128 && Cout <= 32) return 2;
    return 0;
}

template <typename T>
static hipError_t rf_T(int cfg, const ConvArgs& a, hipStream_t s, int* bp, int* records) {
    if (cfg == 1) return rf_launch<T, 32, 2, 4, 2>(a, s, bp, records);
    if (cfg == 2) return rf_launch<T, 64, 1, 8, 1>(a, s, bp, records);
    return hipErrorInvalidValue;
}
hipError_t launch_conv_rf(int dtype, const ConvArgs& a, hipStream_t s, int* bp, int* records) {
    const int cfg = conv_rf_config(a.taps, a.W, a.C * (int)dtype_size(dtype), a.Cout, a.M);
    if (dtype == 1) return rf_T<half_t>(cfg, a, s, bp, records);
    if (dtype == 2) return rf_T<bf16_t>(cfg, a, s, bp, records);
    return hipErrorInvalidValue;
}

}  // namespace y2
