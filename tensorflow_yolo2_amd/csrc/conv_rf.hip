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
//     contiguous stream, staged by LDS-DMA into a four-slot ring of BP-row groups, each input pixel once per
//     workgroup; border positions are computed and dropped (pitch / W - 1 = 0.5 % more work at 208);
//   * the group two tiles ahead is in flight through a whole tile (counted vmcnt: LDS-DMA pieces and the epilogue's
//     stores retire in issue order, so every lane stores every sweep -- masked rows to the slack row behind y --
//     and every stage issues its full piece count), and two workgroups per CU in different phases keep the matrix
//     pipe busy through each other's epilogues.
// Epilogue = conv_epilogue.h's: bias, rounding, wave-private [pixel][cout] patch, full-line stores, batch-norm
// partials of the values as stored from the matrix pipe (S1 = ones x y, S2 = diag(y y^T)); here a record covers
// the valid pixels of one BP-position tile (border rows of the patch are written as zeros), and the
// bordered -> NHWC index of a patch row is carried incrementally (no division per tile).
#include <stdlib.h>
#include "common.h"
#include "conv_epilogue.h"
#include "kernels.h"

namespace y2 {

#ifdef Y2_DEVBUILD
// diagnostic build only: s_memtime deltas per phase of the 128-cout kernel (workgroup 0; [wave][phase])
__device__ unsigned long long g_rf_stamps[8][8];
#define RF_STAMP(k)                                                                                  \
    do {                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        unsigned long long t_;                                                                       \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                   \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        ph[k] += t_ - tlast;                                                                         \
        tlast = t_;                                                                                  \
    } while (0)
#else
#define RF_STAMP(k)
#endif

struct RfGeom {
    int pitch, rows_img;     // W + 1, H + 1
    int ntiles, tiles_per_block;
    int qmax;                // readable bordered rows (bbody_pixels)
};

// PR = pixel rows of the wave's epilogue patch (16 where two workgroups must share a CU's LDS, else 32)
template <typename T, int C, int NCT, int WP, int TP, int PR>
struct RfCfg {
    static constexpr int SZ = sizeof(T);
    static constexpr int NW = WP, NT = NW * 64;
    static constexpr int ROWB = C * SZ, LPR = ROWB / 16, RPI = 64 / LPR, RPB = 256 / ROWB;
    static constexpr int G = C * SZ / 32, KGT = 9 * G;       // 32-byte k-groups per tap / in all
    static constexpr int BP = WP * TP * 32, BC = NCT * 32;
    static constexpr int NSLOT = 4, R = NSLOT * BP, RINGB = R * ROWB;
    static constexpr int EROW = BC * SZ + 16, PATCHB = PR * EROW;
    static constexpr int SCR = (2 * BC + 2) * 8;             // per wave: S1[BC], S2[BC], count (doubles)
    static constexpr int LDS = RINGB + NW * PATCHB + NW * SCR + BC * 4;   // + the bias slice
    static constexpr int PW = BP / RPI / NW;                 // LDS-DMA pieces per wave and group
    static constexpr int CPR = BC / 8, NIT = PR * CPR / 64;  // 16-byte chunks per patch row; store sweeps per patch
    static constexpr int NST = TP * (32 / PR) * NIT;         // store instructions per wave and tile
    static_assert((R & (R - 1)) == 0, "ring rows: power of two");
    static_assert(BP % (RPI * NW) == 0 && (PR * CPR) % 64 == 0, "even split of the pieces / patch chunks over lanes");
    static_assert(PW + NST < 63, "counted vmcnt");
};

template <typename T, int C, int NCT, int WP, int TP, int PR, int PD>
__global__ __launch_bounds__(WP * 64, 2) void conv_rf_kernel(ConvArgs a, RfGeom gm) {
    typedef RfCfg<T, C, NCT, WP, TP, PR> Cfg;
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
    wait_vmcnt<0>();    // the filter fetch must not sit in the counted waits below

    // ---- LDS-DMA of one BP-row group into its ring slot: ALWAYS Cfg::PW pieces per wave (the waits below count
    // them).  Rows below 0 / past the tensor only feed dropped outputs: their pieces re-read an in-range row.
    const int lrow = lane / LPR, lslot = lane % LPR;
    auto stage = [&](int gi) {
        const int slot = gi & (Cfg::NSLOT - 1);
        char* dst = ring + slot * BP * ROWB;
#pragma unroll
        for (int k = 0; k < Cfg::PW; ++k) {
            const int i = k * NW + w;
            int row0 = gi * BP + i * RPI;
            row0 = row0 < 0 ? 0 : (row0 + RPI > gm.qmax ? gm.qmax - RPI : row0);
            const int rr = slot * BP + i * RPI + lrow;
            glds16((const char*)a.x + (size_t)(row0 + lrow) * ROWB + ((lslot ^ ((rr / RPB) % LPR)) * 16), dst + i * 1024);
        }
    };

    // ---- per lane and pixel sub-tile: ring row of the pixel, and its (col, row-in-image, image) for the epilogue
    int qm[TP], pcol[TP], prow_[TP], pimg[TP];
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const int q = T0 * BP + (w * TP + j) * 32 + r32;
        qm[j] = q & (R - 1);
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
    char* const ydump = (char*)a.y + (size_t)a.M * a.ldy * SZ;   // the slack row behind y: target of the masked stores

    // ONE (count, mean, M2) record per workgroup: the per-tile sums (fp32, from the matrix pipe) are added over the
    // workgroup's tiles in double, so the record is as exact as per-tile records merged in double -- and the merge
    // kernel sees a few hundred records instead of ten thousand
    double S1d[NCT], S2d[NCT];
#pragma unroll
    for (int i = 0; i < NCT; ++i) S1d[i] = S2d[i] = 0.0;
    int cntd = 0;
    auto emit_record = [&](int rec) {     // tid < BC: from the waves' sums
        double S1 = 0.0, S2 = 0.0;
        float cnt = 0.f;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const double* q = (const double*)(scr_all + k * (Cfg::SCR / 4));
            S1 += q[tid];
            S2 += q[BC + tid];
            cnt += (float)q[2 * BC];
        }
        const int co = n0 + tid;
        if (co < a.ldy) {
            const double md = cnt > 0.f ? S1 / (double)cnt : 0.0;
            const double m2 = S2 - S1 * md;
            a.part_mean[(size_t)rec * a.ldy + co] = (float)md;
            a.part_m2[(size_t)rec * a.ldy + co] = (float)(m2 > 0.0 ? m2 : 0.0);
        }
        if (tid == 0 && blockIdx.y == 0) a.part_cnt[rec] = cnt;
    };

    // ONE workgroup barrier per tile: at the top of tile t every wave has left tile t-1's K loop, so group t-2 is dead
    // and its ring slot takes group t+2 (needed at the top of t+1: a whole tile of lead); the epilogue is wave-private
    stage(T0 - 1);
    stage(T0);
    stage(T0 + 1);
    for (int tile = T0; tile < T1; ++tile) {
        // groups tile-1 .. tile+1 have landed when only the last epilogue's stores (issued after them) are outstanding
        if (tile == T0) wait_vmcnt<0>();
        else wait_vmcnt<Cfg::NST>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stage(tile + 2);

        // the accumulators start at the bias (rows = couts): no add in the epilogue
        f32x16 acc[NCT][TP];
#pragma unroll
        for (int i = 0; i < NCT; ++i)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 b4 = *(const f32x4*)(biasl + i * 32 + 8 * q4 + 4 * hh);
#pragma unroll
                for (int j = 0; j < TP; ++j)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[i][j][4 * q4 + k] = b4[k];
            }

        // pixel fragments PD steps ahead of their MFMAs (a step is only NCT * TP MFMAs long: one step of lead
        // does not cover the LDS latency with two waves per SIMD)
        frag_t fa[PD + 1][TP];
        auto load_frags = [&](int s, frag_t (&f)[TP]) {
            const int t = s / G, g = s % G;
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const int rr = (qm[j] + sh[t]) & (R - 1);
                f[j] = *(const frag_t*)(ring + rr * ROWB + (((2 * g + hh) ^ ((rr / RPB) % LPR)) * 16));
            }
        };
#pragma unroll
        for (int s = 0; s < PD; ++s) load_frags(s, fa[s]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KGT; ++s) {
            if (s + PD < KGT) load_frags(s + PD, fa[(s + PD) % (PD + 1)]);
#pragma unroll
            for (int i = 0; i < NCT; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) mma32(acc[i][j], wreg[i][s], fa[s % (PD + 1)][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue, PR pixel rows at a time through the wave's own patch
        float S1w[NCT], S2w[NCT];
#pragma unroll
        for (int i = 0; i < NCT; ++i) S1w[i] = S2w[i] = 0.f;
        int cntw = 0;
        bool bad = false;
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const bool valid = pcol[j] >= 1 && prow_[j] >= 1 && pimg[j] < a.N;
            const int p = valid ? (pimg[j] * a.H + prow_[j] - 1) * a.W + pcol[j] - 1 : -1;
            cntw += __popcll(__ballot(valid && hh == 0));
#pragma unroll
            for (int hp = 0; hp < 32 / PR; ++hp) {
                if (PR == 32 || (r32 / PR) == hp) {
                    // border positions are rare: everything is converted and written, then their rows are cleared
#pragma unroll
                    for (int i = 0; i < NCT; ++i)
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            const u32x2 o = {pack2<T>(acc[i][j][4 * q4], acc[i][j][4 * q4 + 1]),
                                             pack2<T>(acc[i][j][4 * q4 + 2], acc[i][j][4 * q4 + 3])};
                            *(u32x2*)(patch + (r32 % PR) * EROW + (i * 32 + 8 * q4 + 4 * hh) * SZ) = o;
                        }
                    if (!valid) {
#pragma unroll
                        for (int i = 0; i < NCT; ++i)
#pragma unroll
                            for (int q4 = 0; q4 < 4; ++q4)
                                *(u32x2*)(patch + (r32 % PR) * EROW + (i * 32 + 8 * q4 + 4 * hh) * SZ) = u32x2{0u, 0u};
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                constexpr int EPC = 8, CPR = Cfg::CPR, RPIe = 64 / CPR;
                const int ch = lane % CPR, prow0 = lane / CPR;
#pragma unroll
                for (int it = 0; it < Cfg::NIT; ++it) {
                    const int prow = it * RPIe + prow0;
                    const int pr = __shfl(p, hp * PR + prow, 64);
                    Chunk<T> c = ld_chunk<T>(patch + prow * EROW + ch * 16);
                    const int cch = n0 + ch * EPC;
                    const bool st = pr >= 0 && cch < a.ldy;
                    // every lane stores (the tile's store count is part of the counted waits): masked rows go to the slack row
                    char* dstp = st ? (char*)a.y + ((size_t)pr * a.ldy + cch) * SZ : ydump + (ch % (a.ldy / EPC)) * 16;
                    st_chunk<T>(dstp, c);
                    if (chk && st) {
#pragma unroll
                        for (int e = 0; e < EPC; ++e)
                            bad |= (__float_as_uint(Elem<T>::to_f32(c.v[e])) & 0x7F800000u) == 0x7F800000u;
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
                        for (int kg = 0; kg < PR / 16; ++kg) {
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
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the patch is rewritten by the next pass
            }
        }
        if (chk && __any(bad) && lane == 0) atomicOr(a.nonfinite, 1u);
        if (stats) {
#pragma unroll
            for (int i = 0; i < NCT; ++i) {
                S1d[i] += (double)S1w[i];
                S2d[i] += (double)S2w[i];
            }
            cntd += cntw;
        }
        // ---- next tile: BP positions on
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            qm[j] = (qm[j] + BP) & (R - 1);
            pcol[j] += BP;
            while (pcol[j] >= pitch) { pcol[j] -= pitch; ++prow_[j]; }
            while (prow_[j] >= gm.rows_img) { prow_[j] -= gm.rows_img; ++pimg[j]; }
        }
    }
    wait_vmcnt<0>();      // the ring's last fills must not outlive the workgroup's LDS
    if (stats) {
        double* sd = (double*)scr;
        if (hh == 0) {
#pragma unroll
            for (int i = 0; i < NCT; ++i) {
                sd[i * 32 + r32] = S1d[i];
                sd[BC + i * 32 + r32] = S2d[i];
            }
        }
        if (lane == 0) sd[2 * BC] = (double)cntd;
        __syncthreads();
        if (tid < BC) emit_record(blockIdx.x);
    }
}

template <typename T, int C, int NCT, int WP, int TP, int PR, int PD>
static hipError_t rf_launch(const ConvArgs& a, hipStream_t s, int* bp, int* records) {
    typedef RfCfg<T, C, NCT, WP, TP, PR> Cfg;
    if (a.ldy > 128 || a.ldy % 8 != 0) return hipErrorInvalidValue;   // masked stores land in y's 256-byte slack row
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
    auto kern = conv_rf_kernel<T, C, NCT, WP, TP, PR, PD>;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS);
        if (e != hipSuccess) return e;
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3(nblk, nct), dim3(Cfg::NT), Cfg::LDS, s, a, g);
    if (bp) *bp = Cfg::BP;
    if (records) *records = nblk;          // one record per workgroup
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Same scheme with the couts spread over WN waves (one 32-cout slice each, the pixel fragments shared through the
// LDS ring) for the 128-cout layers: the workgroup's output tile goes through ONE [pixel][cout] patch so the stores
// are whole 256-byte rows; statistics from the matrix pipe as above, each wave over its own columns and rows.
// ---------------------------------------------------------------------------
// NSLOT ring slots of BP rows; AH = groups of halo on each side (1: pitch + 1 <= BP, 2: <= 2 BP); the NSLOT - 2 AH - 1
// youngest groups may still be in flight when a tile starts
template <typename T, int C, int WP, int WN, int TP, int NSLOT_, int AH_>
struct RfnCfg {
    static constexpr int SZ = sizeof(T);
    static constexpr int NW = WP * WN, NT = NW * 64;
    static constexpr int ROWB = C * SZ, LPR = ROWB / 16, RPI = 64 / LPR, RPB = 256 / ROWB;
    static constexpr int G = C * SZ / 32, KGT = 9 * G;
    static constexpr int BP = WP * TP * 32, BC = WN * 32;
    static constexpr int NSLOT = NSLOT_, AH = AH_, DF = NSLOT - 2 * AH - 1, R = NSLOT * BP, RINGB = R * ROWB;
    static constexpr int EROW = BC * SZ + 16, PATCHB = BP * EROW;
    static constexpr int SCRF = 2 * (NW * 2 * BC + NW);      // floats (= NW x {S1[BC], S2[BC]} + NW counts as doubles)
    static constexpr int YTILE = BP * BC * SZ;               // that mode's tile of the layer below's conv output
    static constexpr int LDS = RINGB + PATCHB + SCRF * 4 + BP * 4 + BC * 4;   // + row table + bias slice
    static constexpr int LDS_BW = LDS + YTILE + 2 * BC * 4;
    static constexpr int PW = BP / RPI / NW;
    static constexpr int CPR = BC / 8, NIT = BP * CPR / NT;
    static constexpr int NST = NIT;
    static_assert(DF >= 1, "one group ahead at least");
    static_assert(BP % (RPI * NW) == 0 && (BP * CPR) % NT == 0 && NT % CPR == 0, "even split over lanes");
    static_assert(DF * (PW + NST) < 63, "counted vmcnt");
};

template <typename T, int C, int WP, int WN, int TP, int PD, int NSLOT, int AH, bool BW>
__global__ __launch_bounds__(WP * WN * 64, 2) void conv_rfn_kernel(ConvArgs a, RfGeom gm) {
    typedef RfnCfg<T, C, WP, WN, TP, NSLOT, AH> Cfg;
    typedef typename Elem<T>::frag frag_t;
    constexpr int SZ = Cfg::SZ, NW = Cfg::NW, NT = Cfg::NT, ROWB = Cfg::ROWB, LPR = Cfg::LPR, RPI = Cfg::RPI, RPB = Cfg::RPB;
    constexpr int G = Cfg::G, KGT = Cfg::KGT, BP = Cfg::BP, BC = Cfg::BC, R = Cfg::R, EROW = Cfg::EROW, CPR = Cfg::CPR;
    static_assert(SZ == 2, "16-bit element types");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ring = smem;
    char* const patch = smem + Cfg::RINGB;
    float* const scr = (float*)(patch + Cfg::PATCHB);          // [WP][2][BC], then WP counts
    int* const ptab = (int*)(scr + Cfg::SCRF);                 // NHWC pixel of every patch row, -1: border position
    float* const biasl = (float*)(ptab + BP);
    char* const ybuf = (char*)(biasl + BC);                    // bw mode only (the launcher sizes the LDS for it)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = w / WN, wn = w % WN;
    const int r32 = lane & 31, hh = lane >> 5;
    const int pitch = gm.pitch;

    const int T0 = blockIdx.x * gm.tiles_per_block;
    int T1 = T0 + gm.tiles_per_block;
    if (T1 > gm.ntiles) T1 = gm.ntiles;
    if (T0 >= T1) return;

    const int n0 = blockIdx.y * BC;
    frag_t wreg[KGT];
    {
        const char* wr = (const char*)a.w + (size_t)(n0 + wn * 32 + r32) * 9 * ROWB + hh * 16;
#pragma unroll
        for (int s = 0; s < KGT; ++s) wreg[s] = *(const frag_t*)(wr + s * 32);
    }
    if (tid < BC) biasl[tid] = (a.bias && n0 + tid < a.Cout) ? a.bias[n0 + tid] : 0.f;
    wait_vmcnt<0>();

    const int lrow = lane / LPR, lslot = lane % LPR;
    auto stage = [&](int gi) {
        const int slot = (gi + 4 * NSLOT) % NSLOT;
        char* dst = ring + slot * BP * ROWB;
#pragma unroll
        for (int k = 0; k < Cfg::PW; ++k) {
            const int i = k * NW + w;
            int row0 = gi * BP + i * RPI;
            row0 = row0 < 0 ? 0 : (row0 + RPI > gm.qmax ? gm.qmax - RPI : row0);
            const int rr = slot * BP + i * RPI + lrow;
            glds16((const char*)a.x + (size_t)(row0 + lrow) * ROWB + ((lslot ^ ((rr / RPB) % LPR)) * 16), dst + i * 1024);
        }
    };

    int qm[TP], pcol[TP], prow_[TP], pimg[TP];
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const int q = T0 * BP + (wp * TP + j) * 32 + r32;
        qm[j] = q % R;
        const int rowi = q / pitch;
        pcol[j] = q - rowi * pitch;
        pimg[j] = rowi / gm.rows_img;
        prow_[j] = rowi - pimg[j] * gm.rows_img;
    }
    int sh[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) sh[t] = (t / 3 - 1) * pitch + (t % 3 - 1);

    const bool stats = !BW && a.part_mean != nullptr;
    const bool chk = a.nonfinite != nullptr;
    constexpr bool bw = BW;                    // dgrad: BN-backward reduce of the layer below (ConvArgs::bw_*)
    // forward / plain launches on the four-slot ring: every wave stores its own 64-byte row segments (no patch barrier,
    // no row table) and the ring slot is refilled at the top of the next tile: ONE workgroup barrier per tile
    constexpr bool WPRIV = !BW && AH == 1 && NSLOT == 4;
    const int qq = (lane & 15) >> 2, pp = lane & 3, g1 = (lane >> 4) & 1;
    // bw mode: scale / shift of the layer below wait in LDS (behind the y tile), registers are for the filters
    float* const bwtab = (float*)(ybuf + Cfg::YTILE);
    if (bw) {
        if (tid < BC) {
            bwtab[tid] = n0 + tid < a.ldy ? a.bw_scale[n0 + tid] : 0.f;
            bwtab[BC + tid] = n0 + tid < a.ldy ? a.bw_shift[n0 + tid] : 0.f;
        }
        wait_vmcnt<0>();
    }
    char* const ydump = (char*)a.y + (size_t)a.M * a.ldy * SZ;

    // ONE record per workgroup (see conv_rf_kernel): the tiles' sums are added in double, statistics in registers,
    // the BN-backward sums of the dgrad mode in the wave's own LDS slots
    double* const scrd = (double*)scr;     // [NW][2][BC] doubles, then NW counts
    double S1d = 0.0, S2d = 0.0;
    int cntd = 0;
    if (bw && lane < 16) {
#pragma unroll
        for (int e = 0; e < 8; ++e) scrd[w * 2 * BC + lane * 8 + e] = scrd[w * 2 * BC + BC + lane * 8 + e] = 0.0;
    }
    auto emit_bw = [&](int rec) {         // tid < BC: the workgroup's S1 / S2 from the waves' sums, fixed order
        double S1 = 0.0, S2 = 0.0;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            S1 += scrd[k * 2 * BC + tid];
            S2 += scrd[k * 2 * BC + BC + tid];
        }
        const int co = n0 + tid;
        if (co < a.ldy) {
            a.bw_psum[((size_t)rec * 2 + 0) * a.ldy + co] = (float)S1;
            a.bw_psum[((size_t)rec * 2 + 1) * a.ldy + co] = (float)S2;
        }
    };
    auto emit_record = [&](int rec) {     // tid < BC
        double S1 = 0.0, S2 = 0.0;
        float cnt = 0.f;
#pragma unroll
        for (int k = 0; k < WP; ++k) {
            S1 += scrd[(k * WN) * 2 * BC + tid];
            S2 += scrd[(k * WN) * 2 * BC + BC + tid];
            cnt += (float)scrd[NW * 2 * BC + k];
        }
        const int co = n0 + tid;
        if (co < a.ldy) {
            const double md = cnt > 0.f ? S1 / (double)cnt : 0.0;
            const double m2 = S2 - S1 * md;
            a.part_mean[(size_t)rec * a.ldy + co] = (float)md;
            a.part_m2[(size_t)rec * a.ldy + co] = (float)(m2 > 0.0 ? m2 : 0.0);
        }
        if (tid == 0 && blockIdx.y == 0) a.part_cnt[rec] = cnt;
    };

#pragma unroll
    for (int g = 0; g < (WPRIV ? NSLOT - 1 : NSLOT); ++g) stage(T0 - AH + g);
#ifdef Y2_DEVBUILD
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast)::"memory");
#endif
    for (int tile = T0; tile < T1; ++tile) {
        if (WPRIV) {   // only the last epilogue's stores (issued after every group this tile needs) may be outstanding
            if (tile == T0) wait_vmcnt<0>();
            else wait_vmcnt<Cfg::NST>();
        } else {   // groups tile - AH .. tile + AH have landed when all but the DF youngest (and the stores between them) are done
            const int done = tile - T0;
            wait_vmcnt_dyn(Cfg::DF * Cfg::PW + (done < Cfg::DF ? done : Cfg::DF) * Cfg::NST);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        RF_STAMP(0);    // wait for the groups
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        RF_STAMP(1);    // barrier A
        if (WPRIV) stage(tile + 2);     // every wave has left tile - 1's K loop: group tile - 2 is dead

        f32x16 acc[TP];     // start at the bias (rows = couts)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const f32x4 b4 = *(const f32x4*)(biasl + wn * 32 + 8 * q4 + 4 * hh);
#pragma unroll
            for (int j = 0; j < TP; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[j][4 * q4 + k] = b4[k];
        }
        frag_t fa[PD + 1][TP];
        auto load_frags = [&](int s, frag_t (&f)[TP]) {
            const int t = s / G, g = s % G;
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                int rr = qm[j] + sh[t];
                if ((R & (R - 1)) == 0) {
                    rr &= R - 1;
                } else {
                    rr = rr < 0 ? rr + R : rr;
                    rr = rr >= R ? rr - R : rr;
                }
                f[j] = *(const frag_t*)(ring + rr * ROWB + (((2 * g + hh) ^ ((rr / RPB) % LPR)) * 16));
            }
        };
#pragma unroll
        for (int s = 0; s < PD; ++s) load_frags(s, fa[s]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KGT; ++s) {
            if (s + PD < KGT) load_frags(s + PD, fa[(s + PD) % (PD + 1)]);
#pragma unroll
            for (int j = 0; j < TP; ++j) mma32(acc[j], wreg[s], fa[s % (PD + 1)][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
        RF_STAMP(2);    // record + K loop
        RF_STAMP(3);

        // ---- the wave's 32-cout columns of its rows into the workgroup's patch
        int cntw = 0;
        int pj[TP], pbj[TP];
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const bool valid = pcol[j] >= 1 && prow_[j] >= 1 && pimg[j] < a.N;
            const int p = valid ? (pimg[j] * a.H + prow_[j] - 1) * a.W + pcol[j] - 1 : -1;
            const int row = (wp * TP + j) * 32 + r32;
            if (!WPRIV && wn == 0 && hh == 0) ptab[row] = p;
            pj[j] = p;
            // bordered position of this pixel in a tensor of the same H x W (the consumer's input): the folded
            // inference batch norm stores there (ConvArgs::aff_out)
            pbj[j] = (pimg[j] * gm.rows_img + prow_[j]) * pitch + pcol[j];
            cntw += __popcll(__ballot(valid && hh == 0));
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const u32x2 o = {pack2<T>(acc[j][4 * q4], acc[j][4 * q4 + 1]), pack2<T>(acc[j][4 * q4 + 2], acc[j][4 * q4 + 3])};
                *(u32x2*)(patch + row * EROW + (wn * 32 + 8 * q4 + 4 * hh) * SZ) = o;
            }
            if (!valid) {     // border positions are rare: their rows are cleared after the fact
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
                    *(u32x2*)(patch + row * EROW + (wn * 32 + 8 * q4 + 4 * hh) * SZ) = u32x2{0u, 0u};
            }
        }
        if (stats) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // own columns, own rows: no barrier needed
            frag_t ones;
#pragma unroll
            for (int k = 0; k < 8; ++k) ones[k] = (T)1.0f;
            f32x16 q1, q2;
#pragma unroll
            for (int q = 0; q < 16; ++q) q1[q] = q2[q] = 0.f;
            const char* pb = patch + (wp * TP * 32 + 8 * hh + qq) * EROW + (wn * 32 + 16 * g1 + 4 * pp) * 2;
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
            dg += __shfl_xor(dg, 32, 64);
            S1d += (double)q1[0];
            S2d += (double)dg;
            cntd += cntw;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        RF_STAMP(4);    // stage + patch + statistics
        bool bad = false;
        if (WPRIV) {
            // ---- the wave's own region out: 64-byte row segments (its 32 couts), lane = (row of 16, 16-byte chunk of 4)
            static_assert(!WPRIV || Cfg::NST == TP * 2, "store count of the wave-private sweep");
#pragma unroll
            for (int it = 0; it < TP * 2; ++it) {
                const int rloc = (it & 1) * 16 + (lane >> 2);              // row inside sub-tile it / 2
                const int pr = __shfl(pj[it >> 1], rloc, 64);
                const int row = (wp * TP + (it >> 1)) * 32 + rloc;
                const int ch = wn * 4 + (lane & 3);
                Chunk<T> c = ld_chunk<T>(patch + row * EROW + ch * 16);
                const int cch = n0 + ch * 8;
                const bool st = pr >= 0 && cch < a.ldy;
                char* dstp = st ? (char*)a.y + ((size_t)pr * a.ldy + cch) * SZ : ydump + (ch % (a.ldy / 8)) * 16;
                if (a.aff_out) {
                    // inference batch norm folded in (kernels.h ConvArgs::aff_*): leaky(T(conv + b) * scale + shift) into
                    // the consumer's bordered tensor -- the arithmetic of bn_act_kernel on the same rounded value; the
                    // per-channel constants come from L1 every sweep (the filter slice owns the registers)
                    const int pb = __shfl(pbj[it >> 1], rloc, 64);
                    if (st) {
                        const f32x4 s0 = *(const f32x4*)(a.aff_scale + cch), s1 = *(const f32x4*)(a.aff_scale + cch + 4);
                        const f32x4 h0 = *(const f32x4*)(a.aff_shift + cch), h1 = *(const f32x4*)(a.aff_shift + cch + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            c.v[e] = Elem<T>::from_f32(leaky_s(Elem<T>::to_f32(c.v[e]) * s0[e] + h0[e], a.aff_slope));
                            c.v[4 + e] = Elem<T>::from_f32(leaky_s(Elem<T>::to_f32(c.v[4 + e]) * s1[e] + h1[e], a.aff_slope));
                        }
                        dstp = (char*)a.aff_out + ((size_t)pb * a.ldy + cch) * SZ;
                    }
                }
                st_chunk<T>(dstp, c);
                if (chk && st) {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        bad |= (__float_as_uint(Elem<T>::to_f32(c.v[e])) & 0x7F800000u) == 0x7F800000u;
                }
            }
        }
        if (!WPRIV) {
        __builtin_amdgcn_s_barrier();          // patch and row table complete
        asm volatile("" ::: "memory");
        }
        RF_STAMP(5);    // barrier C: every wave has also left the K loop, the oldest group's slot is free
        // ---- whole rows out: lane = (row, 16-byte chunk), consecutive lanes along a row
        if (!bw && !WPRIV) {
            stage(tile - AH + NSLOT);
#pragma unroll
            for (int it = 0; it < Cfg::NIT; ++it) {
                const int idx = it * NT + tid;
                const int row = idx / CPR, ch = idx % CPR;
                const int pr = ptab[row];
                Chunk<T> c = ld_chunk<T>(patch + row * EROW + ch * 16);
                const int cch = n0 + ch * 8;
                const bool st = pr >= 0 && cch < a.ldy;
                char* dstp = st ? (char*)a.y + ((size_t)pr * a.ldy + cch) * SZ : ydump + (ch % (a.ldy / 8)) * 16;
                st_chunk<T>(dstp, c);
                if (chk && st) {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        bad |= (__float_as_uint(Elem<T>::to_f32(c.v[e])) & 0x7F800000u) == 0x7F800000u;
                }
            }
        } else if (bw) {
            // the layer below's conv output at this tile's pixels comes in by LDS-DMA too (an ordinary load would make
            // the compiler drain the ring's DMA at its first use): issued BEFORE the ring's next group, waited for with
            // a count that leaves that group in flight; a wave reads back only its own pieces
            int prv[Cfg::NIT];
#pragma unroll
            for (int it = 0; it < Cfg::NIT; ++it) {
                const int idx = it * NT + tid;
                prv[it] = ptab[idx / CPR];
                const int cch = n0 + (idx % CPR) * 8;
                const size_t src = ((size_t)(prv[it] < 0 ? 0 : prv[it]) * a.ldy + (cch < a.ldy ? cch : 0)) * SZ;
                glds16((const char*)a.bw_y + src, ybuf + (it * NT + w * 64) * 16);
            }
            stage(tile - AH + NSLOT);
            wait_vmcnt<Cfg::PW>();
            float s1[8], s2[8], bsc[8], bsh[8];     // this lane's chunk column is fixed (NT % CPR == 0)
#pragma unroll
            for (int e = 0; e < 8; e += 4) {
                const f32x4 v1 = *(const f32x4*)(bwtab + (tid % CPR) * 8 + e), v2 = *(const f32x4*)(bwtab + BC + (tid % CPR) * 8 + e);
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) { bsc[e + k2] = v1[k2]; bsh[e + k2] = v2[k2]; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
#pragma unroll
            for (int it = 0; it < Cfg::NIT; ++it) {
                const int idx = it * NT + tid;
                const int row = idx / CPR, ch = idx % CPR;
                const int pr = prv[it];
                Chunk<T> c = ld_chunk<T>(patch + row * EROW + ch * 16);
                Chunk<T> yv = ld_chunk<T>(ybuf + idx * 16);
                const int cch = n0 + ch * 8;
                const bool st = pr >= 0 && cch < a.ldy;
                char* dstp = st ? (char*)a.y + ((size_t)pr * a.ldy + cch) * SZ : ydump + (ch % (a.ldy / 8)) * 16;
                st_chunk<T>(dstp, c);
                if (st) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float yf = Elem<T>::to_f32(yv.v[e]);
                        const float g = Elem<T>::to_f32(c.v[e]) * leaky_slope_s(fmaf(yf, bsc[e], bsh[e]), a.bw_slope);
                        s1[e] += g;
                        s2[e] = fmaf(g, yf, s2[e]);
                    }
                }
            }
            // lanes l, l+16, l+32, l+48 share a chunk column: add them, lanes 0-15 hold the wave's sums
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s1[e] += __shfl_xor(s1[e], 16, 64);
                s1[e] += __shfl_xor(s1[e], 32, 64);
                s2[e] += __shfl_xor(s2[e], 16, 64);
                s2[e] += __shfl_xor(s2[e], 32, 64);
            }
            if (lane < 16) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    scrd[w * 2 * BC + lane * 8 + e] += (double)s1[e];
                    scrd[w * 2 * BC + BC + lane * 8 + e] += (double)s2[e];
                }
            }
        }
        if (chk && __any(bad) && lane == 0) atomicOr(a.nonfinite, 1u);
        RF_STAMP(6);    // sweep
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            qm[j] += BP;
            qm[j] = qm[j] >= R ? qm[j] - R : qm[j];
            pcol[j] += BP;
            while (pcol[j] >= pitch) { pcol[j] -= pitch; ++prow_[j]; }
            while (prow_[j] >= gm.rows_img) { prow_[j] -= gm.rows_img; ++pimg[j]; }
        }
    }
    wait_vmcnt<0>();
#ifdef Y2_DEVBUILD
    if (blockIdx.x == 0 && lane == 0)
        for (int k2 = 0; k2 < 8; ++k2) g_rf_stamps[w][k2] = ph[k2];
#endif
    if (stats || bw) {
        if (stats) {      // rows (wp, wn = 0 .. WN-1) hold the wave's 32 columns at their place in a BC-wide row of wave wp * WN
            if (hh == 0) {
                scrd[(wp * WN) * 2 * BC + wn * 32 + r32] = S1d;
                scrd[(wp * WN) * 2 * BC + BC + wn * 32 + r32] = S2d;
            }
            if (wn == 0 && lane == 0) scrd[NW * 2 * BC + wp] = (double)cntd;
        }
        __syncthreads();
        if (stats && tid < BC) emit_record(blockIdx.x);
        if (bw && tid < BC) emit_bw(blockIdx.x);
    }
}


#ifdef Y2_DEVBUILD
hipError_t rf_read_stamps(unsigned long long* dst) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_rf_stamps), sizeof(unsigned long long) * 64);
}
#endif

template <typename T, int C, int WP, int WN, int TP, int PD, int NSLOT, int AH>
static hipError_t rfn_launch(const ConvArgs& a, hipStream_t s, int* bp, int* records) {
    typedef RfnCfg<T, C, WP, WN, TP, NSLOT, AH> Cfg;
    if (a.ldy > 128 || a.ldy % 8 != 0) return hipErrorInvalidValue;
    RfGeom g{};
    g.pitch = a.W + 1;
    g.rows_img = a.H + 1;
    if (g.pitch + 1 > Cfg::AH * Cfg::BP) return hipErrorInvalidValue;     // the halo must stay inside the AH neighbouring groups
    const long qtot = (long)a.N * g.rows_img * g.pitch;
    g.ntiles = (int)((qtot + Cfg::BP - 1) / Cfg::BP);
    g.qmax = (int)bbody_pixels(a.N, a.H, a.W);
    const int nct = (a.Cout + Cfg::BC - 1) / Cfg::BC;
    const int lds = a.bw_psum ? Cfg::LDS_BW : Cfg::LDS;
    if (lds > 160 * 1024) return hipErrorOutOfMemory;
    int nblk = (lds <= 80 * 1024 ? 512 : 256) / nct;
    if (nblk > g.ntiles) nblk = g.ntiles;
    g.tiles_per_block = (g.ntiles + nblk - 1) / nblk;
    nblk = (g.ntiles + g.tiles_per_block - 1) / g.tiles_per_block;
    auto kern = a.bw_psum ? conv_rfn_kernel<T, C, WP, WN, TP, PD, NSLOT, AH, true>
                          : conv_rfn_kernel<T, C, WP, WN, TP, PD, NSLOT, AH, false>;
    static int attr[2] = {0, 0};
    if (lds > attr[a.bw_psum ? 1 : 0]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr[a.bw_psum ? 1 : 0] = lds;
    }
    hipLaunchKernelGGL(kern, dim3(nblk, nct), dim3(Cfg::NT), lds, s, a, g);
    if (bp) *bp = Cfg::BP;
    if (records) *records = nblk;          // one record per workgroup
    return hipGetLastError();
}

// 0: not this form (16-bit launches only: row_bytes = input channels * 2).
//   1: 32 -> 64 on 208-wide maps (forward of the second layer)      8 waves x 32 pixels x 64 couts, one workgroup per CU
//   2: 64 -> 32 on 208-wide maps (its dgrad)                        8 waves x 32 pixels x 32 couts, one workgroup per CU
int conv_rf_config(int taps, int W, int row_bytes, int Cout, int M) {
    static const bool off = getenv("Y2_NO_CONV_RF") != nullptr;
    if (off || taps != 9 || W <= 104 || W + 2 >= 256 || M < 256 * 1024) return 0;
    if (row_bytes == 64 && Cout > 32 && Cout <= 64) return 1;
    if (row_bytes == 128 && Cout <= 32) return 2;
    return 0;
}
// the 128-cout form: forward (with statistics), plain, and dgrad with the fused BN-backward reduce
int conv_rfn_config(int taps, int W, int row_bytes, int Cout, int M, int dgrad) {
    static const bool off = getenv("Y2_NO_CONV_RF") != nullptr;
    static const bool nodg = getenv("Y2_NO_CONV_RFN_DGRAD") != nullptr;
    if (off || (dgrad && nodg) || taps != 9 || W <= 52 || W + 2 > 128 || M < 128 * 1024) return 0;
    if (row_bytes == 128 && Cout > 64 && Cout <= 128) return 3;
    return 0;
}

template <typename T>
static hipError_t rf_T(int cfg, const ConvArgs& a, hipStream_t s, int* bp, int* records) {
    // measured (scripts/profile_layers.py): the lead of the fragment reads (1..4 steps) does not matter -- two waves
    // per SIMD cover each other's LDS latency; two 4-wave workgroups per CU (64-position tiles, six-slot ring) are 5 %
    // SLOWER than one 8-wave workgroup on the 128-cout layers (a lone wave cannot keep the matrix pipe busy)
    if (cfg == 1) return rf_launch<T, 32, 2, 8, 1, 32, 2>(a, s, bp, records);
    if (cfg == 2) return rf_launch<T, 64, 1, 8, 1, 32, 3>(a, s, bp, records);
#ifdef Y2_DEVBUILD
    static const int alt = getenv("Y2DEV_RF_ALT") ? atoi(getenv("Y2DEV_RF_ALT")) : 0;
    if (cfg == 3 && alt == 1) return rfn_launch<T, 64, 1, 4, 2, 2, 6, 2>(a, s, bp, records);
#endif
    if (cfg == 3) return rfn_launch<T, 64, 2, 4, 2, 2, 4, 1>(a, s, bp, records);
    return hipErrorInvalidValue;
}
hipError_t launch_conv_rf(int dtype, const ConvArgs& a, hipStream_t s, int* bp, int* records) {
    int cfg = conv_rf_config(a.taps, a.W, a.C * (int)dtype_size(dtype), a.Cout, a.M);
    if (!cfg) cfg = conv_rfn_config(a.taps, a.W, a.C * (int)dtype_size(dtype), a.Cout, a.M, a.is_dgrad);
    if (dtype == 1) return rf_T<half_t>(cfg, a, s, bp, records);
    if (dtype == 2) return rf_T<bf16_t>(cfg, a, s, bp, records);
    return hipErrorInvalidValue;
}

}  // namespace y2
