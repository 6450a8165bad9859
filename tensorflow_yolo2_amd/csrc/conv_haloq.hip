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
//
// COMPACT image (CPT; Y2_HALO_COMPACT=1 -- built, measured, NOT the default): the LDS rows follow the NHW pixel index,
// not the bordered cell index, and every border tap reads one of 16 zero rows.  The bordered image skips a cell at
// every image-row wrap, so the 16 LDS rows of a ds_read_b128 lane group ({0-3,12-15,20-27}, ...) stop being
// distinct mod 16 whenever a 32-pixel fragment crosses an image row -- at W = 13 / 26 always: a 2-way bank conflict
// on nearly every fragment read (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.50 measured, 0.50 in a bank
// simulation of the address stream).  In pixel order a tap shift is uniform (kh W + kw) wherever the tap stays
// inside the image, so a lane group's rows are lambda0 + {0-3,12-15,20-27}: all 16 residues, conflict-free under
// the row-keyed XOR swizzle for every tap; a tap that leaves the image reads zero row (lambda & 15): the bank
// position the in-image cell would have had.  The DMA source of an LDS row is a per-row cell index computed once
// per workgroup (table in LDS); border cells are not staged.  16x16 MFMA tiles mix two k-chunks in a lane group:
// there the pixel columns of a 16-pixel fragment are dealt so that the first chunk's lanes take the even pixels
// and the other's the odd ones (perm16), conflict-free for odd and even lambda0 alike; the epilogue patch is
// written through the same map.  MEASURED (13x13, 1024 -> 1024, batch 64, same box, round 3): conflict ratio
// 0.50 -> 0.02, LDS-active cycles halved -- and the kernel 4 % SLOWER (181 vs 174 us): the per-(fragment, tap)
// border select costs 36 more VALU instructions per tap step (87 vs 51 beside 48 MFMAs), and this loop is bound by
// vector ISSUE, not by the LDS: SQ_WAIT_INST_LDS is 0.9 % of the wave cycles with the conflicts in place.  The
// conflicts were never on the critical path; the bordered image (uniform shift, no select) stays the default.
//
// What did pay in round 3 (both images): the next tap's fragment addresses are computed one step ahead and pinned
// between the MFMAs (mfma_interleave), k-group g of a fragment row is address ^ (g * 64) instead of a second swizzle,
// and the filter fragments are loaded by inline asm with hand-counted vmcnt (frag_load) -- together -3 % on the
// 13x13 / 26x26 layers against the round-2 kernel on the same box.
#include <stdlib.h>
#include "common.h"
#include "conv_epilogue.h"
#include "kernels.h"

namespace y2 {

template <int V> struct IntC { static constexpr int value = V; };

constexpr int kZeroRows = 16;   // compact image: zero rows in front of the image buffers (one per bank position)

// Filter-fragment load the compiler does not see: 16 bytes per lane from a per-lane 64-bit address + immediate.  hipcc waits vmcnt(0) at the first use of ANY load it knows of while an LDS-DMA is in flight
// (cdna_hip_programming.md, "Three .s-level traps" (b)): with plain loads every other tap step began by draining the
// fragments it had just requested for the NEXT step -- one exposed L2 round trip per two steps.  These loads are counted
// by hand instead (frag_wait: `s_waitcnt vmcnt(N)` leaves the N youngest operations in flight; they retire in order).
template <int IMM>
Y2_DEV void frag_load(u32x4& dst, const char* lane_ptr) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(lane_ptr), "n"(IMM) : "memory");
}
// after the counted wait: the registers now hold the data (orders every consumer behind the wait)
Y2_DEV void frag_ready(u32x4& v) { asm volatile("" : "+v"(v)); }
// Order of one scheduling region that holds NM MFMAs, ND ds_reads and address VALU (the next tap's addresses): the
// VALU and the LDS reads go BETWEEN the MFMAs (which leave vector-issue slots free) instead of in front of them --
// left alone, hipcc puts all the VALU ahead of the first MFMA, on the critical path behind the LDS wait
template <int NM, int ND>
Y2_DEV void mfma_interleave() {
#pragma unroll
    for (int k = 0; k < NM; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 // one MFMA
        if (k < ND) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // one LDS read (next k-group's fragments)
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                 // up to three VALU
    }
}
Y2_DEV u32x4 lds_read16(uint32_t lds_addr) {
    return *(const __attribute__((address_space(3))) u32x4*)(uintptr_t)lds_addr;
}

// KS: the K range (input-channel chunks) is split over a.ks_splits workgroups per tile; every one leaves its fp32 partial
// tile in a.ks_scratch [split][M][ldy] and conv_ks_finish_kernel adds them in split order (bias, rounding, statistics
// or the folded inference batch norm there): launches of a few hundred pixels are bound by the SERIAL K loop of their
// few workgroups (~110 ns per tap step, 72 us for K = 9216 whatever the tile), not by anything a roofline names.
// PL2 (f16x2 mode): as conv_haloq16_kernel -- an LDS image row is [BKB/2 bytes of the hi plane | BKB/2 of the lo plane], the
// first half of a row's 32-byte k-groups is x_hi, the second half x_lo, and a tap step runs the three plane products
template <typename T, int WP, int WC, int TP, int TC, int BKB, bool ADB, bool CPT, int TAPS, bool KS = false, bool PL2 = false>
__global__ __launch_bounds__(WP* WC * 64) void conv_haloq_kernel(ConvArgs a, int arows) {
    static_assert(TAPS == 9 || (TAPS == 1 && CPT), "1x1 filters run on the compact image (no halo, no border taps)");
    typedef typename Elem<T>::frag frag_t;
    typedef typename Types<T>::op_t OT;      // f16x2 mode: the operands are half planes of fp32-width rows (common.h)
    typedef typename Types<T>::out_t YT;
    constexpr bool SPLIT = Types<T>::kSplit;
    static_assert(!(SPLIT && KS), "the K split of small launches is not built for the split-operand mode");
    static_assert(!PL2 || (SPLIT && Types<T>::kPasses == 3 && !CPT && (BKB == 128 || BKB == 64)), "the two-plane form: split operands, bordered image");
    constexpr int NW = WP * WC, BP = WP * TP * 32, BC = WC * TC * 32, SZ = sizeof(T);
    constexpr int LPR = BKB / 16, RPI = 64 / LPR, RPB = 256 / BKB, KG = BKB / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
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
    const int pitch = a.W + 1, hw = a.H * a.W;
    const char* __restrict__ xg = (const char*)a.x;

    auto bpos = [&](int p) -> long {
        const int n = p / hw, rem = p - n * hw;
        const int h = rem / a.W, ww = rem - h * a.W;
        return (long)bpix(n, h, ww, a.H, a.W);
    };
    // pooled layers in the inference fold (ConvArgs::aff_pool): the tile's positions are in window-major order
    auto pixq = [&](int q) -> int { return a.aff_pool ? pool_order_pixel(q, a.H, a.W) : q; };
    const int p_last = (m0 + BP - 1 < a.M) ? m0 + BP - 1 : a.M - 1;
    const long lo = CPT ? 0 : bpos(pixq(m0)) - pitch - 1;
    const int nrows = CPT ? arows : (int)(bpos(pixq(p_last)) + pitch + 1 - lo) + 1;
    const int npieces = (nrows + RPI - 1) / RPI;
    const int abytes = arows * BKB;
    const int rowbytes = a.C * SZ;
    const int npl = a.C * (int)sizeof(OT) / BKB;     // K chunks per operand plane (f16x2: three plane passes, common.h)
    // compact image: [16 zero rows][image buffer(s): row lambda = pixel (m0 - W - 1 + lambda)][cell index per row]
    char* const img0 = smem + (CPT ? kZeroRows * BKB : 0);
    const uint32_t* const cell_tab = (const uint32_t*)(img0 + (ADB ? 2 : 1) * abytes);
    if (CPT) {
        uint32_t* tab = (uint32_t*)(img0 + (ADB ? 2 : 1) * abytes);
        for (int r = tid; r < arows; r += NW * 64) {
            int q = m0 - (TAPS == 9 ? a.W + 1 : 0) + r;
            q = q < 0 ? 0 : (q > a.M - 1 ? a.M - 1 : q);      // rows outside the tensor are never read as image cells
            tab[r] = (uint32_t)bpos(q);
        }
        for (int o = tid * 16; o < kZeroRows * BKB; o += NW * 64 * 16) *(u32x4*)(smem + o) = u32x4{0u, 0u, 0u, 0u};
        __syncthreads();
    }

    const int smem_lds = (int)(uintptr_t)(__attribute__((address_space(3))) char*)smem;   // LDS address of smem
    const int lrow = lane / LPR, lslot = lane % LPR;
    constexpr int KGH = KG / 2;                            // PL2: 32-byte k-groups per plane and chunk
    auto issueA = [&](int c, int ab) {
        const char* xs = xg + lo * (long)rowbytes + (PL2 ? (long)c * (BKB / 2) : (long)split_act_chunk<SPLIT>(c, npl) * BKB);
        char* dst = img0 + ab * abytes;
        if (CPT) {
            // eight table entries first, then their DMAs: an LDS-DMA is an LDS write to the compiler, so a table read
            // placed behind one waits for it -- piece by piece that was a chain of LDS latencies per chunk
            for (int i0 = w; i0 < npieces; i0 += 8 * NW) {
                uint32_t cell[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = i0 + k * NW;
                    cell[k] = cell_tab[(i < npieces ? i : i0) * RPI + lrow];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = i0 + k * NW;
                    if (i < npieces) {
                        const int row = i * RPI + lrow;
                        const uint32_t sw = (uint32_t)((lslot ^ ((row / RPB) % LPR)) * 16);
                        glds16(xs + (size_t)cell[k] * (size_t)rowbytes + sw, dst + i * 1024);
                    }
                }
            }
        } else {
            for (int i = w; i < npieces; i += NW) {
                const int row = i * RPI + lrow;
                const uint32_t src = (uint32_t)(lslot ^ ((row / RPB) % LPR));     // 16-byte chunk of the row this slot holds
                // PL2: the first LPR / 2 chunks come from the hi plane, the others from the lo plane (a.C halves further on)
                const uint32_t off = (uint32_t)row * (uint32_t)rowbytes +
                                     (PL2 ? (src % (LPR / 2)) * 16u + (src / (LPR / 2)) * (uint32_t)(a.C * 2) : src * 16u);
                glds16(xs + off, dst + i * 1024);
            }
        }
    };
    // filter fragments: [cout tile of 32][tap][k-group of 32 bytes][lane][16 B]
    const int kgrow = rowbytes / 32;                     // k-groups per tap
    const char* wbase[TC];                               // this lane's 16 bytes of the wave's cout tiles
#pragma unroll
    for (int i = 0; i < TC; ++i)
        wbase[i] = (const char*)a.w + ((size_t)(n0 / 32 + wc * TC + i) * TAPS * kgrow * 64 + lane) * 16;
    static_assert(KG <= 4, "immediate offsets of the fragment loads");
    auto loadB = [&](int c, int t, u32x4 (&fb)[TC][KG]) {
        if constexpr (PL2) {      // k-groups c KGH .. of the hi plane, then the same of the lo plane (kgrow / 2 groups further on)
            const size_t off = (size_t)(t * kgrow + c * KGH) * 1024, lod = (size_t)(kgrow / 2) * 1024;
#pragma unroll
            for (int i = 0; i < TC; ++i) {
                const char* b = wbase[i] + off;
                frag_load<0>(fb[i][0], b);
                if constexpr (KGH > 1) frag_load<1024>(fb[i][1], b);
                frag_load<0>(fb[i][KGH], b + lod);
                if constexpr (KGH > 1) frag_load<1024>(fb[i][KGH + 1], b + lod);
            }
            return;
        }
        const size_t off = (size_t)(t * kgrow + split_flt_chunk<SPLIT>(c, npl) * KG) * 1024;
#pragma unroll
        for (int i = 0; i < TC; ++i) {
            const char* b = wbase[i] + off;
            frag_load<0>(fb[i][0], b);
            if constexpr (KG > 1) frag_load<1024>(fb[i][1], b);
            if constexpr (KG > 2) frag_load<2048>(fb[i][2], b);
            if constexpr (KG > 3) frag_load<3072>(fb[i][3], b);
        }
    };
    constexpr int NBL = TC * KG;                         // loads per loadB
    const int kA = w < npieces ? (npieces - 1 - w) / NW + 1 : 0;   // LDS-DMA pieces this wave issues per issueA

    const int r32 = lane & 31, hh = lane >> 5;
    int rowtlB[TP], fmk[TP];   // byte offset of the top-left tap's image row; CPT: which borders the pixel touches
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        int p = m0 + (wp * TP + j) * 32 + r32;
        if (p > a.M - 1) p = a.M - 1;
        if (CPT) {
            const int rem = p % hw, h = rem / a.W, ww = rem - h * a.W;
            fmk[j] = (ww == 0 ? 1 : 0) | (ww == a.W - 1 ? 2 : 0) | (h == 0 ? 4 : 0) | (h == a.H - 1 ? 8 : 0);
            rowtlB[j] = (p - m0) * BKB;
        } else {
            fmk[j] = 0;
            rowtlB[j] = (int)(bpos(pixq(p)) - pitch - 1 - lo) * BKB;
        }
    }
    // LDS byte offset (from smem) and swizzle key of every pixel fragment row for tap (kh_, kw_) of chunk cc.  A tap that
    // leaves the image reads zero row (lambda & 15): the bank position of the cell the uniform shift points at
    auto tap_addr = [&](int kh_, int kw_, int cc, int (&ao)[TP]) {
        const int shiftB = (kh_ * (CPT ? a.W : pitch) + kw_) * BKB;
        const int tapm = TAPS == 9 ? ((kw_ == 0 ? 1 : 0) | (kw_ == 2 ? 2 : 0) | (kh_ == 0 ? 4 : 0) | (kh_ == 2 ? 8 : 0)) : 0;
        const int bufB = smem_lds + (CPT ? kZeroRows * BKB : 0) + (ADB ? (cc & 1) : 0) * abytes;
#pragma unroll
        for (int j = 0; j < TP; ++j) {
            const int lamB = rowtlB[j] + shiftB;
            const int sw = (lamB >> 8) & (LPR - 1);              // (row / RPB) % LPR: RPB rows = one 256-byte bank row
            const int rowB = (CPT && (fmk[j] & tapm)) ? smem_lds + (lamB & (15 * BKB)) : lamB + bufB;
            // address of k-group 0; group g sits at this address ^ (g * 32): (2g + hh) ^ sw = (hh ^ sw) ^ 2g
            ao[j] = rowB + ((hh ^ sw) << 4);
        }
    };

    f32x16 acc[TC][TP];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    int c_begin = 0, nchunks = PL2 ? a.C * (int)sizeof(OT) / (BKB / 2) : Types<T>::kPasses * npl;     // this workgroup's chunk range [c_begin, nchunks)
    if constexpr (KS) {
        const int per = (nchunks + a.ks_splits - 1) / a.ks_splits;
        c_begin = split * per;
        nchunks = c_begin + per < nchunks ? c_begin + per : nchunks;
        if (c_begin > nchunks) c_begin = nchunks;
    }
    const int steps = (nchunks - c_begin) * TAPS;
    u32x4 fbq[2][TC][KG];
    int aoffq[2][TP];                          // fragment-row addresses of the current / the next tap step
    if (steps > 0) {
        issueA(c_begin, ADB ? (c_begin & 1) : 0);
        loadB(c_begin, 0, fbq[0]);
    }
    tap_addr(0, 0, c_begin, aoffq[0]);
    int c = c_begin, t = 0, kh = 0, kw = 0;
    auto step = [&](auto par, int s) {
        constexpr int P = decltype(par)::value;
        if (t == 0) {
            if (!ADB && c > c_begin) {
                __builtin_amdgcn_s_barrier();      // everyone is done with the single image buffer
                asm volatile("" ::: "memory");
                issueA(c, 0);
            }
            wait_vmcnt<0>();                       // this chunk's image pieces (and all older loads) have landed
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        // next step's filter fragments first (so they never queue behind an image), then the next image
        int tn = t + 1, cn = c, khn = kh, kwn = kw + 1;
        if (kwn == 3) { kwn = 0; ++khn; }
        if (tn == TAPS) { tn = 0; ++cn; khn = 0; kwn = 0; }
        const bool more = s + 1 < steps;
        if (more) loadB(cn, tn, fbq[P ^ 1]);
        const bool dma = ADB && t == 0 && c + 1 < nchunks;
        if (dma) issueA(c + 1, (c + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);

        auto load_frags = [&](int g, frag_t (&fp)[TP]) {
#pragma unroll
            for (int j = 0; j < TP; ++j)
                fp[j] = __builtin_bit_cast(frag_t, lds_read16((uint32_t)(aoffq[P][j] ^ (g * 32))));
        };
        frag_t fp0[TP], fp1[TP];
        load_frags(0, fp0);
        // this step's filter fragments were requested one step ago; younger than them are the image pieces of that
        // step (t == 1 now) and the fragments just requested.  (t == 0: drained by the chunk wait above.)
        if (t != 0) {
            if (t == 1 && ADB && c + 1 < nchunks) wait_vmcnt_dyn((more ? NBL : 0) + kA);
            else if (more) wait_vmcnt<NBL>();
            else wait_vmcnt<0>();
        }
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int g = 0; g < KG; ++g) frag_ready(fbq[P][i][g]);
        __builtin_amdgcn_sched_barrier(0);
        // the next step's addresses: VALU work that issues beside this step's first MFMAs instead of in front of
        // the next step's first LDS reads
        tap_addr(khn, kwn, cn, aoffq[P ^ 1]);
        if constexpr (PL2) {
            // fp0 = x_hi of k-group 0 (read above); per k-group q: w_hi x_hi, then w_hi x_lo and w_lo x_hi; the next group's
            // x_hi goes to a third register set while the small terms run
            frag_t fp2[TP];
#pragma unroll
            for (int q = 0; q < KGH; ++q) {
                frag_t(&xh)[TP] = (q & 1) ? fp2 : fp0;
                frag_t(&xn)[TP] = (q & 1) ? fp0 : fp2;
                load_frags(KGH + q, fp1);                                        // x_lo of this k-group
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int j = 0; j < TP; ++j) mma32(acc[i][j], __builtin_bit_cast(frag_t, fbq[P][i][q]), xh[j]);
                if (q == 0) mfma_interleave<TC * TP, TP>();
                __builtin_amdgcn_sched_barrier(0);
                if (q + 1 < KGH) load_frags(q + 1, xn);
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int j = 0; j < TP; ++j) {
                        mma32(acc[i][j], __builtin_bit_cast(frag_t, fbq[P][i][q]), fp1[j]);
                        mma32(acc[i][j], __builtin_bit_cast(frag_t, fbq[P][i][KGH + q]), xh[j]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else
#pragma unroll
        for (int g = 0; g < KG; g += 2) {
            if (g + 1 < KG) load_frags(g + 1, fp1);
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) mma32(acc[i][j], __builtin_bit_cast(frag_t, fbq[P][i][g]), fp0[j]);
            if (g == 0) mfma_interleave<TC * TP, TP>();
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < KG) {
                if (g + 2 < KG) load_frags(g + 2, fp0);
#pragma unroll
                for (int i = 0; i < TC; ++i)
#pragma unroll
                    for (int j = 0; j < TP; ++j) mma32(acc[i][j], __builtin_bit_cast(frag_t, fbq[P][i][g + 1]), fp1[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (++kw == 3) { kw = 0; ++kh; }
        if (++t == TAPS) { t = 0; kh = 0; kw = 0; ++c; }
    };
    for (int s = 0; s < steps; s += 2) {
        step(IntC<0>{}, s);
        if (s + 1 < steps) step(IntC<1>{}, s + 1);
    }
    if constexpr (KS) {
        // fp32 partial tile: 4 consecutive couts (registers 4 q4 .. 4 q4 + 3) per 16-byte store
        float* const part = a.ks_scratch + (size_t)split * a.M * a.ldy;
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                const int p = m0 + (wp * TP + j) * 32 + r32;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int co = n0 + (wc * TC + i) * 32 + 8 * q4 + 4 * hh;
                    if (p < a.M && co < a.ldy)
                        *(f32x4*)(part + (size_t)p * a.ldy + co) =
                            f32x4{acc[i][j][4 * q4], acc[i][j][4 * q4 + 1], acc[i][j][4 * q4 + 2], acc[i][j][4 * q4 + 3]};
                }
            }
        return;
    } else {
        __syncthreads();
        if constexpr (SPLIT) {      // the filters were packed times kSplitWScale (a power of two)
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int j = 0; j < TP; ++j) acc[i][j] *= kSplitWScaleInv;
        }
        conv_epilogue<YT, WP, WC, TP, TC, 0, Types<T>::kBwF32>(a, acc, smem, w, lane, m0, n0, pt, ct);
    }
}

// K-split partial tiles -> the launch's result.  One thread = one pixel x 4 couts: the splits added in order, bias, the
// rounding to T of the un-split epilogue; then either the folded inference batch norm into the consumer's bordered
// tensor (ConvArgs::aff_*) or y [M][ldy].  Batch-norm statistics: conv_ks_stats_kernel over the stored y.
template <typename T>
__global__ __launch_bounds__(256) void conv_ks_finish_kernel(ConvArgs a, int splits) {
    const int c4n = a.ldy / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)a.M * c4n) return;
    const int m = (int)(idx / c4n), co = (int)(idx % c4n) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < splits; ++sp) v += *(const f32x4*)(a.ks_scratch + ((size_t)sp * a.M + m) * a.ldy + co);
    T o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float b = (a.bias && co + e < a.Cout) ? a.bias[co + e] : 0.f;
        o[e] = Elem<T>::from_f32(v[e] + b);
    }
    if (a.aff_out) {
        const uint32_t r = (uint32_t)(((uint64_t)(uint32_t)m * a.aff_magW) >> a.aff_shW);
        const uint32_t n = (uint32_t)(((uint64_t)r * a.aff_magH) >> a.aff_shH);
        const size_t bp = (size_t)m + r + (size_t)(n + 1) * (uint32_t)(a.W + 1) + 1;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            o[e] = Elem<T>::from_f32(leaky_s(Elem<T>::to_f32(o[e]) * a.aff_scale[co + e] + a.aff_shift[co + e], a.aff_slope));
        T* dst = (T*)a.aff_out + bp * a.ldy + co;
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[e] = o[e];
    } else {
        T* dst = (T*)a.y + (size_t)m * a.ldy + co;
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[e] = o[e];
    }
}
// (count, mean, M2) records over the stored values, one per 128 pixels (bn_finalize merges them as it merges the tile
// records of the un-split epilogue): block = 64 channels x 4 pixel slices of one 128-pixel group, sums about the bias
constexpr int kKsRec = 128;
template <typename T>
__global__ __launch_bounds__(256) void conv_ks_stats_kernel(ConvArgs a) {
    __shared__ float red[4][64][2];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
    const int m0 = blockIdx.y * kKsRec;
    const int m1 = m0 + kKsRec < a.M ? m0 + kKsRec : a.M;
    const bool cv = c < a.ldy;
    const float piv = (cv && a.bias && c < a.Cout) ? a.bias[c] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    if (cv) {
        const T* yp = (const T*)a.y + c;
#pragma unroll 8
        for (int m = m0 + sl; m < m1; m += 4) {
            const float d = Elem<T>::to_f32(yp[(size_t)m * a.ldy]) - piv;
            s1 += d;
            s2 = fmaf(d, d, s2);
        }
    }
    red[sl][threadIdx.x & 63][0] = s1;
    red[sl][threadIdx.x & 63][1] = s2;
    __syncthreads();
    if (sl == 0 && cv) {
        const int k = threadIdx.x & 63;
        const double S1 = ((double)red[0][k][0] + red[1][k][0]) + ((double)red[2][k][0] + red[3][k][0]);
        const double S2 = ((double)red[0][k][1] + red[1][k][1]) + ((double)red[2][k][1] + red[3][k][1]);
        const double n = (double)(m1 - m0), md = S1 / n, m2 = S2 - S1 * md;
        a.part_mean[(size_t)blockIdx.y * a.ldy + c] = (float)((double)piv + md);
        a.part_m2[(size_t)blockIdx.y * a.ldy + c] = (float)(m2 > 0.0 ? m2 : 0.0);
        if (c == 0) a.part_cnt[blockIdx.y] = (float)(m1 - m0);
    }
}

// ---------------------------------------------------------------------------
// Same kernel on 16x16x32 MFMA tiles (v_mfma_f32_16x16x32_f16/bf16, 16x16x4 f32): identical FLOPs, cycles
// and operand traffic per wave tile, but the chip holds a higher clock on this shape under load
// (MI355X_MICROARCH.md "DVFS give-back" (7); measured here with the results discarded: +7 % on the
// 1024-channel layers).  Filters in the 16-row fragment order (pack.hip frag_chunk16):
//   [cout tile of 16][tap][k-group of 64 bytes][lane = (16-byte chunk)*16 + cout%16][16 B].
// TP / TC still count 32-wide units, so tiles, LDS image and epilogue patch are those of the kernel above.
// ---------------------------------------------------------------------------
// PL2 (f16x2 mode, round 5): BOTH operand planes of a K chunk are staged together -- an LDS image row is [64 B of the hi
// plane | 64 B of the lo plane], the two 64-byte k-groups of a row ARE the two planes -- and a tap step runs the three
// plane products of its 32 channels on them (w_hi x_hi, w_hi x_lo, w_lo x_hi) instead of the K loop running three plane
// passes: the hi plane is staged, read from LDS and its filter fragments fetched ONCE instead of twice (2/3 of the LDS-DMA
// bytes, fragment reads, filter loads, tap steps and barriers per matrix instruction).
template <typename T, int WP, int WC, int TP, int TC, int BKB, bool ADB, bool CPT, int TAPS, bool PL2 = false>
__global__ __launch_bounds__(WP* WC * 64) void conv_haloq16_kernel(ConvArgs a, int arows) {
    static_assert(TAPS == 9 || (TAPS == 1 && CPT), "1x1 filters run on the compact image (no halo, no border taps)");
    typedef typename Elem<T>::frag frag_t;
    typedef typename Types<T>::op_t OT;      // f16x2 mode: the operands are half planes of fp32-width rows (common.h)
    typedef typename Types<T>::out_t YT;
    constexpr bool SPLIT = Types<T>::kSplit;
    constexpr int NW = WP * WC, BP = WP * TP * 32, BC = WC * TC * 32, SZ = sizeof(T);
    constexpr int LPR = BKB / 16, RPI = 64 / LPR, RPB = 256 / BKB, KG = BKB / 64;   // k-groups of 64 bytes
    constexpr int TP16 = 2 * TP, TC16 = 2 * TC;
    static_assert(!PL2 || (SPLIT && Types<T>::kPasses == 3 && BKB == 128 && !CPT), "the two-plane form: split operands, 128-byte image rows");
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
    // pooled layers in the inference fold (ConvArgs::aff_pool): the tile's positions are in window-major order
    auto pixq = [&](int q) -> int { return a.aff_pool ? pool_order_pixel(q, a.H, a.W) : q; };
    const int p_last = (m0 + BP - 1 < a.M) ? m0 + BP - 1 : a.M - 1;
    const long lo = CPT ? 0 : bpos(pixq(m0)) - pitch - 1;
    const int nrows = CPT ? arows : (int)(bpos(pixq(p_last)) + pitch + 1 - lo) + 1;
    const int npieces = (nrows + RPI - 1) / RPI;
    const int abytes = arows * BKB;
    const int rowbytes = a.C * SZ;
    const int npl = a.C * (int)sizeof(OT) / BKB;     // K chunks per operand plane (f16x2: three plane passes, common.h)
    // compact image: [16 zero rows][image buffer(s): row lambda = pixel (m0 - W - 1 + lambda)][cell index per row]
    char* const img0 = smem + (CPT ? kZeroRows * BKB : 0);
    const uint32_t* const cell_tab = (const uint32_t*)(img0 + (ADB ? 2 : 1) * abytes);
    if (CPT) {
        uint32_t* tab = (uint32_t*)(img0 + (ADB ? 2 : 1) * abytes);
        for (int r = tid; r < arows; r += NW * 64) {
            int q = m0 - (TAPS == 9 ? a.W + 1 : 0) + r;
            q = q < 0 ? 0 : (q > a.M - 1 ? a.M - 1 : q);      // rows outside the tensor are never read as image cells
            tab[r] = (uint32_t)bpos(q);
        }
        for (int o = tid * 16; o < kZeroRows * BKB; o += NW * 64 * 16) *(u32x4*)(smem + o) = u32x4{0u, 0u, 0u, 0u};
        __syncthreads();
    }

    const int smem_lds = (int)(uintptr_t)(__attribute__((address_space(3))) char*)smem;   // LDS address of smem
    const int lrow = lane / LPR, lslot = lane % LPR;
    auto issueA = [&](int c, int ab) {
        const char* xs = xg + lo * (long)rowbytes + (PL2 ? (long)c * 64 : (long)split_act_chunk<SPLIT>(c, npl) * BKB);
        char* dst = img0 + ab * abytes;
        if (CPT) {
            // eight table entries first, then their DMAs: an LDS-DMA is an LDS write to the compiler, so a table read
            // placed behind one waits for it -- piece by piece that was a chain of LDS latencies per chunk
            for (int i0 = w; i0 < npieces; i0 += 8 * NW) {
                uint32_t cell[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = i0 + k * NW;
                    cell[k] = cell_tab[(i < npieces ? i : i0) * RPI + lrow];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = i0 + k * NW;
                    if (i < npieces) {
                        const int row = i * RPI + lrow;
                        const uint32_t sw = (uint32_t)((lslot ^ ((row / RPB) % LPR)) * 16);
                        glds16(xs + (size_t)cell[k] * (size_t)rowbytes + sw, dst + i * 1024);
                    }
                }
            }
        } else {
            for (int i = w; i < npieces; i += NW) {
                const int row = i * RPI + lrow;
                const uint32_t src = (uint32_t)(lslot ^ ((row / RPB) % LPR));     // 16-byte chunk of the row this slot holds
                // PL2: chunks 0-3 come from the hi plane, 4-7 from the lo plane (a.C halves further on)
                const uint32_t off = (uint32_t)row * (uint32_t)rowbytes +
                                     (PL2 ? (src & 3u) * 16u + (src >> 2) * (uint32_t)(a.C * 2) : src * 16u);
                glds16(xs + off, dst + i * 1024);
            }
        }
    };
    const int kgrow = rowbytes / 64;                     // 64-byte k-groups per tap
    const char* wbase[TC16];                             // this lane's 16 bytes of the wave's cout tiles
#pragma unroll
    for (int i = 0; i < TC16; ++i)
        wbase[i] = (const char*)a.w + ((size_t)(n0 / 16 + wc * TC16 + i) * TAPS * kgrow * 64 + lane) * 16;
    static_assert(KG <= 4, "immediate offsets of the fragment loads");
    auto loadB = [&](int c, int t, u32x4 (&fb)[TC16][KG]) {
        if constexpr (PL2) {      // k-group c of the hi plane and of the lo plane (kgrow / 2 groups further on)
            const size_t off = (size_t)(t * kgrow + c) * 1024, lod = (size_t)(kgrow / 2) * 1024;
#pragma unroll
            for (int i = 0; i < TC16; ++i) {
                const char* b = wbase[i] + off;
                frag_load<0>(fb[i][0], b);
                frag_load<0>(fb[i][1], b + lod);
            }
            return;
        }
        const size_t off = (size_t)(t * kgrow + split_flt_chunk<SPLIT>(c, npl) * KG) * 1024;
#pragma unroll
        for (int i = 0; i < TC16; ++i) {
            const char* b = wbase[i] + off;
            frag_load<0>(fb[i][0], b);
            if constexpr (KG > 1) frag_load<1024>(fb[i][1], b);
            if constexpr (KG > 2) frag_load<2048>(fb[i][2], b);
            if constexpr (KG > 3) frag_load<3072>(fb[i][3], b);
        }
    };
    constexpr int NBL = TC16 * KG;                       // loads per loadB
    const int kA = w < npieces ? (npieces - 1 - w) / NW + 1 : 0;   // LDS-DMA pieces this wave issues per issueA

    const int r16 = lane & 15, kc = lane >> 4;
    const int c16 = CPT ? perm16(r16) : r16;      // pixel offset of this lane's MFMA column
    int rowtlB[TP16], fmk[TP16];   // byte offset of the top-left tap's image row; CPT: which borders the pixel touches
#pragma unroll
    for (int j = 0; j < TP16; ++j) {
        int p = m0 + (wp * TP16 + j) * 16 + c16;
        if (p > a.M - 1) p = a.M - 1;
        if (CPT) {
            const int rem = p % hw, h = rem / a.W, ww = rem - h * a.W;
            fmk[j] = (ww == 0 ? 1 : 0) | (ww == a.W - 1 ? 2 : 0) | (h == 0 ? 4 : 0) | (h == a.H - 1 ? 8 : 0);
            rowtlB[j] = (p - m0) * BKB;
        } else {
            fmk[j] = 0;
            rowtlB[j] = (int)(bpos(pixq(p)) - pitch - 1 - lo) * BKB;
        }
    }
    // LDS byte offset (from smem) and swizzle key of every pixel fragment row for tap (kh_, kw_) of chunk cc.  A tap that
    // leaves the image reads zero row (lambda & 15): the bank position of the cell the uniform shift points at
    auto tap_addr = [&](int kh_, int kw_, int cc, int (&ao)[TP16]) {
        const int shiftB = (kh_ * (CPT ? a.W : pitch) + kw_) * BKB;
        const int tapm = TAPS == 9 ? ((kw_ == 0 ? 1 : 0) | (kw_ == 2 ? 2 : 0) | (kh_ == 0 ? 4 : 0) | (kh_ == 2 ? 8 : 0)) : 0;
        const int bufB = smem_lds + (CPT ? kZeroRows * BKB : 0) + (ADB ? (cc & 1) : 0) * abytes;
#pragma unroll
        for (int j = 0; j < TP16; ++j) {
            const int lamB = rowtlB[j] + shiftB;
            const int sw = (lamB >> 8) & (LPR - 1);              // (row / RPB) % LPR: RPB rows = one 256-byte bank row
            const int rowB = (CPT && (fmk[j] & tapm)) ? smem_lds + (lamB & (15 * BKB)) : lamB + bufB;
            // address of k-group 0; group g sits at this address ^ (g * 64): (4g + kc) ^ sw = (kc ^ sw) ^ 4g
            ao[j] = rowB + ((kc ^ sw) << 4);
        }
    };

    f32x4 acc[TC16][TP16];
#pragma unroll
    for (int i = 0; i < TC16; ++i)
#pragma unroll
        for (int j = 0; j < TP16; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = PL2 ? a.C * (int)sizeof(OT) / 64 : Types<T>::kPasses * npl;
    const int steps = nchunks * TAPS;
    u32x4 fbq[2][TC16][KG];
    int aoffq[2][TP16];                        // fragment-row addresses of the current / the next tap step
    issueA(0, 0);
    loadB(0, 0, fbq[0]);
    tap_addr(0, 0, 0, aoffq[0]);
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
        int tn = t + 1, cn = c, khn = kh, kwn = kw + 1;
        if (kwn == 3) { kwn = 0; ++khn; }
        if (tn == TAPS) { tn = 0; ++cn; khn = 0; kwn = 0; }
        const bool more = s + 1 < steps;
        if (more) loadB(cn, tn, fbq[P ^ 1]);
        const bool dma = ADB && t == 0 && c + 1 < nchunks;
        if (dma) issueA(c + 1, (c + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);

        auto load_frags = [&](int g, frag_t (&fp)[TP16]) {
#pragma unroll
            for (int j = 0; j < TP16; ++j)
                fp[j] = __builtin_bit_cast(frag_t, lds_read16((uint32_t)(aoffq[P][j] ^ (g * 64))));
        };
        frag_t fp0[TP16], fp1[TP16];
        load_frags(0, fp0);
        // this step's filter fragments were requested one step ago; younger than them are the image pieces of that
        // step (t == 1 now) and the fragments just requested.  (t == 0: drained by the chunk wait above.)
        if (t != 0) {
            if (t == 1 && ADB && c + 1 < nchunks) wait_vmcnt_dyn((more ? NBL : 0) + kA);
            else if (more) wait_vmcnt<NBL>();
            else wait_vmcnt<0>();
        }
#pragma unroll
        for (int i = 0; i < TC16; ++i)
#pragma unroll
            for (int g = 0; g < KG; ++g) frag_ready(fbq[P][i][g]);
        __builtin_amdgcn_sched_barrier(0);
        tap_addr(khn, kwn, cn, aoffq[P ^ 1]);   // next step's addresses beside this step's first MFMAs
        if constexpr (PL2) {
            // fp0 = this chunk's x_hi fragments, fp1 = x_lo; fbq[.][i][0] = w_hi, [1] = w_lo: hi hi, then the two small terms
            load_frags(1, fp1);
#pragma unroll
            for (int i = 0; i < TC16; ++i)
#pragma unroll
                for (int j = 0; j < TP16; ++j) mma16(acc[i][j], __builtin_bit_cast(frag_t, fbq[P][i][0]), fp0[j]);
            mfma_interleave<TC16 * TP16, TP16>();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TC16; ++i)
#pragma unroll
                for (int j = 0; j < TP16; ++j) {
                    mma16(acc[i][j], __builtin_bit_cast(frag_t, fbq[P][i][0]), fp1[j]);
                    mma16(acc[i][j], __builtin_bit_cast(frag_t, fbq[P][i][1]), fp0[j]);
                }
            __builtin_amdgcn_sched_barrier(0);
        } else
#pragma unroll
        for (int g = 0; g < KG; g += 2) {
            if (g + 1 < KG) load_frags(g + 1, fp1);
#pragma unroll
            for (int i = 0; i < TC16; ++i)
#pragma unroll
                for (int j = 0; j < TP16; ++j) mma16(acc[i][j], __builtin_bit_cast(frag_t, fbq[P][i][g]), fp0[j]);
            if (g == 0) mfma_interleave<TC16 * TP16, TP16>();
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < KG) {
                if (g + 2 < KG) load_frags(g + 2, fp0);
#pragma unroll
                for (int i = 0; i < TC16; ++i)
#pragma unroll
                    for (int j = 0; j < TP16; ++j) mma16(acc[i][j], __builtin_bit_cast(frag_t, fbq[P][i][g + 1]), fp1[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (++kw == 3) { kw = 0; ++kh; }
        if (++t == TAPS) { t = 0; kh = 0; kw = 0; ++c; }
    };
    for (int s = 0; s < steps; s += 2) {
        step(IntC<0>{}, s);
        if (s + 1 < steps) step(IntC<1>{}, s + 1);
    }
    __syncthreads();
    if constexpr (SPLIT) {
#pragma unroll
        for (int i = 0; i < TC16; ++i)
#pragma unroll
            for (int j = 0; j < TP16; ++j) acc[i][j] *= kSplitWScaleInv;
    }
    // The fp32 patch of a 384 x 128 tile (8 waves x 96 pixels x 64 couts x 4 B) does not fit LDS: the modes with fp32 outputs
    // (split operands, exact f32) run the epilogue in TWO passes over halves of every wave's couts
    constexpr bool EPI2 = sizeof(YT) == 4 && (TC % 2 == 0) && EpiCfg<YT, WP, WC, TP, TC>::LDS > 160 * 1024;
    if constexpr (EPI2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 acch[TC16 / 2][TP16];
#pragma unroll
            for (int i = 0; i < TC16 / 2; ++i)
#pragma unroll
                for (int j = 0; j < TP16; ++j) acch[i][j] = acc[h * (TC16 / 2) + i][j];
            if (h) __syncthreads();      // pass 0's patch and the statistics scratch that aliases it are dead
            conv_epilogue16<YT, WP, WC, TP, TC / 2, CPT, Types<T>::kBwF32>(a, acch, smem, w, lane, m0, n0, pt, ct, TC * 32, h * (TC / 2) * 32);
        }
    } else {
        conv_epilogue16<YT, WP, WC, TP, TC, CPT, Types<T>::kBwF32>(a, acc, smem, w, lane, m0, n0, pt, ct);
    }
}

// compact image: pixels [m0 - W - 1, m0 + BP + W], rounded up to whole 16-row groups
static int haloq_rows_compact(int W, int BP, int taps = 9) { return (BP + (taps == 9 ? 2 * W + 2 : 0) + 15) / 16 * 16; }
static bool halo_compact() {
    static const bool on = getenv("Y2_HALO_COMPACT") && atoi(getenv("Y2_HALO_COMPACT")) != 0;
    return on;
}
// pool: the tile's pixels in window-major order (ConvArgs::aff_pool) -- a run of BP / 4 windows touches at most
// ceil((Wo - 1 + BP / 4) / Wo) row pairs
static int haloq_rows(int H, int W, int BP, int RPI, int pool = 0) {
    if (pool) {
        const int pitch = W + 1, Wo = W / 2, Ho = H / 2, nwin = BP / 4;
        const int pairs = (nwin + 2 * Wo - 2) / Wo;
        const int img_cross = (nwin - 1) / (Ho * Wo) + 1;
        const int nrows = 2 * pairs * pitch + img_cross * pitch + 2 * (pitch + 1) + 1;
        return (nrows + RPI - 1) / RPI * RPI;
    }
    const int pitch = W + 1;
    const int rows_cross = (BP - 1) / W + 1;
    const int img_cross = (BP - 1) / (H * W) + 1;
    const int span = (BP - 1) + rows_cross + img_cross * pitch;
    const int nrows = span + 2 * (pitch + 1) + 1;
    return (nrows + RPI - 1) / RPI * RPI;
}

template <int BKB>
static size_t haloq_lds(int arows, bool adb, bool cpt) {
    return (size_t)(adb ? 2 : 1) * arows * BKB + (cpt ? (size_t)kZeroRows * BKB + (size_t)arows * 4 : 0);
}
template <typename T, int WP, int WC, int TP, int TC, int BKB, bool ADB, bool M16, bool CPT, int TAPS = 9>
static hipError_t haloq_launch(const ConvArgs& a, hipStream_t s) {
    typedef typename Types<T>::out_t YT_;
    constexpr bool EPI2 = sizeof(YT_) == 4 && M16 && (TC % 2 == 0) && EpiCfg<YT_, WP, WC, TP, TC>::LDS > 160 * 1024;
    typedef EpiCfg<YT_, WP, WC, TP, EPI2 ? TC / 2 : TC> Epi;     // (EPI2: the epilogue runs in two passes, conv_haloq16_kernel)
    constexpr int BP = WP * TP * 32, BC = WC * TC * 32, RPI = 64 / (BKB / 16);
    if ((a.C * (int)sizeof(typename Types<T>::op_t)) % BKB != 0) return hipErrorInvalidValue;
    const int arows = CPT ? haloq_rows_compact(a.W, BP, TAPS) : haloq_rows(a.H, a.W, BP, RPI, a.aff_pool);
    if (CPT && a.aff_pool) return hipErrorInvalidValue;
    if (CPT && arows > 0xFFFF) return hipErrorOutOfMemory;
    size_t lds = haloq_lds<BKB>(arows, ADB, CPT);
    if (lds < (size_t)Epi::LDS) lds = Epi::LDS;
    if (lds > 160 * 1024) return hipErrorOutOfMemory;
    void (*kern)(ConvArgs, int);
    // f16x2 on the 16x16 tiles: both operand planes per K chunk (conv_haloq16_kernel PL2); Y2_NO_CONV_PL2=1: three plane passes
    static const bool no_pl2 = getenv("Y2_NO_CONV_PL2") != nullptr;
    constexpr bool kPL2 = Types<T>::kPasses == 3 && M16 && BKB == 128 && !CPT && TAPS == 9;
    bool pl2 = false;
    if constexpr (M16) {
        kern = conv_haloq16_kernel<T, WP, WC, TP, TC, BKB, ADB, CPT, TAPS>;
        if constexpr (kPL2) {
            if (!no_pl2) { kern = conv_haloq16_kernel<T, WP, WC, TP, TC, BKB, ADB, CPT, TAPS, true>; pl2 = true; }
        }
    } else {
        kern = conv_haloq_kernel<T, WP, WC, TP, TC, BKB, ADB, CPT, TAPS>;
        constexpr bool kPL2q = Types<T>::kPasses == 3 && !CPT && TAPS == 9 && (BKB == 128 || BKB == 64);
        if constexpr (kPL2q) {
            if (!no_pl2) { kern = conv_haloq_kernel<T, WP, WC, TP, TC, BKB, ADB, CPT, TAPS, false, true>; pl2 = true; }
        }
    }
    static size_t attr[2] = {0, 0};
    if (lds > attr[pl2]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr[pl2] = lds;
    }
    const int nPT = (a.M + BP - 1) / BP;
    const int nCT = (a.Cout + BC - 1) / BC;
    hipLaunchKernelGGL(kern, dim3(nPT * nCT), dim3(WP * WC * 64), lds, s, a, arows);
    return hipGetLastError();
}

template <typename T, int WP, int WC, int TP, int TC, int BKB, bool M16 = false>
static hipError_t haloq_pick(const ConvArgs& a, hipStream_t s) {
    constexpr int BP = WP * TP * 32, RPI = 64 / (BKB / 16);
    const int nchunks = a.C * (int)sizeof(typename Types<T>::op_t) / BKB * Types<T>::kPasses;
    if (halo_compact()) {
        const int arows = haloq_rows_compact(a.W, BP);
        if (nchunks > 1 && haloq_lds<BKB>(arows, true, true) <= 150 * 1024)
            return haloq_launch<T, WP, WC, TP, TC, BKB, true, M16, true>(a, s);
        return haloq_launch<T, WP, WC, TP, TC, BKB, false, M16, true>(a, s);
    }
    const size_t arows = haloq_rows(a.H, a.W, BP, RPI, a.aff_pool);
    if (nchunks > 1 && 2 * arows * BKB <= 150 * 1024) return haloq_launch<T, WP, WC, TP, TC, BKB, true, M16, false>(a, s);
    return haloq_launch<T, WP, WC, TP, TC, BKB, false, M16, false>(a, s);
}

// 1x1 filters (round 3, Y2_HALOQ_1X1=1 -- built, measured, NOT the default): the same kernels with ONE tap per K-chunk on
// the compact image (rows = the tile's pixels, no halo, no border taps, conflict-free); the pixel tile crosses LDS once
// per cout tile, the filter fragments come straight from L2: 131 FLOP per staged byte (384 x 128 tile) against 65 of
// conv_igemm's 128 x 128 tiles.  MEASURED against conv_igemm on the same box (C4 shapes, fwd / dgrad us): 52x52 256->128
// 41.5 / 70.2 vs 42.3 / 63.9; 26x26 512->256 29.7 / 44.7 vs 31.5 / 42.2; 13x13 1024->512 30.8 / 31.6 vs 27.8 / 35.3 --
// a wash: with one tap per chunk every 0.4-us step waits for its own 48-KB image piece (the nine-tap loop had nine steps
// to hide it), and at 13x13 the 384 x 64 tiles that fill the chip halve the reuse again.  These layers are bound by
// the per-CU global->LDS fill rate and by M (10,816 pixels): the fix is a K split across workgroups, not another tile.
template <typename T, int WP, int WC, int TP, int TC, bool M16>
static hipError_t haloq_pick1(const ConvArgs& a, hipStream_t s) {
    constexpr int BP = WP * TP * 32, BKB = 128;
    const int nchunks = a.C * (int)sizeof(T) / BKB;
    const int arows = haloq_rows_compact(a.W, BP, 1);
    if (nchunks > 1 && haloq_lds<BKB>(arows, true, true) <= 150 * 1024)
        return haloq_launch<T, WP, WC, TP, TC, BKB, true, M16, true, 1>(a, s);
    return haloq_launch<T, WP, WC, TP, TC, BKB, false, M16, true, 1>(a, s);
}
template <typename T>
static hipError_t haloq_T1(const ConvArgs& a, hipStream_t s, int* bp) {
    const int kb = a.C * (int)sizeof(T);
    if ((kb % 128) != 0 || a.Cout <= 64) return hipErrorInvalidValue;
    *bp = 384;
    if (conv_filter_layout(1, a.W, kb, a.Cout, a.M, a.is_dgrad, (int)sizeof(T)) == 2) {
        if (sizeof(T) == 4) {     // the f32 epilogue patch of a 384 x 128 tile does not fit LDS
            *bp = 256;
            return haloq_pick1<T, 4, 2, 2, 2, true>(a, s);
        }
        return haloq_pick1<T, 4, 2, 3, 2, true>(a, s);
    }
    return haloq_pick1<T, 4, 2, 3, 1, false>(a, s);
}

// K split of a small launch (see conv_haloq_kernel<.., KS>): fewer than 128 workgroups of the 256 x 64 tile and at least
// four 128-byte K chunks; the depth fills ~one round of the chip.  hipErrorNotSupported: not this form.
static int haloq_ks_depth(int M, int Cout, int nchunks) {
    static const bool off = getenv("Y2_NO_KSPLIT") != nullptr;
    const int wgs = ((M + 255) / 256) * ((Cout + 63) / 64);
    if (off || wgs >= 128 || nchunks < 4) return 1;
    int d = 256 / wgs;
    d = d > 8 ? 8 : d;
    d = d > nchunks / 2 ? nchunks / 2 : d;
    return d < 2 ? 1 : d;
}
int conv_igemm_ks_depth(int M, int Cout, int row_bytes);      // conv_igemm.hip: the 1x1 launches' K split (round 6)
int conv_ks_depth(int taps, int M, int Cout, int row_bytes) {
    if (taps == 1) return conv_igemm_ks_depth(M, Cout, row_bytes);
    if (taps != 9 || M >= 384 * 8 || (row_bytes % 128) != 0 || Cout <= 64 || halo_compact()) return 1;
    return haloq_ks_depth(M, Cout, row_bytes / 128);
}
size_t conv_ks_scratch_floats(int taps, int M, int ldy, int row_bytes) {
    if ((taps != 9 && taps != 1) || M >= 384 * 8 || (row_bytes % 128) != 0 || ldy <= 64) return 0;
    return (size_t)8 * M * ldy;
}
// partial tiles -> result (+ statistics records of 128 pixels) of a K-split launch of either kernel family
template <typename T>
static hipError_t ks_finish_T(const ConvArgs& a, int depth, hipStream_t s) {
    const long quads = (long)a.M * (a.ldy / 4);
    hipLaunchKernelGGL(conv_ks_finish_kernel<T>, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, a, depth);
    if (a.part_mean && !a.aff_out)
        hipLaunchKernelGGL(conv_ks_stats_kernel<T>, dim3((a.ldy + 63) / 64, (a.M + kKsRec - 1) / kKsRec), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_conv_ks_finish(int dtype, const ConvArgs& a, int depth, hipStream_t s) {
    switch (dtype) {
        case 0: return ks_finish_T<float>(a, depth, s);
        case 1: return ks_finish_T<half_t>(a, depth, s);
        case 2: return ks_finish_T<bf16_t>(a, depth, s);
    }
    return hipErrorInvalidValue;
}
template <typename T>
static hipError_t haloq_ks(const ConvArgs& a0, hipStream_t s, int* bp) {
    constexpr int WP = 4, WC = 2, TP = 2, TC = 1, BKB = 128, BP = WP * TP * 32, BC = WC * TC * 32, RPI = 64 / (BKB / 16);
    const int nchunks = a0.C * (int)sizeof(T) / BKB;
    int depth = haloq_ks_depth(a0.M, a0.Cout, nchunks);
    if (depth < 2 || !a0.ks_scratch || a0.bw_psum || a0.nonfinite || halo_compact() || (a0.ldy % 4) != 0)
        return hipErrorNotSupported;
    while (depth > 1 && (size_t)depth * a0.M * a0.ldy > a0.ks_floats) --depth;
    if (depth < 2) return hipErrorNotSupported;
    ConvArgs a = a0;
    a.ks_splits = depth;
    const int arows = haloq_rows(a.H, a.W, BP, RPI);
    const bool adb = (nchunks + depth - 1) / depth > 1 && 2 * (size_t)arows * BKB <= 150 * 1024;
    const size_t lds = haloq_lds<BKB>(arows, adb, false);
    if (lds > 160 * 1024) return hipErrorNotSupported;
    void (*kern)(ConvArgs, int) = adb ? conv_haloq_kernel<T, WP, WC, TP, TC, BKB, true, false, 9, true>
                                      : conv_haloq_kernel<T, WP, WC, TP, TC, BKB, false, false, 9, true>;
    static size_t attr[2] = {0, 0};
    if (lds > attr[adb]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr[adb] = lds;
    }
    const int tiles = ((a.M + BP - 1) / BP) * ((a.Cout + BC - 1) / BC);
    hipLaunchKernelGGL(kern, dim3(tiles * depth), dim3(WP * WC * 64), lds, s, a, arows);
    const long quads = (long)a.M * (a.ldy / 4);
    hipLaunchKernelGGL(conv_ks_finish_kernel<T>, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, a, depth);
    if (a.part_mean && !a.aff_out)
        hipLaunchKernelGGL(conv_ks_stats_kernel<T>, dim3((a.ldy + 63) / 64, (a.M + kKsRec - 1) / kKsRec), dim3(256), 0, s, a);
    *bp = kKsRec;    // batch-norm records of the launch (launch_conv: records = ceil(M / bp))
    return hipGetLastError();
}

template <typename T>
static hipError_t haloq_T(const ConvArgs& a, hipStream_t s, int* bp) {
    typedef typename Types<T>::out_t YT;        // the epilogue patch is sized by what is stored
    const int kb = a.C * (int)sizeof(typename Types<T>::op_t);      // bytes of one operand plane per pixel
    const bool k128 = (kb % 128) == 0;
    if (!k128 && (kb % 64) != 0) return hipErrorInvalidValue;
    if (a.W > 52) {
        // long rows (104, 208): 512-pixel tiles amortise the two-row halo; one K-chunk per tile where it fits
        hipError_t e = hipErrorOutOfMemory;
        if (a.Cout > 64) {
            *bp = 256;
            // (split-operand forward, round 6: the fp32 epilogue patch sizes the LDS either way, so the two-plane form takes
            //  128-byte chunks -- 32 channels of both planes, half the tap steps of the 64-byte form; Y2_HALOQ_104_K64=1: A/B)
            static const bool k64 = getenv("Y2_HALOQ_104_K64") != nullptr;
            // (the hi-plane dgrads of f16x2f keep the 64-byte chunks: one 128-byte chunk is their whole K -- no second image to
            //  load behind the first -- and measured 216 against 200 us)
            if (Types<T>::kPasses == 3 && k128 && !k64) e = haloq_pick<T, 4, 2, 2, 2, 128>(a, s);
            else e = haloq_pick<T, 4, 2, 2, 2, 64>(a, s);
        } else if (a.Cout > 32) {
            *bp = 512;
            // (64-byte K chunks here, so that the 512-pixel image at W = 104 can be double-buffered, measured 12 %
            //  SLOWER than the single-buffered 128-byte ones: half the MFMAs per tap step for the same step overhead)
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
        const int tile = haloq_tile_choice(a.W, kb, a.Cout, a.M, Types<T>::kSplit ? 6 : (int)sizeof(YT));
        if (tile != HQ_NONE) {
            // the tile (and with it the filter pack: 16-row fragments for the _M16 kernels, 32-row ones otherwise) is
            // decided by ONE function for bind and launch time (conv_halo.hip haloq_tile_choice).  (Round 2 fell through
            // to a 32x32-tile kernel on the 16-row pack in the f32 mode -- wrong outputs from batch 24 up at 416x416.)
            switch (tile) {
                case HQ_384x128_M16:
                    *bp = 384;       // (fp32 outputs: the epilogue runs in two passes, conv_haloq16_kernel EPI2)
                    return haloq_pick<T, 4, 2, 3, 2, 128, true>(a, s);
                case HQ_256x128_M16: *bp = 256; return haloq_pick<T, 4, 2, 2, 2, 128, true>(a, s);
                case HQ_384x64: *bp = 384; return haloq_pick<T, 4, 2, 3, 1, 128>(a, s);
                case HQ_512x128:
                    if constexpr (sizeof(YT) == 2) { *bp = 512; return haloq_pick<T, 4, 2, 4, 2, 128>(a, s); }
                    return hipErrorInvalidValue;
                case HQ_256x128: *bp = 256; return haloq_pick<T, 4, 2, 2, 2, 128>(a, s);
                case HQ_512x64: *bp = 512; return haloq_pick<T, 4, 2, 4, 1, 128>(a, s);
                case HQ_256x64: *bp = 256; return haloq_pick<T, 4, 2, 2, 1, 128>(a, s);
            }
            return hipErrorInvalidValue;
        }
        // Fewer than 3072 pixels (the reference's own training shape, 224x224 at batch 24: 1176 on the 7x7 maps; single
        // images: 49): the launch is a stream of the FILTERS through a few workgroups, each bound by its serial K loop
        // (~110 ns per tap step: 72 us for K = 9216 whatever the tile).  64-cout tiles double the workgroups of the 128 x 128
        // form of rounds 1-3 on the same fragment pack: 7x7 1024 -> 1024 at batch 24 148 -> 81 us, one image 147 -> 72
        // (profiles/r04_sweep_small_m.txt; 128 x 64, 256 x 32 and 256 x 64 tie -- what is left there is a K split)
        if (k128 && a.M < 384 * 8) {
            if constexpr (!Types<T>::kSplit) {
                const hipError_t e = haloq_ks<T>(a, s, bp);
                if (e != hipErrorNotSupported) return e;
            }
            *bp = 256;
            return haloq_pick<T, 4, 2, 2, 1, 128>(a, s);
        }
        if (a.M >= 384 * 8) {   // what the cost model does not cover (64-byte K chunks; long rows that did not fit above)
            *bp = 384;
            e = k128 ? haloq_pick<T, 4, 2, 3, 2, 128>(a, s) : haloq_pick<T, 4, 2, 3, 2, 64>(a, s);
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
    if (a.taps == 1) {
        switch (dtype) {
            case 0: return haloq_T1<float>(a, s, bp);
            case 1: return haloq_T1<half_t>(a, s, bp);
            case 2: return haloq_T1<bf16_t>(a, s, bp);
        }
        return hipErrorInvalidValue;
    }
    if (a.taps != 9) return hipErrorInvalidValue;
    switch (dtype) {
        case 0: return haloq_T<float>(a, s, bp);
        case 1: return haloq_T<half_t>(a, s, bp);
        case 2: return haloq_T<bf16_t>(a, s, bp);
        case 3: return haloq_T<hsplit_t>(a, s, bp);
        case 4: return haloq_T<hsplith_t>(a, s, bp);     // f16x2f backward launches: the hi planes of split tensors
        case 5: return haloq_T<hsplithh_t>(a, s, bp);    // ... with dA stored in f16 (common.h hsplithh_t)
    }
    return hipErrorInvalidValue;
}

#ifdef Y2_DEVBUILD
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
        // round 4: 64-cout layers on the long rows (104x104 128 -> 64 and the 64-cout dgrads): one wave column, every
        // pixel fragment feeds TWO MFMAs (TC = 2) instead of one -- half the LDS reads per MFMA of <4,2,4,1>
        HQ(140, 8, 1, 2, 2, 128)      // 512 x 64, 8 waves of 64 px x 64 co
        HQ(141, 8, 1, 2, 2, 64)
        HQ(142, 8, 1, 4, 2, 64)       // 1024 x 64, 8 waves of 128 px x 64 co
        HQ(143, 4, 1, 4, 2, 128)      // 512 x 64, 4 waves of 128 px x 64 co
        HQ(144, 4, 1, 4, 2, 64)
        HQ(145, 8, 1, 3, 2, 128)      // 768 x 64
        // round 4: a few hundred to a few thousand pixels (224x224 at the reference's batch 24, single images)
        HQ(146, 4, 1, 2, 1, 128)      // 256 x 32, 4 waves
        HQ(147, 2, 2, 2, 1, 128)      // 128 x 64, 4 waves
        HQ(148, 2, 1, 2, 1, 128)      // 128 x 32, 2 waves
        HQ(149, 2, 2, 2, 2, 128)      // 128 x 128, 4 waves, 128-byte chunks
    }
#undef HQ
    return hipErrorInvalidValue;
}
#endif  // Y2_DEVBUILD

}  // namespace y2
