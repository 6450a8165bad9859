// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
// wave = 64 lanes; MFMA 32x32 tiles; LDS 160 KiB/CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace y2 {

typedef _Float16 half_t;
typedef __bf16 bf16_t;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define Y2_DEV __device__ __forceinline__

constexpr int kWave = 64;

// ---------------------------------------------------------------------------
// Element traits.  A "fragment" is always 16 bytes per lane: 8 halfs / 8 bf16 /
// 4 floats.  One 32-byte k-group of a row feeds lanes h=0 (bytes 0..15) and
// h=1 (bytes 16..31) of a 32x32 MFMA:
//   f16/bf16: one v_mfma_f32_32x32x16  (k = 16 per group)
//   f32     : four v_mfma_f32_32x32x2  (k = 8 per group; component s of the
//             fragment pairs k = s (h=0) with k = 4+s (h=1): a k-permutation
//             applied identically to both operands, so the sum is unchanged)
// ---------------------------------------------------------------------------
template <typename T> struct Elem;

template <> struct Elem<float> {
    typedef f32x4 frag;
    static constexpr int kPerFrag = 4;
    static constexpr int kId = 0;
    static Y2_DEV float to_f32(float v) { return v; }
    static Y2_DEV float from_f32(float v) { return v; }
};
template <> struct Elem<half_t> {
    typedef f16x8 frag;
    static constexpr int kPerFrag = 8;
    static constexpr int kId = 1;
    static Y2_DEV float to_f32(half_t v) { return (float)v; }
    static Y2_DEV half_t from_f32(float v) { return (half_t)v; }
};
template <> struct Elem<bf16_t> {
    typedef bf16x8 frag;
    static constexpr int kPerFrag = 8;
    static constexpr int kId = 2;
    static Y2_DEV float to_f32(bf16_t v) { return (float)v; }
    static Y2_DEV bf16_t from_f32(float v) { return (bf16_t)v; }
};

// ---------------------------------------------------------------------------
// Split-operand element ("f16x2", dtype 3 -- round 5): one fp32-width value kept as TWO halves, hi = f16(v) and
// lo = f16(v - hi), in two PLANES of a channel row: a pixel (or filter row) of C elements is C*4 bytes,
// [C halves hi][C halves lo].  A product a*b is formed on the f16 matrix pipe as hi*hi + lo*hi + hi*lo (fp32
// accumulate; the lo*lo term is 2^-22 of the product and dropped): ~22 mantissa bits at a third of the f16 MFMA rate
// instead of the exact-f32 MFMA's sixteenth.  sizeof(hsplit_t) = 4 on purpose: every byte count of the executor
// (row pitch, tensor size) is that of the f32 mode; what differs is the operand type the kernels read (op_t) and the
// K loop (three plane passes).  Conv outputs, dgrad outputs and everything the batch-norm passes read are plain fp32.
// Range: f16 keeps subnormals on this hardware (MFMA A/B inputs un-flushed, scripts/probes/mfma_denorm.hip), so the
// lo plane degrades gracefully: |v| >= 2^-3 keeps all 22 bits, below that the absolute error floor is 2^-25.
// Filters are pre-scaled by kSplitWScale (a power of two: exact) so that typical weights (|w| ~ 0.01 .. 0.1) sit in
// the full-precision range; the conv epilogues multiply the accumulators by its reciprocal.
// ---------------------------------------------------------------------------
struct hsplit_t { uint32_t bits; };
template <> struct Elem<hsplit_t> {
    typedef f16x8 frag;
    static constexpr int kPerFrag = 8;
    static constexpr int kId = 3;
};
constexpr float kSplitWScale = 64.0f, kSplitWScaleInv = 1.0f / 64.0f;
// Round 6 ("f16x2f", dtype 4 -- backward launches only): the SAME split tensors read through their HI planes alone, one
// f16 MFMA per product.  With the forward decisions (leaky branches, pool arg-max, responsible boxes) fixed by the
// split-operand forward pass, the backward pass is linear in dY: rounding dY, W and x to f16 per contraction is a relative
// perturbation of ~3e-4 per layer that nothing amplifies chaotically.  A cell is still [C halves hi][C halves lo]
// (4 bytes per element: the executor's byte counts do not change); a launch walks ONE plane pass of the K range.
struct hsplith_t { uint32_t bits; };
template <> struct Elem<hsplith_t> {
    typedef f16x8 frag;
    static constexpr int kPerFrag = 8;
    static constexpr int kId = 4;
};
// ... and (launch dtype 5, dgrad launches of the f16x2f mode above the second layer) the same contraction with its OUTPUT
// stored in f16 as well: dA, the gradient with respect to a layer's input, is consumed once, by the batch-norm backward
// pass of the layer below, which rounds its own result (dY) to f16 for the next contraction anyway -- 2 of 4 bytes per element
// on the write and on the read; the batch-norm backward reduce fused into this epilogue still reads that layer's fp32 conv
// output (kBwF32)
struct hsplithh_t { uint32_t bits; };
template <> struct Elem<hsplithh_t> {
    typedef f16x8 frag;
    static constexpr int kPerFrag = 8;
    static constexpr int kId = 5;
};
// operand type of the MFMA kernels / type of what their epilogues store; kPasses = plane passes of the K loop; kBwF32: the
// conv output the fused batch-norm backward reduce reads (ConvArgs::bw_y) is fp32 although the epilogue stores 16-bit values
template <typename T> struct Types { typedef T op_t; typedef T out_t; static constexpr bool kSplit = false; static constexpr int kPasses = 1; static constexpr bool kBwF32 = false; };
template <> struct Types<hsplit_t> { typedef half_t op_t; typedef float out_t; static constexpr bool kSplit = true; static constexpr int kPasses = 3; static constexpr bool kBwF32 = false; };
template <> struct Types<hsplith_t> { typedef half_t op_t; typedef float out_t; static constexpr bool kSplit = true; static constexpr int kPasses = 1; static constexpr bool kBwF32 = false; };
template <> struct Types<hsplithh_t> { typedef half_t op_t; typedef half_t out_t; static constexpr bool kSplit = true; static constexpr int kPasses = 1; static constexpr bool kBwF32 = true; };
// K chunk c of a launch whose K range is [hi plane | lo plane | hi plane again] x [filter hi | filter hi | filter lo]:
// n = chunks per plane.  Activation chunk: c mod 2n; filter chunk: c < n ? c : c - n (hi, hi, lo).
template <bool SPLIT> Y2_DEV int split_act_chunk(int c, int n) { return SPLIT ? (c >= 2 * n ? c - 2 * n : c) : c; }
template <bool SPLIT> Y2_DEV int split_flt_chunk(int c, int n) { return SPLIT ? (c >= n ? c - n : c) : c; }
Y2_DEV void split_f16(float v, half_t& hi, half_t& lo) {
    hi = (half_t)v;
    lo = (half_t)(v - (float)hi);
}
// four consecutive channels c0.. of one cell of a split tensor with C elements per cell: 8 bytes into each plane
Y2_DEV void st_split4(char* cell, int C, int c0, const float* r) {
    half_t h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split_f16(r[e], h[e], l[e]);
    *(u32x2*)(cell + (size_t)c0 * 2) = *(const u32x2*)h;
    *(u32x2*)(cell + (size_t)(C + c0) * 2) = *(const u32x2*)l;
}
Y2_DEV void ld_split4(const char* cell, int C, int c0, float* r) {
    half_t h[4], l[4];
    *(u32x2*)h = *(const u32x2*)(cell + (size_t)c0 * 2);
    *(u32x2*)l = *(const u32x2*)(cell + (size_t)(C + c0) * 2);
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = (float)h[e] + (float)l[e];
}

// XCD-aware block index (8 XCDs, each with its own L2; the dispatcher deals consecutive workgroups round-robin over
// them): workgroup b -> logical index such that each XCD works on ONE contiguous run of the logical grid, so the
// tiles that share an operand panel meet in one L2.  Bijective for any grid size.  mode 0: identity.
Y2_DEV int xcd_block(int b, int n, int mode) {
    if (!mode) return b;
    const int q = n >> 3, r = n & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// two floats -> one dword of two T (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32: round to nearest even, as the scalar casts)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
template <typename T> Y2_DEV uint32_t pack2(float a, float b);
template <> Y2_DEV uint32_t pack2<half_t>(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){a, b}, f16x2));
}
template <> Y2_DEV uint32_t pack2<bf16_t>(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){a, b}, bf16x2));
}

// acc[rows(regs)][cols(lanes)] += A(rows x k) * B(k x cols)
// C/D layout (all dtypes): col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
Y2_DEV void mma32(f32x16& acc, const f32x4& a, const f32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc, 0, 0, 0);
}
Y2_DEV void mma32(f32x16& acc, const f16x8& a, const f16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
}
Y2_DEV void mma32(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}

// row index inside a 32x32 accumulator tile held by (reg q, lane half h)
// 16x16 MFMA tiles: D[row 4*(lane>>4) + reg][col lane&15]; A rows / B columns = lane & 15, k-chunk = lane >> 4
Y2_DEV void mma16(f32x4& acc, const f32x4& a, const f32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc, 0, 0, 0);
}
Y2_DEV void mma16(f32x4& acc, const f16x8& a, const f16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
}
Y2_DEV void mma16(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}

// Split-operand products formed IN REGISTERS from fp32 data (the 3-channel first layer of the f16x2 mode, whose
// operands are fp32 in memory: conv1.hip / conv1_wgrad.hip XS forms): eight fp32 values -> one hi and one lo fragment,
// and the three plane products of one 32x32x16 step (small terms first).  The two f32x4 halves are the two 16-byte
// fragments a lane holds of a 32-wide fp32 k range (Elem<float>): the k order inside the MFMA is a permutation applied
// identically to both operands, so the sum is unchanged.
Y2_DEV void split_frag8(const f32x4& a, const f32x4& b, float scale, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float va = a[k] * scale, vb = b[k] * scale;
        hi[k] = (half_t)va;
        hi[4 + k] = (half_t)vb;
        lo[k] = (half_t)(va - (float)hi[k]);
        lo[4 + k] = (half_t)(vb - (float)hi[4 + k]);
    }
}
Y2_DEV void mma32_split(f32x16& acc, const f16x8& ah, const f16x8& al, const f16x8& bh, const f16x8& bl) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
}

Y2_DEV int acc_row(int q, int h) { return (q & 3) + 8 * (q >> 2) + 4 * h; }

// 16-byte vector of T (a "chunk"): load/store + per-element float access
template <typename T> struct Chunk {
    static constexpr int N = 16 / sizeof(T);
    T v[N];
};
template <typename T> Y2_DEV Chunk<T> ld_chunk(const void* p) {
    Chunk<T> c;
    *reinterpret_cast<u32x4*>(c.v) = *reinterpret_cast<const u32x4*>(p);
    return c;
}
template <typename T> Y2_DEV void st_chunk(void* p, const Chunk<T>& c) {
    *reinterpret_cast<u32x4*>(p) = *reinterpret_cast<const u32x4*>(c.v);
}

// async global -> LDS, 16 B per lane.  LDS destination = lds_base + lane*16
// (wave-uniform base: hardware rule); the GLOBAL address is per lane.
Y2_DEV void glds16(const void* gptr, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)gptr,
        (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

Y2_DEV float wave_sum_xor(float v, int mask) { return v + __shfl_xor(v, mask, 64); }

Y2_DEV float leaky01(float z) { return fmaxf(0.1f * z, z); }
// the same with the slope as a parameter (round 4: layer options -- 0.1 the reference's leaky ReLU, darknet.py:5,45;
// 0 = ReLU and 1 = no activation for slim's resnet_v1 bottlenecks, slim_dir/nets/resnet_v1.py:99-112)
Y2_DEV float leaky_s(float z, float s) { return fmaxf(s * z, z); }
Y2_DEV float leaky_slope_s(float z, float s) { return (s * z >= z) ? s : 1.0f; }
// TF maximum(alpha*z, z): gradient to alpha*z where alpha*z >= z, i.e. z <= 0
Y2_DEV float leaky01_slope(float z) { return (0.1f * z >= z) ? 0.1f : 1.0f; }

// two ds_read_b64_tr_b16 (hardware-transposed LDS reads, 16-bit elements) -> one 8-element
// MFMA fragment: p0 addresses k = 0..3 of this lane's half, p1 the next four
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <typename T>
Y2_DEV typename Elem<T>::frag tr_frag(const char* p0, const char* p1) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
    s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(typename Elem<T>::frag, both);
}

// ---------------------------------------------------------------------------
// Zero-bordered NHWC activation layout with SHARED borders ("bordered" tensors):
//   pitch = W + 1 pixels per row, H + 1 rows per image (+ one closing row):
//   pixel (n, h, w) lives at  n*(H+1)*pitch + (h+1)*pitch + (w+1).
// The zero cell left of (h, 0) is also the cell right of (h-1, W-1); the zero row
// above image n is the row below image n-1.  Every 3x3 tap of every interior pixel
// is therefore in bounds and reads 0 outside the image, with only
// (H+1)(W+1)/(HW) storage / linear-K overhead (1.16x at 13x13 instead of 1.33x).
// Nothing ever writes a border cell; the allocation is zeroed once at bind time.
// ---------------------------------------------------------------------------
__host__ __device__ inline size_t bpix(int n, int h, int w, int H, int W) {
    return ((size_t)n * (H + 1) + (size_t)(h + 1)) * (size_t)(W + 1) + (size_t)(w + 1);
}
__host__ __device__ inline size_t bbody_pixels(int N, int H, int W) {
    return (size_t)N * (H + 1) * (W + 1) + (size_t)(W + 1) + 1;
}

// window-major pixel order of a pooled layer's convolution tiles (kernels.h ConvArgs::aff_pool): position q of the
// order -> NHW pixel index; H and W even
Y2_DEV int pool_order_pixel(int q, int H, int W) {
    const int Wo = W >> 1, Ho = H >> 1;
    const int win = q >> 2, d = q & 3;
    const int t = win / Wo, wo = win - t * Wo;
    const int n = t / Ho, ho = t - n * Ho;
    return (n * H + 2 * ho + (d >> 1)) * W + 2 * wo + (d & 1);
}

}  // namespace y2
