// Internal launcher interface between the network executor (net.hip) and the
// gfx950 kernels.  dtype: 0 = f32 (parity mode, exact-f32 MFMA), 1 = f16, 2 = bf16,
// 3 = f16x2 (round 5: split-operand mode -- every MFMA operand is a (hi, lo) pair of halves in two planes of its
// channel row, three f16 MFMAs per product, fp32-width storage everywhere: common.h hsplit_t).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace y2 {

//   4 = f16x2f (round 6): LAUNCH dtype of the backward contractions of a split-mode context created with Y2_F16X2F -- the
//       same split tensors, hi planes only, one f16 MFMA per product (common.h hsplith_t).  A context's own dtype is 3.
//   5 = the same contraction (dgrad launches only) with its output dA stored in f16 instead of fp32 (common.h hsplithh_t)
inline size_t dtype_size(int dtype) { return (dtype == 0 || dtype >= 3) ? 4 : 2; }   // bytes per stored element
inline int dtype_kbytes(int dtype) { return dtype == 0 ? 4 : 2; }                  // bytes per element of ONE K plane (MFMA operand)
inline bool dtype_split(int dtype) { return dtype >= 3; }
// dtype of the kernels whose arithmetic is elementwise fp32 in the split mode (first layer, casts): f32
inline int dtype_plain(int dtype) { return dtype >= 3 ? 0 : dtype; }

struct ConvArgs {
    const void* x;      // zero-bordered NHWC [N][H+2][W+2][C]
    const void* w;      // packed [Cout_pad][taps][C]
    void* y;            // [M][ldy]
    const float* bias;  // [Cout] or null
    float* part_cnt;    // [P]            BN partials (null: no statistics)
    float* part_mean;   // [P][ldy]
    float* part_m2;     // [P][ldy]
    int N, H, W, C;
    int M;              // N*H*W
    int Cout;
    int ldy;
    int taps;           // 9 (3x3) or 1 (1x1)
    // dgrad launches only: the batch-norm backward REDUCE pass of the layer below, fused into this epilogue.
    // y (this launch's output) is dA of that layer; with bw_y = its conv output at the same pixels (a pooled
    // layer: at the window's arg-max, BnActArgs::ysel) and its scale / shift the epilogue accumulates
    // S1 = sum g and S2 = sum g * y, g = dA * leaky'(y * scale + shift), per pixel tile:
    // bw_psum[tile][2][ldy] -- what bn_bwd_kernel<.., APPLY = false> would write.
    const void* bw_y = nullptr;       // [M][ldy]
    const float* bw_scale = nullptr;
    const float* bw_shift = nullptr;
    float* bw_psum = nullptr;
    float bw_slope = 0.1f;            // activation slope of that layer (leaky_slope_s)
    // non-null (plain-store launches): set to 1 when a stored value is inf / NaN -- the early overflow guard of
    // y2_backward_adam / _momentum watches the dgrad that feeds the first layer this way
    unsigned* nonfinite = nullptr;
    int xcd = 0;            // XCD-aware workgroup order (common.h xcd_block)
    int is_dgrad = 0;       // the launch computes an input gradient (filters from the dgrad copy): kernel policy only
    // Inference-mode batch norm FOLDED into the epilogue (forward launches of un-pooled layers whose statistics are
    // the moving ones: tf.layers.batch_normalization(training=False) is a fixed per-channel affine,
    // src/yolo2_nets/darknet.py:39-46): the stored value is leaky(T(conv + bias) * scale + shift) -- the same
    // arithmetic on the same rounded conv output as the two-pass form (bn_act_kernel), bit for bit -- written straight
    // into the consumer's bordered tensor aff_out [N][H+2][W+2][ldy]; y is NOT written.  Set with conv_set_affine().
    const float* aff_scale = nullptr;
    const float* aff_shift = nullptr;
    void* aff_out = nullptr;
    float aff_slope = 0.1f;           // activation slope of this layer
    uint32_t aff_magW = 0, aff_magH = 0;   // m / W = (m * magW) >> shW for m < 2^31 (Granlund-Montgomery), same for H
    int aff_shW = 0, aff_shH = 0;
    // POOLED layers in the fold (round 5; conv_haloq kernels, even H and W): the tile's pixels run in WINDOW-MAJOR order
    // q = ((n Ho + ho) Wo + wo) 4 + 2 dh + dw (common.h pool_order_pixel), so a tile holds whole 2x2 windows -- four
    // consecutive rows of the epilogue patch -- and the stored value is leaky(max over the window of T(conv + b) * scale
    // + shift) at the pooled position of aff_out [N][H/2+2][W/2+2][ldy]: bn_act_kernel's pool_window arithmetic
    int aff_pool = 0;
    // K split over workgroups for launches of a few hundred to a few thousand pixels (conv_haloq.hip: haloq_ks): the
    // caller lends ks_floats floats of scratch; the launcher decides whether and how deep to split (ks_splits is its own)
    float* ks_scratch = nullptr;
    size_t ks_floats = 0;
    int ks_splits = 0;
};
// scratch floats a launch of M pixels x ldy couts may ask for (0: the K split never applies)
size_t conv_ks_scratch_floats(int taps, int M, int ldy, int row_bytes);
// split depth a launch of this shape takes when it is lent scratch (< 2: it runs un-split)
int conv_ks_depth(int taps, int M, int Cout, int row_bytes);
inline void conv_div_magic(uint32_t d, uint32_t* mag, int* sh) {
    int l = 0;
    while ((1u << l) < d) ++l;                    // ceil(log2 d)
    *sh = 31 + l;
    *mag = (uint32_t)((((uint64_t)1 << (31 + l)) + d - 1) / d);
}
inline void conv_set_affine(ConvArgs& a, const float* scale, const float* shift, void* out_bordered) {
    a.aff_scale = scale; a.aff_shift = shift; a.aff_out = out_bordered;
    conv_div_magic((uint32_t)a.W, &a.aff_magW, &a.aff_shW);
    conv_div_magic((uint32_t)a.H, &a.aff_magH, &a.aff_shH);
}
// which launches take the folded form (else: y + the bn_act pass)
bool conv_affine_ok(int dtype, const ConvArgs& a);
// ... and which POOLED layers do (ConvArgs::aff_pool; conv_affine_ok holds too)
bool conv_affine_pool_ok(int dtype, const ConvArgs& a);
hipError_t launch_conv_igemm(int dtype, const ConvArgs& a, hipStream_t s);   // per-tap staging (used for 1x1)
// 1x1 (round 6, conv_gemm1.hip): filter fragments straight from L2 (pack layout 1), the pixel tile in a deep LDS ring
bool conv_gemm1_ok(int taps, int row_bytes, int Cout, int M);
hipError_t launch_conv_gemm1(int dtype, const ConvArgs& a, hipStream_t s, int* block_pixels);
hipError_t launch_conv_halo(int dtype, const ConvArgs& a, hipStream_t s, int* block_pixels);  // 3x3: LDS halo image
hipError_t launch_conv_haloq(int dtype, const ConvArgs& a, hipStream_t s, int* block_pixels); // + filters via registers
// filter layout launch_conv expects (0/1/2).  row_bytes = bytes of one operand plane per pixel (dtype_kbytes), elem_size
// = dtype_size (sizes the epilogue patch), split = dtype_split
int conv_filter_layout(int taps, int W, int row_bytes, int Cout, int M, int dgrad = 0, int elem_size = 2, int split = 0);
// Pixel x cout tile of conv_haloq on the short-row 3x3 layers (W <= 52, more than 64 couts, 128-byte K chunks), chosen by
// a cost model of the workgroup rounds on the 256 CUs (conv_halo.hip: haloq_tile_choice).  ONE function decides both the
// kernel (launch time) and the filter pack it reads (bind time): HQ_384x128_M16 reads 16-row fragments (layout 2),
// every other tile 32-row fragments (layout 1).
enum HqTile { HQ_NONE = 0, HQ_384x128_M16, HQ_256x128_M16, HQ_384x64, HQ_512x128, HQ_256x128, HQ_512x64, HQ_256x64 };
int haloq_tile_choice(int W, int row_bytes, int Cout, int M, int elem_size);
// policy; *records = rows of the BN partial list written (one per pixel tile)
hipError_t launch_conv(int dtype, const ConvArgs& a, hipStream_t s, int* block_pixels = nullptr, int* records = nullptr);
// 3x3, filters resident in registers, persistent workgroups over the bordered pixel space (conv_rf.hip)
int conv_rf_config(int taps, int W, int row_bytes, int Cout, int M);   // 0: not this form
int conv_rfn_config(int taps, int W, int row_bytes, int Cout, int M, int dgrad);
hipError_t launch_conv_rf(int dtype, const ConvArgs& a, hipStream_t s, int* block_pixels, int* records);
int conv_block_pixels(int Cout);
int conv_block_couts(int Cout);

// ---- first layer (Cin = 3, stored as 4 channels)
struct Conv1Args {
    const void* x4;     // [N][H+2][W+2][4]
    const void* w;      // packed [32][3][16] (kh, then kw*4+c, 12 real + 4 zero)
    void* y;            // [M][32]
    const float* bias;
    float* part_cnt;
    float* part_mean;
    float* part_m2;
    int N, H, W, M;
    int nblocks;        // persistent grid size == number of partials
    int stats_only;     // 1: batch-norm partials only, y is not written (first pass of the pooled form)
    // f16x2 mode, stats_only pass: the fp32 operands are split into half planes in registers and a filter row is three f16
    // matrix instructions (conv1.hip XS forms; set together with Conv1PoolArgs::xs / Conv1WgradLinArgs::xs)
    int xs = 0;
};
hipError_t launch_conv1_fwd(int dtype, const Conv1Args& a, hipStream_t s);
// pooled first layer, second pass: conv again + BN + leaky + 2x2 max pool -> y (optional) and the pooled output
struct Conv1PoolArgs {
    const void* x4;
    const void* w;
    void* y;            // [M][32] (written when store_y)
    const float* bias;
    const float *scale, *shift;
    void* out;          // zero-bordered [N][Ho+2][Wo+2][32] of T
    int N, H, W;
    int nblocks;
    int store_y;
    // training, linear form of the backward pass (conv1_wgrad.hip): instead of the conv output (64 B/pixel) keep,
    // per pooled pixel, the conv output at the window's first arg-max (ysel [Mout][32] of T) and WHICH of the four
    // positions it was (idx [Mout][chunks] u16: 2 bits per channel of a 16-byte chunk)
    void* ysel = nullptr;
    unsigned short* idx = nullptr;
    // round 4, 16-bit types: NO conv output at all is kept.  idx3 [Mout][chunks] u32 holds 3 bits per channel of a chunk:
    // the window position (2) and whether the activation there took the leaky branch (1: 0.1 * z >= z).  That is all the
    // backward pass needs per element (g = dA * slope, scattered to that position); its sum of g * y follows from the
    // linearity of y in the filter: sum_p dz y = sum_k W[k] X(dz)[k] + b sum dz, with X(dz) the matrix the weight
    // gradient forms anyway (conv1_wgrad.hip conv1_lin_s2_kernel).  Set instead of ysel / idx.
    unsigned* idx3 = nullptr;
    int out_split = 0;  // f16x2 mode (T = float kernels): `out` is a split tensor ([32 halves hi][32 halves lo] per cell)
    int xs = 0;         // f16x2 mode: split-operand products formed in registers (Conv1Args::xs)
};
// backward reduce pass of the same layer with the conv output recomputed (x4 + dA in, psum out)
struct Conv1BnBwdArgs {
    const void* x4;
    const void* w;
    const float* bias;
    const float *scale, *shift;
    const void* dA;     // grad wrt the pooled output [N*Ho*Wo][32] of T
    float* psum;        // [nblocks][2][32]
    int N, H, W;
    int nblocks;
};
hipError_t launch_conv1_bnbwd_reduce(int dtype, const Conv1BnBwdArgs& a, hipStream_t s);
bool conv1_pool_ok(int H, int W, int pool, int cout);
hipError_t launch_conv1_pool(int dtype, const Conv1PoolArgs& a, hipStream_t s);
struct Conv1WgradArgs {
    const void* x4;     // [N][H+2][W+2][4]
    const void* dy;     // zero-bordered [N][H+2][W+2][32]
    float* dW;          // [3][3][3][32] fp32, accumulated with atomics (pre-zeroed)
    int N, H, W, M;
    float scale;        // 1 / grad_scale
};
hipError_t launch_conv1_wgrad(int dtype, const Conv1WgradArgs& a, hipStream_t s);
// pooled first layer: BN-backward apply fused into the weight gradient (dy never reaches HBM)
struct Conv1WgradFusedArgs {
    const void* x4;       // [N][H+2][W+2][4]
    const void* y;        // conv output [M][32]
    const void* dA;       // grad wrt the pooled layer output [N*Ho*Wo][32]
    const float *scale, *shift;
    const float* coef;    // [2][32]: ka, kb (bn_bwd_finalize)
    float* dW;            // [3][3][3][32] fp32, atomics (pre-zeroed)
    int N, H, W;
    float inv_grad_scale;
};
bool conv1_wgrad_fused_ok(int H, int W, int pool, int ldy, int elem_size);
hipError_t launch_conv1_wgrad_fused(int dtype, const Conv1WgradFusedArgs& a, hipStream_t s);
// Linear form (no conv output of the first layer in HBM at all).  With dy = scale dz - (ka + kb y):
//     dW = scale * X(dz) - ka * X(1) - kb * X(y),   X(v)[t][c][co] = sum_p x[p + t][c] v[p][co]
// and y = conv(x, W) + b:  X(y) = G W + b X(1),  G = the Gram matrix of the 27-element input patches
// (weights-independent).  The kernel accumulates X(dz) [48][32] and G [48][48] (rows kh*16 + kw*4 + c; channel 3
// of the stored input is 1 inside the image, so G's row of the centre tap's channel 3 IS X(1)); a one-block
// finalize combines them with the BN-backward constants.
struct Conv1WgradLinArgs {
    const void* x4;             // [N][H+2][W+2][4], channel 3 = 1 inside the image
    const void* dA;             // [Mout][32] of T
    const void* ysel;           // [Mout][32] of T
    const unsigned short* idx;  // [Mout][chunks]
    const float *scale, *shift;
    const float* gram = nullptr; // [48][48] totals of the forward pass's Gram matrix (launch_conv1_gram_stats): G is not rebuilt here
    // Conv1PoolArgs::idx3 form (ysel / idx null): Wf (fp32 HWIO [3][3][3][32]) and bias for sum g * y = W . X(dz) + b sum dz,
    // written as one more psum record (S1 = 0) behind the blocks' records
    const unsigned* idx3 = nullptr;
    const float* Wf = nullptr;
    const float* bias = nullptr;
    float* acc;                 // 16 slice sums of [48*32 + 48*48], followed by the per-block partials (conv1_wgrad_lin_scratch_floats)
    float* psum;                // out: BN-backward partial sums [blocks][2][32] (S1, S2) -- the reduce pass rides here
    int* nblocks_out;           // host: number of partial records written
    int N, H, W;
    int xs = 0;                 // f16x2 mode (T = float kernel): X(dz) and G from split-operand f16 products (Conv1Args::xs);
                                // 2 (f16x2f): from the hi planes alone, one product each
    // set by the launcher: a row pair is worked in nseg column segments of ws pixels (a multiple of 16) so that the LDS row
    // images of the fp32-wide form leave room for two workgroups per CU
    int nseg = 1, ws = 0;
};
struct Conv1DwFinalizeArgs {
    const float* acc;           // the 16 slice sums (added here)
    const float* W;             // fp32 HWIO [3][3][3][32]
    const float* bias;
    const float* scale;
    const float* coef;          // [2][32] ka, kb
    float* dW;                  // out [3][3][3][32]
    float inv_grad_scale;
    const float* gram = nullptr; // [48][48] Gram totals of the forward pass (else: the G part of acc)
};
// Forward pass of the pooled first layer, training: Gram matrix of the input patches -> batch-norm statistics of the
// layer (conv1_wgrad.hip: replaces the statistics-only convolution pass + bn_finalize) and the totals the backward
// pass reuses.  mid: conv1_gram_scratch_floats() floats = [1 + 16 + 768][48*48]: gram totals, slices, block partials.
struct Conv1GramStatsArgs {
    const void* x4;             // [N][H+2][W+2][4], channel 3 = 1 inside the image
    int N, H, Wd;
    const float* W;             // fp32 HWIO [3][3][3][32]
    const float* bias;
    const float *gamma, *beta;
    float *moving_mean, *moving_var;
    float *scale, *shift, *mean, *invstd, *var;
    float eps, momentum;
    int update_moving, bessel;
    float* mid;                 // slices + partials (scratch)
    float* gram;                // out: [48][48] totals
};
bool conv1_gram_ok(int H, int W, int elem_size);
size_t conv1_gram_scratch_floats();
hipError_t launch_conv1_gram_stats(int dtype, const Conv1GramStatsArgs& a, hipStream_t s);
bool conv1_wgrad_lin_ok(int H, int W, int pool, int ldy, int elem_size);
size_t conv1_wgrad_lin_scratch_floats();   // acc: totals + per-block partials
hipError_t launch_conv1_wgrad_lin(int dtype, const Conv1WgradLinArgs& a, hipStream_t s);
hipError_t launch_conv1_dw_finalize(const Conv1DwFinalizeArgs& a, hipStream_t s);

// ---- weight-gradient GEMM  dW[t][ci][co] += sum_p X[p+t][ci] * dY[p][co]
struct WgradArgs {
    const void* x;      // zero-bordered [N][H+2][W+2][Cin]
    const void* dy;     // zero-bordered [N][H+2][W+2][Cdy]
    float* dW;          // [taps][Cin][Cout] fp32 (HWIO), accumulated with atomics
    int N, H, W, M;
    int Cin, Cdy, Cout; // Cdy = channel stride of dy (>= Cout)
    int taps;
    int splitk;
    float scale;
    int xcd = 0;        // XCD-aware workgroup order (common.h xcd_block)
    // split-K partial tiles: with a slab of at least splitk * taps*Cin*Cout floats the splits store their partials
    // there (plain stores) and wgrad_split_finish sums them in a fixed order into dW -- deterministic, and no
    // zero-fill of dW; without one they are added into a zeroed dW with float atomics (≈1.3 TB/s chip-wide)
    float* slab = nullptr;
    size_t slab_floats = 0;
    // f16x2 mode: x / dy are split tensors.  The launchers run the 16-bit kernels once per operand-plane pair (hi hi,
    // lo hi, hi lo -- `quads` = 3) with xpitch / ypitch = elements of the operand type per pixel (2 Cin / 2 Cdy) and
    // every launch's partial tiles in its own range of the slab; ONE fixed-order sum over quads * splitk partials
    int xpitch = 0, ypitch = 0;     // 0: Cin / Cdy
    int quads = 1;
    // Round 5: the split-K sum INSIDE the weight-gradient kernel (wgrad_finish.h).  tile_cnt != null (>= cnt_ints zeroed
    // ints, self-cleaning): every block adds 1 to its tile's counter after its partial tile is in the slab; the block
    // that completes a group of <= 16 partials sums that group in index order (a fixed tree of groups: deterministic
    // whoever arrives last, no spin-wait, so no co-residency requirement) and the one that completes the top group
    // writes dW -- the separate wgrad_reduce launches (18 per detector step) disappear.
    int* tile_cnt = nullptr;
    size_t cnt_ints = 0;
    int part0 = 0;          // slab slot of this launch's split 0 (quads: q * splitk); set by wgrad_launch_quads
    int cnt_stride = 0;     // counters per tile (wgrad_cnt_per_tile); 0: the separate sum kernel runs after the launch
    int fin_lds_off = 0;    // byte offset of the finish flag in the dynamic LDS (behind the staging buffers: the ring form
                            // needs the buffers at an aligned LDS base, so no static __shared__ in these kernels)
};
constexpr int kWgFinG = 16;      // partials per group of the in-kernel sum
// counters per tile for `parts` partials: one per group of every level of the tree
inline int wgrad_cnt_per_tile(int parts) {
    int c = 0;
    for (int n = parts; ; n = (n + kWgFinG - 1) / kWgFinG) {
        if (n <= kWgFinG) return c + 1;
        c += (n + kWgFinG - 1) / kWgFinG;
    }
}
// f16x2: the split form of a (x, dy) pair for the 16-bit kernels (dtype 3 -> 1)
inline int wgrad_split_args(int dtype, WgradArgs& a) {
    if (!a.xpitch) a.xpitch = a.Cin;
    if (!a.ypitch) a.ypitch = a.Cdy;
    if (!dtype_split(dtype)) return dtype;
    a.xpitch = 2 * a.Cin; a.ypitch = 2 * a.Cdy;
    a.quads = dtype >= 4 ? 1 : 3;      // f16x2f: the hi-plane pair alone -- the f16 kernels on cells of twice the pitch
    return 1;
}
// one launch per operand-plane pair (WgradArgs::quads): hi hi, x lo, dy lo -- each with its own slab range
template <typename K, typename... Extra>
inline hipError_t wgrad_launch_quads(K kern, dim3 grid, dim3 block, size_t lds, hipStream_t s, const WgradArgs& a, Extra... extra) {
    for (int q = 0; q < a.quads; ++q) {
        WgradArgs b = a;
        if (q == 1) b.x = (const char*)a.x + (size_t)a.Cin * 2;
        if (q == 2) b.dy = (const char*)a.dy + (size_t)a.Cdy * 2;
        b.part0 = q * a.splitk;
        hipLaunchKernelGGL(kern, grid, block, lds, s, b, extra...);
    }
    return hipGetLastError();
}
hipError_t launch_wgrad(int dtype, const WgradArgs& a, hipStream_t s);       // one tap per block
hipError_t launch_wgrad9(int dtype, const WgradArgs& a, hipStream_t s);      // 3x3: nine taps per block
hipError_t launch_wgrad_auto(int dtype, const WgradArgs& a, hipStream_t s);
// around a split-K launch (a.splitk resolved): chooses slab or atomics (zero-filling dW for the latter) / sums the slab
int wgrad_finish_max_parts();      // Y2_WGRAD_FINISH (0 = the in-kernel sum is off: the default)
hipError_t wgrad_split_prepare(WgradArgs& a, hipStream_t s);
hipError_t wgrad_split_finish(const WgradArgs& a, hipStream_t s);

// ---- packing
hipError_t launch_pack_input(int dtype, const float* img, void* x4, int N, int H, int W, hipStream_t s);
// same from uint8 pixels, with image_read's conversion x / 255 * 2 - 1 (src/img_dataset/pascal_voc.py:63-64) fused
hipError_t launch_pack_input_u8(int dtype, const uint8_t* img, void* x4, int N, int H, int W, hipStream_t s);
// fwd:  wf[co][t][ci] = W[t][ci][co]           rows co >= Cout zero (Cout_pad rows)
// dgrad: wd[ci][t'][co] = W[8-t'][ci][co]       rows ci >= Cin zero, cols co >= Cout zero (Cdy cols)
//   Kc = row length per tap of wf (>= Cin, zero beyond Cin)
hipError_t launch_pack_weights(int dtype, const float* W, void* wf, void* wd, int taps, int Cin, int Cout,
                               int Cout_pad, int Kc, int Cin_pad, int Cdy, int frag, hipStream_t s);
hipError_t launch_pack_conv1_weights(int dtype, const float* W, void* wp, hipStream_t s);
// all layers in one launch: table entry per layer (device copy lives in the workspace)
struct PackLayer {
    const float* W;
    size_t w_off;       // offset of W in the flat parameter buffer (adam_pack: the same offset into m, v, grads)
    void* wf;
    void* wd;           // null: no dgrad copy
    int taps, Cin, Cout, Cout_pad, Kc, Cin_pad, Cdy;
    int wf_bx, wf_by, wf_blocks, wd_blocks, first_block;
    int opt_first;      // first tile block of this layer in launch_opt_pack's grid (tiles only)
    int wf_frag, wd_frag;   // conv_filter_layout(): 0 K-contiguous rows, 1 / 2 MFMA-fragment order (32 / 16 rows)
};
void pack_layer_plan(PackLayer& L, int first_block, int elem_size);
hipError_t launch_pack_all(int dtype, const PackLayer* tab_dev, int nlayers, int total_blocks, hipStream_t s);
// Optimizer step + filter re-pack in ONE pass over the parameters (the update reads and writes every filter
// anyway: the packed f16 / bf16 copies leave from the same registers instead of a second 193 MB read).
// kind 0: Adam (slot0 = m, slot1 = v), 1: Momentum (slot0 = accum).  ctrl != null: guarded (optim.hip), lr_t from
// ctrl; else lr_t = hyper[0].  hyper = {lr_t or lr, b1 or momentum, b2, eps, grad_mult}.
// small: [offset, count] ranges of the parameters that are not filter tiles (b, gamma, beta; a 3-channel first filter)
struct OptPackArgs {
    float* p; float* slot0; float* slot1; const float* g;
    const void* ctrl;
    float lr_t, b1, b2, eps, gmult;
    int kind;
    const PackLayer* tab; int nlayers; int tile_blocks;
    const unsigned* small; int nsmall;
};
hipError_t launch_opt_pack(int dtype, const OptPackArgs& a, hipStream_t s);
hipError_t launch_convert_grad(int dtype, const float* src, void* dst, int M, int C, int ldd, float scale,
                               hipStream_t s);
hipError_t launch_unpack_act(int dtype, const void* xp, float* out, int N, int H, int W, int C, int Cs,
                             hipStream_t s);
// whole allocation of a bordered tensor (guards, borders, body, padding channels) in one pass; hipErrorNotSupported
// where the 16-byte form does not apply (the caller then zeroes the allocation and uses launch_pack_act)
hipError_t launch_pack_act_region(int dtype, const float* in, void* region, size_t region_bytes, size_t front_px, int N,
                                  int H, int W, int C, int Cs, hipStream_t s);
hipError_t launch_pack_act(int dtype, const float* in, void* xp, int N, int H, int W, int C, int Cs,
                           hipStream_t s);
hipError_t launch_cast_to_f32(int dtype, const void* src, float* dst, size_t rows, int C, int lds, hipStream_t s,
                              float scale = 1.0f);

// ---- batch norm
struct BnFinalizeArgs {
    const float* part_cnt;
    const float* part_mean;
    const float* part_m2;
    int P, C, ldp;
    const float* gamma;
    const float* beta;
    float* moving_mean;   // updated in place when update_moving
    float* moving_var;
    float* scale;         // out: gamma * rsqrt(var + eps)
    float* shift;         // out: beta - mean * scale
    float* mean;          // out (saved for backward)
    float* invstd;        // out
    float* var;           // out: biased (or Bessel) batch variance as the moving update uses it
    float eps, momentum;
    int update_moving;
    int bessel;
    float* scratch = nullptr;   // >= 64 * (1 + 2 * ldp) floats: long partial lists are compressed to 64 records first
};
hipError_t launch_bn_finalize(const BnFinalizeArgs& a, hipStream_t s);
// per layer: moving statistics + gamma / beta -> scale, shift, mean, invstd of the apply pass
struct BnInferLayer {
    const float *gamma, *beta, *mm, *mv;
    float *scale, *shift, *mean, *invstd;
    int C, is_core;
};
hipError_t launch_bn_infer_prepare_all(const BnInferLayer* tab, int nlayers, int max_c, int train_core, int train_head,
                                       float eps, hipStream_t s);
hipError_t launch_bn_update_moving(const float* mean, const float* var, float* mm, float* mv, int C, float momentum,
                                   hipStream_t s);
struct BnActArgs {
    const void* y;        // [M][ldy]
    const float* scale;
    const float* shift;
    void* out;            // zero-bordered [N][Ho+2][Wo+2][C] of T, or (out_f32) float [M][C] compact
    int N, H, W, C, ldy;
    int pool;             // 1: 2x2/2 SAME max pool after the activation.  2 (round 5, the ResNet swap's stride-2 3x3
                          // convolutions, slim conv2d_same: resnet_utils.py:77-122): SUBSAMPLE -- the layer keeps window
                          // position 0 (the stride-1 output at even rows / columns) and its batch norm runs over the kept
                          // positions only; even H and W
    int out_f32;
    float slope = 0.1f;   // activation: max(slope * z, z)
    void* ysel = nullptr; // pooled layers, training: the conv output at the window's (first) arg-max, [Mout][ldy] of T --
                          // what the BN-backward reduce needs of y (fused into the dgrad epilogue above this layer)
    const float* join = nullptr;   // out_f32 only, [M][C] like out: the stored value is max(act + join, 0) -- the join of a
                                   // ResNet bottleneck unit (y2_forward_join)
    const void* join_t = nullptr;  // the same join read from a BORDERED tensor of T with the output's geometry (round 5,
                                   // y2_link: bottleneck units chained without an fp32 hand-over); either output form
};
hipError_t launch_bn_act(int dtype, const BnActArgs& a, hipStream_t s);
// merge of a short partial list (P <= 128) + apply in one launch (64-channel slabs); bn_fin_act_ok says whether it applies
bool bn_fin_act_ok(const BnActArgs& a, const BnFinalizeArgs& f);
hipError_t launch_bn_fin_act(int dtype, const BnActArgs& a, const BnFinalizeArgs& f, hipStream_t s);
// subsampling layers (BnActArgs::pool == 2): (count, mean, M2) records of the conv output y [N*H*W][ldy] over the KEPT
// positions (even rows and columns), one record per kBnSubRec kept pixels; *records = their number
constexpr int kBnSubRec = 256;
hipError_t launch_bn_stats_sub(int dtype, const void* y, int N, int H, int W, int ldy, float* part_cnt, float* part_mean,
                               float* part_m2, int* records, hipStream_t s);

struct BnBwdArgs {
    const void* dA;       // grad wrt layer output [M_out][ldd] of T (scaled by grad_scale)
    const void* y;        // conv output [M][ldy]
    const float* scale;   // gamma*invstd
    const float* shift;
    const float* mean;
    const float* invstd;
    float* psum;          // [P][2][C] partial sums (dz, dz*y)
    float* dgamma;        // out (unscaled)
    float* dbeta;
    float* dbias;         // out: sum(dy), analytic (see bn.hip)
    float* coef;          // [2][ldy]: ka, kb of dy = scale*dz - (ka + kb*y)   (scaled domain)
    void* dyp;            // out: zero-bordered [N][H+2][W+2][ldy] of T
    int N, H, W, C, ldy, ldd;
    int pool;
    int training;         // batch statistics (1) or moving statistics (0)
    float inv_grad_scale;
    int P;                // number of partial blocks (set by launcher)
    float slope = 0.1f;   // activation slope of the forward pass
    int hi_only = 0;      // f16x2f (split dyp): the consumers read the hi plane of dY alone -- the lo plane is not written
    int dA_half = 0;      // f16x2f (T = float kernels): dA [M_out][ldd] is f16 (written by a launch-dtype-5 dgrad, common.h hsplithh_t)
};
int bn_bwd_partials(const BnBwdArgs& a);
hipError_t launch_bn_bwd_reduce(int dtype, BnBwdArgs& a, hipStream_t s);
hipError_t launch_bn_bwd_finalize(const BnBwdArgs& a, hipStream_t s);
hipError_t launch_bn_bwd_apply(int dtype, const BnBwdArgs& a, hipStream_t s);
// finalize of a short partial list (P <= 128) + apply in one launch (64-channel slabs)
bool bn_bwd_fin_apply_ok(const BnBwdArgs& a);
hipError_t launch_bn_bwd_fin_apply(int dtype, const BnBwdArgs& a, hipStream_t s);

// ---- loss / heads
struct LossArgs {
    const float* net;     // [N][S][S][C + 5B]
    const float* labels;  // [N][S][S][5 + C]
    float* loss;          // [5]: class, object, noobject, coord, total
    float* ious;          // [N][S][S][B]
    float* mask;          // [N][S][S][B]
    float* dnet;          // [N][S][S][C+5B] or null
    float* partial;       // [nblocks][4]
    int N, S, B, C;
    float image_size;
    float lambda_coord, lambda_noobj;
};
int loss_blocks(int N, int S);
hipError_t launch_yolo_loss(const LossArgs& a, hipStream_t s);
hipError_t launch_get_iou(const float* b1, const float* b2, float* out, int n, hipStream_t s);
hipError_t launch_decode(const float* pred, int S, int B, int C, int im_w, int im_h, float thresh, int* out,
                         float* out_conf, hipStream_t s);
hipError_t launch_avgpool_fwd(const float* h, float* out, int N, int H, int W, int C, int k, hipStream_t s);
hipError_t launch_avgpool_bwd(const float* dout, float* dh, int N, int H, int W, int C, int k, hipStream_t s);
hipError_t launch_softmax_ce(const float* logits, const int* labels, float* loss, float* dlogits, int N, int C,
                             hipStream_t s);

hipError_t launch_accuracy(const float* logits, const int* labels, float* acc, int N, int C, hipStream_t s);

// ---- optimizers (flat buffers)
hipError_t launch_adam(float* p, float* m, float* v, const float* g, size_t n, float lr_t, float b1, float b2,
                       float eps, float gscale, hipStream_t s);
hipError_t launch_momentum(float* p, float* acc, const float* g, size_t n, float lr, float mom, float gscale,
                           hipStream_t s);
// dynamic loss scaling: ctrl = {int found_inf, int step, int skipped, float lr_t}
hipError_t launch_grad_check(const float* g, size_t n, void* ctrl, hipStream_t s);
hipError_t launch_grad_check_ranges(const float* g, const void* ranges_dev, int nranges, void* ctrl, hipStream_t s,
                                    unsigned* flag = nullptr, bool keep = false);   // keep: do not clear found_inf first
hipError_t launch_range_check(const float* x, size_t n, float limit, void* ctrl, hipStream_t s);   // found_inf |= !(|x| <= limit)
hipError_t launch_opt_ctrl_advance(void* ctrl, float lr, float b1, float b2, hipStream_t s);
hipError_t launch_adam_guarded(float* p, float* m, float* v, const float* g, size_t n, const void* ctrl, float b1,
                               float b2, float eps, float gscale, hipStream_t s);
hipError_t launch_momentum_guarded(float* p, float* acc, const float* g, size_t n, const void* ctrl, float lr, float mom,
                                   float gscale, hipStream_t s);
hipError_t launch_init_trunc_normal(float* p, size_t n, float stddev, uint64_t seed, uint64_t stream_id,
                                    hipStream_t s);
hipError_t launch_fill(float* p, size_t n, float v, hipStream_t s);

// per-(device, stream) scratch of the graph-level operators (split partial sums); grows on demand, never shrinks
void* op_scratch(hipStream_t s, size_t bytes);
int op_scratch_error();      // code of the last null return of op_scratch on this thread (its message is already in y2_last_error)
int* op_counters(hipStream_t s, size_t ints);      // zeroed, self-cleaning tile counters (WgradArgs::tile_cnt)
constexpr size_t kWgCntInts = 65536;

}  // namespace y2
