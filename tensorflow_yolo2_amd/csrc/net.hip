// Network executor + C ABI (include/yolo2_hip.h).
//
// Replaces the TF1 graph that src/yolo2_nets/darknet.py builds out of
// conv_bn_layer (conv2d + bias -> batch_normalization -> leaky -> max_pool) and
// the autodiff graph behind tf.train.*Optimizer().minimize
// (src/pascal/pascal_train_darknet.py:49-51).  The executor owns no memory: it
// plans offsets inside ONE caller-provided workspace sized for MI355X's 288 GB
// HBM (every layer keeps its own bordered activation, conv output and gradient
// buffers: nothing is re-zeroed or re-laid-out between steps).
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/yolo2_hip.h"
#include "common.h"
#include "kernels.h"

using namespace y2;

static thread_local std::string g_err;
namespace y2 {
int set_error(int code, const char* msg) {   // shared with ext.hip
    g_err = msg;
    return code;
}
}
static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) return fail(Y2_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                          __FILE__, __LINE__);                                        \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int round_up(int v, int a) { return (v + a - 1) / a * a; }

static inline size_t part_rows_bound(int N, int H, int W, int cout) {
    size_t bp = conv_block_pixels(cout);
    if (bp > 64) bp = 64;     // the smallest pixel tile of any kernel form (conv_rf.hip: 64 bordered positions)
    return ((size_t)N * (H + 1) * (W + 1) + bp - 1) / bp;
}


// zero-bordered NHWC tensor with guard bands (see wgrad.hip): geometry helper
struct PadGeom {
    int N, H, W, C;
    size_t front_px() const { return (size_t)W + 3; }
    size_t body_px() const { return bbody_pixels(N, H, W); }
    size_t back_px() const { return (size_t)2 * W + 6 + 256; }
    size_t bytes(size_t sz) const { return align_up((front_px() + body_px() + back_px()) * C * sz + 64, 256); }
    size_t base_off(size_t sz) const { return front_px() * C * sz; }
};

struct Layer {
    int k, cin, cout, pool;
    int H, W, Ho, Wo, M;
    int cin_s;       // channel stride of the bordered input tensor
    int ldy;         // channel stride of the conv output / its gradient
    int cout_pad;    // rows of packed forward filter
    int cin_pad;     // rows of packed dgrad filter
    bool first3;     // Cin = 3 special layer
    float slope = 0.1f;   // activation slope (y2_set_layer_options): 0.1 leaky (the reference), 0 ReLU, 1 none
    size_t pW, pb, pg, pbeta;  // float offsets into params / grads
    size_t smm, smv;           // float offsets into state
    // byte offsets into the workspace
    size_t xin, y, stat, wf, wd, dyp, ysel, idx0;
};

struct y2_ctx {
    int N, H, W, dtype, tail, tail_k, core_layers;
    // Y2_F16X2F (round 6): dtype == 3 (split tensors, split-operand forward) and the backward contractions -- dgrad and
    // weight gradients -- read the hi planes only (launch dtype 4, common.h hsplith_t): one f16 MFMA per product
    int bwd_dtype = 0;
    std::vector<Layer> L;
    size_t nparams = 0, nstate = 0;
    int outN, outH, outW, outC;
    float grad_scale = 1.f;
    int bessel = 0;
    float bn_eps = 1e-3f, bn_momentum = 0.99f;   // tf.layers.batch_normalization defaults (y2_set_layer_options)
    int zero_bias_grad = 0;                       // the conv biases are not variables of the graph (slim conv2d + batch_norm)
    // bound memory
    float* params = nullptr;
    float* grads = nullptr;
    float* state = nullptr;
    char* ws = nullptr;
    size_t ws_bytes = 0;
    int bound_training = 0;
    int bind_gen = 0;              // incremented by every y2_bind: a pack-group table (absolute pointers) is valid for ONE binding
    int group_gen = -1;            // bind_gen recorded by y2_pack_group_table
    bool weights_dirty = true;
    bool fwd_saved = false;
    bool moving_pending = false;   // last forward ran with update_moving = 0
    std::vector<int> fwd_training;  // per layer BN mode of the last forward
    std::vector<int> fwd_folded;    // per layer: the last forward folded BN + leaky into the conv epilogue (no y)
    // shared scratch offsets
    size_t o_part_scratch = 0;
    size_t o_part_cnt, o_part_mean, o_part_m2, o_psum, o_dA0, o_dA1, o_h32, o_dh32, o_xin_last_end;
    size_t part_rows, part_ld;
    size_t total_infer = 0, total_train = 0;
    int dA_cur = 0;
    int dA_half = 0;            // f16x2f: the dA buffer in use holds f16 values (written by a launch-dtype-5 dgrad)
    // pooled 3-channel first layer, training: the linear form of its backward pass (conv1_wgrad.hip) -- its conv
    // output is never stored
    bool lin1() const { return bound_training && !L.empty() && L[0].idx0 != 0; }
    // f16x2: the 3-channel layer's passes form split-operand products in registers (conv1.hip XS forms) wherever no
    // exact-fp32 fallback recomputes its conv output (the linear-form backward, or no backward at all)
    bool xs1() const {
        static const bool off = getenv("Y2_CONV1_NO_XS") != nullptr;
        return dtype_split(dtype) && !off && (lin1() || !bound_training);
    }
    bool nosel1() const { return lin1() && L[0].ysel == 0; }     // ... and not even the arg-max outputs (Conv1PoolArgs::idx3)
    size_t o_infertab = 0;      // BnInferLayer per layer (one prepare launch for all inference-mode layers)
    size_t o_packtab = 0, o_chkranges = 0, o_smallranges = 0, o_lin = 0, o_nfflag = 0, o_slab = 0;
    size_t o_wgcnt = 0;         // tile counters of the in-kernel split-K sums (WgradArgs::tile_cnt; zero from y2_bind on)
    size_t o_ks = 0, ks_floats = 0;   // K-split partial tiles of small convolution launches (ConvArgs::ks_scratch)
    size_t o_gram = 0;          // first layer: Gram matrix of the input patches [48][48] + its slices / block partials
    bool gram_valid = false;    // the last forward computed it (training mode, pooled first layer): backward reuses it
    size_t slab_floats = 0;     // split-K partial tiles of the weight gradients (WgradArgs::slab)
    int n_chkranges = 0, n_smallranges = 0, opt_tile_blocks = 0;
    // both range tables list the layers above the first one first: the optimizer step fused into the backward pass
    // (y2_backward_adam / _momentum) checks and updates that part while the first layer's gradient is still
    // being computed
    int n_chk_upper = 0, n_small_upper = 0;
    struct FusedOpt {
        bool on = false;
        int kind = 0;
        float* slot0 = nullptr; float* slot1 = nullptr;
        void* ctrl = nullptr;
        float lr = 0.f, b1 = 0.f, b2 = 0.f, eps = 0.f, gmult = 1.f;
        int step = 0;
    } fopt;
    std::vector<PackLayer> packtab;
    std::vector<BnInferLayer> infertab;
    int pack_blocks = 0;
    // optional per-launch HIP-event bracketing (bench.py roofline leg)
    int prof = 0;   // 0 off, 1 every launch, 2 only the dominant kernel (conv forward + dgrad)
    struct ProfRec { int cat, layer; hipEvent_t a, b; };
    int prof_layer = -1;
    // weight gradients run on a side stream beside the dgrad of the same layer (both only read dY)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int overlap_wgrad = 1;
    // y2_backward_marks: (main, side) event pairs recorded when every layer >= mark_layers[k] is complete
    std::vector<hipEvent_t> mark_main, mark_side;
    int n_marks = 0;
    const int* cur_marks = nullptr;
    float* dinput = nullptr;        // y2_backward_input: gradient wrt the stack's input, fp32 [N,H,W,cin]
    // y2_link (round 5): tensors of the arithmetic type handed from / to the neighbouring stacks of a composed graph --
    // no fp32 round trip between the bottleneck units of the ResNet swap.  Bordered pointers address cell 0.
    void* ext_xin = nullptr;        // layer 0's bordered input (else: the fp32 images are packed into the workspace)
    void* ext_out = nullptr;        // the last layer's output, bordered [N][Ho+1][Wo+1][cout] (else: fp32 `out`)
    const void* ext_join = nullptr; // added before a final ReLU: bordered, the output's geometry
    int ext_join_self = 0;          // ... or the stack's own layer-0 input (identity shortcut)
    const void* ext_dout = nullptr; // gradient wrt the (pre-join) output, [M][ldy] of T, already times grad_scale
    void* ext_dx = nullptr;         // gradient wrt the stack's input, [M][cin] of T, times grad_scale
    std::vector<ProfRec> prof_recs;
    size_t prof_used = 0;
    size_t sz() const { return dtype_size(dtype); }
    PadGeom in_geom(int l) const { return PadGeom{N, L[l].H, L[l].W, L[l].cin_s}; }
    PadGeom dy_geom(int l) const { return PadGeom{N, L[l].H, L[l].W, L[l].ldy}; }
};

enum { CAT_CONV_FWD = 0, CAT_CONV1_FWD, CAT_DGRAD, CAT_WGRAD, CAT_CONV1_WGRAD, CAT_BN_FWD, CAT_BN_BWD, CAT_MISC,
       CAT_COUNT };

struct ProfScope {
    y2_ctx* c; hipStream_t s; int idx = -1;
    ProfScope(y2_ctx* c_, hipStream_t s_, int cat) : c(c_), s(s_) {
        if (!c->prof) return;
        if (c->prof == 2 && cat != CAT_CONV_FWD && cat != CAT_DGRAD && cat != CAT_WGRAD) return;
        if (c->prof_used == c->prof_recs.size()) {
            y2_ctx::ProfRec r; r.cat = cat;
            if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
            c->prof_recs.push_back(r);
        }
        idx = (int)c->prof_used++;
        c->prof_recs[idx].cat = cat;
        c->prof_recs[idx].layer = c->prof_layer;
        (void)hipEventRecord(c->prof_recs[idx].a, s);
    }
    ~ProfScope() { if (idx >= 0) (void)hipEventRecord(c->prof_recs[idx].b, s); }
};
#define PROF(cat) ProfScope _prof_scope(c, s, cat)

static void plan(y2_ctx* c) {
    const size_t sz = c->sz();
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += align_up(bytes, 256);
        return o;
    };
    size_t max_part_rows = 1, max_ld = 32, max_dA = 0;
    for (size_t l = 0; l < c->L.size(); ++l) {
        Layer& y = c->L[l];
        y.xin = take(c->in_geom((int)l).bytes(sz));
        y.y = take((size_t)y.M * y.ldy * sz + 256);
        y.stat = take((size_t)7 * y.ldy * sizeof(float));  // scale, shift, mean, invstd, coef[2], batch variance
        if (y.first3) {
            y.wf = take((size_t)32 * 3 * 16 * sz);
            y.wd = 0;
        } else {
            y.wf = take((size_t)y.cout_pad * y.k * y.k * y.cin_s * sz);
            y.wd = take((size_t)y.cin_pad * y.k * y.k * y.ldy * sz);
        }
        // one record per pixel tile; the persistent form tiles the bordered positions (conv_rf.hip)
        size_t prow = y.first3 ? 2048 : part_rows_bound(c->N, y.H, y.W, y.cout);
        if (prow > max_part_rows) max_part_rows = prow;
        if ((size_t)y.ldy > max_ld) max_ld = y.ldy;
        size_t mo = (size_t)c->N * y.Ho * y.Wo * y.ldy;
        if (mo > max_dA) max_dA = mo;
        size_t mi = (size_t)y.M * y.cin_s;
        if (mi > max_dA) max_dA = mi;
    }
    c->part_rows = max_part_rows;
    c->part_ld = max_ld;
    c->o_part_cnt = take(max_part_rows * sizeof(float));
    // partial slabs are sized for the worst (rows x channels) product over layers
    size_t max_slab = 0;
    for (auto& y : c->L) {
        size_t prow = y.first3 ? 2048 : part_rows_bound(c->N, y.H, y.W, y.cout);
        size_t s = prow * y.ldy;
        if (s > max_slab) max_slab = s;
    }
    c->o_part_mean = take(max_slab * sizeof(float));
    c->o_part_m2 = take(max_slab * sizeof(float));
    c->o_part_scratch = take((size_t)64 * (1 + 2 * max_ld) * sizeof(float));
    c->o_h32 = take(c->tail == Y2_TAIL_AVGPOOL ? (size_t)c->L.back().M * c->L.back().cout * sizeof(float) : 0);
    c->o_packtab = take(c->L.size() * sizeof(PackLayer));
    c->o_infertab = take(c->L.size() * sizeof(BnInferLayer));
    c->o_chkranges = take((c->L.size() + 1) * 2 * sizeof(unsigned));
    c->o_smallranges = take((c->L.size() + 1) * 2 * sizeof(unsigned));
    c->o_nfflag = take(256);
    {   // K-split partial tiles of the small launches (conv_haloq.hip haloq_ks): forward and dgrad of every layer
        size_t ks = 0;
        for (auto& y : c->L) {
            if (y.first3 || dtype_split(c->dtype)) continue;     // (the K split is not built for the split-operand mode)
            ks = std::max(ks, conv_ks_scratch_floats(y.k * y.k, y.M, y.ldy, y.cin_s * (int)sz));
            ks = std::max(ks, conv_ks_scratch_floats(y.k * y.k, y.M, y.cin, y.ldy * (int)sz));
        }
        c->ks_floats = ks;
        c->o_ks = take(ks * sizeof(float));
    }
    c->total_infer = off;
    // ---- training-only buffers
    for (size_t l = 0; l < c->L.size(); ++l) c->L[l].dyp = take(c->dy_geom((int)l).bytes(sz));
    for (size_t l = 0; l < c->L.size(); ++l) {   // pooled layers: conv output at the arg-max (BnActArgs::ysel)
        Layer& y = c->L[l];
        const bool lin1 = y.first3 && c->L.size() > 1 && conv1_pool_ok(y.H, y.W, y.pool, y.cout) &&
                          conv1_wgrad_lin_ok(y.H, y.W, y.pool, y.ldy, (int)sz);
        // round 4, 16-bit types: the linear form keeps NO conv output of the first layer, only 3 index bits per element
        // (kernels.h Conv1PoolArgs::idx3); Y2_CONV1_YSEL=1 (and the f32 parity mode) keep ysel + 2 index bits
        static const bool keep_ysel = getenv("Y2_CONV1_YSEL") != nullptr;
        // (f16x2: the layer's fp32-operand kernels form split products in registers -- conv1.hip / conv1_wgrad.hip XS forms -- and
        //  take the 3-bit form too: 241 -> 210 us forward, 657 -> 617 us backward on one box; on the exact-fp32 matrix
        //  instructions, Y2_CONV1_NO_XS=1, it had changed nothing.  Y2_CONV1_XS_YSEL=1 keeps ysel there for A/B)
        static const bool xs_off = getenv("Y2_CONV1_NO_XS") != nullptr;
        static const bool xs_sel = getenv("Y2_CONV1_XS_YSEL") != nullptr;
        const bool nosel = lin1 && (dtype_plain(c->dtype) != 0 || (dtype_split(c->dtype) && !xs_off && !xs_sel)) && !keep_ysel;
        y.ysel = (y.pool && (!y.first3 || (lin1 && !nosel))) ? take((size_t)c->N * y.Ho * y.Wo * y.ldy * sz + 256) : 0;
        y.idx0 = lin1 ? take((size_t)c->N * y.Ho * y.Wo * (y.ldy * sz / 16) * (nosel ? sizeof(unsigned) : sizeof(unsigned short)) + 256) : 0;
        if (lin1) c->o_lin = take(conv1_wgrad_lin_scratch_floats() * sizeof(float));
        if (lin1 && conv1_gram_ok(y.H, y.W, (int)sz)) c->o_gram = take(conv1_gram_scratch_floats() * sizeof(float));
    }
    // BN-backward partial sums [P][2][ldy]: P <= 2048 from the reduce kernel, or one record per 128+ pixel tile
    // of the dgrad above when the reduce is fused into that dgrad's epilogue
    size_t psum_floats = (size_t)2048 * 2 * max_ld;
    for (size_t l = 1; l < c->L.size(); ++l) {
        // one record per 128+ position tile; the persistent form tiles the bordered positions (conv_rf.hip)
        const size_t rows = ((size_t)c->N * (c->L[l].H + 1) * (c->L[l].W + 1) + 127) / 128;
        psum_floats = std::max(psum_floats, rows * 2 * (size_t)c->L[l - 1].ldy);
    }
    c->o_psum = take(psum_floats * sizeof(float));
    // split-K partials of one weight-gradient launch: at most ~1,000 workgroups x one 64 x 32 x 9 (or 128 x 128) tile
    // (f16x2: three operand-plane pairs per launch, each with its own partial tiles)
    c->slab_floats = (size_t)1024 * 18432 * (dtype_split(c->dtype) ? 3 : 1);
    c->o_slab = take(c->slab_floats * sizeof(float));
    c->o_wgcnt = take(kWgCntInts * sizeof(int));
    c->o_dA0 = take(max_dA * sz + 256);
    c->o_dA1 = take(max_dA * sz + 256);
    c->o_dh32 = take(c->tail == Y2_TAIL_AVGPOOL ? (size_t)c->L.back().M * c->L.back().cout * sizeof(float) : 0);
    c->total_train = off;
}

extern "C" {

const char* y2_last_error(void) { return g_err.c_str(); }
int y2_version(void) { return 1; }

int y2_darknet19_spec(int kind, int output_filter, int* spec, int max_layers) {
    static const int core[18][4] = {
        {3, 3, 32, 1},     {3, 32, 64, 1},    {3, 64, 128, 0},   {3, 128, 64, 0},   {3, 64, 128, 1},
        {3, 128, 256, 0},  {1, 256, 128, 0},  {3, 128, 256, 1},  {3, 256, 512, 0},  {1, 512, 256, 0},
        {3, 256, 512, 0},  {1, 512, 256, 0},  {3, 256, 512, 1},  {3, 512, 1024, 0}, {1, 1024, 512, 0},
        {3, 512, 1024, 0}, {1, 1024, 512, 0}, {3, 512, 1024, 0}};
    int n = 18 + (kind == 1 ? 4 : (kind == 2 ? 1 : 0));
    if (kind < 0 || kind > 2) return fail(Y2_ERR_ARG, "unknown net kind %d", kind);
    if (n > max_layers) return fail(Y2_ERR_ARG, "spec buffer too small (%d > %d)", n, max_layers);
    memcpy(spec, core, sizeof(core));
    if (kind == 1) {
        for (int i = 0; i < 3; ++i) {
            int* s = spec + (18 + i) * 4;
            s[0] = 3; s[1] = 1024; s[2] = 1024; s[3] = 0;
        }
        int* s = spec + 21 * 4;
        s[0] = 1; s[1] = 1024; s[2] = output_filter; s[3] = 0;
    } else if (kind == 2) {
        int* s = spec + 18 * 4;
        s[0] = 1; s[1] = 1024; s[2] = 1000; s[3] = 0;
    }
    return n;
}

int y2_ctx_create(y2_ctx** out, const int* spec, int num_layers, int core_layers, int tail, int tail_k, int batch,
                  int height, int width, int dtype) {
    if (!out || !spec || num_layers <= 0) return fail(Y2_ERR_ARG, "bad arguments");
    if (dtype < 0 || dtype > 4) return fail(Y2_ERR_ARG, "dtype must be 0 (f32), 1 (f16), 2 (bf16), 3 (f16x2) or 4 (f16x2f)");
    if (batch <= 0 || height <= 0 || width <= 0) return fail(Y2_ERR_ARG, "bad input shape");
    y2_ctx* c = new y2_ctx();
    c->N = batch; c->H = height; c->W = width;
    c->bwd_dtype = dtype;                       // what the dgrad / weight-gradient launches run in
    c->dtype = dtype = (dtype == 4 ? 3 : dtype);   // tensors, forward pass, batch norm, optimizer: the split-operand mode's
    c->tail = tail; c->tail_k = tail_k; c->core_layers = core_layers;
    int h = height, w = width, cprev = 3;
    size_t po = 0, so = 0;
    for (int l = 0; l < num_layers; ++l) {
        Layer y{};
        y.k = spec[l * 4 + 0]; y.cin = spec[l * 4 + 1]; y.cout = spec[l * 4 + 2]; y.pool = spec[l * 4 + 3];
        if (y.pool < 0 || y.pool > 2 || (y.pool == 2 && ((h & 1) || (w & 1) || l == 0 || l + 1 == num_layers))) {
            delete c;
            return fail(Y2_ERR_ARG, "layer %d: pool is 0, 1 (2x2 max pool) or 2 (subsample: an inner layer on an even map)", l);
        }
        if ((y.k != 1 && y.k != 3) || y.cin <= 0 || y.cout <= 0) {
            delete c;
            return fail(Y2_ERR_ARG, "layer %d: only 1x1 / 3x3 stride-1 SAME convolutions exist in this network", l);
        }
        if (l > 0 && y.cin != cprev) {
            delete c;
            return fail(Y2_ERR_ARG, "layer %d: in_chl %d does not match previous out_chl %d", l, y.cin, cprev);
        }
        if (y.cout > (dtype_size(dtype) == 4 ? 1024 : 2048)) {
            delete c;
            return fail(Y2_ERR_ARG, "layer %d: out_chl %d exceeds the batch-norm kernels' row width (%d)", l, y.cout,
                        dtype_size(dtype) == 4 ? 1024 : 2048);
        }
        y.first3 = (l == 0 && y.cin == 3);
        if (y.first3 && !(y.k == 3 && y.cout == 32)) {
            delete c;
            return fail(Y2_ERR_ARG, "the 3-channel input layer must be 3x3, 3->32 (darknet.py:150)");
        }
        if (!y.first3 && (y.cin % 32) != 0) {
            delete c;
            return fail(Y2_ERR_ARG, "layer %d: in_chl must be a multiple of 32", l);
        }
        if (!y.first3 && y.cin > 128 && (y.cin % 128) != 0) {
            delete c;
            return fail(Y2_ERR_ARG, "layer %d: in_chl above 128 must be a multiple of 128", l);
        }
        if (l + 1 < num_layers && (y.cout % 32) != 0) {
            delete c;
            return fail(Y2_ERR_ARG, "layer %d: inner out_chl must be a multiple of 32", l);
        }
        y.H = h; y.W = w; y.M = batch * h * w;
        y.Ho = y.pool ? (h + 1) / 2 : h;
        y.Wo = y.pool ? (w + 1) / 2 : w;
        y.cin_s = y.first3 ? 4 : y.cin;
        y.ldy = round_up(y.cout, 32);   // 30 -> 32, 1000 -> 1024: K of the dgrad GEMM in 64-byte multiples
        y.cout_pad = round_up(y.cout, conv_block_couts(y.cout));
        y.cin_pad = round_up(y.cin, conv_block_couts(y.cin));
        y.pW = po; po += (size_t)y.k * y.k * y.cin * y.cout;
        y.pb = po; po += y.cout;
        y.pg = po; po += y.cout;
        y.pbeta = po; po += y.cout;
        y.smm = so; so += y.cout;
        y.smv = so; so += y.cout;
        c->L.push_back(y);
        h = y.Ho; w = y.Wo; cprev = y.cout;
    }
    c->nparams = po; c->nstate = so;
    const Layer& last = c->L.back();
    if (tail == Y2_TAIL_AVGPOOL) {
        if (tail_k <= 0 || last.Ho < tail_k || last.Wo < tail_k) {
            delete c;
            return fail(Y2_ERR_ARG, "avgpool window %d does not fit %dx%d", tail_k, last.Ho, last.Wo);
        }
        c->outN = batch; c->outH = last.Ho / tail_k; c->outW = last.Wo / tail_k; c->outC = last.cout;
    } else {
        c->outN = batch; c->outH = last.Ho; c->outW = last.Wo; c->outC = last.cout;
    }
    if (last.pool) {
        delete c;
        return fail(Y2_ERR_ARG, "the last layer must not pool");
    }
    plan(c);
    c->fwd_training.assign(c->L.size(), 0);
    c->fwd_folded.assign(c->L.size(), 0);
    *out = c;
    return Y2_OK;
}

void y2_ctx_destroy(y2_ctx* ctx) {
    if (!ctx) return;
    for (auto& r : ctx->prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    if (ctx->side) (void)hipStreamDestroy(ctx->side);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    for (auto e : ctx->mark_main) (void)hipEventDestroy(e);
    for (auto e : ctx->mark_side) (void)hipEventDestroy(e);
    delete ctx;
}

// on: 0 stop (records are kept until collected), 1 / 2 start afresh, 3 resume mode 2 without clearing
// (sampling: bracket only some steps of a timed region)
int y2_profile_enable(y2_ctx* c, int on) {
    if (on == 1 || on == 2) c->prof_used = 0;
    c->prof = on == 3 ? 2 : on;
    return Y2_OK;
}
int y2_profile_collect(y2_ctx* c, double* ms, int* count, int ncat) {
    if (ncat < CAT_COUNT) return fail(Y2_ERR_ARG, "need room for %d categories", (int)CAT_COUNT);
    for (int i = 0; i < ncat; ++i) { ms[i] = 0.0; count[i] = 0; }
    for (size_t i = 0; i < c->prof_used; ++i) {
        auto& r = c->prof_recs[i];
        HIPCHK(hipEventSynchronize(r.b));
        float t = 0.f;
        HIPCHK(hipEventElapsedTime(&t, r.a, r.b));
        ms[r.cat] += t;
        count[r.cat] += 1;
    }
    c->prof_used = 0;
    return Y2_OK;
}
// per-layer, per-category milliseconds [num_layers][CAT_COUNT] of the records so far; does not reset
int y2_profile_layers(y2_ctx* c, double* ms) {
    const int nl = (int)c->L.size();
    for (int i = 0; i < nl * CAT_COUNT; ++i) ms[i] = 0.0;
    for (size_t i = 0; i < c->prof_used; ++i) {
        auto& r = c->prof_recs[i];
        HIPCHK(hipEventSynchronize(r.b));
        float t = 0.f;
        HIPCHK(hipEventElapsedTime(&t, r.a, r.b));
        if (r.layer >= 0 && r.layer < nl) ms[r.layer * CAT_COUNT + r.cat] += t;
    }
    return Y2_OK;
}
// Busy time of the launches of the categories in `cat_mask`: the length of the UNION of their
// [start, end] intervals (the weight gradients run on a side stream beside the dgrads, so the sum of
// the individual durations would count shared machine time twice).  Does not reset.
int y2_profile_busy(y2_ctx* c, int cat_mask, double* busy_ms, int* launches) {
    std::vector<std::pair<float, float>> iv;
    if (c->prof_used == 0) { *busy_ms = 0.0; *launches = 0; return Y2_OK; }
    const hipEvent_t ref = c->prof_recs[0].a;
    for (size_t i = 0; i < c->prof_used; ++i) {
        auto& r = c->prof_recs[i];
        if (!((cat_mask >> r.cat) & 1)) continue;
        HIPCHK(hipEventSynchronize(r.b));
        float t0 = 0.f, t1 = 0.f;
        HIPCHK(hipEventElapsedTime(&t0, ref, r.a));
        HIPCHK(hipEventElapsedTime(&t1, ref, r.b));
        iv.emplace_back(t0, t1);
    }
    std::sort(iv.begin(), iv.end());
    double busy = 0.0;
    float cur0 = 0.f, cur1 = -1.f;
    for (auto& p : iv) {
        if (cur1 < cur0 || p.first > cur1) {
            if (cur1 >= cur0) busy += cur1 - cur0;
            cur0 = p.first; cur1 = p.second;
        } else if (p.second > cur1) {
            cur1 = p.second;
        }
    }
    if (cur1 >= cur0) busy += cur1 - cur0;
    *busy_ms = busy;
    *launches = (int)iv.size();
    return Y2_OK;
}
int y2_num_layers(const y2_ctx* c) { return (int)c->L.size(); }
int y2_layer_info(const y2_ctx* c, int l, int info[8]) {
    if (l < 0 || l >= (int)c->L.size()) return fail(Y2_ERR_ARG, "layer out of range");
    const Layer& y = c->L[l];
    info[0] = y.k; info[1] = y.cin; info[2] = y.cout; info[3] = y.pool;
    info[4] = y.H; info[5] = y.W; info[6] = y.Ho; info[7] = y.Wo;
    return Y2_OK;
}
size_t y2_param_count(const y2_ctx* c) { return c->nparams; }
size_t y2_state_count(const y2_ctx* c) { return c->nstate; }
int y2_param_offsets(const y2_ctx* c, int l, size_t off[6]) {
    if (l < 0 || l >= (int)c->L.size()) return fail(Y2_ERR_ARG, "layer out of range");
    const Layer& y = c->L[l];
    off[0] = y.pW; off[1] = y.pb; off[2] = y.pg; off[3] = y.pbeta; off[4] = y.smm; off[5] = y.smv;
    return Y2_OK;
}
int y2_output_shape(const y2_ctx* c, int shape[4]) {
    shape[0] = c->outN; shape[1] = c->outH; shape[2] = c->outW; shape[3] = c->outC;
    return Y2_OK;
}
size_t y2_workspace_bytes(const y2_ctx* c, int training) { return training ? c->total_train : c->total_infer; }

int y2_bind(y2_ctx* c, float* params, float* grads, float* state, void* workspace, size_t workspace_bytes,
            int training, void* stream) {
    if (!params || !state || !workspace) return fail(Y2_ERR_ARG, "null buffer");
    if (training && !grads) return fail(Y2_ERR_ARG, "training needs a gradient buffer");
    const size_t need = y2_workspace_bytes(c, training);
    if (workspace_bytes < need) return fail(Y2_ERR_ARG, "workspace too small: %zu < %zu", workspace_bytes, need);
    c->params = params; c->grads = grads; c->state = state;
    c->ws = (char*)workspace; c->ws_bytes = workspace_bytes; c->bound_training = training;
    c->weights_dirty = true; c->fwd_saved = false;
    ++c->bind_gen;
    // borders, guard bands and channel padding must be (and then stay) zero
    HIPCHK(hipMemsetAsync(workspace, 0, need, (hipStream_t)stream));
    // filter re-pack job table (one launch per step for all layers)
    c->packtab.clear();
    int nb = 0, ntile = 0;
    for (size_t l = 0; l < c->L.size(); ++l) {
        const Layer& y = c->L[l];
        if (y.first3) continue;
        PackLayer p{};
        p.W = params + y.pW;
        p.w_off = y.pW;
        p.wf = c->ws + y.wf;
        p.wd = training ? (void*)(c->ws + y.wd) : nullptr;   // layer 0 too: y2_backward_input wants its dgrad
        p.taps = y.k * y.k; p.Cin = y.cin; p.Cout = y.cout; p.Cout_pad = y.cout_pad; p.Kc = y.cin_s;
        p.Cin_pad = y.cin_pad; p.Cdy = y.ldy;
        const int kb = dtype_kbytes(c->dtype), split = dtype_split(c->dtype) ? 1 : 0;
        p.wf_frag = conv_filter_layout(p.taps, y.W, y.cin_s * kb, y.cout, y.M, 0, (int)c->sz(), split);   // forward launch
        p.wd_frag = conv_filter_layout(p.taps, y.W, y.ldy * kb, y.cin, y.M, 1, (int)c->sz(), split);   // dgrad launch: Cout = cin
        pack_layer_plan(p, nb, split ? 2 : (int)c->sz());     // (f16x2: the pack kernels run their 16-bit form on two planes)
        nb += p.wf_blocks + p.wd_blocks;
        p.opt_first = ntile;
        ntile += p.wf_blocks;
        c->packtab.push_back(p);
    }
    c->pack_blocks = nb;
    c->opt_tile_blocks = ntile;
    {   // parameters outside the filter tiles of the fused optimizer + re-pack pass
        std::vector<unsigned> rg;
        for (size_t l = 1; l <= c->L.size(); ++l) {      // layer 0 last
            const Layer& y = c->L[l % c->L.size()];
            rg.push_back((unsigned)y.pb);
            rg.push_back((unsigned)(3 * y.cout));
        }
        c->n_small_upper = (int)c->L.size() - 1;
        if (c->L[0].first3) {
            rg.push_back((unsigned)c->L[0].pW);
            rg.push_back((unsigned)(27 * c->L[0].cout));
        }
        c->n_smallranges = (int)(rg.size() / 2);
        HIPCHK(hipMemcpyAsync(c->ws + c->o_smallranges, rg.data(), rg.size() * sizeof(unsigned), hipMemcpyHostToDevice,
                              (hipStream_t)stream));
    }
    {   // sentinel ranges of y2_grad_check: b, gamma, beta of every layer (contiguous) + the first filter
        std::vector<unsigned> rg;
        for (size_t l = 1; l <= c->L.size(); ++l) {      // layer 0 last
            const Layer& y = c->L[l % c->L.size()];
            rg.push_back((unsigned)y.pb);
            rg.push_back((unsigned)(3 * y.cout));
        }
        c->n_chk_upper = (int)c->L.size() - 1;
        // first filter: a dy_0 that overflows at its own store makes every dW_0[t][ci][co] of its channel co
        // non-finite (inf * x, or inf * 0 = NaN), so one (tap 0, ci 0) row of couts is a complete sentinel
        rg.push_back((unsigned)c->L[0].pW);
        rg.push_back((unsigned)(c->L[0].first3 ? 27 * c->L[0].cout : c->L[0].cout));
        c->n_chkranges = (int)(rg.size() / 2);
        HIPCHK(hipMemcpyAsync(c->ws + c->o_chkranges, rg.data(), rg.size() * sizeof(unsigned), hipMemcpyHostToDevice,
                              (hipStream_t)stream));
    }
    if (!c->packtab.empty())
        HIPCHK(hipMemcpyAsync(c->ws + c->o_packtab, c->packtab.data(), c->packtab.size() * sizeof(PackLayer),
                              hipMemcpyHostToDevice, (hipStream_t)stream));
    {
        std::vector<BnInferLayer>& tab = c->infertab;      // a member: the asynchronous copy reads it after this returns
        tab.clear();
        for (size_t l = 0; l < c->L.size(); ++l) {
            const Layer& y = c->L[l];
            float* stat = (float*)(c->ws + y.stat);
            BnInferLayer t{};
            t.gamma = params + y.pg; t.beta = params + y.pbeta; t.mm = state + y.smm; t.mv = state + y.smv;
            t.scale = stat; t.shift = stat + y.ldy; t.mean = stat + 2 * y.ldy; t.invstd = stat + 3 * y.ldy;
            t.C = y.cout; t.is_core = (int)l < c->core_layers ? 1 : 0;
            tab.push_back(t);
        }
        HIPCHK(hipMemcpyAsync(c->ws + c->o_infertab, tab.data(), tab.size() * sizeof(BnInferLayer), hipMemcpyHostToDevice,
                              (hipStream_t)stream));
    }
    return Y2_OK;
}

int y2_set_options(y2_ctx* c, float grad_scale, int bessel) {
    if (!(grad_scale > 0.f)) return fail(Y2_ERR_ARG, "grad_scale must be positive");
    c->grad_scale = grad_scale; c->bessel = bessel;
    return Y2_OK;
}

// Per-layer activation slopes and the batch-norm constants of the stack (round 4).  Defaults = the reference's
// darknet.py: leaky 0.1 (:5,45), tf.layers.batch_normalization eps 1e-3 / momentum 0.99 (:39-44), a conv bias per layer
// (:33-35).  slim's resnet_v1 bottlenecks (src/slim_dir/nets/resnet_v1.py:99-112, resnet_utils.py:230-257 arg scope) are
// conv2d(no bias) + batch_norm(decay 0.997, epsilon 1e-5) + ReLU, the last one of a unit without activation: slopes 0
// (ReLU) / 1 (none), eps 1e-5, momentum 0.997, zero_bias_grad = 1 (the bias slots stay in the flat layout, the caller
// keeps them at zero and they receive no gradient).  slopes == NULL keeps the current slopes.
int y2_set_layer_options(y2_ctx* c, const float* slopes, int num_layers, float bn_eps, float bn_momentum,
                         int zero_bias_grad) {
    if (slopes && num_layers != (int)c->L.size()) return fail(Y2_ERR_ARG, "one slope per layer (%d layers)", (int)c->L.size());
    if (!(bn_eps > 0.f) || !(bn_momentum >= 0.f && bn_momentum < 1.f)) return fail(Y2_ERR_ARG, "bad batch-norm constants");
    if (slopes)
        for (size_t l = 0; l < c->L.size(); ++l) {
            if (!(slopes[l] >= 0.f && slopes[l] <= 1.f)) return fail(Y2_ERR_ARG, "layer %d: slope must be in [0, 1]", (int)l);
            if (c->L[l].first3 && slopes[l] != 0.1f)
                return fail(Y2_ERR_ARG, "the 3-channel image layer's kernels implement the reference's leaky 0.1 only");
        }
    if (slopes)
        for (size_t l = 0; l < c->L.size(); ++l) c->L[l].slope = slopes[l];
    c->bn_eps = bn_eps; c->bn_momentum = bn_momentum; c->zero_bias_grad = zero_bias_grad ? 1 : 0;
    return Y2_OK;
}

int y2_init_params(y2_ctx* c, uint64_t seed, void* stream) {
    if (!c->params) return fail(Y2_ERR_STATE, "bind buffers first");
    hipStream_t s = (hipStream_t)stream;
    for (size_t l = 0; l < c->L.size(); ++l) {
        const Layer& y = c->L[l];
        HIPCHK(launch_init_trunc_normal(c->params + y.pW, (size_t)y.k * y.k * y.cin * y.cout, 0.1f, seed, l, s));
        HIPCHK(launch_fill(c->params + y.pb, y.cout, 0.1f, s));
        HIPCHK(launch_fill(c->params + y.pg, y.cout, 1.0f, s));
        HIPCHK(launch_fill(c->params + y.pbeta, y.cout, 0.0f, s));
        HIPCHK(launch_fill(c->state + y.smm, y.cout, 0.0f, s));
        HIPCHK(launch_fill(c->state + y.smv, y.cout, 1.0f, s));
    }
    c->weights_dirty = true;
    return Y2_OK;
}
int y2_params_changed(y2_ctx* c) {
    c->weights_dirty = true;
    return Y2_OK;
}

static int pack_all_weights(y2_ctx* c, hipStream_t s) {
    PROF(CAT_MISC);
    if (!c->L.empty() && c->L[0].first3)
        HIPCHK(launch_pack_conv1_weights(c->dtype, c->params + c->L[0].pW, c->ws + c->L[0].wf, s));
    if (c->pack_blocks > 0)
        HIPCHK(launch_pack_all(c->dtype, (const PackLayer*)(c->ws + c->o_packtab), (int)c->packtab.size(),
                               c->pack_blocks, s));
    c->weights_dirty = false;
    return Y2_OK;
}

// ONE filter-pack launch for a GROUP of contexts (round 5: the ResNet swap binds 20 small stacks to one flat parameter
// buffer; after every optimizer step each of them used to re-pack its three filters in its own 10-us launch).
// y2_pack_group_table: the concatenated pack tables of the contexts (first blocks re-based) into device memory the caller
// owns -- a synchronous copy, once, outside any capture; returns layers / blocks of the group launch.
// y2_pack_group_run: that launch; the contexts' packed copies are current afterwards (their own lazy pack is skipped).
int y2_pack_group_table(y2_ctx** ctxs, int n, void* table_dev, size_t table_bytes, int* nlayers, int* blocks) {
    if (!ctxs || n < 1 || !table_dev || !nlayers || !blocks) return fail(Y2_ERR_ARG, "y2_pack_group_table: bad arguments");
    std::vector<PackLayer> all;
    int nb = 0;
    for (int i = 0; i < n; ++i) {
        y2_ctx* c = ctxs[i];
        if (!c || !c->ws) return fail(Y2_ERR_STATE, "y2_pack_group_table: context %d is not bound", i);
        if (c->dtype != ctxs[0]->dtype) return fail(Y2_ERR_ARG, "y2_pack_group_table: one arithmetic type per group");
        if (!c->L.empty() && c->L[0].first3) return fail(Y2_ERR_ARG, "y2_pack_group_table: a 3-channel first layer packs on its own");
        for (PackLayer p : c->packtab) {
            p.first_block += nb;
            all.push_back(p);
        }
        nb += c->pack_blocks;
        c->group_gen = c->bind_gen;
    }
    if (all.size() * sizeof(PackLayer) > table_bytes) return fail(Y2_ERR_ARG, "y2_pack_group_table: %zu bytes needed", all.size() * sizeof(PackLayer));
    if (!all.empty()) HIPCHK(hipMemcpy(table_dev, all.data(), all.size() * sizeof(PackLayer), hipMemcpyHostToDevice));
    *nlayers = (int)all.size();
    *blocks = nb;
    return Y2_OK;
}
int y2_pack_group_run(y2_ctx** ctxs, int n, const void* table_dev, int nlayers, int blocks, void* stream) {
    if (!ctxs || n < 1 || !table_dev) return fail(Y2_ERR_ARG, "y2_pack_group_run: bad arguments");
    // the table holds absolute pointers copied at y2_pack_group_table time: a member re-bound since then (y2_bind /
    // Network.rebind) would silently run on stale packed filters -- refuse instead
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i] || !ctxs[i]->ws) return fail(Y2_ERR_STATE, "y2_pack_group_run: context %d is not bound", i);
        if (ctxs[i]->group_gen != ctxs[i]->bind_gen)
            return fail(Y2_ERR_STATE, "y2_pack_group_run: context %d was re-bound after y2_pack_group_table: build the table again", i);
    }
    if (nlayers > 0 && blocks > 0)
        HIPCHK(launch_pack_all(ctxs[0]->dtype, (const PackLayer*)table_dev, nlayers, blocks, (hipStream_t)stream));
    for (int i = 0; i < n; ++i) ctxs[i]->weights_dirty = false;
    return Y2_OK;
}

static int forward_impl(y2_ctx* c, const float* images, const uint8_t* images_u8, int train_core, int train_head,
                        int update_moving, float* out, void* stream, const float* join = nullptr);
int y2_forward(y2_ctx* c, const float* images, int train_core, int train_head, int update_moving, float* out,
               void* stream) {
    if (!images && !c->ext_xin) return fail(Y2_ERR_ARG, "null tensor");
    return forward_impl(c, images, nullptr, train_core, train_head, update_moving, out, stream);
}
// The stack as the residual branch of a ResNet bottleneck unit: out = relu(join + stack(images)) written by the last
// layer's apply pass (src/slim_dir/nets/resnet_v1.py:112) -- the same values as y2_forward followed by y2_add_relu, bit
// for bit, without storing and re-reading the branch output
int y2_forward_join(y2_ctx* c, const float* images, const float* join, int train_core, int train_head, int update_moving,
                    float* out, void* stream) {
    if (!images || !join) return fail(Y2_ERR_ARG, "null tensor");
    if (c->tail == Y2_TAIL_AVGPOOL) return fail(Y2_ERR_ARG, "y2_forward_join: not for the average-pool tail");
    if (join == out) return fail(Y2_ERR_ARG, "y2_forward_join: join and out must not alias");
    return forward_impl(c, images, nullptr, train_core, train_head, update_moving, out, stream, join);
}
int y2_forward_u8(y2_ctx* c, const uint8_t* images_u8, int train_core, int train_head, int update_moving, float* out,
                  void* stream) {
    if (!images_u8) return fail(Y2_ERR_ARG, "null tensor");
    if (c->L.empty() || !c->L[0].first3) return fail(Y2_ERR_ARG, "uint8 input needs the 3-channel image layer first");
    if (((uintptr_t)images_u8 & 3) != 0) return fail(Y2_ERR_ARG, "uint8 images must be 4-byte aligned");
    return forward_impl(c, nullptr, images_u8, train_core, train_head, update_moving, out, stream);
}
static int forward_impl(y2_ctx* c, const float* images, const uint8_t* images_u8, int train_core, int train_head,
                        int update_moving, float* out, void* stream, const float* join) {
    if (!c->ws) return fail(Y2_ERR_STATE, "bind buffers first");
    if (!out && !c->ext_out) return fail(Y2_ERR_ARG, "null tensor");
    hipStream_t s = (hipStream_t)stream;
    const size_t sz = c->sz();
    if (c->weights_dirty) {
        int r = pack_all_weights(c, s);
        if (r) return r;
    }
    float* part_cnt = (float*)(c->ws + c->o_part_cnt);
    float* part_mean = (float*)(c->ws + c->o_part_mean);
    float* part_m2 = (float*)(c->ws + c->o_part_m2);
    const int nl = (int)c->L.size();
    if (!train_core || (!train_head && c->core_layers < nl)) {     // some layer normalises with its moving statistics
        int max_c = 0;
        for (const Layer& y : c->L) max_c = y.cout > max_c ? y.cout : max_c;
        HIPCHK(launch_bn_infer_prepare_all((const BnInferLayer*)(c->ws + c->o_infertab), nl, max_c, train_core, train_head,
                                           c->bn_eps, s));
    }
    for (int l = 0; l < nl; ++l) {
        c->prof_layer = l;
        const Layer& y = c->L[l];
        const int training = (l < c->core_layers) ? train_core : train_head;
        c->fwd_training[l] = training;
        char* xin = (l == 0 && c->ext_xin) ? (char*)c->ext_xin : c->ws + y.xin + c->in_geom(l).base_off(sz);
        float* stat = (float*)(c->ws + y.stat);
        float *scale = stat, *shift = stat + y.ldy, *mean = stat + 2 * y.ldy, *invstd = stat + 3 * y.ldy;
        int P = 0;
        bool folded = false, gram1 = false;
        // pooled first layer: statistics-only conv, then conv again fused with BN + leaky + pool
        const bool pool1 = y.first3 && l + 1 < nl && y.ldy == 32 && conv1_pool_ok(y.H, y.W, y.pool, y.cout);
        if (y.first3) {
            {
                PROF(CAT_MISC);
                if (images_u8) HIPCHK(launch_pack_input_u8(c->dtype, images_u8, xin, c->N, y.H, y.W, s));
                else HIPCHK(launch_pack_input(c->dtype, images, xin, c->N, y.H, y.W, s));
            }
            Conv1Args a{};
            a.x4 = xin; a.w = c->ws + y.wf; a.y = c->ws + y.y; a.bias = c->params + y.pb;
            a.part_cnt = part_cnt; a.part_mean = part_mean; a.part_m2 = part_m2;
            a.N = c->N; a.H = y.H; a.W = y.W; a.M = y.M;
            int nb = (y.M + 127) / 128;
            a.nblocks = nb > 1024 ? 1024 : nb;      // = statistics records (the plan reserves 2048 rows; 1024 vs 2048: -4 us)
            P = a.nblocks;
            a.stats_only = pool1 ? 1 : 0;
            a.xs = (pool1 && c->xs1()) ? 1 : 0;
            // round 4: the statistics of the pooled first layer come from the Gram matrix of the input patches (below)
            static const bool no_gram = getenv("Y2_NO_CONV1_GRAM") != nullptr;
            // Half-precision modes only: their stored activations carry 5e-4 of rounding noise, against which the ~1e-6
            // of the fp32 MFMA sums behind the Gram moments is nothing.  The f32 parity mode keeps the statistics-only
            // convolution pass (moments of the exact fp32 outputs, double merge): a randomly initialised 20-layer
            // stack amplifies a 1e-6 perturbation of the first scale / shift to 1e-3 at the top, enough to flip
            // leaky / arg-max decisions the f32 tests compare element-wise with the oracle.
            gram1 = pool1 && training && c->lin1() && c->o_gram != 0 && !no_gram && dtype_plain(c->dtype) != 0;
            if ((!pool1 || training) && !gram1) { PROF(CAT_CONV1_FWD); HIPCHK(launch_conv1_fwd(c->dtype, a, s)); }
            c->gram_valid = gram1;
        } else {
            if (l == 0 && !c->ext_xin) HIPCHK(launch_pack_act(c->dtype, images, xin, c->N, y.H, y.W, y.cin, y.cin_s, s));
            ConvArgs a{};
            a.x = xin; a.w = c->ws + y.wf; a.y = c->ws + y.y; a.bias = c->params + y.pb;
            // (subsampling layers, pool == 2: their batch norm runs over the kept positions -- launch_bn_stats_sub below)
            if (training && y.pool != 2) { a.part_cnt = part_cnt; a.part_mean = part_mean; a.part_m2 = part_m2; }
            a.N = c->N; a.H = y.H; a.W = y.W; a.C = y.cin_s; a.M = y.M; a.Cout = y.cout; a.ldy = y.ldy;
            a.taps = y.k * y.k;
            if (c->ks_floats) { a.ks_scratch = (float*)(c->ws + c->o_ks); a.ks_floats = c->ks_floats; }
            int bp = 0, rec = 0;
            // inference statistics, no pool, a consumer layer: scale / shift / leaky ride in the conv epilogue and the
            // activation goes straight into the consumer's bordered input (no y, no bn_act pass).  Training
            // bindings keep y: a later y2_backward of a frozen-core graph reads it.
            // (round 5: pooled layers on the conv_haloq kernels too -- window-major tiles, ConvArgs::aff_pool)
            if (!training && l + 1 < nl && !c->bound_training && y.ldy == c->L[l + 1].cin_s &&
                (y.pool == 1 ? conv_affine_pool_ok(c->dtype, a) : (y.pool == 0 && conv_affine_ok(c->dtype, a)))) {
                conv_set_affine(a, scale, shift, c->ws + c->L[l + 1].xin + c->in_geom(l + 1).base_off(sz));
                a.aff_slope = y.slope;
                a.aff_pool = y.pool == 1 ? 1 : 0;
                folded = true;
            }
            { PROF(CAT_CONV_FWD); HIPCHK(launch_conv(c->dtype, a, s, &bp, &rec)); }
            P = rec;
            if (training && y.pool == 2) {
                PROF(CAT_BN_FWD);
                HIPCHK(launch_bn_stats_sub(c->dtype, c->ws + y.y, c->N, y.H, y.W, y.ldy, part_cnt, part_mean, part_m2, &P, s));
            }
        }
        c->fwd_folded[l] = folded ? 1 : 0;
        if (folded) continue;
        PROF(CAT_BN_FWD);
        BnFinalizeArgs f{};
        if (training) {
            f.part_cnt = part_cnt; f.part_mean = part_mean; f.part_m2 = part_m2;
            f.P = P; f.C = y.cout; f.ldp = y.first3 ? 32 : y.ldy;
            f.gamma = c->params + y.pg; f.beta = c->params + y.pbeta;
            f.moving_mean = c->state + y.smm; f.moving_var = c->state + y.smv;
            f.scale = scale; f.shift = shift; f.mean = mean; f.invstd = invstd;
            f.var = stat + 6 * y.ldy;
            f.scratch = (float*)(c->ws + c->o_part_scratch);
            f.eps = c->bn_eps; f.momentum = c->bn_momentum; f.update_moving = update_moving ? 1 : 0; f.bessel = c->bessel;
        }
        // short partial lists: the merge rides in the apply pass (bn.hip bn_fin_act_kernel)
        static const bool no_fin_fuse = getenv("Y2_NO_BN_FIN_FUSE") != nullptr;
        bool fin_fused = false;
        if (training && !pool1 && !no_fin_fuse && (l + 1 < nl || c->ext_out)) {
            BnActArgs t{};
            t.C = y.cout; t.ldy = y.ldy; t.out_f32 = 0;
            fin_fused = bn_fin_act_ok(t, f);
        }
        if (gram1) {
            Conv1GramStatsArgs q{};
            q.x4 = xin; q.N = c->N; q.H = y.H; q.Wd = y.W;
            q.W = c->params + y.pW; q.bias = c->params + y.pb; q.gamma = f.gamma; q.beta = f.beta;
            q.moving_mean = f.moving_mean; q.moving_var = f.moving_var;
            q.scale = scale; q.shift = shift; q.mean = mean; q.invstd = invstd; q.var = f.var;
            q.eps = f.eps; q.momentum = f.momentum; q.update_moving = f.update_moving; q.bessel = f.bessel;
            q.gram = (float*)(c->ws + c->o_gram); q.mid = q.gram + 48 * 48;
            HIPCHK(launch_conv1_gram_stats(c->dtype, q, s));
        } else if (training && !fin_fused) HIPCHK(launch_bn_finalize(f, s));     // (inference: prepared for every layer above)
        if (pool1) {
            Conv1PoolArgs q{};
            q.x4 = xin; q.w = c->ws + y.wf; q.y = c->ws + y.y; q.bias = c->params + y.pb;
            q.scale = scale; q.shift = shift;
            q.out = c->ws + c->L[l + 1].xin + c->in_geom(l + 1).base_off(sz);
            q.N = c->N; q.H = y.H; q.W = y.W;
            const int tiles = c->N * (y.H / 2) * ((y.W + 31) / 32);
            q.nblocks = (tiles + 3) / 4 > 2048 ? 2048 : (tiles + 3) / 4;
            q.store_y = (c->bound_training && !c->lin1()) ? 1 : 0;
            q.out_split = dtype_split(c->dtype) ? 1 : 0;
            q.xs = c->xs1() ? 1 : 0;
            if (c->nosel1()) q.idx3 = (unsigned*)(c->ws + y.idx0);
            else if (c->lin1()) { q.ysel = c->ws + y.ysel; q.idx = (unsigned short*)(c->ws + y.idx0); }
            HIPCHK(launch_conv1_pool(c->dtype, q, s));
            continue;
        }
        BnActArgs b{};
        b.y = c->ws + y.y; b.scale = scale; b.shift = shift;
        b.N = c->N; b.H = y.H; b.W = y.W; b.C = y.cout; b.ldy = y.ldy; b.pool = y.pool;
        b.slope = y.slope;
        if (l + 1 < nl) {
            b.out = c->ws + c->L[l + 1].xin + c->in_geom(l + 1).base_off(sz);
            b.out_f32 = 0;
        } else if (c->ext_out) {      // linked: the consumer stack's bordered input, in the arithmetic type
            b.out = c->ext_out;
            b.out_f32 = 0;
            b.join_t = c->ext_join_self ? (const void*)(c->ext_xin ? (char*)c->ext_xin : c->ws + c->L[0].xin + c->in_geom(0).base_off(sz))
                                        : c->ext_join;
        } else {
            b.out = (c->tail == Y2_TAIL_AVGPOOL) ? (void*)(c->ws + c->o_h32) : (void*)out;
            b.out_f32 = 1;
            b.join = join;
            if (!join) b.join_t = c->ext_join_self ? (const void*)(c->ext_xin ? (char*)c->ext_xin : c->ws + c->L[0].xin + c->in_geom(0).base_off(sz))
                                                   : c->ext_join;
        }
        if (y.pool && c->bound_training && y.ysel) b.ysel = c->ws + y.ysel;
        if (fin_fused) HIPCHK(launch_bn_fin_act(c->dtype, b, f, s));
        else HIPCHK(launch_bn_act(c->dtype, b, s));
    }
    if (c->tail == Y2_TAIL_AVGPOOL) {
        const Layer& y = c->L.back();
        HIPCHK(launch_avgpool_fwd((const float*)(c->ws + c->o_h32), out, c->N, y.Ho, y.Wo, y.cout, c->tail_k, s));
    }
    c->fwd_saved = true;
    c->moving_pending = !update_moving;
    return Y2_OK;
}

// The reference updates the moving statistics through UPDATE_OPS attached to train_op
// (src/pascal/pascal_train_darknet.py:49-51): a forward that only evaluates the loss leaves them alone.
// y2_forward(update_moving = 0) keeps the batch mean / variance of its training-mode layers; this applies
// the momentum-0.99 update from them (once), for callers that decide to train after the forward.
int y2_update_moving_stats(y2_ctx* c, void* stream) {
    if (!c->ws) return fail(Y2_ERR_STATE, "bind buffers first");
    if (!c->fwd_saved) return fail(Y2_ERR_STATE, "run y2_forward first");
    if (!c->moving_pending) return Y2_OK;
    hipStream_t s = (hipStream_t)stream;
    for (size_t l = 0; l < c->L.size(); ++l) {
        if (!c->fwd_training[l]) continue;
        const Layer& y = c->L[l];
        const float* stat = (const float*)(c->ws + y.stat);
        HIPCHK(launch_bn_update_moving(stat + 2 * y.ldy, stat + 6 * y.ldy, c->state + y.smm, c->state + y.smv, y.cout,
                                       c->bn_momentum, s));
    }
    c->moving_pending = false;
    return Y2_OK;
}

static int fused_opt_part(y2_ctx* c, int part, hipStream_t s);

int y2_backward(y2_ctx* c, const float* dout, int layer_lo, int layer_hi, void* stream) {
    if (!c->ws || !c->bound_training) return fail(Y2_ERR_STATE, "bind with training=1 first");
    if (!c->fwd_saved) return fail(Y2_ERR_STATE, "run y2_forward before y2_backward");
    const int nl = (int)c->L.size();
    if (layer_lo < 0 || layer_hi > nl || layer_lo >= layer_hi) return fail(Y2_ERR_ARG, "bad layer range");
    hipStream_t s = (hipStream_t)stream;
    const size_t sz = c->sz();
    const float inv_gs = 1.0f / c->grad_scale;
    char* dA[2] = {c->ws + c->o_dA0, c->ws + c->o_dA1};
    const bool ext_top = layer_hi == nl && c->ext_dout != nullptr;
    if (layer_hi == nl) c->dA_half = 0;        // the top layer's dA is the fp32 output gradient (times grad_scale)
    if (ext_top) {
        c->dA_cur = 0;
        if (layer_lo > 0 && c->grads) HIPCHK(hipMemsetAsync(c->grads, 0, c->L[layer_lo].pW * sizeof(float), s));
    } else if (layer_hi == nl) {
        if (!dout) return fail(Y2_ERR_ARG, "null output gradient");
        const Layer& y = c->L.back();
        const float* src = dout;
        if (c->tail == Y2_TAIL_AVGPOOL) {
            float* dh = (float*)(c->ws + c->o_dh32);
            HIPCHK(launch_avgpool_bwd(dout, dh, c->N, y.Ho, y.Wo, y.cout, c->tail_k, s));
            src = dh;
        }
        c->dA_cur = 0;
        PROF(CAT_MISC);
        HIPCHK(launch_convert_grad(c->dtype, src, dA[0], y.M, y.cout, y.ldy, c->grad_scale, s));
        // every gradient element is written with a plain store each step (split-K partials go through the slab
        // and a fixed-order sum: wgrad.hip); the kernels that still add with atomics zero their own target
        // A pass that starts at the top but stops above layer 0 leaves the gradients of [0, layer_lo) unwritten: zero
        // them, so that an optimizer step over the whole flat buffer does not re-apply the previous step's values
        // (the per-step zero-fill of the whole buffer went away with the split-K slab).  Sliced passes
        // continue downwards with layer_hi < nl and overwrite their own ranges.
        if (layer_lo > 0 && c->grads) HIPCHK(hipMemsetAsync(c->grads, 0, c->L[layer_lo].pW * sizeof(float), s));
    }
    float* psum = (float*)(c->ws + c->o_psum);
    bool forked = false;
    bool opt_early = false;   // fused optimizer: the layers above the first one were updated on the side stream
    int fused_P = 0;          // > 0: the dgrad of the layer above already reduced this layer's BN-backward sums
    static const bool no_fuse = getenv("Y2_NO_BNBWD_FUSE") != nullptr;
    if (c->overlap_wgrad && !c->side) {
        static const bool off = getenv("Y2_NO_WGRAD_OVERLAP") != nullptr;
        if (off) c->overlap_wgrad = 0;
        else {
            // Y2_SIDE_PRIORITY=low|high: the weight-gradient stream at the device's least / greatest queue priority (A/B:
            // the dgrad -> BN-backward chain on the caller's stream is the critical path of the backward pass)
            static const char* prio = getenv("Y2_SIDE_PRIORITY");
            int least = 0, greatest = 0;
            // Y2_SIDE_CUS=<n>: the weight-gradient stream may use only the first n compute units of the mask (A/B: the
            // small kernels of the dgrad -> BN-backward chain wait for wave slots behind its long-running workgroups).
            // Such a stream synchronises with the NULL stream: the caller's stream must be another one.
            static const int side_cus = getenv("Y2_SIDE_CUS") ? atoi(getenv("Y2_SIDE_CUS")) : 0;
            if (side_cus > 0) {
                uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int i = 0; i < side_cus && i < 256; ++i) mask[i >> 5] |= 1u << (i & 31);
                HIPCHK(hipExtStreamCreateWithCUMask(&c->side, 8, mask));
            } else if (prio && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest) {
                HIPCHK(hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, prio[0] == 'l' ? least : greatest));
            } else {
                (void)hipGetLastError();
                HIPCHK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
            }
            HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
        }
    }
    for (int l = layer_hi - 1; l >= layer_lo; --l) {
        c->prof_layer = l;
        const Layer& y = c->L[l];
        float* stat = (float*)(c->ws + y.stat);
        char* dyp = c->ws + y.dyp + c->dy_geom(l).base_off(sz);
        BnBwdArgs b{};
        b.dA = (ext_top && l == nl - 1) ? c->ext_dout : (const void*)dA[c->dA_cur]; b.y = c->ws + y.y;
        b.scale = stat; b.shift = stat + y.ldy; b.mean = stat + 2 * y.ldy; b.invstd = stat + 3 * y.ldy;
        b.coef = stat + 4 * y.ldy;
        b.psum = psum;
        b.dgamma = c->grads + y.pg; b.dbeta = c->grads + y.pbeta; b.dbias = c->grads + y.pb;
        b.dyp = dyp;
        b.N = c->N; b.H = y.H; b.W = y.W; b.C = y.cout; b.ldy = y.ldy;
        b.ldd = y.ldy;
        b.pool = y.pool; b.training = c->fwd_training[l]; b.inv_grad_scale = inv_gs;
        b.slope = y.slope;
        // f16x2f: dgrad and weight gradient read the hi plane of this dY alone (the 3-channel layer's own dy goes to fp32 kernels)
        b.hi_only = (c->bwd_dtype == 4 && !y.first3) ? 1 : 0;
        b.dA_half = c->dA_half;
        if (c->zero_bias_grad) b.dbias = nullptr;      // stays zero from y2_bind: the bias is not a variable of this graph
        const bool fused1 = y.first3 && conv1_wgrad_fused_ok(y.H, y.W, y.pool, y.ldy, (int)c->sz());
        const bool rec1 = y.first3 && (size_t)l + 1 < c->L.size() && y.ldy == 32 &&
                          conv1_pool_ok(y.H, y.W, y.pool, y.cout);
        const bool lin1 = y.first3 && c->lin1();
        // f16x2: the 3-channel layer's own dy (un-pooled / odd-sized fallbacks) is consumed by fp32 kernels
        const int bn_dtype = y.first3 ? dtype_plain(c->dtype) : c->dtype;
        if (lin1 && c->fopt.on && forked && l == 0) {
            // Every gradient above this layer is complete once the side stream has passed the dgrad that was just
            // queued: check and update those layers there, beside this layer's (compute-bound) gradient kernel.
            HIPCHK(hipEventRecord(c->ev_fork, s));
            HIPCHK(hipStreamWaitEvent(c->side, c->ev_fork, 0));
            const int rc = fused_opt_part(c, 1, c->side);
            if (rc != Y2_OK) return rc;
            opt_early = true;
        }
        {
            PROF(CAT_BN_BWD);
            if (lin1) {
                // linear form: the reduce pass rides in the weight-gradient kernel, which does not need its result
                Conv1WgradLinArgs g{};
                g.x4 = c->ws + y.xin + c->in_geom(l).base_off(sz); g.dA = b.dA;
                if (c->nosel1()) {
                    g.idx3 = (const unsigned*)(c->ws + y.idx0); g.Wf = c->params + y.pW; g.bias = c->params + y.pb;
                } else {
                    g.ysel = c->ws + y.ysel;
                    g.idx = (const unsigned short*)(c->ws + y.idx0);
                }
                g.scale = b.scale; g.shift = b.shift; g.acc = (float*)(c->ws + c->o_lin); g.psum = psum;
                if (c->gram_valid) g.gram = (const float*)(c->ws + c->o_gram);
                int nbl = 0;
                g.nblocks_out = &nbl;
                g.N = c->N; g.H = y.H; g.W = y.W;
                // f16x2f: this layer's backward contractions on the hi planes too (conv1_wgrad.hip XS == 2); Y2_F16X2F_CONV1_XS3=1: A/B
                static const bool xs3 = getenv("Y2_F16X2F_CONV1_XS3") != nullptr;
                g.xs = c->xs1() ? ((c->bwd_dtype == 4 && !xs3) ? 2 : 1) : 0;
                HIPCHK(launch_conv1_wgrad_lin(c->dtype, g, s));
                b.P = nbl;
            } else if (rec1) {   // pooled first layer: recompute the conv output instead of reading it (80 -> 24 B/pixel)
                Conv1BnBwdArgs q{};
                q.x4 = c->ws + y.xin + c->in_geom(l).base_off(sz); q.w = c->ws + y.wf; q.bias = c->params + y.pb;
                q.scale = b.scale; q.shift = b.shift; q.dA = b.dA; q.psum = psum;
                q.N = c->N; q.H = y.H; q.W = y.W;
                const int tiles = c->N * (y.H / 2) * ((y.W + 31) / 32);
                q.nblocks = (tiles + 3) / 4 > 2048 ? 2048 : (tiles + 3) / 4;
                b.P = q.nblocks;
                HIPCHK(launch_conv1_bnbwd_reduce(c->dtype, q, s));
            } else if (fused_P > 0) {
                b.P = fused_P;
            } else {
                HIPCHK(launch_bn_bwd_reduce(bn_dtype, b, s));
            }
            fused_P = 0;
            // short partial lists: the finalize rides in the apply pass (bn.hip bn_bwd_fin_apply_kernel)
            static const bool no_fin_fuse = getenv("Y2_NO_BN_FIN_FUSE") != nullptr;
            if (!fused1 && !lin1 && !no_fin_fuse && bn_bwd_fin_apply_ok(b)) {
                HIPCHK(launch_bn_bwd_fin_apply(bn_dtype, b, s));
            } else {
                HIPCHK(launch_bn_bwd_finalize(b, s));
                if (!fused1 && !lin1) HIPCHK(launch_bn_bwd_apply(bn_dtype, b, s));
            }
        }
        char* xin = (l == 0 && c->ext_xin) ? (char*)c->ext_xin : c->ws + y.xin + c->in_geom(l).base_off(sz);
        if (lin1) {
            // no conv output of this layer exists: dW = scale X(dz) - ka X(1) - kb (G W + b X(1))  (conv1_wgrad.hip)
            Conv1DwFinalizeArgs f{};
            f.acc = (float*)(c->ws + c->o_lin); f.W = c->params + y.pW; f.bias = c->params + y.pb; f.scale = b.scale;
            f.coef = b.coef; f.dW = c->grads + y.pW; f.inv_grad_scale = inv_gs;
            if (c->gram_valid) f.gram = (const float*)(c->ws + c->o_gram);
            PROF(CAT_CONV1_WGRAD);
            HIPCHK(launch_conv1_dw_finalize(f, s));
        } else if (fused1) {
            // the first layer's dy has one consumer: apply pass and weight gradient in one kernel
            Conv1WgradFusedArgs g{};
            g.x4 = xin; g.y = b.y; g.dA = b.dA;
            g.scale = b.scale; g.shift = b.shift; g.coef = b.coef;
            g.dW = c->grads + y.pW;
            g.N = c->N; g.H = y.H; g.W = y.W; g.inv_grad_scale = inv_gs;
            HIPCHK(hipMemsetAsync(g.dW, 0, (size_t)27 * y.cout * sizeof(float), s));     // atomics
            { PROF(CAT_CONV1_WGRAD); HIPCHK(launch_conv1_wgrad_fused(c->dtype, g, s)); }
        } else if (y.first3) {
            Conv1WgradArgs g{};
            g.x4 = xin; g.dy = dyp; g.dW = c->grads + y.pW;
            g.N = c->N; g.H = y.H; g.W = y.W; g.M = y.M; g.scale = inv_gs;
            HIPCHK(hipMemsetAsync(g.dW, 0, (size_t)27 * y.cout * sizeof(float), s));     // atomics
            { PROF(CAT_CONV1_WGRAD); HIPCHK(launch_conv1_wgrad(c->dtype, g, s)); }
        } else {
            WgradArgs g{};
            g.x = xin; g.dy = dyp; g.dW = c->grads + y.pW;
            g.N = c->N; g.H = y.H; g.W = y.W; g.M = y.M;
            g.Cin = y.cin_s; g.Cdy = y.ldy; g.Cout = y.cout; g.taps = y.k * y.k; g.splitk = 0; g.scale = inv_gs;
            g.slab = (float*)(c->ws + c->o_slab); g.slab_floats = c->slab_floats;
            g.tile_cnt = (int*)(c->ws + c->o_wgcnt); g.cnt_ints = kWgCntInts;
            hipStream_t ws_ = s;
            if (c->overlap_wgrad && c->prof != 1) {
                // fork: the filter gradient only reads x and dY; it fills the bubbles of the dgrad beside it.  The
                // lowest layer forks too although nothing runs beside it: every weight gradient shares ONE split-K slab,
                // so they must all queue on one stream (on the caller's stream it raced the side stream's sum kernel)
                HIPCHK(hipEventRecord(c->ev_fork, s));
                HIPCHK(hipStreamWaitEvent(c->side, c->ev_fork, 0));
                ws_ = c->side;
                forked = true;
            }
            { ProfScope _p(c, ws_, CAT_WGRAD); HIPCHK(launch_wgrad_auto(c->bwd_dtype, g, ws_)); }
            if (l > 0 || c->dinput || c->ext_dx) {
                ConvArgs a{};
                a.x = dyp; a.w = c->ws + y.wd; a.y = (l == 0 && c->ext_dx) ? (char*)c->ext_dx : dA[c->dA_cur ^ 1];
                a.N = c->N; a.H = y.H; a.W = y.W; a.C = y.ldy; a.M = y.M; a.Cout = y.cin; a.ldy = y.cin;
                a.taps = y.k * y.k;
                a.is_dgrad = 1;
                if (c->ks_floats) { a.ks_scratch = (float*)(c->ws + c->o_ks); a.ks_floats = c->ks_floats; }
                int bp = 0;
                const Layer& z = c->L[l > 0 ? l - 1 : 0];
                // the BN-backward reduce of the layer below rides in this dgrad's epilogue (it needs that layer's
                // conv output, scale and shift beside the dA tile the epilogue holds anyway); the first layer
                // keeps its own recomputing reduce
                // (a launch of a few hundred pixels splits its K range over workgroups instead -- conv_haloq.hip haloq_ks --
                //  and leaves the reduce to the standalone kernel: 7x7 1024 -> 512 at batch 24: 81 us fused and un-split)
                const bool ks = c->ks_floats && conv_ks_depth(a.taps, a.M, a.Cout, a.C * dtype_kbytes(c->dtype)) >= 2;
                const bool fuse = !no_fuse && !ks && l > 0 && l - 1 >= layer_lo && !z.first3 && z.ldy == y.cin;
                if (l == 1 && z.first3 && c->fopt.on && c->fopt.ctrl && c->lin1() && forked)
                    a.nonfinite = (unsigned*)(c->ws + c->o_nfflag);   // this launch stores dA_0: the early guard's view of layer 0
                if (fuse) {
                    float* zs = (float*)(c->ws + z.stat);
                    a.bw_y = c->ws + (z.pool ? z.ysel : z.y);   // same pixel grid as this launch's output either way
                    a.bw_scale = zs; a.bw_shift = zs + z.ldy; a.bw_psum = psum;
                    a.bw_slope = z.slope;
                }
                int rec = 0;
                // f16x2f: dA is consumed once, by the batch-norm backward pass of the layer below, which rounds its own result
                // to f16 for the next contraction: store it in f16 (launch dtype 5) wherever that consumer is one of the
                // fp32-wide batch-norm kernels (not the 3-channel layer's own kernels, not an external input gradient)
                static const bool da32 = getenv("Y2_F16X2F_DA32") != nullptr;      // A/B switch: fp32 dA everywhere
                const bool half_out = c->bwd_dtype == 4 && !da32 && l > 0 && !z.first3;
                { PROF(CAT_DGRAD); HIPCHK(launch_conv(half_out ? 5 : c->bwd_dtype, a, s, &bp, &rec)); }
                if (fuse) fused_P = rec;
                c->dA_cur ^= 1;
                c->dA_half = half_out ? 1 : 0;
                if (l == 0 && c->dinput)   // the stack's input gradient leaves in fp32 NHWC, loss scale divided out
                    HIPCHK(launch_cast_to_f32(c->dtype, c->ext_dx ? (const void*)c->ext_dx : (const void*)dA[c->dA_cur], c->dinput,
                                              (size_t)y.M, y.cin, y.cin, s, inv_gs));
            }
        }
        for (int k = 0; k < c->n_marks; ++k)
            if (c->cur_marks[k] == l) {
                HIPCHK(hipEventRecord(c->mark_main[k], s));
                HIPCHK(hipEventRecord(c->mark_side[k], forked ? c->side : s));
            }
    }
    if (forked) {   // join: every gradient is complete when the caller's stream gets past this call
        HIPCHK(hipEventRecord(c->ev_join, c->side));
        HIPCHK(hipStreamWaitEvent(s, c->ev_join, 0));
    }
    if (c->fopt.on) return fused_opt_part(c, opt_early ? 2 : 0, s);
    return Y2_OK;
}

// One backward pass over all layers that records, for each mark k, an event pair when every layer
// >= mark_layers[k] is complete (main-stream chain and side-stream weight gradients).  A consumer stream
// (the gradient all-reduce of that slice) waits for the pair with y2_wait_mark -- no join on the caller's
// stream until the end of the pass, unlike one y2_backward call per slice.
int y2_backward_marks(y2_ctx* c, const float* dout, int n_marks, const int* mark_layers, void* stream) {
    if (n_marks < 0 || (n_marks > 0 && !mark_layers)) return fail(Y2_ERR_ARG, "bad marks");
    for (int k = 0; k < n_marks; ++k)   // an unrecorded event would let y2_wait_mark return at once
        if (mark_layers[k] < 0 || mark_layers[k] >= (int)c->L.size())
            return fail(Y2_ERR_ARG, "mark %d: layer %d out of range", k, mark_layers[k]);
    while ((int)c->mark_main.size() < n_marks) {
        hipEvent_t a = nullptr, b = nullptr;
        HIPCHK(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        c->mark_main.push_back(a);
        c->mark_side.push_back(b);
    }
    c->n_marks = n_marks;
    c->cur_marks = mark_layers;
    const int rc = y2_backward(c, dout, 0, (int)c->L.size(), stream);
    c->n_marks = 0;
    c->cur_marks = nullptr;
    return rc;
}
// Full backward pass that also returns the gradient with respect to the stack's INPUT (fp32 NHWC [N,H,W,cin]):
// what a composed graph needs of a stack that is not fed by a placeholder (the YOLOv2 detector's 13x13 and head
// stacks).  Not available for the 3-channel first layer (the reference never differentiates the image either).
int y2_backward_input(y2_ctx* c, const float* dout, float* dinput, void* stream) {
    if (!dinput) return fail(Y2_ERR_ARG, "null input gradient");
    if (c->L.empty() || c->L[0].first3) return fail(Y2_ERR_ARG, "the 3-channel image layer has no input gradient");
    c->dinput = dinput;
    const int rc = y2_backward(c, dout, 0, (int)c->L.size(), stream);
    c->dinput = nullptr;
    return rc;
}
// ---------------------------------------------------------------------------
// Linked stacks (round 5): the bottleneck units of the ResNet swap (src/slim_dir/nets/resnet_v1.py:99-112) hand their
// activations and gradients to each other in the arithmetic type -- bordered tensors forward, [M][C] tensors backward --
// instead of through fp32 NHWC tensors (cast / pack / convert passes on both sides of every join).
// ---------------------------------------------------------------------------
size_t y2_bordered_bytes(int N, int H, int W, int C, int dtype, size_t* cell0_offset) {
    const PadGeom g{N, H, W, C};
    if (cell0_offset) *cell0_offset = g.base_off(dtype_size(dtype));
    return g.bytes(dtype_size(dtype));
}
int y2_link(y2_ctx* c, void* x_bordered, void* out_bordered, const void* join_bordered, int join_self, const void* dout_t,
            void* dx_t) {
    if (!c || c->L.empty()) return fail(Y2_ERR_ARG, "y2_link: null or empty context");
    if (dtype_split(c->dtype) && (x_bordered || out_bordered || join_bordered || join_self || dout_t || dx_t))
        return fail(Y2_ERR_ARG, "y2_link: not built for the split-operand mode");
    const Layer& first = c->L.front();
    const Layer& last = c->L.back();
    if ((x_bordered || dx_t) && first.first3) return fail(Y2_ERR_ARG, "y2_link: the 3-channel image layer takes fp32 / uint8 images");
    // (ldy = out_chl rounded up to 32: the linked tensors have no padding channels)
    if ((out_bordered || join_bordered || join_self || dout_t) && (c->tail == Y2_TAIL_AVGPOOL || last.cout != last.ldy))
        return fail(Y2_ERR_ARG, "y2_link: the linked output / its gradient need out_chl in multiples of 32 and no average-pool tail");
    if ((x_bordered || dx_t) && first.cin_s != first.cin)
        return fail(Y2_ERR_ARG, "y2_link: the linked input / its gradient need in_chl equal to the tensor's channel stride");
    if (join_self && (first.cin != last.cout || first.H != last.Ho || first.W != last.Wo))
        return fail(Y2_ERR_ARG, "y2_link: an identity shortcut needs input and output of one shape");
    if (join_self && join_bordered) return fail(Y2_ERR_ARG, "y2_link: one join");
    c->ext_xin = x_bordered; c->ext_out = out_bordered; c->ext_join = join_bordered; c->ext_join_self = join_self ? 1 : 0;
    c->ext_dout = dout_t; c->ext_dx = dx_t;
    return Y2_OK;
}

int y2_wait_mark(y2_ctx* c, int k, void* stream) {
    if (k < 0 || k >= (int)c->mark_main.size()) return fail(Y2_ERR_ARG, "no such mark");
    HIPCHK(hipStreamWaitEvent((hipStream_t)stream, c->mark_main[k], 0));
    HIPCHK(hipStreamWaitEvent((hipStream_t)stream, c->mark_side[k], 0));
    return Y2_OK;
}

// what: 0 = bordered input of `layer` (post BN+leaky+pool of the previous layer), [N,H,W,cin]
//       1 = conv output (+bias) of `layer`, [N,H,W,cout]
//       2 = dy (gradient wrt the conv output, times grad_scale), [N,H,W,cout]
//       3 = the layer's normalisation constants of the last forward, [4][cout]: mean, invstd, scale, shift
int y2_debug_read(y2_ctx* c, int l, int what, float* dst, void* stream) {
    if (l < 0 || l >= (int)c->L.size()) return fail(Y2_ERR_ARG, "layer out of range");
    const Layer& y = c->L[l];
    hipStream_t s = (hipStream_t)stream;
    const size_t sz = c->sz();
    if (what == 0) {
        const int C = y.first3 ? 3 : y.cin;
        HIPCHK(launch_unpack_act(y.first3 ? dtype_plain(c->dtype) : c->dtype, c->ws + y.xin + c->in_geom(l).base_off(sz), dst, c->N, y.H, y.W, C,
                                 y.cin_s, s));
    } else if (what == 1) {
        if (c->fwd_folded[l])
            return fail(Y2_ERR_STATE, "layer %d: inference batch norm was folded into the convolution, its conv output "
                                      "is not stored (Y2_NO_INFER_FOLD=1 keeps the two-pass form)", l);
        if (y.first3 && !c->bound_training && (size_t)l + 1 < c->L.size() && y.ldy == 32 &&
            conv1_pool_ok(y.H, y.W, y.pool, y.cout))
            return fail(Y2_ERR_STATE, "inference binding: the pooled first layer does not store its conv output");
        if (y.first3 && c->lin1()) {   // the linear form never stores this layer's conv output: recompute it (tests)
            Conv1Args a{};
            a.x4 = c->ws + y.xin + c->in_geom(l).base_off(sz); a.w = c->ws + y.wf; a.y = c->ws + y.y;
            a.bias = c->params + y.pb;
            a.part_cnt = (float*)(c->ws + c->o_part_cnt); a.part_mean = (float*)(c->ws + c->o_part_mean);
            a.part_m2 = (float*)(c->ws + c->o_part_m2);
            a.N = c->N; a.H = y.H; a.W = y.W; a.M = y.M;
            const int nb = (y.M + 127) / 128;
            a.nblocks = nb > 2048 ? 2048 : nb;
            a.stats_only = 0;
            HIPCHK(launch_conv1_fwd(c->dtype, a, s));
        }
        HIPCHK(launch_cast_to_f32(c->dtype, c->ws + y.y, dst, (size_t)y.M, y.cout, y.ldy, s));
    } else if (what == 2) {
        if (!c->bound_training) return fail(Y2_ERR_STATE, "no gradients in inference binding");
        if (y.first3 && (c->lin1() || conv1_wgrad_fused_ok(y.H, y.W, y.pool, y.ldy, (int)c->sz())))
            return fail(Y2_ERR_STATE, "the first layer's dy is fused into its weight gradient and never stored");
        HIPCHK(launch_unpack_act(y.first3 ? dtype_plain(c->dtype) : c->dtype, c->ws + y.dyp + c->dy_geom(l).base_off(sz), dst, c->N, y.H, y.W, y.cout,
                                 y.ldy, s));
    } else if (what == 3) {
        // the per-channel constants the last forward normalised this layer with: dst [4][cout] = mean, 1 / sqrt(var + eps),
        // scale = gamma * invstd, shift = beta - mean * scale (batch statistics of a training-mode layer, else the moving
        // ones) -- tests hand them to the oracle as INPUTS where the device's moments are formed differently from the
        // oracle's (the Gram-matrix statistics of the 3-channel layer) instead of teaching the oracle that form
        if (!c->fwd_saved) return fail(Y2_ERR_STATE, "run y2_forward first");
        const float* stat = (const float*)(c->ws + y.stat);
        const float* src[4] = {stat + 2 * y.ldy, stat + 3 * y.ldy, stat, stat + y.ldy};
        for (int k = 0; k < 4; ++k)
            HIPCHK(hipMemcpyAsync(dst + (size_t)k * y.cout, src[k], (size_t)y.cout * sizeof(float), hipMemcpyDeviceToDevice, s));
    } else {
        return fail(Y2_ERR_ARG, "unknown selector");
    }
    return Y2_OK;
}

// ---------------------------------------------------------------------------
size_t y2_yolo_loss_workspace_bytes(int batch, int S) { return (size_t)loss_blocks(batch, S) * 4 * sizeof(float) + 256; }

int y2_yolo_loss(const float* net, const float* labels, int num_class, int batch, float image_size, int S, int B,
                 float lambda_coord, float lambda_noobj, float* loss, float* ious, float* object_mask, float* dnet,
                 void* workspace, void* stream) {
    if (!net || !labels || !loss || !ious || !object_mask || !workspace) return fail(Y2_ERR_ARG, "null tensor");
    if (B < 1 || B > 8 || num_class < 1 || S < 1 || batch < 1) return fail(Y2_ERR_ARG, "bad loss geometry");
    LossArgs a{};
    a.net = net; a.labels = labels; a.loss = loss; a.ious = ious; a.mask = object_mask; a.dnet = dnet;
    a.partial = (float*)workspace;
    a.N = batch; a.S = S; a.B = B; a.C = num_class; a.image_size = image_size;
    a.lambda_coord = lambda_coord; a.lambda_noobj = lambda_noobj;
    HIPCHK(launch_yolo_loss(a, (hipStream_t)stream));
    return Y2_OK;
}
int y2_get_iou(const float* b1, const float* b2, float* iou, int n, void* stream) {
    if (!b1 || !b2 || !iou || n < 0) return fail(Y2_ERR_ARG, "bad arguments");
    if (n == 0) return Y2_OK;
    HIPCHK(launch_get_iou(b1, b2, iou, n, (hipStream_t)stream));
    return Y2_OK;
}
int y2_decode_detections(const float* predict, int S, int B, int num_class, int im_w, int im_h, float thresh,
                         int* det, float* conf, void* stream) {
    if (!predict || !det || !conf) return fail(Y2_ERR_ARG, "null tensor");
    HIPCHK(launch_decode(predict, S, B, num_class, im_w, im_h, thresh, det, conf, (hipStream_t)stream));
    return Y2_OK;
}
int y2_softmax_cross_entropy(const float* logits, const int* labels, int batch, int classes, float* loss,
                             float* dlogits, void* stream) {
    if (!logits || !labels || !loss) return fail(Y2_ERR_ARG, "null tensor");
    HIPCHK(launch_softmax_ce(logits, labels, loss, dlogits, batch, classes, (hipStream_t)stream));
    return Y2_OK;
}

int y2_accuracy(const float* logits, const int* labels, int batch, int classes, float* accuracy, void* stream) {
    if (!logits || !labels || !accuracy || batch < 1 || classes < 1) return fail(Y2_ERR_ARG, "bad arguments");
    HIPCHK(launch_accuracy(logits, labels, accuracy, batch, classes, (hipStream_t)stream));
    return Y2_OK;
}

int y2_adam_step(float* params, float* m, float* v, const float* grads, size_t n, int step, float lr, float beta1,
                 float beta2, float eps, float grad_mult, void* stream) {
    if (!params || !m || !v || !grads || step < 1) return fail(Y2_ERR_ARG, "bad arguments");
    // TF: lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, step)) / (1.0 - pow((double)beta1, step));
    HIPCHK(launch_adam(params, m, v, grads, n, (float)lr_t, beta1, beta2, eps, grad_mult, (hipStream_t)stream));
    return Y2_OK;
}
// Loss-scale-safe forms of the two optimizers: `ctrl` = 8 caller-owned, zero-initialised 32-bit words on the
// device {found_inf, step, skipped, -, lr_t, -, -, -}.  First one of the two scans sets ctrl.found_inf
// (y2_grad_check: the context's sentinel ranges, ~30 k floats; y2_grad_check_full: every element of any buffer),
// then the guarded step: a gradient buffer flagged non-finite leaves params, slots and the step counter
// untouched (ctrl.skipped += 1).
int y2_grad_check(y2_ctx* c, void* ctrl, void* stream) {
    if (!c->grads || !ctrl) return fail(Y2_ERR_ARG, "bind with a gradient buffer first");
    HIPCHK(launch_grad_check_ranges(c->grads, c->ws + c->o_chkranges, c->n_chkranges, ctrl, (hipStream_t)stream));
    return Y2_OK;
}
int y2_grad_check_more(y2_ctx* c, void* ctrl, void* stream) {
    if (!c->grads || !ctrl) return fail(Y2_ERR_ARG, "bind with a gradient buffer first");
    HIPCHK(launch_grad_check_ranges(c->grads, c->ws + c->o_chkranges, c->n_chkranges, ctrl, (hipStream_t)stream, nullptr,
                                    true));
    return Y2_OK;
}
int y2_grad_check_full(const float* grads, size_t n, void* ctrl, void* stream) {
    if (!grads || !ctrl) return fail(Y2_ERR_ARG, "bad arguments");
    HIPCHK(launch_grad_check(grads, n, ctrl, (hipStream_t)stream));
    return Y2_OK;
}
int y2_range_check(const float* x, size_t n, float limit, void* ctrl, void* stream) {
    if (!x || !ctrl || !(limit > 0.f)) return fail(Y2_ERR_ARG, "bad arguments");
    if (n) HIPCHK(launch_range_check(x, n, limit, ctrl, (hipStream_t)stream));
    return Y2_OK;
}
int y2_adam_step_guarded(float* params, float* m, float* v, const float* grads, size_t n, void* ctrl, float lr,
                         float beta1, float beta2, float eps, float grad_mult, void* stream) {
    if (!params || !m || !v || !grads || !ctrl) return fail(Y2_ERR_ARG, "bad arguments");
    HIPCHK(launch_opt_ctrl_advance(ctrl, lr, beta1, beta2, (hipStream_t)stream));
    HIPCHK(launch_adam_guarded(params, m, v, grads, n, ctrl, beta1, beta2, eps, grad_mult, (hipStream_t)stream));
    return Y2_OK;
}
int y2_momentum_step_guarded(float* params, float* accum, const float* grads, size_t n, void* ctrl, float lr,
                             float momentum, float grad_mult, void* stream) {
    if (!params || !accum || !grads || !ctrl) return fail(Y2_ERR_ARG, "bad arguments");
    HIPCHK(launch_opt_ctrl_advance(ctrl, lr, 0.9f, 0.999f, (hipStream_t)stream));
    HIPCHK(launch_momentum_guarded(params, accum, grads, n, ctrl, lr, momentum, grad_mult, (hipStream_t)stream));
    return Y2_OK;
}
// Optimizer step of the context's bound params / grads fused with the filter re-pack (pack.hip opt_pack_kernel):
// every filter is read and written by the update anyway, so its MFMA-operand copies (forward and dgrad layouts)
// leave in the same pass -- the separate re-pack of the next y2_forward (193 MB read again) is not needed.
// ctrl (nullable): the overflow guard of y2_adam_step_guarded (run y2_grad_check first); NULL: plain step `step`.
// part: 0 = every parameter; 1 = the filter tiles and the small ranges of the layers above the first one;
// 2 = the rest (the first layer's b / gamma / beta and a 3-channel first filter) -- 1 then 2 is 0.
static int opt_step_packed(y2_ctx* c, int kind, float* slot0, float* slot1, void* ctrl, float lr_t_or_lr, float b1,
                           float b2, float eps, float grad_mult, hipStream_t s, int part = 0) {
    if (!c->ws || !c->grads || !c->bound_training) return fail(Y2_ERR_STATE, "bind with training=1 first");
    OptPackArgs a{};
    a.p = c->params; a.slot0 = slot0; a.slot1 = slot1; a.g = c->grads; a.ctrl = ctrl;
    a.lr_t = lr_t_or_lr; a.b1 = b1; a.b2 = b2; a.eps = eps; a.gmult = grad_mult; a.kind = kind;
    a.tab = (const PackLayer*)(c->ws + c->o_packtab); a.nlayers = (int)c->packtab.size();
    a.tile_blocks = part == 2 ? 0 : c->opt_tile_blocks;
    const unsigned* small = (const unsigned*)(c->ws + c->o_smallranges);
    a.small = part == 2 ? small + 2 * c->n_small_upper : small;
    a.nsmall = part == 0 ? c->n_smallranges : (part == 1 ? c->n_small_upper : c->n_smallranges - c->n_small_upper);
    HIPCHK(launch_opt_pack(c->dtype, a, s));
    if (part != 1 && !c->L.empty() && c->L[0].first3)
        HIPCHK(launch_pack_conv1_weights(c->dtype, c->params + c->L[0].pW, c->ws + c->L[0].wf, s));
    if (part != 1) c->weights_dirty = false;
    return Y2_OK;
}
// The optimizer step of y2_backward_adam / y2_backward_momentum, whole (part 0) or in its two parts.  With a guard,
// parts 0 and 1 decide the step: part 1 from the sentinel ranges of the layers above the first one and the
// non-finite marker of the dgrad that produced dA_0 (optim.hip: every overflow above reaches one of them; the first
// layer's own sums are fp32 functions of that finite dA_0), so part 2 only follows the decision.
static int fused_opt_part(y2_ctx* c, int part, hipStream_t s) {
    const y2_ctx::FusedOpt& f = c->fopt;
    double lr_t = f.lr;
    if (f.ctrl) {
        if (part != 2) {
            const bool upper = part == 1;
            HIPCHK(launch_grad_check_ranges(c->grads, c->ws + c->o_chkranges, upper ? c->n_chk_upper : c->n_chkranges,
                                            f.ctrl, s, upper ? (unsigned*)(c->ws + c->o_nfflag) : nullptr));
            HIPCHK(launch_opt_ctrl_advance(f.ctrl, f.lr, f.kind == 0 ? f.b1 : 0.9f, f.kind == 0 ? f.b2 : 0.999f, s));
        }
    } else if (f.kind == 0) {
        lr_t = (double)f.lr * sqrt(1.0 - pow((double)f.b2, f.step)) / (1.0 - pow((double)f.b1, f.step));
    }
    return opt_step_packed(c, f.kind, f.slot0, f.slot1, f.ctrl, (float)lr_t, f.b1, f.b2, f.eps, f.gmult, s, part);
}
// Backward pass + guarded optimizer step + filter re-pack as ONE call (the reference's train_op =
// optimizer.minimize(loss), src/pascal/pascal_train_darknet.py:49-51).  Same results as y2_backward followed by
// y2_grad_check and y2_adam_step_packed; the difference is the schedule: with the pooled 3-channel first layer the
// update of every layer above it runs on the weight-gradient stream while the first layer's gradient is computed.
int y2_backward_adam(y2_ctx* c, const float* dout, float* m, float* v, void* ctrl, int step, float lr, float beta1,
                     float beta2, float eps, float grad_mult, void* stream) {
    if (!m || !v || (!ctrl && step < 1)) return fail(Y2_ERR_ARG, "bad arguments");
    if (!c->grads) return fail(Y2_ERR_STATE, "bind with a gradient buffer first");
    c->fopt = y2_ctx::FusedOpt{true, 0, m, v, ctrl, lr, beta1, beta2, eps, grad_mult, step};
    const int rc = y2_backward(c, dout, 0, (int)c->L.size(), stream);
    c->fopt.on = false;
    return rc;
}
int y2_backward_momentum(y2_ctx* c, const float* dout, float* accum, void* ctrl, float lr, float momentum,
                         float grad_mult, void* stream) {
    if (!accum) return fail(Y2_ERR_ARG, "bad arguments");
    if (!c->grads) return fail(Y2_ERR_STATE, "bind with a gradient buffer first");
    c->fopt = y2_ctx::FusedOpt{true, 1, accum, nullptr, ctrl, lr, momentum, 0.f, 0.f, grad_mult, 0};
    const int rc = y2_backward(c, dout, 0, (int)c->L.size(), stream);
    c->fopt.on = false;
    return rc;
}
int y2_adam_step_packed(y2_ctx* c, float* m, float* v, void* ctrl, int step, float lr, float beta1, float beta2,
                        float eps, float grad_mult, void* stream) {
    if (!m || !v || (!ctrl && step < 1)) return fail(Y2_ERR_ARG, "bad arguments");
    double lr_t = lr;
    // ctrl with step < 0: the control block was advanced by another stack's call of this step (one composed graph,
    // one step counter): use its lr_t / found_inf as they stand
    if (ctrl && step >= 0) HIPCHK(launch_opt_ctrl_advance(ctrl, lr, beta1, beta2, (hipStream_t)stream));
    else if (!ctrl) lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, step)) / (1.0 - pow((double)beta1, step));
    return opt_step_packed(c, 0, m, v, ctrl, (float)lr_t, beta1, beta2, eps, grad_mult, (hipStream_t)stream);
}
int y2_momentum_step_packed(y2_ctx* c, float* accum, void* ctrl, float lr, float momentum, float grad_mult,
                            void* stream) {
    if (!accum) return fail(Y2_ERR_ARG, "bad arguments");
    if (ctrl) HIPCHK(launch_opt_ctrl_advance(ctrl, lr, 0.9f, 0.999f, (hipStream_t)stream));
    return opt_step_packed(c, 1, accum, nullptr, ctrl, lr, momentum, 0.f, 0.f, grad_mult, (hipStream_t)stream);
}
int y2_momentum_step(float* params, float* accum, const float* grads, size_t n, float lr, float momentum,
                     float grad_mult, void* stream) {
    if (!params || !accum || !grads) return fail(Y2_ERR_ARG, "bad arguments");
    HIPCHK(launch_momentum(params, accum, grads, n, lr, momentum, grad_mult, (hipStream_t)stream));
    return Y2_OK;
}

// ---------------------------------------------------------------------------
// single-op conv2d (+ backward) on fp32 NHWC / HWIO tensors
// ---------------------------------------------------------------------------
struct OpPlan {
    int Cin_p, Cdy, Cout_pad, Cin_pad, ldy;
    size_t xp, wf, wd, y, dyp, dx, dw, slab, ks, ks_floats, total;
};
// split-K partial tiles of the op-level weight gradient (WgradArgs::slab): at most ~1,000 workgroups x one 64 x 32 x 9
// (or 128 x 128) tile, as in the network plan -- the op-level gradients are then summed in a fixed order too
// (no float atomics: bit-reproducible run to run, like the network's)
constexpr size_t kOpSlabFloats1 = (size_t)1024 * 18432;
static size_t op_slab_floats(int dtype) { return kOpSlabFloats1 * (dtype_split(dtype) ? 3 : 1); }
static OpPlan op_plan(int N, int H, int W, int Cin, int Cout, int k, int dtype) {
    OpPlan p{};
    const size_t sz = dtype_size(dtype);
    p.Cin_p = Cin <= 32 ? 32 : (Cin <= 64 ? 64 : round_up(Cin, 128));
    p.ldy = round_up(Cout, 32);
    p.Cdy = p.ldy;
    p.Cout_pad = round_up(Cout, conv_block_couts(Cout));
    p.Cin_pad = round_up(p.Cin_p, conv_block_couts(p.Cin_p));
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = off; off += align_up(b, 256); return o; };
    p.xp = take(PadGeom{N, H, W, p.Cin_p}.bytes(sz));
    p.wf = take((size_t)p.Cout_pad * k * k * p.Cin_p * sz);
    p.wd = take((size_t)p.Cin_pad * k * k * p.Cdy * sz);
    p.y = take((size_t)N * H * W * p.ldy * sz + 256);
    p.dyp = take(PadGeom{N, H, W, p.Cdy}.bytes(sz));
    p.dx = take((size_t)N * H * W * p.Cin_p * sz + 256);
    p.dw = take((size_t)k * k * p.Cin_p * Cout * sizeof(float));
    p.slab = take(op_slab_floats(dtype) * sizeof(float));
    p.ks_floats = dtype_split(dtype) ? 0 : std::max(conv_ks_scratch_floats(k * k, N * H * W, p.ldy, p.Cin_p * (int)sz),
                                                     conv_ks_scratch_floats(k * k, N * H * W, p.Cin_p, p.Cdy * (int)sz));
    p.ks = take(p.ks_floats * sizeof(float));
    p.total = off;
    return p;
}
// fp32 NHWC -> the bordered tensor of the op-level entries.  Only the bordered activations rely on zeros (guards, borders,
// channel padding; the filter packs write their own padding): one pass writes the whole allocation where the 16-byte
// form applies, else memset + the scalar pack.
static hipError_t op_pack_bordered(int dtype, const float* x, char* region, size_t region_bytes, const PadGeom& g, int C,
                                   hipStream_t s) {
    hipError_t e = launch_pack_act_region(dtype, x, region, region_bytes, g.front_px(), g.N, g.H, g.W, C, g.C, s);
    if (e != hipErrorNotSupported) return e;
    (void)hipGetLastError();
    e = hipMemsetAsync(region, 0, region_bytes, s);
    if (e != hipSuccess) return e;
    return launch_pack_act(dtype, x, region + g.base_off(dtype_size(dtype)), g.N, g.H, g.W, C, g.C, s);
}
size_t y2_conv2d_workspace_bytes(int N, int H, int W, int Cin, int Cout, int k, int dtype) {
    return op_plan(N, H, W, Cin, Cout, k, dtype).total;
}
int y2_conv2d(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cin, int Cout,
              int k, int dtype, void* workspace, void* stream) {
    if (!x || !w || !y || !workspace) return fail(Y2_ERR_ARG, "null tensor");
    if (k != 1 && k != 3) return fail(Y2_ERR_ARG, "filter size must be 1 or 3");
    if (dtype < 0 || dtype > 4) return fail(Y2_ERR_ARG, "bad dtype");
    if (dtype == 4) dtype = 3;      // f16x2f: the forward pass is the split-operand mode's
    hipStream_t s = (hipStream_t)stream;
    const size_t sz = dtype_size(dtype);
    OpPlan p = op_plan(N, H, W, Cin, Cout, k, dtype);
    char* ws = (char*)workspace;
    PadGeom g{N, H, W, p.Cin_p};
    char* xp = ws + p.xp + g.base_off(sz);
    HIPCHK(op_pack_bordered(dtype, x, ws + p.xp, p.wf - p.xp, g, Cin, s));
    HIPCHK(launch_pack_weights(dtype, w, ws + p.wf, nullptr, k * k, Cin, Cout, p.Cout_pad, p.Cin_p, 0, 0,
                               conv_filter_layout(k * k, W, p.Cin_p * dtype_kbytes(dtype), Cout, N * H * W, 0, (int)sz, dtype_split(dtype)), s));
    ConvArgs a{};
    a.x = xp; a.w = ws + p.wf; a.y = ws + p.y; a.bias = bias;
    a.N = N; a.H = H; a.W = W; a.C = p.Cin_p; a.M = N * H * W; a.Cout = Cout; a.ldy = p.ldy; a.taps = k * k;
    if (p.ks_floats) { a.ks_scratch = (float*)(ws + p.ks); a.ks_floats = p.ks_floats; }
    HIPCHK(launch_conv(dtype, a, s));
    HIPCHK(launch_cast_to_f32(dtype, ws + p.y, y, (size_t)N * H * W, Cout, p.ldy, s));
    return Y2_OK;
}
int y2_conv2d_backward(const float* x, const float* w, const float* dy, float* dx, float* dw, int N, int H, int W,
                       int Cin, int Cout, int k, int dtype, void* workspace, void* stream) {
    if (!x || !w || !dy || !workspace) return fail(Y2_ERR_ARG, "null tensor");
    if (k != 1 && k != 3) return fail(Y2_ERR_ARG, "filter size must be 1 or 3");
    if (dtype < 0 || dtype > 4) return fail(Y2_ERR_ARG, "bad dtype");
    const int ldt = dtype;          // launch dtype of the two contractions (4: hi planes of the split tensors)
    if (dtype == 4) dtype = 3;      // tensors and packs: the split-operand mode's
    hipStream_t s = (hipStream_t)stream;
    const size_t sz = dtype_size(dtype);
    OpPlan p = op_plan(N, H, W, Cin, Cout, k, dtype);
    char* ws = (char*)workspace;
    PadGeom gx{N, H, W, p.Cin_p}, gy{N, H, W, p.Cdy};
    char* xp = ws + p.xp + gx.base_off(sz);
    char* dyp = ws + p.dyp + gy.base_off(sz);
    HIPCHK(op_pack_bordered(dtype, x, ws + p.xp, p.wf - p.xp, gx, Cin, s));
    HIPCHK(op_pack_bordered(dtype, dy, ws + p.dyp, p.dx - p.dyp, gy, Cout, s));
    if (dx) {
        HIPCHK(launch_pack_weights(dtype, w, nullptr, ws + p.wd, k * k, Cin, Cout, 0, 0, p.Cin_pad, p.Cdy,
                                   conv_filter_layout(k * k, W, p.Cdy * dtype_kbytes(dtype), p.Cin_p, N * H * W, 1, (int)sz, dtype_split(dtype)), s));
        ConvArgs a{};
        a.x = dyp; a.w = ws + p.wd; a.y = ws + p.dx;
        a.N = N; a.H = H; a.W = W; a.C = p.Cdy; a.M = N * H * W; a.Cout = p.Cin_p; a.ldy = p.Cin_p; a.taps = k * k;
        a.is_dgrad = 1;
        if (p.ks_floats) { a.ks_scratch = (float*)(ws + p.ks); a.ks_floats = p.ks_floats; }
        HIPCHK(launch_conv(ldt, a, s));
        HIPCHK(launch_cast_to_f32(dtype, ws + p.dx, dx, (size_t)N * H * W, Cin, p.Cin_p, s));
    }
    if (dw) {
        WgradArgs g{};
        const bool direct = p.Cin_p == Cin;      // no channel padding: the kernel writes the caller's tensor
        g.x = xp; g.dy = dyp; g.dW = direct ? dw : (float*)(ws + p.dw);
        g.N = N; g.H = H; g.W = W; g.M = N * H * W; g.Cin = p.Cin_p; g.Cdy = p.Cdy; g.Cout = Cout;
        g.taps = k * k; g.splitk = 0; g.scale = 1.f;
        g.slab = (float*)(ws + p.slab); g.slab_floats = op_slab_floats(dtype);
        g.tile_cnt = op_counters(s, kWgCntInts); g.cnt_ints = g.tile_cnt ? kWgCntInts : 0;
        HIPCHK(launch_wgrad_auto(ldt, g, s));
        for (int t = 0; t < k * k && !direct; ++t)
            HIPCHK(hipMemcpyAsync(dw + (size_t)t * Cin * Cout, (float*)(ws + p.dw) + (size_t)t * p.Cin_p * Cout,
                                  (size_t)Cin * Cout * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    return Y2_OK;
}

// CRC-32C of a host buffer: the crc32 instruction where the CPU has it, one table otherwise (host code only)
static uint32_t crc32c_table(const unsigned char* p, size_t n, uint32_t c) {
    static uint32_t tab[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t v = i;
            for (int k = 0; k < 8; ++k) v = (v >> 1) ^ ((v & 1) ? 0x82F63B78u : 0u);
            tab[i] = v;
        }
        init = true;
    }
    for (size_t i = 0; i < n; ++i) c = tab[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c;
}
#if defined(__x86_64__)
__attribute__((target("sse4.2"))) static uint32_t crc32c_hw(const unsigned char* p, size_t n, uint32_t c) {
    uint64_t c64 = c;
    while (n && ((uintptr_t)p & 7)) { c64 = __builtin_ia32_crc32qi((uint32_t)c64, *p++); --n; }
    for (; n >= 8; n -= 8, p += 8) c64 = __builtin_ia32_crc32di(c64, *(const uint64_t*)p);
    while (n--) c64 = __builtin_ia32_crc32qi((uint32_t)c64, *p++);
    return (uint32_t)c64;
}
#endif
uint32_t y2_crc32c(const void* data, size_t n, uint32_t crc) {
    const unsigned char* p = (const unsigned char*)data;
    uint32_t c = crc ^ 0xFFFFFFFFu;
#if defined(__x86_64__)
    if (__builtin_cpu_supports("sse4.2")) return crc32c_hw(p, n, c) ^ 0xFFFFFFFFu;
#endif
    return crc32c_table(p, n, c) ^ 0xFFFFFFFFu;
}

}  // extern "C"
