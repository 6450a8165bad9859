// YOLOv2 pieces that north_star names but the reference does NOT contain (SURVEY §8 rows a-x1,
// a-x2): stride-2 reorg / passthrough concat, anchor-box decode, per-image greedy NMS.
// There is no reference code to follow; the specification is oracle/ext_ref.py (this repo),
// and parity is bit-exact for the index work (reorg, NMS keep lists) -- the file is compiled
// with -ffp-contract=off and the IoU uses the spec's fp32 operation order.
#include <stdio.h>
#include "common.h"
#include "kernels.h"
#include "../../include/yolo2_hip.h"

#include <stdarg.h>
namespace y2 {
int set_error(int code, const char* msg);   // net.hip (y2_last_error)
}
using namespace y2;
static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return set_error(code, buf);
}

#define EXTCHK(expr)                                                                        \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) return fail(Y2_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// ---------------------------------------------------------------------------
// reorg (space-to-depth, block s): y[n, h/s, w/s, ((h%s)*s + w%s)*C + c] = x[n, h, w, c]
// forward = 0 runs the inverse permutation (its gradient).  16-byte moves when C % 4 == 0.
// ---------------------------------------------------------------------------
template <int VEC>
__global__ void reorg_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C, int s,
                             int forward, int ldy, int coff) {
    const int Cv = C / VEC;
    const size_t total = (size_t)N * H * W * Cv;
    const int Ho = H / s, Wo = W / s;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % Cv);
        size_t p = i / Cv;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H);
        const int n = (int)(p / H);
        const size_t fine = (((size_t)n * H + h) * W + w) * C + (size_t)cv * VEC;
        const size_t coarse = (((size_t)n * Ho + h / s) * Wo + w / s) * ldy + coff +
                              (size_t)((h % s) * s + (w % s)) * C + (size_t)cv * VEC;
        if (VEC == 4) {
            if (forward) *(f32x4*)(y + coarse) = *(const f32x4*)(x + fine);
            else *(f32x4*)(y + fine) = *(const f32x4*)(x + coarse);
        } else {
            if (forward) y[coarse] = x[fine];
            else y[fine] = x[coarse];
        }
    }
}

static int reorg_launch(const float* x, float* y, int N, int H, int W, int C, int s, int forward, int ldy, int coff,
                        hipStream_t st) {
    const bool vec = (C % 4) == 0 && (ldy % 4) == 0 && (coff % 4) == 0;
    const size_t total = (size_t)N * H * W * (vec ? C / 4 : C);
    size_t nb = (total + 255) / 256;
    if (nb > 65536) nb = 65536;
    if (nb < 1) nb = 1;
    if (vec) hipLaunchKernelGGL(reorg_kernel<4>, dim3((unsigned)nb), dim3(256), 0, st, x, y, N, H, W, C, s, forward, ldy, coff);
    else hipLaunchKernelGGL(reorg_kernel<1>, dim3((unsigned)nb), dim3(256), 0, st, x, y, N, H, W, C, s, forward, ldy, coff);
    EXTCHK(hipGetLastError());
    return Y2_OK;
}

// copy channels [0, C) of src [M][lds] (offset soff) into dst [M][ldd] (offset doff)
__global__ void chan_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t M, int C, int lds,
                                 int soff, int ldd, int doff) {
    const size_t total = M * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t m = i / C;
        dst[m * ldd + doff + c] = src[m * lds + soff + c];
    }
}
static int chan_copy(const float* src, float* dst, size_t M, int C, int lds, int soff, int ldd, int doff, hipStream_t st) {
    size_t nb = (M * C + 255) / 256;
    if (nb > 65536) nb = 65536;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(chan_copy_kernel, dim3((unsigned)nb), dim3(256), 0, st, src, dst, M, C, lds, soff, ldd, doff);
    EXTCHK(hipGetLastError());
    return Y2_OK;
}

// dst += src (the 26x26x512 activation of the YOLOv2 graph has two consumers -- the pool and the passthrough --
// so its gradient is the sum of two paths)
__global__ void accumulate_kernel(float4* __restrict__ dst, const float4* __restrict__ src, size_t n4, float* dt,
                                  const float* st, size_t tail0, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 a = dst[i];
        const float4 b = src[i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        dst[i] = a;
    }
    if (blockIdx.x == 0)
        for (size_t i = tail0 + threadIdx.x; i < n; i += blockDim.x) dt[i] += st[i];
}

// ---------------------------------------------------------------------------
// anchor decode: net [N,S,S,B,5+C] = (tx, ty, tw, th, to, class logits)
//   bx = (sigmoid(tx) + col)/S, by = (sigmoid(ty) + row)/S, bw = pw*exp(tw)/S, bh = ph*exp(th)/S
//   score[c] = sigmoid(to) * softmax(class logits)[c]
// ---------------------------------------------------------------------------
__global__ void decode_anchors_kernel(const float* __restrict__ net, const float* __restrict__ anchors,
                                      float* __restrict__ boxes, float* __restrict__ scores, int N, int S, int B, int C) {
    const int total = N * S * S * B;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int b = i % B;
    const int cell = (i / B) % (S * S);
    const int row = cell / S, col = cell % S;
    const float* p = net + (size_t)i * (5 + C);
    const float sx = 1.f / (1.f + expf(-p[0])), sy = 1.f / (1.f + expf(-p[1]));
    const float so = 1.f / (1.f + expf(-p[4]));
    const float fs = (float)S;
    boxes[(size_t)i * 4 + 0] = (sx + (float)col) / fs;
    boxes[(size_t)i * 4 + 1] = (sy + (float)row) / fs;
    boxes[(size_t)i * 4 + 2] = anchors[2 * b] * expf(p[2]) / fs;
    boxes[(size_t)i * 4 + 3] = anchors[2 * b + 1] * expf(p[3]) / fs;
    float mx = p[5];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, p[5 + c]);
    float sum = 0.f;
    for (int c = 0; c < C; ++c) sum += expf(p[5 + c] - mx);
    for (int c = 0; c < C; ++c) scores[(size_t)i * C + c] = so * (expf(p[5 + c] - mx) / sum);
}

// ---------------------------------------------------------------------------
// per-image greedy NMS.  One block per image, K <= 4096 candidates.
//   order: score descending, ties by ascending index (bitonic sort of (score, index) in LDS);
//   candidates with score < score_thresh are dropped;
//   walk the order; a kept box suppresses every later box with IoU > iou_thresh
//   (and, if class_aware, the same class id).  keep[n][0..count) = original indices.
// IoU (fp32, this exact operation order -- oracle/ext_ref.py: nms_iou):
//   x1 = cx - w*0.5, x2 = cx + w*0.5 (same for y); iw = max(0, min(x2a,x2b) - max(x1a,x1b));
//   inter = iw*ih; union = (wa*ha + wb*hb) - inter; iou = union > 0 ? inter/union : 0
// ---------------------------------------------------------------------------
template <int KP>
__global__ __launch_bounds__(1024) void nms_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                    const int* __restrict__ classes, int K, float iou_thresh,
                                                    float score_thresh, int max_out, int class_aware,
                                                    int* __restrict__ keep, int* __restrict__ count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 33 bytes per candidate (132 KB at 4096)
    float* skey = (float*)smem;
    int* sidx = (int*)(skey + KP);
    float *bx1 = (float*)(sidx + KP), *by1 = bx1 + KP, *bx2 = by1 + KP, *by2 = bx2 + KP, *barea = by2 + KP;
    int* bcls = (int*)(barea + KP);
    unsigned char* sup = (unsigned char*)(bcls + KP);
    __shared__ int s_cnt, s_valid;
    const int n = blockIdx.x, tid = threadIdx.x;
    const float* bb = boxes + (size_t)n * K * 4;
    const float* sc = scores + (size_t)n * K;
    for (int i = tid; i < KP; i += 1024) {
        const bool v = i < K && sc[i] >= score_thresh;
        skey[i] = v ? sc[i] : -INFINITY;
        sidx[i] = i;
    }
    __syncthreads();
    // bitonic sort, descending by (score, -index)
    for (int k = 2; k <= KP; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < KP; i += 1024) {
                const int l = i ^ j;
                if (l > i) {
                    const float a = skey[i], b = skey[l];
                    const int ia = sidx[i], ib = sidx[l];
                    const bool a_first = (a > b) || (a == b && ia < ib);   // a precedes b in the final order
                    const bool desc = (i & k) == 0;
                    if (desc ? !a_first : a_first) {
                        skey[i] = b; skey[l] = a;
                        sidx[i] = ib; sidx[l] = ia;
                    }
                }
            }
            __syncthreads();
        }
    if (tid == 0) { s_cnt = 0; s_valid = 0; }
    __syncthreads();
    for (int i = tid; i < KP; i += 1024) {
        const bool v = skey[i] > -INFINITY;
        if (v) {
            const int o = sidx[i];
            const float cx = bb[o * 4 + 0], cy = bb[o * 4 + 1], w = bb[o * 4 + 2], h = bb[o * 4 + 3];
            const float hw = w * 0.5f, hh = h * 0.5f;
            bx1[i] = cx - hw; bx2[i] = cx + hw; by1[i] = cy - hh; by2[i] = cy + hh;
            barea[i] = w * h;
            bcls[i] = classes ? classes[(size_t)n * K + o] : 0;
            atomicAdd(&s_valid, 1);
        }
        sup[i] = 0;
    }
    __syncthreads();
    const int nvalid = s_valid;   // valid entries are the first nvalid of the sorted order
    for (int i = 0; i < nvalid; ++i) {
        if (s_cnt >= max_out) break;
        if (!sup[i]) {
            if (tid == 0) keep[(size_t)n * max_out + s_cnt] = sidx[i];
            const float ax1 = bx1[i], ay1 = by1[i], ax2 = bx2[i], ay2 = by2[i], aa = barea[i];
            const int ac = bcls[i];
            for (int j = i + 1 + tid; j < nvalid; j += 1024) {
                if (sup[j]) continue;
                const float iw = fmaxf(0.f, fminf(ax2, bx2[j]) - fmaxf(ax1, bx1[j]));
                const float ih = fmaxf(0.f, fminf(ay2, by2[j]) - fmaxf(ay1, by1[j]));
                const float inter = iw * ih;
                const float uni = (aa + barea[j]) - inter;
                const float iou = uni > 0.f ? inter / uni : 0.f;
                if (iou > iou_thresh && (!class_aware || bcls[j] == ac)) sup[j] = 1;
            }
            __syncthreads();
            if (tid == 0) s_cnt = s_cnt + 1;
        }
        __syncthreads();
    }
    __syncthreads();
    if (tid == 0) count[n] = s_cnt;
    for (int i = s_cnt + tid; i < max_out; i += 1024) keep[(size_t)n * max_out + i] = -1;
}

// ---------------------------------------------------------------------------
// standalone 2x2/2 SAME max pool on fp32 NHWC (tf.nn.max_pool, reference darknet.py:24-25) and its gradient
// (first maximum in row-major window order, like MaxPoolGrad and like the fused pooling of bn.hip).  The
// network executor pools inside its BN pass; this op exists for composed graphs (the YOLOv2 passthrough
// needs the UN-pooled 26x26 activation as well as the pooled one).
// ---------------------------------------------------------------------------
__global__ void maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ dy,
                               float* __restrict__ dx, int N, int H, int W, int C) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const size_t total = (size_t)N * Ho * Wo * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t p = i / C;
        const int wo = (int)(p % Wo); p /= Wo;
        const int ho = (int)(p % Ho);
        const int n = (int)(p / Ho);
        float best = -INFINITY;
        int arg = 0;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int h = 2 * ho + (d >> 1), w = 2 * wo + (d & 1);
            if (h < H && w < W) {
                const float v = x[(((size_t)n * H + h) * W + w) * C + c];
                if (v > best) { best = v; arg = d; }
            }
        }
        if (y) y[i] = best;
        if (dx) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int h = 2 * ho + (d >> 1), w = 2 * wo + (d & 1);
                if (h < H && w < W) dx[(((size_t)n * H + h) * W + w) * C + c] = (d == arg) ? dy[i] : 0.f;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// YOLOv2 anchor-box loss, forward + gradient in one kernel (specification: oracle/ext_ref.py yolov2_loss; the
// reference has no anchor model).  One block per image: the image's ground-truth boxes (one per responsible
// cell of the reference's label grid) go to LDS, then one thread per (cell, anchor) pair decodes its
// prediction, finds its best IoU over those boxes and writes its whole gradient row.
// ---------------------------------------------------------------------------
struct V2LossArgs {
    const float* net;      // [N][S][S][B][5+C]
    const float* labels;   // [N][S][S][5+C]
    const float* anchors;  // [B][2] cell units
    float* dnet;           // same shape as net (nullable)
    float* partial;        // [N][4] coord, object, noobject, class of one image
    int N, S, B, C;
    float image_size, coord, obj, noobj, cls, thresh;
};
constexpr int kV2MaxTruth = 1024;

Y2_DEV float v2_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }
Y2_DEV float v2_iou(float ax, float ay, float aw, float ah, float bx, float by, float bw, float bh) {
    const float iw = fmaxf(0.0f, fminf(ax + aw / 2, bx + bw / 2) - fmaxf(ax - aw / 2, bx - bw / 2));
    const float ih = fmaxf(0.0f, fminf(ay + ah / 2, by + bh / 2) - fmaxf(ay - ah / 2, by - bh / 2));
    const float inter = iw * ih;
    const float uni = aw * ah + bw * bh - inter;
    return uni > 0.0f ? inter / uni : 0.0f;
}

__global__ __launch_bounds__(256) void yolov2_loss_kernel(V2LossArgs a) {
    __shared__ float tr[kV2MaxTruth][4];
    __shared__ int ntr;
    __shared__ float red[4][256];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int cells = a.S * a.S, D = 5 + a.C, LD = 5 + a.C;
    const float fs = (float)a.S;
    if (tid == 0) ntr = 0;
    __syncthreads();
    for (int cell = tid; cell < cells; cell += 256) {
        const float* lab = a.labels + ((size_t)n * cells + cell) * LD;
        if (lab[0] > 0.0f) {
            const int k = atomicAdd(&ntr, 1);
            tr[k][0] = lab[1] / a.image_size * fs; tr[k][1] = lab[2] / a.image_size * fs;
            tr[k][2] = lab[3] / a.image_size * fs; tr[k][3] = lab[4] / a.image_size * fs;
        }
    }
    __syncthreads();
    const int nt = ntr;
    float t_coord = 0.f, t_obj = 0.f, t_noobj = 0.f, t_cls = 0.f;
    const float invN = 1.0f / (float)a.N;
    for (int p = tid; p < cells * a.B; p += 256) {
        const int cell = p / a.B, b = p - cell * a.B;
        const int row = cell / a.S, col = cell - row * a.S;
        const float* t = a.net + (((size_t)n * cells + cell) * a.B + b) * D;
        float* g = a.dnet ? a.dnet + (((size_t)n * cells + cell) * a.B + b) * D : nullptr;
        const float* lab = a.labels + ((size_t)n * cells + cell) * LD;
        const float aw = a.anchors[2 * b], ah = a.anchors[2 * b + 1];
        const float sx = v2_sigmoid(t[0]), sy = v2_sigmoid(t[1]), so = v2_sigmoid(t[4]);
        const float px = sx + (float)col, py = sy + (float)row, pw = aw * expf(t[2]), ph = ah * expf(t[3]);
        bool responsible = false;
        float gx = 0.f, gy = 0.f, gw = 1.f, gh = 1.f;
        if (lab[0] > 0.0f) {
            gx = lab[1] / a.image_size * fs; gy = lab[2] / a.image_size * fs;
            gw = lab[3] / a.image_size * fs; gh = lab[4] / a.image_size * fs;
            int bs = 0;
            float bi = -1.0f;
            for (int k = 0; k < a.B; ++k) {   // the anchor whose shape fits best, first one on ties
                const float kw = a.anchors[2 * k], kh = a.anchors[2 * k + 1];
                const float inter = fminf(gw, kw) * fminf(gh, kh);
                const float si = inter / (gw * gh + kw * kh - inter);
                if (si > bi) { bi = si; bs = k; }
            }
            responsible = (bs == b);
        }
        if (responsible) {
            const float ex = sx - (gx - (float)col), ey = sy - (gy - (float)row);
            const float ew = t[2] - logf(gw / aw), eh = t[3] - logf(gh / ah);
            t_coord += a.coord * (ex * ex + ey * ey + ew * ew + eh * eh);
            const float iou = v2_iou(px, py, pw, ph, gx, gy, gw, gh);
            const float eo = so - iou;
            t_obj += a.obj * eo * eo;
            int k = 0;
            float bestl = lab[5], m = t[5];
            for (int c = 1; c < a.C; ++c) {
                if (lab[5 + c] > bestl) { bestl = lab[5 + c]; k = c; }     // class = argmax of the one-hot
                m = fmaxf(m, t[5 + c]);
            }
            float se = 0.f;
            for (int c = 0; c < a.C; ++c) se += expf(t[5 + c] - m);
            const float lse = m + logf(se);
            t_cls += a.cls * (lse - t[5 + k]);
            if (g) {
                g[0] = a.coord * 2.0f * ex * sx * (1.0f - sx) * invN;
                g[1] = a.coord * 2.0f * ey * sy * (1.0f - sy) * invN;
                g[2] = a.coord * 2.0f * ew * invN;
                g[3] = a.coord * 2.0f * eh * invN;
                g[4] = a.obj * 2.0f * eo * so * (1.0f - so) * invN;
                for (int c = 0; c < a.C; ++c)
                    g[5 + c] = a.cls * (expf(t[5 + c] - lse) - (c == k ? 1.0f : 0.0f)) * invN;
            }
        } else {
            float best = 0.f;
            for (int k = 0; k < nt; ++k) best = fmaxf(best, v2_iou(px, py, pw, ph, tr[k][0], tr[k][1], tr[k][2], tr[k][3]));
            const bool pen = best <= a.thresh;
            if (pen) t_noobj += a.noobj * so * so;
            if (g) {
                g[0] = g[1] = g[2] = g[3] = 0.f;
                g[4] = pen ? a.noobj * 2.0f * so * so * (1.0f - so) * invN : 0.f;
                for (int c = 0; c < a.C; ++c) g[5 + c] = 0.f;
            }
        }
    }
    red[0][tid] = t_coord; red[1][tid] = t_obj; red[2][tid] = t_noobj; red[3][tid] = t_cls;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s)
            for (int k = 0; k < 4; ++k) red[k][tid] += red[k][tid + s];
        __syncthreads();
    }
    if (tid < 4) a.partial[n * 4 + tid] = red[tid][0];
}
__global__ void yolov2_loss_finalize_kernel(const float* partial, float* loss, int N) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float p[4] = {0.f, 0.f, 0.f, 0.f};
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < 4; ++k) p[k] += partial[n * 4 + k];
    float tot = 0.f;
    for (int k = 0; k < 4; ++k) { loss[k] = p[k] / (float)N; tot += loss[k]; }
    loss[4] = tot;
}

extern "C" {

int y2_maxpool2x2(const float* x, float* y, int N, int H, int W, int C, void* stream) {
    if (!x || !y) return fail(Y2_ERR_ARG, "null tensor");
    const size_t total = (size_t)N * ((H + 1) / 2) * ((W + 1) / 2) * C;
    size_t nb = (total + 255) / 256;
    if (nb > 65536) nb = 65536;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, y, nullptr, nullptr, N, H,
                       W, C);
    EXTCHK(hipGetLastError());
    return Y2_OK;
}

int y2_maxpool2x2_backward(const float* x, const float* dy, float* dx, int N, int H, int W, int C, void* stream) {
    if (!x || !dy || !dx) return fail(Y2_ERR_ARG, "null tensor");
    const size_t total = (size_t)N * ((H + 1) / 2) * ((W + 1) / 2) * C;
    size_t nb = (total + 255) / 256;
    if (nb > 65536) nb = 65536;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, nullptr, dy, dx, N, H,
                       W, C);
    EXTCHK(hipGetLastError());
    return Y2_OK;
}

int y2_reorg(const float* x, float* y, int N, int H, int W, int C, int stride, int forward, void* stream) {
    if (!x || !y) return fail(Y2_ERR_ARG, "null tensor");
    if (stride < 1 || H % stride || W % stride) return fail(Y2_ERR_ARG, "H and W must be multiples of the stride");
    // forward: x fine [N,H,W,C] -> y coarse; inverse: x coarse -> y fine [N,H,W,C]
    return reorg_launch(x, y, N, H, W, C, stride, forward ? 1 : 0, stride * stride * C, 0, (hipStream_t)stream);
}

int y2_passthrough_concat(const float* fine, const float* coarse, float* out, int N, int H, int W, int Cf, int Cc,
                          void* stream) {
    if (!fine || !coarse || !out) return fail(Y2_ERR_ARG, "null tensor");
    const int ld = 4 * Cf + Cc;
    int rc = reorg_launch(fine, out, N, 2 * H, 2 * W, Cf, 2, 1, ld, 0, (hipStream_t)stream);
    if (rc) return rc;
    return chan_copy(coarse, out, (size_t)N * H * W, Cc, Cc, 0, ld, 4 * Cf, (hipStream_t)stream);
}

int y2_passthrough_concat_backward(const float* dout, float* dfine, float* dcoarse, int N, int H, int W, int Cf, int Cc,
                                   void* stream) {
    if (!dout || !dfine || !dcoarse) return fail(Y2_ERR_ARG, "null tensor");
    const int ld = 4 * Cf + Cc;
    int rc = reorg_launch(dout, dfine, N, 2 * H, 2 * W, Cf, 2, 0, ld, 0, (hipStream_t)stream);
    if (rc) return rc;
    return chan_copy(dout, dcoarse, (size_t)N * H * W, Cc, ld, 4 * Cf, Cc, 0, (hipStream_t)stream);
}

int y2_decode_anchors(const float* net, const float* anchors, float* boxes, float* scores, int N, int S, int B, int C,
                      void* stream) {
    if (!net || !anchors || !boxes || !scores) return fail(Y2_ERR_ARG, "null tensor");
    if (N < 1 || S < 1 || B < 1 || C < 1) return fail(Y2_ERR_ARG, "bad shape");
    const int total = N * S * S * B;
    hipLaunchKernelGGL(decode_anchors_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, net, anchors,
                       boxes, scores, N, S, B, C);
    EXTCHK(hipGetLastError());
    return Y2_OK;
}

int y2_nms(const float* boxes, const float* scores, const int* classes, int N, int K, float iou_thresh,
           float score_thresh, int max_out, int class_aware, int* keep, int* count, void* stream) {
    if (!boxes || !scores || !keep || !count) return fail(Y2_ERR_ARG, "null tensor");
    if (N < 1 || K < 1 || K > 4096 || max_out < 1) return fail(Y2_ERR_ARG, "need 1 <= K <= 4096 candidates per image");
    if (class_aware && !classes) return fail(Y2_ERR_ARG, "class-aware NMS needs class ids");
    hipStream_t st = (hipStream_t)stream;
#define NMS_LAUNCH(KP)                                                                                          \
    do {                                                                                                        \
        const int lds = KP * 33 + 64;                                                                           \
        EXTCHK(hipFuncSetAttribute((const void*)nms_kernel<KP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
        hipLaunchKernelGGL(nms_kernel<KP>, dim3(N), dim3(1024), lds, st, boxes, scores, classes, K, iou_thresh, \
                           score_thresh, max_out, class_aware, keep, count);                                    \
    } while (0)
    if (K <= 1024) NMS_LAUNCH(1024);
    else if (K <= 2048) NMS_LAUNCH(2048);
    else NMS_LAUNCH(4096);
#undef NMS_LAUNCH
    EXTCHK(hipGetLastError());
    return Y2_OK;
}

int y2_accumulate(float* dst, const float* src, size_t n, void* stream) {
    if (!dst || !src) return fail(Y2_ERR_ARG, "null tensor");
    const size_t n4 = n / 4;
    size_t nb = (n4 + 255) / 256;
    if (nb > 16384) nb = 16384;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(accumulate_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (float4*)dst,
                       (const float4*)src, n4, dst, src, n4 * 4, n);
    EXTCHK(hipGetLastError());
    return Y2_OK;
}

// best class score and its index per box: scores [R][C] -> best [R], cls [R] (ties: the smallest index, as argmax)
__global__ void class_argmax_kernel(const float* __restrict__ scores, float* __restrict__ best, int* __restrict__ cls,
                                    int R, int C) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float* p = scores + (size_t)r * C;
    float m = p[0];
    int k = 0;
    for (int c = 1; c < C; ++c) {
        const float v = p[c];
        if (v > m) { m = v; k = c; }
    }
    best[r] = m;
    cls[r] = k;
}
int y2_class_argmax(const float* scores, float* best, int* cls, int rows, int classes, void* stream) {
    if (!scores || !best || !cls || rows < 0 || classes < 1) return fail(Y2_ERR_ARG, "bad arguments");
    if (rows == 0) return Y2_OK;
    hipLaunchKernelGGL(class_argmax_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, scores, best, cls,
                       rows, classes);
    EXTCHK(hipGetLastError());
    return Y2_OK;
}

// x *= s (loss scaling of an output gradient in front of a half-precision backward pass)
__global__ void scale_kernel(float* __restrict__ x, size_t n, float s) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] *= s;
}
int y2_scale(float* x, size_t n, float s, void* stream) {
    if (!x) return fail(Y2_ERR_ARG, "null tensor");
    if (n == 0) return Y2_OK;
    size_t nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, n, s);
    EXTCHK(hipGetLastError());
    return Y2_OK;
}

size_t y2_yolov2_loss_workspace_bytes(int batch) { return (size_t)batch * 4 * sizeof(float) + 256; }

int y2_yolov2_loss(const float* net, const float* labels, const float* anchors, int batch, int S, int B, int num_class,
                   float image_size, const float* scales, float* loss, float* dnet, void* workspace, void* stream) {
    if (!net || !labels || !anchors || !loss || !workspace) return fail(Y2_ERR_ARG, "null tensor");
    if (batch < 1 || S < 1 || S * S > kV2MaxTruth || B < 1 || num_class < 1) return fail(Y2_ERR_ARG, "bad loss geometry");
    V2LossArgs a{};
    a.net = net; a.labels = labels; a.anchors = anchors; a.dnet = dnet; a.partial = (float*)workspace;
    a.N = batch; a.S = S; a.B = B; a.C = num_class; a.image_size = image_size;
    a.coord = scales ? scales[0] : 1.0f; a.obj = scales ? scales[1] : 5.0f; a.noobj = scales ? scales[2] : 1.0f;
    a.cls = scales ? scales[3] : 1.0f; a.thresh = scales ? scales[4] : 0.6f;
    hipLaunchKernelGGL(yolov2_loss_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, a);
    EXTCHK(hipGetLastError());
    hipLaunchKernelGGL(yolov2_loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a.partial, loss, batch);
    EXTCHK(hipGetLastError());
    return Y2_OK;
}

}  // extern "C"
