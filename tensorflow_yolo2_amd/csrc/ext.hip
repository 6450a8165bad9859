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

}  // extern "C"
