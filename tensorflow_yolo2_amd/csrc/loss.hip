// YOLO grid loss (forward + backward in ONE kernel), IoU, detection decode,
// classifier head tail (7x7 average pool, softmax cross-entropy).
// Reference: src/yolo2_nets/net_utils.py:222-439, src/config.py:37-45,
// src/imagenet/imagenet_train_darknet.py:51-53.
//
// The reference builds ~120 tiny TF ops for the loss (launch-bound); here one
// thread owns one grid cell and keeps every intermediate in registers.  This
// file is compiled with -ffp-contract=off: the fp32 operation ORDER mirrors the
// reference graph so that `ious` and the 0/1 `object_mask` (index work) come
// out bit-identical to an op-by-op fp32 evaluation.
#include "common.h"
#include "kernels.h"

namespace y2 {

constexpr int kMaxB = 8;

struct IouFwd {
    float x1a, y1a, x2a, y2a, x1b, y1b, x2b, y2b;
    float ddx, ddy, ix, iy, inter, u_raw, uni, ratio, iou;
};

// boxes are (x_center, y_center, w, h) -- net_utils.py:231-260
Y2_DEV IouFwd iou_forward(float ax, float ay, float aw, float ah, float bx, float by, float bw, float bh) {
    IouFwd f;
    f.x1a = ax - aw / 2.0f; f.y1a = ay - ah / 2.0f; f.x2a = ax + aw / 2.0f; f.y2a = ay + ah / 2.0f;
    f.x1b = bx - bw / 2.0f; f.y1b = by - bh / 2.0f; f.x2b = bx + bw / 2.0f; f.y2b = by + bh / 2.0f;
    const float lux = fmaxf(f.x1a, f.x1b), luy = fmaxf(f.y1a, f.y1b);
    const float rdx = fminf(f.x2a, f.x2b), rdy = fminf(f.y2a, f.y2b);
    f.ddx = rdx - lux; f.ddy = rdy - luy;
    f.ix = fmaxf(0.0f, f.ddx); f.iy = fmaxf(0.0f, f.ddy);
    f.inter = f.ix * f.iy;
    const float sq1 = (f.x2a - f.x1a) * (f.y2a - f.y1a);
    const float sq2 = (f.x2b - f.x1b) * (f.y2b - f.y1b);
    f.u_raw = sq1 + sq2 - f.inter;
    f.uni = fmaxf(f.u_raw, 1e-10f);
    f.ratio = f.inter / f.uni;
    f.iou = fminf(fmaxf(f.ratio, 0.0f), 1.0f);
    return f;
}

__global__ void get_iou_kernel(const float* b1, const float* b2, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* a = b1 + (size_t)i * 4;
    const float* b = b2 + (size_t)i * 4;
    out[i] = iou_forward(a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]).iou;
}
hipError_t launch_get_iou(const float* b1, const float* b2, float* out, int n, hipStream_t s) {
    hipLaunchKernelGGL(get_iou_kernel, dim3((n + 255) / 256), dim3(256), 0, s, b1, b2, out, n);
    return hipGetLastError();
}

// one wave per block: 10,816 cells at configs[3] then spread over 169 CUs instead of 43 (the kernel is a chain of
// dependent scalar math per cell: latency, not throughput)
constexpr int kLossThreads = 64;
int loss_blocks(int N, int S) { return (N * S * S + kLossThreads - 1) / kLossThreads; }

__global__ __launch_bounds__(kLossThreads) void yolo_loss_kernel(LossArgs a) {
    __shared__ float red[4][kLossThreads];
    const int cells = a.N * a.S * a.S;
    const int cell = blockIdx.x * kLossThreads + threadIdx.x;
    float t_class = 0.f, t_obj = 0.f, t_noobj = 0.f, t_coord = 0.f;
    if (cell < cells) {
        const int B = a.B, C = a.C, D = C + 5 * B;
        const int col = cell % a.S, row = (cell / a.S) % a.S;
        const float* net = a.net + (size_t)cell * D;
        const float* lab = a.labels + (size_t)cell * (5 + C);
        float* dnet = a.dnet ? a.dnet + (size_t)cell * D : nullptr;
        const float nb = (float)a.N;
        const float resp = lab[0];
        // ---- class term (net_utils.py:290-297)
        for (int c = 0; c < C; ++c) {
            const float d = resp * (net[c] - lab[5 + c]);
            t_class += d * d;
            if (dnet) dnet[c] = 2.0f * resp * resp * (net[c] - lab[5 + c]) / nb;
        }
        // ---- boxes (net_utils.py:302-334)
        const float Sf = (float)a.S;
        const float gx = lab[1] / a.image_size, gy = lab[2] / a.image_size;
        const float gw = lab[3] / a.image_size, gh = lab[4] / a.image_size;
        const float offx = (float)col, offy = (float)row;
        IouFwd f[kMaxB];
        float iou_max = -INFINITY;
        for (int b = 0; b < B; ++b) {
            const float* pb = net + C + B + 4 * b;
            const float px = (pb[0] + offx) / Sf, py = (pb[1] + offy) / Sf;
            const float pw = pb[2] * pb[2], ph = pb[3] * pb[3];
            f[b] = iou_forward(px, py, pw, ph, gx, gy, gw, gh);
            iou_max = fmaxf(iou_max, f[b].iou);
        }
        const float tx = gx * Sf - offx, ty = gy * Sf - offy;
        const float tw = sqrtf(gw), th = sqrtf(gh);
        for (int b = 0; b < B; ++b) {
            const float* pb = net + C + B + 4 * b;
            const float conf = net[C + b];
            const float iou = f[b].iou;
            const float mask = ((iou >= iou_max) ? 1.0f : 0.0f) * resp;      // :323-324
            const float nomask = 1.0f - mask;
            a.ious[(size_t)cell * B + b] = iou;
            a.mask[(size_t)cell * B + b] = mask;
            const float d0 = mask * (pb[0] - tx), d1 = mask * (pb[1] - ty);
            const float d2 = mask * (pb[2] - tw), d3 = mask * (pb[3] - th);
            t_coord += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
            const float od = mask * (conf - iou);
            t_obj += od * od;
            const float nd = nomask * conf;
            t_noobj += nd * nd;
            if (dnet) {
                dnet[C + b] = 2.0f * mask * mask * (conf - iou) / nb +
                              a.lambda_noobj * 2.0f * (nomask * nomask) * conf / nb;
                // gradient through ious (TF tie conventions, see oracle/loss_ref.py)
                const float diou = -2.0f * mask * mask * (conf - iou) / nb;
                const IouFwd& q = f[b];
                const float g_ratio = (q.ratio >= 0.0f && q.ratio <= 1.0f) ? diou : 0.0f;
                float g_inter = g_ratio / q.uni;
                const float g_union = -g_ratio * q.inter / (q.uni * q.uni);
                const float g_uraw = (q.u_raw >= 1e-10f) ? g_union : 0.0f;
                const float g_sq1 = g_uraw;
                g_inter = g_inter - g_uraw;
                const float g_ix = g_inter * q.iy, g_iy = g_inter * q.ix;
                const float g_ddx = (0.0f >= q.ddx) ? 0.0f : g_ix;
                const float g_ddy = (0.0f >= q.ddy) ? 0.0f : g_iy;
                float g_x1a = (q.x1a >= q.x1b) ? -g_ddx : 0.0f;
                float g_y1a = (q.y1a >= q.y1b) ? -g_ddy : 0.0f;
                float g_x2a = (q.x2a <= q.x2b) ? g_ddx : 0.0f;
                float g_y2a = (q.y2a <= q.y2b) ? g_ddy : 0.0f;
                g_x2a = g_x2a + g_sq1 * (q.y2a - q.y1a);
                g_x1a = g_x1a - g_sq1 * (q.y2a - q.y1a);
                g_y2a = g_y2a + g_sq1 * (q.x2a - q.x1a);
                g_y1a = g_y1a - g_sq1 * (q.x2a - q.x1a);
                const float g_px = g_x1a + g_x2a, g_py = g_y1a + g_y2a;
                const float g_pw = (g_x2a - g_x1a) / 2.0f, g_ph = (g_y2a - g_y1a) / 2.0f;
                const float m2 = mask * mask;
                float* db = dnet + C + B + 4 * b;
                db[0] = g_px / Sf + a.lambda_coord * 2.0f * m2 * (pb[0] - tx) / nb;
                db[1] = g_py / Sf + a.lambda_coord * 2.0f * m2 * (pb[1] - ty) / nb;
                db[2] = g_pw * 2.0f * pb[2] + a.lambda_coord * 2.0f * m2 * (pb[2] - tw) / nb;
                db[3] = g_ph * 2.0f * pb[3] + a.lambda_coord * 2.0f * m2 * (pb[3] - th) / nb;
            }
        }
    }
    red[0][threadIdx.x] = t_class;
    red[1][threadIdx.x] = t_obj;
    red[2][threadIdx.x] = t_noobj;
    red[3][threadIdx.x] = t_coord;
    __syncthreads();
    for (int s = kLossThreads / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s)
            for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 4) a.partial[blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}

__global__ void yolo_loss_finalize_kernel(LossArgs a, int nblocks) {
    // one wave; every lane sums a strided share of the block partials of all four terms in double, then a
    // fixed butterfly: the result does not depend on timing
    const int lane = threadIdx.x;
    double t[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = lane; b < nblocks; b += 64)
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] += (double)a.partial[b * 4 + k];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        for (int off = 32; off > 0; off >>= 1) t[k] += __shfl_xor(t[k], off, 64);
    if (lane == 0) {
        float l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double v = t[k] / (double)a.N;   // reduce_mean over the batch
            if (k == 2) v *= (double)a.lambda_noobj;
            if (k == 3) v *= (double)a.lambda_coord;
            l[k] = (float)v;
            a.loss[k] = l[k];
        }
        a.loss[4] = l[0] + l[1] + l[2] + l[3];   // net_utils.py:372 order
    }
}

hipError_t launch_yolo_loss(const LossArgs& a, hipStream_t s) {
    if (a.B > kMaxB) return hipErrorInvalidValue;
    const int nb = loss_blocks(a.N, a.S);
    hipLaunchKernelGGL(yolo_loss_kernel, dim3(nb), dim3(kLossThreads), 0, s, a);
    hipLaunchKernelGGL(yolo_loss_finalize_kernel, dim3(1), dim3(64), 0, s, a, nb);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// show_yolo_detection's arithmetic (net_utils.py:393-421): float64 products as
// numpy-1.x scalar promotion gives, int() truncation, py2 floor division.
// out[(cell*B+b)*8 + {0..7}] = keep, ulx, uly, w, h, cls, cell_row, cell_col
// ---------------------------------------------------------------------------
__global__ void decode_kernel(const float* pred, int S, int B, int C, int im_w, int im_h, float thresh, int* out,
                              float* out_conf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * S * B) return;
    const int b = i % B, cell = i / B;
    const int r = cell % S, c = cell / S;  // reference loop names: c = first (row) index, r = second (column)
    const int D = C + 5 * B;
    const float* p = pred + (size_t)cell * D;
    const float conf = p[C + b];
    const float* pb = p + C + B + 4 * b;
    const double xs = ((double)pb[0] + (double)r) / (double)S;   // offset[c, r, b] = r (column index)
    const double ys = ((double)pb[1] + (double)c) / (double)S;
    const float ws = pb[2] * pb[2], hs = pb[3] * pb[3];          // np.square on float32
    const int x = (int)(xs * (double)im_w), y = (int)(ys * (double)im_h);
    const int w = (int)((double)ws * (double)im_w), h = (int)((double)hs * (double)im_h);
    int cls = 0;
    float best = p[0];
    for (int k = 1; k < C; ++k)
        if (p[k] > best) {
            best = p[k];
            cls = k;
        }
    auto floordiv2 = [](int v) { return (v >= 0) ? v / 2 : -((-v + 1) / 2); };
    int* o = out + (size_t)i * 8;
    o[0] = conf > thresh ? 1 : 0;
    o[1] = x - floordiv2(w);
    o[2] = y - floordiv2(h);
    o[3] = w;
    o[4] = h;
    o[5] = cls;
    o[6] = c;
    o[7] = r;
    out_conf[i] = conf;
}
hipError_t launch_decode(const float* pred, int S, int B, int C, int im_w, int im_h, float thresh, int* out,
                         float* out_conf, hipStream_t s) {
    const int n = S * S * B;
    hipLaunchKernelGGL(decode_kernel, dim3((n + 63) / 64), dim3(64), 0, s, pred, S, B, C, im_w, im_h, thresh, out,
                       out_conf);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// classifier tail: tf.layers.average_pooling2d(h, [k,k], [k,k]) VALID + reshape
// (darknet.py:116-117) and sparse softmax cross-entropy + reduce_mean.
// ---------------------------------------------------------------------------
__global__ void avgpool_fwd_kernel(const float* h, float* out, int N, int H, int W, int C, int k) {
    const int Ho = H / k, Wo = W / k;
    const size_t total = (size_t)N * Ho * Wo * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t p = i / C;
        const int wo = (int)(p % Wo), ho = (int)((p / Wo) % Ho), n = (int)(p / ((size_t)Wo * Ho));
        float s = 0.f;
        for (int dy = 0; dy < k; ++dy)
            for (int dx = 0; dx < k; ++dx) s += h[(((size_t)n * H + ho * k + dy) * W + wo * k + dx) * C + c];
        out[i] = s / (float)(k * k);
    }
}
__global__ void avgpool_bwd_kernel(const float* dout, float* dh, int N, int H, int W, int C, int k) {
    const int Ho = H / k, Wo = W / k;
    const size_t total = (size_t)N * H * W * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t p = i / C;
        const int w = (int)(p % W), hh = (int)((p / W) % H), n = (int)(p / ((size_t)W * H));
        const int ho = hh / k, wo = w / k;
        dh[i] = (ho < Ho && wo < Wo) ? dout[(((size_t)n * Ho + ho) * Wo + wo) * C + c] / (float)(k * k) : 0.f;
    }
}
hipError_t launch_avgpool_fwd(const float* h, float* out, int N, int H, int W, int C, int k, hipStream_t s) {
    size_t total = (size_t)N * (H / k) * (W / k) * C;
    size_t nb = (total + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3((unsigned)nb), dim3(256), 0, s, h, out, N, H, W, C, k);
    return hipGetLastError();
}
hipError_t launch_avgpool_bwd(const float* dout, float* dh, int N, int H, int W, int C, int k, hipStream_t s) {
    size_t total = (size_t)N * H * W * C;
    size_t nb = (total + 255) / 256;
    if (nb > 8192) nb = 8192;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, s, dout, dh, N, H, W, C, k);
    return hipGetLastError();
}

// one block per row; loss[0] accumulates mean CE with an atomic (pre-zeroed by the launcher)
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* logits, const int* labels, float* loss,
                                                         float* dlogits, int N, int C) {
    __shared__ float red[256];
    const int n = blockIdx.x;
    const float* z = logits + (size_t)n * C;
    float m = -INFINITY;
    for (int c = threadIdx.x; c < C; c += 256) m = fmaxf(m, z[c]);
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    m = red[0];
    __syncthreads();
    float sum = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) sum += expf(z[c] - m);
    red[threadIdx.x] = sum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const float lse = logf(red[0]);
    const int lab = labels[n];
    if (dlogits)
        for (int c = threadIdx.x; c < C; c += 256) {
            const float p = expf(z[c] - m - lse);
            dlogits[(size_t)n * C + c] = (p - (c == lab ? 1.0f : 0.0f)) / (float)N;
        }
    if (threadIdx.x == 0) atomicAdd(loss, -(z[lab] - m - lse) / (float)N);
}
hipError_t launch_softmax_ce(const float* logits, const int* labels, float* loss, float* dlogits, int N, int C,
                             hipStream_t s) {
    hipError_t e = hipMemsetAsync(loss, 0, sizeof(float), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(softmax_ce_kernel, dim3(N), dim3(256), 0, s, logits, labels, loss, dlogits, N, C);
    return hipGetLastError();
}

// accuracy = mean(argmax(logits, 1) == labels) (imagenet_train_darknet.py:60-61); tf.argmax returns the
// smallest index among equal maxima.  One wave per row, one block for the batch: exact integer count.
__global__ __launch_bounds__(256) void accuracy_kernel(const float* logits, const int* labels, float* acc, int N, int C) {
    __shared__ int hits[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int mine = 0;
    for (int n = wave; n < N; n += 4) {
        const float* z = logits + (size_t)n * C;
        float best = -INFINITY;
        int arg = 0x7fffffff;
        for (int c = lane; c < C; c += 64) {
            const float v = z[c];
            if (v > best || (v == best && c < arg)) { best = v; arg = c; }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const float ob = __shfl_xor(best, off, 64);
            const int oa = __shfl_xor(arg, off, 64);
            if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
        }
        mine += (arg == labels[n]) ? 1 : 0;
    }
    if (lane == 0) hits[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) *acc = (float)(hits[0] + hits[1] + hits[2] + hits[3]) / (float)N;
}
hipError_t launch_accuracy(const float* logits, const int* labels, float* acc, int N, int C, hipStream_t s) {
    hipLaunchKernelGGL(accuracy_kernel, dim3(1), dim3(256), 0, s, logits, labels, acc, N, C);
    return hipGetLastError();
}

}  // namespace y2
