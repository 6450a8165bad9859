// slim.fully_connected of the ResNet swap's grid head (src/pascal/pascal_train_resnet.py:41-46: flatten ->
// fully_connected(4096) -> dropout -> fully_connected(S*S*(5B+C))), forward and backward, for a BATCH of at most 128 rows.
// With M = batch the three products
//     y  [M][N] = x [M][K]  W[K][N]          (+ bias, ReLU)
//     dx [M][K] = dy[M][N]  W[K][N]^T
//     dW [K][N] = x [M][K]^T dy[M][N]
// are bound by ONE pass over the weight matrix (fc1: 100352 x 4096 fp32 = 1.64 GB; 26 GFLOP per product), not by
// arithmetic.  The 1x1-convolution route of round 2 re-packed W to half precision twice per step, zeroed a 3.3 GB
// workspace and ran a 32-row GEMM through a 128-row tile (4.6 ms per step for fc1).  Here every kernel streams the
// fp32 master weights exactly once, converts in registers to the compute type (f16 / bf16 MFMA 32x32x16, or
// fp32 MFMA 32x32x2) and accumulates in fp32; no LDS, no packed copy:
//   forward : wave = 64 columns x one K slice; the B fragment of a lane is 8 (4) rows of one column -- per row the 32
//             lanes of a half-wave read one 128-byte line; split-K partials are added in slice order (deterministic)
//             by the reduce kernel, which also applies bias + ReLU;
//   dx      : wave = 64 rows of W, the whole N range (8 consecutive floats per lane and step);
//   dW      : wave = 32 rows x a range of column tiles; the x^T fragments stay in registers, the 32 x 32 fp32 tiles
//             are stored straight to dW (write-bound: 1.64 GB).
#include "../../include/yolo2_hip.h"
#include "common.h"
#include "kernels.h"
#include "optim_math.h"

namespace y2 {
int set_error(int code, const char* msg);

template <typename T> Y2_DEV typename Elem<T>::frag fc_frag(const float* v);
template <> Y2_DEV f32x4 fc_frag<float>(const float* v) { return (f32x4){v[0], v[1], v[2], v[3]}; }
template <> Y2_DEV f16x8 fc_frag<half_t>(const float* v) {
    u32x4 u = {pack2<half_t>(v[0], v[1]), pack2<half_t>(v[2], v[3]), pack2<half_t>(v[4], v[5]), pack2<half_t>(v[6], v[7])};
    return __builtin_bit_cast(f16x8, u);
}
template <> Y2_DEV bf16x8 fc_frag<bf16_t>(const float* v) {
    u32x4 u = {pack2<bf16_t>(v[0], v[1]), pack2<bf16_t>(v[2], v[3]), pack2<bf16_t>(v[4], v[5]), pack2<bf16_t>(v[6], v[7])};
    return __builtin_bit_cast(bf16x8, u);
}

// KPL consecutive floats (KPL = 4 or 8), 16-byte loads when VEC
template <int KPL, bool VEC>
Y2_DEV void fc_load_row(const float* p, float* v) {
    if constexpr (VEC) {
#pragma unroll
        for (int e = 0; e < KPL; e += 4) {
            const float4 t = *(const float4*)(p + e);
            v[e] = t.x; v[e + 1] = t.y; v[e + 2] = t.z; v[e + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int e = 0; e < KPL; ++e) v[e] = p[e];
    }
}

// ---------------------------------------------------------------------------
// forward partials: part[slice][M][N].  Rows m >= M and columns n >= N read clamped (valid) addresses; their products
// land only in accumulator rows / columns that are never stored.
// ---------------------------------------------------------------------------
template <typename T, int MT, bool VEC>
__global__ __launch_bounds__(256) void fc_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                     float* __restrict__ part, int M, int K, int N, int steps_per_slice) {
    constexpr int KPL = Elem<T>::kPerFrag, KS = 2 * KPL, CW = 2;
    typedef typename Elem<T>::frag frag_t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l32 = lane & 31, kg = lane >> 5;
    const int n0 = (blockIdx.x * 4 + wave) * 32 * CW;
    if (n0 >= N) return;
    const int ksteps = K / KS;
    const int s0 = blockIdx.y * steps_per_slice;
    int s1 = s0 + steps_per_slice;
    if (s1 > ksteps) s1 = ksteps;
    int ncol[CW];
#pragma unroll
    for (int j = 0; j < CW; ++j) ncol[j] = min(n0 + j * 32 + l32, N - 1);
    const float* xrow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xrow[mt] = x + (size_t)min(mt * 32 + l32, M - 1) * K + kg * KPL;
    f32x16 acc[MT][CW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < CW; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[mt][j][q] = 0.f;
    for (int st = s0; st < s1; ++st) {
        const size_t kb = (size_t)st * KS + kg * KPL;
        frag_t fb[CW];
#pragma unroll
        for (int j = 0; j < CW; ++j) {
            float v[KPL];
            const float* wp = W + kb * N + ncol[j];
#pragma unroll
            for (int e = 0; e < KPL; ++e) v[e] = wp[(size_t)e * N];
            fb[j] = fc_frag<T>(v);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float v[KPL];
            fc_load_row<KPL, VEC>(xrow[mt] + (size_t)st * KS, v);
            const frag_t fa = fc_frag<T>(v);
#pragma unroll
            for (int j = 0; j < CW; ++j) mma32(acc[mt][j], fa, fb[j]);
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < CW; ++j) {
            const int n = n0 + j * 32 + l32;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int m = mt * 32 + acc_row(q, kg);
                if (m < M && n < N) part[((size_t)blockIdx.y * M + m) * N + n] = acc[mt][j][q];
            }
        }
}
// y = act(sum over the slices in order + bias)
__global__ void fc_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ y,
                                 int slices, size_t MN, int N, int relu) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < MN; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < slices; ++k) s += part[(size_t)k * MN + i];
        if (bias) s += bias[i % N];
        y[i] = relu ? fmaxf(s, 0.f) : s;
    }
}

// ---------------------------------------------------------------------------
// dx[M][K]: wave = 64 consecutive rows of W (output columns), reduction over n.  The tail step (N % KS) zero-fills.
// ---------------------------------------------------------------------------
template <typename T, int MT, bool VEC>
__global__ __launch_bounds__(256) void fc_dx_kernel(const float* __restrict__ dy, const float* __restrict__ W,
                                                    float* __restrict__ dx, int M, int K, int N) {
    constexpr int KPL = Elem<T>::kPerFrag, KS = 2 * KPL, CW = 2;
    typedef typename Elem<T>::frag frag_t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l32 = lane & 31, kg = lane >> 5;
    const int k0 = (blockIdx.x * 4 + wave) * 32 * CW;
    if (k0 >= K) return;
    const float* wrow[CW];
#pragma unroll
    for (int j = 0; j < CW; ++j) wrow[j] = W + (size_t)min(k0 + j * 32 + l32, K - 1) * N + kg * KPL;
    const float* drow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) drow[mt] = dy + (size_t)min(mt * 32 + l32, M - 1) * N + kg * KPL;
    f32x16 acc[MT][CW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < CW; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[mt][j][q] = 0.f;
    const int full = N / KS;
    for (int st = 0; st < full; ++st) {
        frag_t fb[CW];
#pragma unroll
        for (int j = 0; j < CW; ++j) {
            float v[KPL];
            fc_load_row<KPL, VEC>(wrow[j] + (size_t)st * KS, v);
            fb[j] = fc_frag<T>(v);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float v[KPL];
            fc_load_row<KPL, VEC>(drow[mt] + (size_t)st * KS, v);
            const frag_t fa = fc_frag<T>(v);
#pragma unroll
            for (int j = 0; j < CW; ++j) mma32(acc[mt][j], fa, fb[j]);
        }
    }
    if (full * KS < N) {
        const int nb = full * KS + kg * KPL;
        frag_t fb[CW];
#pragma unroll
        for (int j = 0; j < CW; ++j) {
            float v[KPL];
#pragma unroll
            for (int e = 0; e < KPL; ++e) v[e] = nb + e < N ? wrow[j][(size_t)full * KS + e] : 0.f;
            fb[j] = fc_frag<T>(v);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float v[KPL];
#pragma unroll
            for (int e = 0; e < KPL; ++e) v[e] = nb + e < N ? drow[mt][(size_t)full * KS + e] : 0.f;
            const frag_t fa = fc_frag<T>(v);
#pragma unroll
            for (int j = 0; j < CW; ++j) mma32(acc[mt][j], fa, fb[j]);
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < CW; ++j) {
            const int k = k0 + j * 32 + l32;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int m = mt * 32 + acc_row(q, kg);
                if (m < M && k < K) dx[(size_t)m * K + k] = acc[mt][j][q];
            }
        }
}

// ---------------------------------------------------------------------------
// dW[K][N]: wave = 32 rows k x column tiles [t0, t1); reduction over the M batch rows (zero-filled to a multiple of KS).
// MS = reduction steps held in registers (MS * KS >= M).
// ---------------------------------------------------------------------------
template <typename T, int MS>
__global__ __launch_bounds__(256) void fc_dw_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                    float* __restrict__ dW, int M, int K, int N, int tiles_per_chunk) {
    constexpr int KPL = Elem<T>::kPerFrag, KS = 2 * KPL;
    typedef typename Elem<T>::frag frag_t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l32 = lane & 31, kg = lane >> 5;
    const int k0 = (blockIdx.x * 4 + wave) * 32;
    if (k0 >= K) return;
    const int kc = min(k0 + l32, K - 1);
    frag_t fa[MS];
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) {
        float v[KPL];
#pragma unroll
        for (int e = 0; e < KPL; ++e) {
            const int m = ms * KS + kg * KPL + e;
            v[e] = m < M ? x[(size_t)m * K + kc] : 0.f;
        }
        fa[ms] = fc_frag<T>(v);
    }
    const int ntiles = (N + 31) / 32;
    const int t0 = blockIdx.y * tiles_per_chunk;
    int t1 = t0 + tiles_per_chunk;
    if (t1 > ntiles) t1 = ntiles;
    for (int t = t0; t < t1; ++t) {
        const int n = t * 32 + l32, nc = min(n, N - 1);
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) {
            float v[KPL];
#pragma unroll
            for (int e = 0; e < KPL; ++e) {
                const int m = ms * KS + kg * KPL + e;
                v[e] = m < M ? dy[(size_t)m * N + nc] : 0.f;
            }
            mma32(acc, fa[ms], fc_frag<T>(v));
        }
        if (n < N) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int k = k0 + acc_row(q, kg);
                if (k < K) dW[(size_t)k * N + n] = acc[q];
            }
        }
    }
}

// The same product with the guarded Adam update of the weight in its epilogue (round 5): dW is NEVER stored.  The
// fully connected layer of the ResNet swap's grid head (src/pascal/pascal_train_resnet.py:41-46) holds 100352 x 4096
// weights -- 1.64 GB; writing its gradient, scanning it for the overflow guard and reading it back in the optimizer
// were 3 of the 10 passes over that size per step.  ctrl: the control block of y2_adam_step_guarded AFTER its
// advance (found_inf decides, lr_t applies); the arithmetic is optim_math.h's, bit for bit what the flat kernel does
// with the stored gradient.
struct FcCtrlView { int found_inf, step, skipped, reserved; float lr_t; };
template <typename T, int MS>
__global__ __launch_bounds__(256) void fc_dw_adam_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ W, float* __restrict__ Mo, float* __restrict__ Vo,
                                                         int M, int K, int N, int tiles_per_chunk, const FcCtrlView* ctrl,
                                                         float b1, float b2, float eps, float gmult) {
    if (ctrl->found_inf) return;          // overflowed gradients somewhere in the step: nothing moves
    const float lr_t = ctrl->lr_t;
    constexpr int KPL = Elem<T>::kPerFrag, KS = 2 * KPL;
    typedef typename Elem<T>::frag frag_t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l32 = lane & 31, kg = lane >> 5;
    const int k0 = (blockIdx.x * 4 + wave) * 32;
    if (k0 >= K) return;
    const int kc = min(k0 + l32, K - 1);
    frag_t fa[MS];
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) {
        float v[KPL];
#pragma unroll
        for (int e = 0; e < KPL; ++e) {
            const int m = ms * KS + kg * KPL + e;
            v[e] = m < M ? x[(size_t)m * K + kc] : 0.f;
        }
        fa[ms] = fc_frag<T>(v);
    }
    const int ntiles = (N + 31) / 32;
    const int t0 = blockIdx.y * tiles_per_chunk;
    int t1 = t0 + tiles_per_chunk;
    if (t1 > ntiles) t1 = ntiles;
    for (int t = t0; t < t1; ++t) {
        const int n = t * 32 + l32, nc = min(n, N - 1);
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) {
            float v[KPL];
#pragma unroll
            for (int e = 0; e < KPL; ++e) {
                const int m = ms * KS + kg * KPL + e;
                v[e] = m < M ? dy[(size_t)m * N + nc] : 0.f;
            }
            mma32(acc, fa[ms], fc_frag<T>(v));
        }
        if (n < N) {
            float pw[16], pm[16], pv[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {      // all loads of the tile first
                const int k = k0 + acc_row(q, kg);
                const size_t o = (size_t)(k < K ? k : K - 1) * N + n;
                pw[q] = W[o]; pm[q] = Mo[o]; pv[q] = Vo[o];
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int k = k0 + acc_row(q, kg);
                if (k < K) {
                    adam_update(pw[q], pm[q], pv[q], acc[q] * gmult, lr_t, b1, b2, eps);
                    const size_t o = (size_t)k * N + n;
                    W[o] = pw[q]; Mo[o] = pm[q]; Vo[o] = pv[q];
                }
            }
        }
    }
}

static int fc_fail(int code, const char* msg) { return set_error(code, msg); }
#define FCHK(expr)                                                    \
    do {                                                              \
        hipError_t _e = (expr);                                       \
        if (_e != hipSuccess) return fc_fail(Y2_ERR_HIP, hipGetErrorString(_e)); \
    } while (0)

static bool aligned16(const void* a, const void* b) { return (((uintptr_t)a | (uintptr_t)b) & 15) == 0; }

template <typename T>
static int fc_forward_T(const float* x, const float* w, const float* bias, float* y, int M, int K, int N, int relu,
                        hipStream_t s) {
    constexpr int KS = 2 * Elem<T>::kPerFrag;
    if (K % 16) return fc_fail(Y2_ERR_ARG, "fully connected: the input width must be a multiple of 16");
    const int ksteps = K / KS;
    const int nwt = (N + 63) / 64;                         // wave tiles
    int slices = 2048 / nwt;
    if (slices > ksteps / 8) slices = ksteps / 8;
    if (slices < 1) slices = 1;
    const int sps = (ksteps + slices - 1) / slices;
    slices = (ksteps + sps - 1) / sps;
    const size_t MN = (size_t)M * N;
    float* part = (float*)op_scratch(s, (size_t)slices * MN * sizeof(float));
    if (!part) return op_scratch_error();     // (op_scratch left the reason in the error state)
    const dim3 grid((nwt + 3) / 4, slices);
    const bool vec = aligned16(x, nullptr);
    const int mt = (M + 31) / 32;
#define FC_FWD(MTv)                                                                                                    \
    do {                                                                                                               \
        if (vec) hipLaunchKernelGGL((fc_fwd_kernel<T, MTv, true>), grid, dim3(256), 0, s, x, w, part, M, K, N, sps);    \
        else hipLaunchKernelGGL((fc_fwd_kernel<T, MTv, false>), grid, dim3(256), 0, s, x, w, part, M, K, N, sps);       \
    } while (0)
    if (mt == 1) FC_FWD(1);
    else if (mt == 2) FC_FWD(2);
    else FC_FWD(4);
#undef FC_FWD
    size_t nb = (MN + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(fc_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, s, part, bias, y, slices, MN, N, relu);
    FCHK(hipGetLastError());
    return Y2_OK;
}

template <typename T>
static int fc_backward_T(const float* x, const float* w, const float* dy, float* dx, float* dw, int M, int K, int N,
                         hipStream_t s) {
    constexpr int KS = 2 * Elem<T>::kPerFrag;
    const int mt = (M + 31) / 32;
    if (dx) {
        const bool vec = N % 4 == 0 && aligned16(dy, w);
        const dim3 grid(((K + 63) / 64 + 3) / 4);
#define FC_DX(MTv)                                                                                          \
    do {                                                                                                    \
        if (vec) hipLaunchKernelGGL((fc_dx_kernel<T, MTv, true>), grid, dim3(256), 0, s, dy, w, dx, M, K, N); \
        else hipLaunchKernelGGL((fc_dx_kernel<T, MTv, false>), grid, dim3(256), 0, s, dy, w, dx, M, K, N);    \
    } while (0)
        if (mt == 1) FC_DX(1);
        else if (mt == 2) FC_DX(2);
        else FC_DX(4);
#undef FC_DX
    }
    if (dw) {
        const int kw = ((K + 31) / 32 + 3) / 4;            // workgroups along K
        const int ntiles = (N + 31) / 32;
        int chunks = 2048 / kw;
        if (chunks > ntiles / 4) chunks = ntiles / 4;
        if (chunks < 1) chunks = 1;
        const int tpc = (ntiles + chunks - 1) / chunks;
        chunks = (ntiles + tpc - 1) / tpc;
        const dim3 grid(kw, chunks);
        const int ms = (M + KS - 1) / KS;
#define FC_DW(MSv) hipLaunchKernelGGL((fc_dw_kernel<T, MSv>), grid, dim3(256), 0, s, x, dy, dw, M, K, N, tpc)
        if (ms <= 1) FC_DW(1);
        else if (ms <= 2) FC_DW(2);
        else if (ms <= 4) FC_DW(4);
        else if (ms <= 8) FC_DW(8);
        else FC_DW(16);
#undef FC_DW
    }
    FCHK(hipGetLastError());
    return Y2_OK;
}
template <typename T>
static int fc_adam_T(const float* x, const float* dy, float* w, float* m, float* v, int M, int K, int N, const void* ctrl,
                     float b1, float b2, float eps, float gmult, hipStream_t s) {
    constexpr int KS = 2 * Elem<T>::kPerFrag;
    const int kw = ((K + 31) / 32 + 3) / 4;            // workgroups along K
    const int ntiles = (N + 31) / 32;
    int chunks = 2048 / kw;
    if (chunks > ntiles / 4) chunks = ntiles / 4;
    if (chunks < 1) chunks = 1;
    const int tpc = (ntiles + chunks - 1) / chunks;
    chunks = (ntiles + tpc - 1) / tpc;
    const dim3 grid(kw, chunks);
    const int ms = (M + KS - 1) / KS;
#define FC_DWA(MSv) hipLaunchKernelGGL((fc_dw_adam_kernel<T, MSv>), grid, dim3(256), 0, s, x, dy, w, m, v, M, K, N, tpc, \
                                       (const FcCtrlView*)ctrl, b1, b2, eps, gmult)
    if (ms <= 1) FC_DWA(1);
    else if (ms <= 2) FC_DWA(2);
    else if (ms <= 4) FC_DWA(4);
    else if (ms <= 8) FC_DWA(8);
    else FC_DWA(16);
#undef FC_DWA
    FCHK(hipGetLastError());
    return Y2_OK;
}
}  // namespace y2

using namespace y2;
extern "C" {
int y2_fully_connected(const float* x, const float* w, const float* bias, float* y, int rows, int in_features,
                       int out_features, int relu, int dtype, void* stream) {
    if (!x || !w || !y) return fc_fail(Y2_ERR_ARG, "fully connected: null tensor");
    if (rows < 1 || rows > 128) return fc_fail(Y2_ERR_ARG, "fully connected: 1..128 rows (the batch) per call");
    if (in_features < 1 || out_features < 1) return fc_fail(Y2_ERR_ARG, "fully connected: bad shape");
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case 0: return fc_forward_T<float>(x, w, bias, y, rows, in_features, out_features, relu, s);
        case 1: return fc_forward_T<half_t>(x, w, bias, y, rows, in_features, out_features, relu, s);
        case 2: return fc_forward_T<bf16_t>(x, w, bias, y, rows, in_features, out_features, relu, s);
    }
    return fc_fail(Y2_ERR_ARG, "bad dtype");
}
int y2_fc_adam_apply_guarded(const float* x, const float* dz, float* w, float* m, float* v, int rows, int in_features,
                             int out_features, int dtype, const void* ctrl, float beta1, float beta2, float eps,
                             float grad_mult, void* stream) {
    if (!x || !dz || !w || !m || !v || !ctrl) return fc_fail(Y2_ERR_ARG, "fully connected: null tensor");
    // rows: the batch -- or, data parallel (round 6), the batches of every replica gathered (tf_resnet.py): the kernel keeps up
    // to 16 row fragments of x in registers, 16 rows each in the 16-bit types, 8 in fp32
    const int max_rows = dtype == 0 ? 128 : 256;
    if (rows < 1 || rows > max_rows) return fc_fail(Y2_ERR_ARG, "fully connected: 1..256 rows per call (128 in fp32)");
    if (in_features < 1 || out_features < 1) return fc_fail(Y2_ERR_ARG, "fully connected: bad shape");
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case 0: return fc_adam_T<float>(x, dz, w, m, v, rows, in_features, out_features, ctrl, beta1, beta2, eps, grad_mult, s);
        case 1: return fc_adam_T<half_t>(x, dz, w, m, v, rows, in_features, out_features, ctrl, beta1, beta2, eps, grad_mult, s);
        case 2: return fc_adam_T<bf16_t>(x, dz, w, m, v, rows, in_features, out_features, ctrl, beta1, beta2, eps, grad_mult, s);
    }
    return fc_fail(Y2_ERR_ARG, "bad dtype");
}
int y2_fully_connected_backward(const float* x, const float* w, const float* dy, float* dx, float* dw, int rows,
                                int in_features, int out_features, int dtype, void* stream) {
    if (!x || !w || !dy) return fc_fail(Y2_ERR_ARG, "fully connected: null tensor");
    if (rows < 1 || rows > 128) return fc_fail(Y2_ERR_ARG, "fully connected: 1..128 rows (the batch) per call");
    if (in_features < 1 || out_features < 1) return fc_fail(Y2_ERR_ARG, "fully connected: bad shape");
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case 0: return fc_backward_T<float>(x, w, dy, dx, dw, rows, in_features, out_features, s);
        case 1: return fc_backward_T<half_t>(x, w, dy, dx, dw, rows, in_features, out_features, s);
        case 2: return fc_backward_T<bf16_t>(x, w, dy, dx, dw, rows, in_features, out_features, s);
    }
    return fc_fail(Y2_ERR_ARG, "bad dtype");
}
}  // extern "C"
