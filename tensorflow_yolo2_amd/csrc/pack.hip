// Layout/precision conversion passes between the reference's tensors
// (fp32 NHWC activations, fp32 HWIO filters -- darknet.py:10-21) and the
// MI355X-side layouts (zero-bordered NHWC of T, K-contiguous packed filters).
#include "common.h"
#include "kernels.h"
#include "optim_math.h"

namespace y2 {

// image fp32 [N][H][W][3] -> x4 [N][H+2][W+2][4] of T; border pre-zeroed.  Channel 3 = 1 inside the image: its
// filter weights are zero (pack_conv1_kernel), so the convolution ignores it, and the Gram matrix of the input
// patches (conv1_wgrad.hip, linear form) gets its "sum of x over the valid taps" row from it for free
template <typename T>
__global__ void pack_input_kernel(const float* __restrict__ img, T* __restrict__ x4, int N, int H, int W) {
    const size_t total = (size_t)N * H * W;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
        const int w = (int)(p % W);
        const int h = (int)((p / W) % H);
        const int n = (int)(p / ((size_t)W * H));
        const float* s = img + p * 3;
        T* d = x4 + bpix(n, h, w, H, W) * 4;
        d[0] = Elem<T>::from_f32(s[0]);
        d[1] = Elem<T>::from_f32(s[1]);
        d[2] = Elem<T>::from_f32(s[2]);
        d[3] = Elem<T>::from_f32(1.f);
    }
}
template <typename T>
static hipError_t pack_input_T(const float* img, void* x4, int N, int H, int W, hipStream_t s) {
    size_t total = (size_t)N * H * W;
    size_t nb = (total + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(pack_input_kernel<T>, dim3((unsigned)nb), dim3(256), 0, s, img, (T*)x4, N, H, W);
    return hipGetLastError();
}
// uint8 pixels -> the same 4-channel bordered image: image.astype(np.float32) / 255.0 * 2.0 - 1.0 in fp32, in the
// reference's operation order (src/img_dataset/pascal_voc.py:63-64; the division is IEEE-exact, x * 2 is exact, so
// the result has the bits numpy produces).  Four pixels (12 bytes) per thread: three aligned dword loads.
Y2_DEV float u8_to_unit(uint32_t b) {
    float v = (float)b;
    v = __fdiv_rn(v, 255.0f);
    v = __fmul_rn(v, 2.0f);
    return __fsub_rn(v, 1.0f);
}
template <typename T>
__global__ void pack_input_u8_kernel(const uint8_t* __restrict__ img, T* __restrict__ x4, int N, int H, int W) {
    const size_t total = (size_t)N * H * W;
    const size_t quads = total / 4;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (size_t)gridDim.x * blockDim.x) {
        const uint32_t* s = (const uint32_t*)(img + q * 12);
        const uint32_t w0 = s[0], w1 = s[1], w2 = s[2];
        const uint32_t by[12] = {w0 & 255u, (w0 >> 8) & 255u, (w0 >> 16) & 255u, w0 >> 24,
                                 w1 & 255u, (w1 >> 8) & 255u, (w1 >> 16) & 255u, w1 >> 24,
                                 w2 & 255u, (w2 >> 8) & 255u, (w2 >> 16) & 255u, w2 >> 24};
        size_t p = q * 4;
        int w = (int)(p % W);
        int h = (int)((p / W) % H);
        int n = (int)(p / ((size_t)W * H));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            T* d = x4 + bpix(n, h, w, H, W) * 4;
            T o[4] = {Elem<T>::from_f32(u8_to_unit(by[3 * j])), Elem<T>::from_f32(u8_to_unit(by[3 * j + 1])),
                      Elem<T>::from_f32(u8_to_unit(by[3 * j + 2])), Elem<T>::from_f32(1.f)};
            if (sizeof(T) == 2) *(u32x2*)d = *(const u32x2*)o;
            else *(u32x4*)d = *(const u32x4*)o;
            if (++w == W) { w = 0; if (++h == H) { h = 0; ++n; } }
        }
    }
    // the 0..3 pixels behind the last whole quad
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)(total - quads * 4)) {
        const size_t p = quads * 4 + threadIdx.x;
        const int w = (int)(p % W), h = (int)((p / W) % H), n = (int)(p / ((size_t)W * H));
        const uint8_t* s = img + p * 3;
        T* d = x4 + bpix(n, h, w, H, W) * 4;
        d[0] = Elem<T>::from_f32(u8_to_unit(s[0]));
        d[1] = Elem<T>::from_f32(u8_to_unit(s[1]));
        d[2] = Elem<T>::from_f32(u8_to_unit(s[2]));
        d[3] = Elem<T>::from_f32(1.f);
    }
}
template <typename T>
static hipError_t pack_input_u8_T(const uint8_t* img, void* x4, int N, int H, int W, hipStream_t s) {
    size_t quads = (size_t)N * H * W / 4;
    size_t nb = (quads + 255) / 256;
    if (nb > 8192) nb = 8192;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(pack_input_u8_kernel<T>, dim3((unsigned)nb), dim3(256), 0, s, img, (T*)x4, N, H, W);
    return hipGetLastError();
}
hipError_t launch_pack_input_u8(int dtype, const uint8_t* img, void* x4, int N, int H, int W, hipStream_t s) {
    if (((uintptr_t)img & 3) != 0) return hipErrorInvalidValue;   // dword loads
    switch (dtype) {
        case 0: return pack_input_u8_T<float>(img, x4, N, H, W, s);
        case 1: return pack_input_u8_T<half_t>(img, x4, N, H, W, s);
        case 2: return pack_input_u8_T<bf16_t>(img, x4, N, H, W, s);
        case 3: return pack_input_u8_T<float>(img, x4, N, H, W, s);     // f16x2: the 3-channel layer runs in exact fp32
    }
    return hipErrorInvalidValue;
}

hipError_t launch_pack_input(int dtype, const float* img, void* x4, int N, int H, int W, hipStream_t s) {
    switch (dtype) {
        case 0: return pack_input_T<float>(img, x4, N, H, W, s);
        case 1: return pack_input_T<half_t>(img, x4, N, H, W, s);
        case 2: return pack_input_T<bf16_t>(img, x4, N, H, W, s);
        case 3: return pack_input_T<float>(img, x4, N, H, W, s);
    }
    return hipErrorInvalidValue;
}

// generic fp32 NHWC [N][H][W][C] <-> zero-bordered [N][H+2][W+2][C] of T (op-level API)
// C = channels of the fp32 tensor, Cs = channel stride of the bordered tensor (>= C, extra = 0)
template <typename T, bool PACK>
__global__ void act_pack_kernel(const float* in, T* xp, float* out, int N, int H, int W, int C, int Cs) {
    const int Ci = PACK ? Cs : C;
    const size_t total = (size_t)N * H * W * Ci;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Ci);
        const size_t p = i / Ci;
        const int w = (int)(p % W);
        const int h = (int)((p / W) % H);
        const int n = (int)(p / ((size_t)W * H));
        const size_t po = bpix(n, h, w, H, W) * Cs + c;
        if (PACK) xp[po] = Elem<T>::from_f32(c < C ? in[p * C + c] : 0.f);
        else out[i] = Elem<T>::to_f32(xp[po]);
    }
}
// the same with eight channels per thread (C, Cs multiples of 8, 16-byte aligned tensors): two 16-byte loads, one
// 16-byte store of T (two for fp32); the scalar form spends its time on the per-element index arithmetic
template <typename T, bool PACK>
__global__ void act_pack8_kernel(const float* __restrict__ in, T* __restrict__ xp, float* __restrict__ out, int N, int H,
                                 int W, int C, int Cs) {
    const int G = (PACK ? Cs : C) / 8;
    const size_t total = (size_t)N * H * W * G;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % G) * 8;
        const size_t p = i / G;
        const int w = (int)(p % W);
        const int h = (int)((p / W) % H);
        const int n = (int)(p / ((size_t)W * H));
        T* q = xp + bpix(n, h, w, H, W) * Cs + c;
        if (PACK) {
            float v[8];
            if (c < C) {
                const float4 a = *(const float4*)(in + p * C + c), b = *(const float4*)(in + p * C + c + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;
            }
            T t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = Elem<T>::from_f32(v[e]);
#pragma unroll
            for (int e = 0; e < 8 * (int)sizeof(T) / 16; ++e) ((u32x4*)q)[e] = ((const u32x4*)t)[e];
        } else {
            T t[8];
#pragma unroll
            for (int e = 0; e < 8 * (int)sizeof(T) / 16; ++e) ((u32x4*)t)[e] = ((const u32x4*)q)[e];
            float* o = out + p * C + c;
            *(float4*)o = make_float4(Elem<T>::to_f32(t[0]), Elem<T>::to_f32(t[1]), Elem<T>::to_f32(t[2]), Elem<T>::to_f32(t[3]));
            *(float4*)(o + 4) = make_float4(Elem<T>::to_f32(t[4]), Elem<T>::to_f32(t[5]), Elem<T>::to_f32(t[6]), Elem<T>::to_f32(t[7]));
        }
    }
}
// f16x2 mode: the bordered tensor is split (cell = [Cs halves hi][Cs halves lo], common.h hsplit_t).  Op-level entries
// and debug reads only (the network's own tensors are written split by the batch-norm passes): one element per thread
template <bool PACK>
__global__ void act_pack_split_kernel(const float* in, char* xp, float* out, int N, int H, int W, int C, int Cs) {
    const int Ci = PACK ? Cs : C;
    const size_t total = (size_t)N * H * W * Ci;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Ci);
        const size_t p = i / Ci;
        const int w = (int)(p % W);
        const int h = (int)((p / W) % H);
        const int n = (int)(p / ((size_t)W * H));
        half_t* cell = (half_t*)(xp + bpix(n, h, w, H, W) * (size_t)Cs * 4);
        if (PACK) {
            half_t hi, lo;
            split_f16(c < C ? in[p * C + c] : 0.f, hi, lo);
            cell[c] = hi;
            cell[Cs + c] = lo;
        } else {
            out[i] = (float)cell[c] + (float)cell[Cs + c];
        }
    }
}
template <bool PACK>
static hipError_t act_pack_split(const float* in, void* xp, float* out, int N, int H, int W, int C, int Cs, hipStream_t s) {
    size_t total = (size_t)N * H * W * (PACK ? Cs : C);
    size_t nb = (total + 255) / 256;
    if (nb > 65536) nb = 65536;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL((act_pack_split_kernel<PACK>), dim3((unsigned)nb), dim3(256), 0, s, in, (char*)xp, out, N, H, W, C, Cs);
    return hipGetLastError();
}

template <typename T, bool PACK>
static hipError_t act_pack_T(const float* in, void* xp, float* out, int N, int H, int W, int C, int Cs,
                             hipStream_t s) {
    if (C % 8 == 0 && Cs % 8 == 0 && (((uintptr_t)in | (uintptr_t)xp | (uintptr_t)out) & 15) == 0) {
        size_t total8 = (size_t)N * H * W * ((PACK ? Cs : C) / 8);
        size_t nb8 = (total8 + 255) / 256;
        if (nb8 > 16384) nb8 = 16384;
        if (nb8 < 1) nb8 = 1;
        hipLaunchKernelGGL((act_pack8_kernel<T, PACK>), dim3((unsigned)nb8), dim3(256), 0, s, in, (T*)xp, out, N, H, W, C,
                           Cs);
        return hipGetLastError();
    }
    size_t total = (size_t)N * H * W * (PACK ? Cs : C);
    size_t nb = (total + 255) / 256;
    if (nb > 16384) nb = 16384;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL((act_pack_kernel<T, PACK>), dim3((unsigned)nb), dim3(256), 0, s, in, (T*)xp, out, N, H, W, C,
                       Cs);
    return hipGetLastError();
}
hipError_t launch_pack_act(int dtype, const float* in, void* xp, int N, int H, int W, int C, int Cs,
                           hipStream_t s) {
    switch (dtype) {
        case 0: return act_pack_T<float, true>(in, xp, nullptr, N, H, W, C, Cs, s);
        case 1: return act_pack_T<half_t, true>(in, xp, nullptr, N, H, W, C, Cs, s);
        case 2: return act_pack_T<bf16_t, true>(in, xp, nullptr, N, H, W, C, Cs, s);
        case 3: return act_pack_split<true>(in, xp, nullptr, N, H, W, C, Cs, s);
    }
    return hipErrorInvalidValue;
}
// The WHOLE bordered tensor of the op-level entries in one pass: front guard, zero borders, body, back guard and the
// padding channels, one 16-byte chunk per thread -- no memset in front of it.  region = first byte of the allocation
// (front_px cells before cell 0 of the bordered space), region_bytes a multiple of 16.
template <typename T>
__global__ void act_pack_region_kernel(const float* __restrict__ in, char* __restrict__ region, size_t chunks,
                                       size_t front_px, int N, int H, int W, int C, int Cs) {
    constexpr int EPC = 16 / sizeof(T);
    const int cpc = Cs / EPC;                       // chunks per cell
    const size_t pitch = (size_t)W + 1, rows_img = (size_t)H + 1;
    const size_t body = (size_t)N * rows_img * pitch;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += (size_t)gridDim.x * blockDim.x) {
        const size_t cell = i / cpc;
        const int c = (int)(i % cpc) * EPC;
        T t[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) t[e] = Elem<T>::from_f32(0.f);
        if (cell >= front_px && cell - front_px < body && c < C) {
            const size_t q = cell - front_px;
            const size_t row = q / pitch;
            const int wc = (int)(q - row * pitch);
            const int hr = (int)(row % rows_img);
            if (hr >= 1 && wc >= 1) {
                const size_t n = row / rows_img;
                const float* src = in + ((n * H + (hr - 1)) * W + (wc - 1)) * C + c;
#pragma unroll
                for (int e = 0; e < EPC; e += 4) {
                    const float4 v = *(const float4*)(src + e);
                    t[e] = Elem<T>::from_f32(v.x); t[e + 1] = Elem<T>::from_f32(v.y);
                    t[e + 2] = Elem<T>::from_f32(v.z); t[e + 3] = Elem<T>::from_f32(v.w);
                }
            }
        }
        *(u32x4*)(region + i * 16) = *(const u32x4*)t;
    }
}
hipError_t launch_pack_act_region(int dtype, const float* in, void* region, size_t region_bytes, size_t front_px, int N,
                                  int H, int W, int C, int Cs, hipStream_t s) {
    if (dtype_split(dtype)) return hipErrorNotSupported;     // the caller zeroes the allocation and packs the body
    const int epc = dtype == 0 ? 4 : 8;
    if (C % epc || Cs % epc || region_bytes % 16 || (((uintptr_t)in | (uintptr_t)region) & 15)) return hipErrorNotSupported;
    const size_t chunks = region_bytes / 16;
    size_t nb = (chunks + 255) / 256;
    if (nb > 16384) nb = 16384;
    dim3 g((unsigned)nb), b(256);
    switch (dtype) {
        case 0: hipLaunchKernelGGL(act_pack_region_kernel<float>, g, b, 0, s, in, (char*)region, chunks, front_px, N, H, W, C, Cs); break;
        case 1: hipLaunchKernelGGL(act_pack_region_kernel<half_t>, g, b, 0, s, in, (char*)region, chunks, front_px, N, H, W, C, Cs); break;
        case 2: hipLaunchKernelGGL(act_pack_region_kernel<bf16_t>, g, b, 0, s, in, (char*)region, chunks, front_px, N, H, W, C, Cs); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
hipError_t launch_unpack_act(int dtype, const void* xp, float* out, int N, int H, int W, int C, int Cs,
                             hipStream_t s) {
    switch (dtype) {
        case 0: return act_pack_T<float, false>(nullptr, (void*)xp, out, N, H, W, C, Cs, s);
        case 1: return act_pack_T<half_t, false>(nullptr, (void*)xp, out, N, H, W, C, Cs, s);
        case 2: return act_pack_T<bf16_t, false>(nullptr, (void*)xp, out, N, H, W, C, Cs, s);
        case 3: return act_pack_split<false>(nullptr, (void*)xp, out, N, H, W, C, Cs, s);
    }
    return hipErrorInvalidValue;
}

// [rows][lds] of T -> fp32 [rows][C]
template <typename T>
__global__ void cast_f32_kernel(const T* src, float* dst, size_t rows, int C, int lds, float scale) {
    const size_t total = rows * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / C;
        const int c = (int)(i % C);
        const float v = Elem<T>::to_f32(src[r * lds + c]);
        dst[i] = scale == 1.0f ? v : v * scale;
    }
}
template <typename T>
__global__ void cast_f32x8_kernel(const T* __restrict__ src, float* __restrict__ dst, size_t rows, int C, int lds,
                                  float scale) {
    const int G = C / 8;
    const size_t total = rows * G;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / G;
        const int c = (int)(i % G) * 8;
        T t[8];
#pragma unroll
        for (int e = 0; e < 8 * (int)sizeof(T) / 16; ++e) ((u32x4*)t)[e] = ((const u32x4*)(src + r * lds + c))[e];
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[e] = Elem<T>::to_f32(t[e]);
            if (scale != 1.0f) v[e] *= scale;
        }
        float* o = dst + r * C + c;
        *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}
hipError_t launch_cast_to_f32(int dtype, const void* src, float* dst, size_t rows, int C, int lds, hipStream_t s,
                              float scale) {
    dtype = dtype_plain(dtype);      // f16x2: conv outputs and gradients wrt activations are fp32
    if (C % 8 == 0 && lds % 8 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        size_t nb8 = (rows * (C / 8) + 255) / 256;
        if (nb8 > 16384) nb8 = 16384;
        if (nb8 < 1) nb8 = 1;
        dim3 g8((unsigned)nb8), b8(256);
        switch (dtype) {
            case 0: hipLaunchKernelGGL(cast_f32x8_kernel<float>, g8, b8, 0, s, (const float*)src, dst, rows, C, lds, scale); break;
            case 1: hipLaunchKernelGGL(cast_f32x8_kernel<half_t>, g8, b8, 0, s, (const half_t*)src, dst, rows, C, lds, scale); break;
            case 2: hipLaunchKernelGGL(cast_f32x8_kernel<bf16_t>, g8, b8, 0, s, (const bf16_t*)src, dst, rows, C, lds, scale); break;
            default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    size_t total = rows * C;
    size_t nb = (total + 255) / 256;
    if (nb > 16384) nb = 16384;
    if (nb < 1) nb = 1;
    dim3 g((unsigned)nb), b(256);
    switch (dtype) {
        case 0: hipLaunchKernelGGL(cast_f32_kernel<float>, g, b, 0, s, (const float*)src, dst, rows, C, lds, scale); break;
        case 1: hipLaunchKernelGGL(cast_f32_kernel<half_t>, g, b, 0, s, (const half_t*)src, dst, rows, C, lds, scale); break;
        case 2: hipLaunchKernelGGL(cast_f32_kernel<bf16_t>, g, b, 0, s, (const bf16_t*)src, dst, rows, C, lds, scale); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// fp32 [M][C] * scale -> T [M][ldd] (columns >= C zero)
template <typename T>
__global__ void convert_grad_kernel(const float* src, T* dst, int M, int C, int ldd, float scale) {
    const size_t total = (size_t)M * ldd;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / ldd;
        const int c = (int)(i % ldd);
        dst[i] = Elem<T>::from_f32(c < C ? src[r * C + c] * scale : 0.f);
    }
}
hipError_t launch_convert_grad(int dtype, const float* src, void* dst, int M, int C, int ldd, float scale,
                               hipStream_t s) {
    dtype = dtype_plain(dtype);
    size_t total = (size_t)M * ldd;
    size_t nb = (total + 255) / 256;
    if (nb > 16384) nb = 16384;
    if (nb < 1) nb = 1;
    dim3 g((unsigned)nb), b(256);
    switch (dtype) {
        case 0: hipLaunchKernelGGL(convert_grad_kernel<float>, g, b, 0, s, src, (float*)dst, M, C, ldd, scale); break;
        case 1: hipLaunchKernelGGL(convert_grad_kernel<half_t>, g, b, 0, s, src, (half_t*)dst, M, C, ldd, scale); break;
        case 2: hipLaunchKernelGGL(convert_grad_kernel<bf16_t>, g, b, 0, s, src, (bf16_t*)dst, M, C, ldd, scale); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// filters: W fp32 HWIO [taps][Cin][Cout]
//   wf[co][t][ci]  (rows co in [0,Cout_pad), zero beyond Cout)           forward
//   wd[ci][t][co]  = W[taps-1-t][ci][co]  (rows ci in [0,Cin_pad), cols co in [0,Cdy)) dgrad
// Tiled transpose through LDS so both the fp32 reads and the T writes coalesce.
// ---------------------------------------------------------------------------
// MFMA-fragment order (conv_haloq.hip): [row tile of 32][tap][k-group of 32 bytes][lane = hh*32 + row%32][16 B];
// returns the 16-byte chunk index of element (row, tap, k), k a multiple of EPC
__host__ __device__ inline size_t frag_chunk(int row, int t, int k, int taps, int krow, int EPC) {
    const int kgrow = krow / (2 * EPC);
    const int kg = k / (2 * EPC), hh = (k / EPC) & 1;
    return ((size_t)(row >> 5) * taps * kgrow + (size_t)t * kgrow + kg) * 64 + hh * 32 + (row & 31);
}

// 16-row form (conv_haloq16): [row tile of 16][tap][k-group of 64 bytes][lane = (16-byte chunk)*16 + row%16][16 B]
__host__ __device__ inline size_t frag_chunk16(int row, int t, int k, int taps, int krow, int EPC) {
    const int kgrow = krow / (4 * EPC);
    const int kg = k / (4 * EPC), kc = (k / EPC) & 3;
    return ((size_t)(row >> 4) * taps * kgrow + (size_t)t * kgrow + kg) * 64 + kc * 16 + (row & 15);
}
__host__ __device__ inline size_t frag_chunk_any(int layout, int row, int t, int k, int taps, int krow, int EPC) {
    return layout == 2 ? frag_chunk16(row, t, k, taps, krow, EPC) : frag_chunk(row, t, k, taps, krow, EPC);
}

// SPLIT (f16x2 mode, T = half_t): a packed row holds two planes per tap, [Kc halves hi][Kc halves lo] of W * kSplitWScale
template <typename T, bool SPLIT = false>
__global__ __launch_bounds__(256) void pack_wf_kernel(const float* __restrict__ W, T* __restrict__ wf, int taps,
                                                      int Cin, int Cout, int Cout_pad, int Kc, int frag) {
    constexpr int EPC = 16 / sizeof(T);
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int ci0 = blockIdx.y * 32, co0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        tile[r][tx] = (ci < Cin && co < Cout) ? W[((size_t)t * Cin + ci) * Cout + co] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        if (co < Cout_pad && ci < Kc) {
            if constexpr (SPLIT) {
                half_t hl[2];
                split_f16(tile[tx][r] * kSplitWScale, hl[0], hl[1]);
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const int k = pl * Kc + ci;
                    const size_t o = frag ? frag_chunk_any(frag, co, t, k - k % EPC, taps, 2 * Kc, EPC) * EPC + k % EPC
                                          : ((size_t)co * taps + t) * 2 * Kc + k;
                    wf[o] = hl[pl];
                }
            } else {
                const size_t o = frag ? frag_chunk_any(frag, co, t, ci - ci % EPC, taps, Kc, EPC) * EPC + ci % EPC
                                      : ((size_t)co * taps + t) * Kc + ci;
                wf[o] = Elem<T>::from_f32(tile[tx][r]);
            }
        }
    }
}
template <typename T, bool SPLIT = false>
__global__ void pack_wd_kernel(const float* __restrict__ W, T* __restrict__ wd, int taps, int Cin, int Cout,
                               int Cin_pad, int Cdy, int frag) {
    constexpr int EPC = 16 / sizeof(T);
    const size_t total = (size_t)Cin_pad * taps * Cdy;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % Cdy);
        const int t = (int)((i / Cdy) % taps);
        const int ci = (int)(i / ((size_t)Cdy * taps));
        float v = 0.f;
        if (ci < Cin && co < Cout) v = W[((size_t)(taps - 1 - t) * Cin + ci) * Cout + co];
        if constexpr (SPLIT) {
            half_t hl[2];
            split_f16(v * kSplitWScale, hl[0], hl[1]);
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const int k = pl * Cdy + co;
                const size_t o = frag ? frag_chunk_any(frag, ci, t, k - k % EPC, taps, 2 * Cdy, EPC) * EPC + k % EPC
                                      : ((size_t)ci * taps + t) * 2 * Cdy + k;
                wd[o] = hl[pl];
            }
        } else {
            const size_t o = frag ? frag_chunk_any(frag, ci, t, co - co % EPC, taps, Cdy, EPC) * EPC + co % EPC : i;
            wd[o] = Elem<T>::from_f32(v);
        }
    }
}
template <typename T, bool SPLIT = false>
static hipError_t pack_weights_T(const float* W, void* wf, void* wd, int taps, int Cin, int Cout, int Cout_pad,
                                 int Kc, int Cin_pad, int Cdy, int frag, hipStream_t s) {
    if (wf) {
        dim3 g((Cout_pad + 31) / 32, (Kc + 31) / 32, taps);
        hipLaunchKernelGGL((pack_wf_kernel<T, SPLIT>), g, dim3(256), 0, s, W, (T*)wf, taps, Cin, Cout, Cout_pad, Kc, frag);
    }
    if (wd) {
        size_t total = (size_t)Cin_pad * taps * Cdy;
        size_t nb = (total + 255) / 256;
        if (nb > 16384) nb = 16384;
        hipLaunchKernelGGL((pack_wd_kernel<T, SPLIT>), dim3((unsigned)nb), dim3(256), 0, s, W, (T*)wd, taps, Cin, Cout, Cin_pad,
                           Cdy, frag);
    }
    return hipGetLastError();
}
hipError_t launch_pack_weights(int dtype, const float* W, void* wf, void* wd, int taps, int Cin, int Cout,
                               int Cout_pad, int Kc, int Cin_pad, int Cdy, int frag, hipStream_t s) {
    switch (dtype) {
        case 0: return pack_weights_T<float>(W, wf, wd, taps, Cin, Cout, Cout_pad, Kc, Cin_pad, Cdy, frag, s);
        case 1: return pack_weights_T<half_t>(W, wf, wd, taps, Cin, Cout, Cout_pad, Kc, Cin_pad, Cdy, frag, s);
        case 2: return pack_weights_T<bf16_t>(W, wf, wd, taps, Cin, Cout, Cout_pad, Kc, Cin_pad, Cdy, frag, s);
        case 3: return pack_weights_T<half_t, true>(W, wf, wd, taps, Cin, Cout, Cout_pad, Kc, Cin_pad, Cdy, frag, s);
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------
// every layer's filters in ONE launch (the per-layer launches were latency-bound:
// 2 x 21 kernels of ~8 us per step).  A block looks its layer up in a small table.
// ---------------------------------------------------------------------------
// one 16-byte chunk (EPC filter values v[], k = first K index of the chunk inside its tap) of packed row `row`, tap t, of
// a pack whose rows hold krow K elements per tap.  layout: 0 K-contiguous rows, 1 / 2 fragment order.
// SPLIT (f16x2 mode, T = half_t): two chunks -- the halves of v * kSplitWScale into the hi plane (k) and the lo plane
// (krow + k) of a row that holds 2 * krow halves per tap
template <typename T, bool SPLIT>
Y2_DEV void st_filter_chunk(T* base, int layout, int row, int t, int k, int taps, int krow, const float* v) {
    constexpr int EPC = 16 / sizeof(T);
    if constexpr (SPLIT) {
        Chunk<T> hi, lo;
#pragma unroll
        for (int e = 0; e < EPC; ++e) split_f16(v[e] * kSplitWScale, hi.v[e], lo.v[e]);
        const int kr2 = 2 * krow;
        if (layout) {
            st_chunk<T>(base + frag_chunk_any(layout, row, t, k, taps, kr2, EPC) * EPC, hi);
            st_chunk<T>(base + frag_chunk_any(layout, row, t, krow + k, taps, kr2, EPC) * EPC, lo);
        } else {
            T* q = base + ((size_t)row * taps + t) * kr2 + k;
            st_chunk<T>(q, hi);
            st_chunk<T>(q + krow, lo);
        }
    } else {
        Chunk<T> o;
#pragma unroll
        for (int e = 0; e < EPC; ++e) o.v[e] = Elem<T>::from_f32(v[e]);
        if (layout) st_chunk<T>(base + frag_chunk_any(layout, row, t, k, taps, krow, EPC) * EPC, o);
        else st_chunk<T>(base + ((size_t)row * taps + t) * krow + k, o);
    }
}

template <typename T, bool SPLIT = false>
__global__ __launch_bounds__(256) void pack_all_kernel(const PackLayer* __restrict__ tab, int nlayers) {
    constexpr int EPC = 16 / sizeof(T);       // elements per 16-byte store
    constexpr int CPR = 64 / EPC;             // chunks per 64-element tile row
    constexpr int RPP = 256 / CPR;            // tile rows per pass of the block
    __shared__ float tile[64][65];
    int l = 0;
    const int b = blockIdx.x;
    while (l + 1 < nlayers && b >= tab[l + 1].first_block) ++l;
    const PackLayer L = tab[l];
    const int local = b - L.first_block;
    const float* __restrict__ W = L.W;
    const int tid = threadIdx.x;
    if (local < L.wf_blocks) {
        // wf[co][t][ci] <- W[t][ci][co]: 64 x 64 transpose through LDS, 256-byte rows in, 16-byte chunks out
        T* __restrict__ wf = (T*)L.wf;
        const int bx = local % L.wf_bx, by = (local / L.wf_bx) % L.wf_by, t = local / (L.wf_bx * L.wf_by);
        const int ci0 = by * 64, co0 = bx * 64;
        const int tx = tid & 63, ty = tid >> 6;
        for (int r = ty; r < 64; r += 4) {
            const int ci = ci0 + r, co = co0 + tx;
            tile[r][tx] = (ci < L.Cin && co < L.Cout) ? W[((size_t)t * L.Cin + ci) * L.Cout + co] : 0.f;
        }
        __syncthreads();
        const int cs = tid % CPR, cr = tid / CPR;
#pragma unroll
        for (int p = 0; p < 64 / RPP; ++p) {
            const int col = p * RPP + cr;
            const int co = co0 + col, ci = ci0 + cs * EPC;
            if (co < L.Cout_pad && ci < L.Kc) {
                float o[EPC];
#pragma unroll
                for (int k = 0; k < EPC; ++k) o[k] = tile[cs * EPC + k][col];
                st_filter_chunk<T, SPLIT>(wf, L.wf_frag, co, t, ci, L.taps, L.Kc, o);
            }
        }
    } else if (L.wd) {
        // wd[ci][t][co] <- W[taps-1-t][ci][co]: rows stay co-contiguous, 16-byte chunks both ways
        T* __restrict__ wd = (T*)L.wd;
        const int cpr = L.Cdy / EPC;                                    // chunks per row
        const uint32_t total = (uint32_t)L.Cin_pad * L.taps * cpr;
        const uint32_t i0 = (uint32_t)(local - L.wf_blocks) * 1024u;
        const bool vec = (L.Cout % 4) == 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = i0 + k * 256 + tid;
            if (i < total) {
                const int cc = (int)(i % (uint32_t)cpr);
                const uint32_t q = i / (uint32_t)cpr;
                const int tt = (int)(q % (uint32_t)L.taps), ci = (int)(q / (uint32_t)L.taps);
                const int co = cc * EPC;
                float o[EPC];
                const float* src = W + ((size_t)(L.taps - 1 - tt) * L.Cin + ci) * L.Cout + co;
                if (ci < L.Cin && vec && co + EPC <= L.Cout) {
#pragma unroll
                    for (int e4 = 0; e4 < EPC; e4 += 4) {
                        const f32x4 v = *(const f32x4*)(src + e4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[e4 + j] = v[j];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < EPC; ++e) o[e] = (ci < L.Cin && co + e < L.Cout) ? src[e] : 0.f;
                }
                st_filter_chunk<T, SPLIT>(wd, L.wd_frag, ci, tt, co, L.taps, L.Cdy, o);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Optimizer update fused with the filter re-pack (kernels.h: OptPackArgs).  One block = one 64 (ci) x 64 (co)
// tile of one tap of one layer: p, slot(s), g in (16-byte loads, 256-byte rows), update, p / slots out, and the
// updated tile leaves a second and third time in the MFMA operand type: co-contiguous rows to the dgrad copy
// (tap-flipped), and through a 64 x 64 LDS transpose to the forward copy.  Blocks past the tiles update the
// small ranges (b, gamma, beta, a 3-channel first filter) with the plain flat form.
// ---------------------------------------------------------------------------
struct OptCtrlView { int found_inf, step, skipped, reserved; float lr_t; };

template <int KIND>
Y2_DEV void opt_update(float& p, float& s0, float& s1, float g, float lr_t, float b1, float b2, float eps) {
    if (KIND == 0) adam_update(p, s0, s1, g, lr_t, b1, b2, eps);
    else momentum_update(p, s0, g, lr_t, b1);
}

template <typename T, int KIND, bool SPLIT = false>
__global__ __launch_bounds__(256) void opt_pack_kernel(OptPackArgs a) {
    constexpr int EPC = 16 / sizeof(T);
    constexpr int CPR = 64 / EPC;
    constexpr int RPP = 256 / CPR;
    float lr_t = a.lr_t;
    if (a.ctrl) {
        const OptCtrlView* c = (const OptCtrlView*)a.ctrl;
        if (c->found_inf) return;          // overflowed gradients: nothing moves, the packed copies stay valid
        if (KIND == 0) lr_t = c->lr_t;
    }
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    if (b >= a.tile_blocks) {
        // small ranges: grid-stride over the concatenation of the ranges
        const int nb = gridDim.x - a.tile_blocks, bi = b - a.tile_blocks;
        for (int r = 0; r < a.nsmall; ++r) {
            const unsigned off = a.small[2 * r], cnt = a.small[2 * r + 1];
            for (unsigned i = bi * 256 + tid; i < cnt; i += nb * 256) {
                float p = a.p[off + i], s0 = a.slot0[off + i], s1 = KIND == 0 ? a.slot1[off + i] : 0.f;
                opt_update<KIND>(p, s0, s1, a.g[off + i] * a.gmult, lr_t, a.b1, a.b2, a.eps);
                a.p[off + i] = p; a.slot0[off + i] = s0;
                if (KIND == 0) a.slot1[off + i] = s1;
            }
        }
        return;
    }
    __shared__ float tile[64][65];
    int l = 0;
    while (l + 1 < a.nlayers && b >= a.tab[l + 1].opt_first) ++l;
    const PackLayer L = a.tab[l];
    const int local = b - L.opt_first;
    const int bx = local % L.wf_bx, by = (local / L.wf_bx) % L.wf_by, t = local / (L.wf_bx * L.wf_by);
    const int ci0 = by * 64, co0 = bx * 64;
    // ---- update: thread = 4 consecutive co of one ci row per pass (16 lanes per 256-byte row, 16 rows per pass)
    const int c4 = (tid & 15) * 4, r0 = tid >> 4;
    const bool vec = (L.Cout & 3) == 0;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int r = pass * 16 + r0;
        const int ci = ci0 + r, co = co0 + c4;
        float pv[4] = {0.f, 0.f, 0.f, 0.f};
        if (ci < L.Cin && co < L.Cout) {
            const size_t o = L.w_off + ((size_t)t * L.Cin + ci) * L.Cout + co;
            if (vec) {
                f32x4 p = *(const f32x4*)(a.p + o), s0 = *(const f32x4*)(a.slot0 + o), g = *(const f32x4*)(a.g + o);
                f32x4 s1 = KIND == 0 ? *(const f32x4*)(a.slot1 + o) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float pk = p[k], ak = s0[k], bk = s1[k];
                    opt_update<KIND>(pk, ak, bk, g[k] * a.gmult, lr_t, a.b1, a.b2, a.eps);
                    p[k] = pk; s0[k] = ak; s1[k] = bk; pv[k] = pk;
                }
                *(f32x4*)(a.p + o) = p; *(f32x4*)(a.slot0 + o) = s0;
                if (KIND == 0) *(f32x4*)(a.slot1 + o) = s1;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (co + k < L.Cout) {
                        float pk = a.p[o + k], ak = a.slot0[o + k], bk = KIND == 0 ? a.slot1[o + k] : 0.f;
                        opt_update<KIND>(pk, ak, bk, a.g[o + k] * a.gmult, lr_t, a.b1, a.b2, a.eps);
                        a.p[o + k] = pk; a.slot0[o + k] = ak;
                        if (KIND == 0) a.slot1[o + k] = bk;
                        pv[k] = pk;
                    }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) tile[r][c4 + k] = pv[k];
    }
    __syncthreads();
    // ---- dgrad copy wd[ci][taps-1-t][co]: rows stay co-contiguous (16-byte chunks of EPC couts)
    if (L.wd) {
        T* __restrict__ wd = (T*)L.wd;
        const int tt = L.taps - 1 - t;
        const int cs = tid % CPR, cr = tid / CPR;
#pragma unroll
        for (int p = 0; p < 64 / RPP; ++p) {
            const int r = p * RPP + cr;
            const int ci = ci0 + r, co = co0 + cs * EPC;
            if (ci < L.Cin_pad && co < L.Cdy) {
                float o[EPC];
#pragma unroll
                for (int k = 0; k < EPC; ++k) o[k] = tile[r][cs * EPC + k];
                st_filter_chunk<T, SPLIT>(wd, L.wd_frag, ci, tt, co, L.taps, L.Cdy, o);
            }
        }
    }
    // ---- forward copy wf[co][t][ci]: the transpose
    {
        T* __restrict__ wf = (T*)L.wf;
        const int cs = tid % CPR, cr = tid / CPR;
#pragma unroll
        for (int p = 0; p < 64 / RPP; ++p) {
            const int col = p * RPP + cr;
            const int co = co0 + col, ci = ci0 + cs * EPC;
            if (co < L.Cout_pad && ci < L.Kc) {
                float o[EPC];
#pragma unroll
                for (int k = 0; k < EPC; ++k) o[k] = tile[cs * EPC + k][col];
                st_filter_chunk<T, SPLIT>(wf, L.wf_frag, co, t, ci, L.taps, L.Kc, o);
            }
        }
    }
}

hipError_t launch_opt_pack(int dtype, const OptPackArgs& a, hipStream_t s) {
    const int small_blocks = a.nsmall > 0 ? 64 : 0;
    dim3 g(a.tile_blocks + small_blocks), b(256);
    if (g.x == 0) return hipSuccess;
#define Y2_OP(T, K) hipLaunchKernelGGL((opt_pack_kernel<T, K>), g, b, 0, s, a)
    switch (dtype * 2 + a.kind) {
        case 0: Y2_OP(float, 0); break;
        case 1: Y2_OP(float, 1); break;
        case 2: Y2_OP(half_t, 0); break;
        case 3: Y2_OP(half_t, 1); break;
        case 4: Y2_OP(bf16_t, 0); break;
        case 5: Y2_OP(bf16_t, 1); break;
        case 6: hipLaunchKernelGGL((opt_pack_kernel<half_t, 0, true>), g, b, 0, s, a); break;     // f16x2
        case 7: hipLaunchKernelGGL((opt_pack_kernel<half_t, 1, true>), g, b, 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
#undef Y2_OP
    return hipGetLastError();
}

void pack_layer_plan(PackLayer& L, int first_block, int elem_size) {
    L.wf_bx = (L.Cout_pad + 63) / 64;
    L.wf_by = (L.Kc + 63) / 64;
    L.wf_blocks = L.wf ? L.wf_bx * L.wf_by * L.taps : 0;
    const size_t total = (size_t)L.Cin_pad * L.taps * L.Cdy / (16 / elem_size);   // 16-byte chunks, 1024 per block
    L.wd_blocks = L.wd ? (int)((total + 1023) / 1024) : 0;
    L.first_block = first_block;
}

hipError_t launch_pack_all(int dtype, const PackLayer* tab_dev, int nlayers, int total_blocks, hipStream_t s) {
    dim3 g(total_blocks), b(256);
    switch (dtype) {
        case 0: hipLaunchKernelGGL(pack_all_kernel<float>, g, b, 0, s, tab_dev, nlayers); break;
        case 1: hipLaunchKernelGGL(pack_all_kernel<half_t>, g, b, 0, s, tab_dev, nlayers); break;
        case 2: hipLaunchKernelGGL(pack_all_kernel<bf16_t>, g, b, 0, s, tab_dev, nlayers); break;
        case 3: hipLaunchKernelGGL((pack_all_kernel<half_t, true>), g, b, 0, s, tab_dev, nlayers); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// conv1: W [3][3][3][32] -> wp[co][kh][16]: element kw*4+c (c<3), zero elsewhere
template <typename T>
__global__ void pack_conv1_kernel(const float* W, T* wp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 32 * 3 * 16) return;
    const int e = i % 16, kh = (i / 16) % 3, co = i / 48;
    const int kw = e / 4, c = e % 4;
    float v = 0.f;
    if (kw < 3 && c < 3) v = W[((kh * 3 + kw) * 3 + c) * 32 + co];
    wp[i] = Elem<T>::from_f32(v);
}
hipError_t launch_pack_conv1_weights(int dtype, const float* W, void* wp, hipStream_t s) {
    dim3 g(6), b(256);
    dtype = dtype_plain(dtype);
    switch (dtype) {
        case 0: hipLaunchKernelGGL(pack_conv1_kernel<float>, g, b, 0, s, W, (float*)wp); break;
        case 1: hipLaunchKernelGGL(pack_conv1_kernel<half_t>, g, b, 0, s, W, (half_t*)wp); break;
        case 2: hipLaunchKernelGGL(pack_conv1_kernel<bf16_t>, g, b, 0, s, W, (bf16_t*)wp); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace y2
