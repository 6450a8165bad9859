// First layer: 3x3 SAME conv 3 -> 32 (reference darknet.py:150, Cin = 3).
// K = 27 is too small for the LDS-tiled implicit GEMM and the layer is
// HBM-bound (12 B in, 64 B out per pixel at fp16), so it gets its own kernel:
//   * input stored as zero-bordered NHWC with 4 channels: the three taps of one
//     filter row are 12 contiguous elements -> each MFMA B-fragment is ONE
//     16-byte global load straight into registers (no LDS staging)
//   * K padded 27 -> 3 x 16 (zero weights), D[cout][pixel] on 32x32 MFMA
//   * persistent waves; epilogue repacks through a wave-private LDS patch so the
//     stores are whole 64-byte pixel rows, 1 KiB contiguous per wave-instruction
//   * BN statistics as bias-shifted sums per wave, Chan-merged per block.
#include "common.h"
#include "kernels.h"

namespace y2 {

template <typename T>
__global__ __launch_bounds__(256) void conv1_fwd_kernel(Conv1Args a) {
    typedef typename Elem<T>::frag frag_t;
    // a pixel is 4*SZ bytes, so a fragment load is only 8-byte aligned at f16/bf16
    struct __attribute__((packed, aligned(8))) UFrag { frag_t v; };
    constexpr int SZ = sizeof(T);
    constexpr int KGC = 16 * SZ / 32;        // k-groups per filter row (1: f16/bf16, 2: f32)
    constexpr int EROW = 32 * SZ + 16;
    constexpr int EPC = 16 / SZ;
    constexpr int CPR = 32 / EPC;            // chunks per pixel row
    constexpr int RPIe = 64 / CPR;
    constexpr int NIT = 32 / RPIe;
    __shared__ __attribute__((aligned(16))) char smem[4 * 32 * EROW + 4 * 32 * 3 * 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, hh = lane >> 5;
    char* ew = smem + w * 32 * EROW;

    // weights: A operand rows = cout r32
    frag_t fw[3][KGC];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int g = 0; g < KGC; ++g)
            fw[kh][g] = *(const frag_t*)((const char*)a.w + ((r32 * 3 + kh) * 16) * SZ + 32 * g + 16 * hh);

    const int ch = lane % CPR, prow0 = lane / CPR;
    float bsh[EPC], s1[EPC], s2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        bsh[e] = a.bias[ch * EPC + e];
        s1[e] = s2[e] = 0.f;
    }
    float b4[4][4];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
        for (int j = 0; j < 4; ++j) b4[q4][j] = a.bias[8 * q4 + 4 * hh + j];

    const int ntiles = (a.M + 31) / 32;
    const int nwaves = gridDim.x * 4;
    const int rowpitch = (a.W + 1) * 4 * SZ;
    int my_cnt = 0;
    for (int tile = blockIdx.x * 4 + w; tile < ntiles; tile += nwaves) {
        const int p = tile * 32 + r32;
        uint32_t base = 0;
        if (p < a.M) {
            const int hw = a.H * a.W;
            const int n = p / hw, rem = p - n * hw;
            const int h = rem / a.W, ww = rem - h * a.W;
            base = (uint32_t)(bpix(n, h, ww, a.H, a.W) - (size_t)(a.W + 2)) * (uint32_t)(4 * SZ);
        }
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        frag_t fx[3][KGC];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int g = 0; g < KGC; ++g)
                fx[kh][g] = ((const UFrag*)((const char*)a.x4 + base + kh * rowpitch + 32 * g + 16 * hh))->v;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int g = 0; g < KGC; ++g) mma32(acc, fw[kh][g], fx[kh][g]);
        // acc[q]: cout = acc_row(q, hh), pixel = r32
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            T o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = Elem<T>::from_f32(acc[4 * q4 + j] + b4[q4][j]);
            char* dst = ew + r32 * EROW + (8 * q4 + 4 * hh) * SZ;
            if (SZ == 2) *(u32x2*)dst = *(const u32x2*)o;
            else *(u32x4*)dst = *(const u32x4*)o;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int prow = it * RPIe + prow0;
            Chunk<T> c = ld_chunk<T>(ew + prow * EROW + ch * 16);
            const int pp = tile * 32 + prow;
            if (pp < a.M) {
                st_chunk<T>((char*)a.y + ((size_t)pp * 32 + ch * EPC) * SZ, c);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float d = Elem<T>::to_f32(c.v[e]) - bsh[e];
                    s1[e] += d;
                    s2[e] += d * d;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        int tc = a.M - tile * 32;
        my_cnt += tc > 32 ? 32 : tc;
    }
    // ---- per-wave (count, mean, M2), then Chan-merge the 4 waves
    float* st = (float*)(smem + 4 * 32 * EROW);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
#pragma unroll
        for (int msk = CPR; msk < 64; msk <<= 1) {
            s1[e] = wave_sum_xor(s1[e], msk);
            s2[e] = wave_sum_xor(s2[e], msk);
        }
    }
    if (prow0 == 0) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float n = (float)my_cnt;
            const float md = n > 0 ? s1[e] / n : 0.f;
            st[(w * 32 + ch * EPC + e) * 3 + 0] = n;
            st[(w * 32 + ch * EPC + e) * 3 + 1] = bsh[e] + md;
            st[(w * 32 + ch * EPC + e) * 3 + 2] = n > 0 ? fmaxf(s2[e] - s1[e] * md, 0.f) : 0.f;
        }
    }
    __syncthreads();
    if (tid < 32) {
        float n_acc = 0.f, mean_acc = 0.f, m2_acc = 0.f;
        for (int k = 0; k < 4; ++k) {
            const float nk = st[(k * 32 + tid) * 3 + 0];
            if (nk == 0.f) continue;
            const float mk = st[(k * 32 + tid) * 3 + 1], vk = st[(k * 32 + tid) * 3 + 2];
            const float nn = n_acc + nk, dlt = mk - mean_acc;
            mean_acc += dlt * (nk / nn);
            m2_acc += vk + dlt * dlt * (n_acc * nk / nn);
            n_acc = nn;
        }
        a.part_mean[blockIdx.x * 32 + tid] = mean_acc;
        a.part_m2[blockIdx.x * 32 + tid] = m2_acc;
        if (tid == 0) a.part_cnt[blockIdx.x] = n_acc;
    }
}

hipError_t launch_conv1_fwd(int dtype, const Conv1Args& a, hipStream_t s) {
    dim3 g(a.nblocks), b(256);
    switch (dtype) {
        case 0: hipLaunchKernelGGL(conv1_fwd_kernel<float>, g, b, 0, s, a); break;
        case 1: hipLaunchKernelGGL(conv1_fwd_kernel<half_t>, g, b, 0, s, a); break;
        case 2: hipLaunchKernelGGL(conv1_fwd_kernel<bf16_t>, g, b, 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace y2
