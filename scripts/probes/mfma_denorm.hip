// probe: do the f16 MFMA forms honour subnormal A/B inputs on gfx950?  (round 5, split-operand mode)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float a, float b, float* out) {
    f16x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (_Float16)a; B[i] = (_Float16)b; }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc, 0, 0, 0);
    f32x4 acc4 = {0, 0, 0, 0};
    acc4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, acc4, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = acc4[0]; out[2] = (float)A[0]; out[3] = (float)((_Float16)a * (_Float16)b); }
}
int main() {
    float* d; hipMalloc(&d, 16);
    const float as[] = {1.0f, 0x1p-16f, 0x1p-20f, 0x1p-24f, 3 * 0x1p-24f};
    for (float a : as) {
        float h[4];
        k<<<1, 64>>>(a, 1024.0f, d);
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("a=%g b=1024: mfma32x32x16 %g (expect %g)  mfma16x16x32 %g (expect %g)  f16(a)=%g valu_f16_mul=%g\n", a, h[0], 16.0 * a * 1024.0, h[1], 32.0 * a * 1024.0, h[2], h[3]);
    }
    return 0;
}
