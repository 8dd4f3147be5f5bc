// f16_denorm.hip -- does v_mfma_f32_16x16x32_f16 honour fp16 subnormal INPUTS, and do v_cvt_pkrtz_f16_f32 / v_cvt_pk_f16_f32 (round
// to nearest: what split2h uses) produce them?
//   hipcc --offload-arch=gfx950 -O3 -w tools_ubench/f16_denorm.hip -o tools_ubench/f16_denorm.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __fp16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float a, float b, float* out) {
    const h2 pa = __builtin_amdgcn_cvt_pkrtz(a, a), pb = __builtin_amdgcn_cvt_pkrtz(b, b);
    h8 va, vb;
    for (int i = 0; i < 8; ++i) { va[i] = (_Float16)pa[0]; vb[i] = (_Float16)pb[0]; }
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, vb, c, 0, 0, 0);
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef _Float16 g2 __attribute__((ext_vector_type(2)));
    const g2 pn = __builtin_convertvector(((f2){a, 1.5f * a}), g2);   // v_cvt_pk_f16_f32
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)pa[0]; out[2] = (float)pn[0]; out[3] = (float)pn[1]; }
}
int main() {
    float* d; hipMalloc(&d, 16);
    const float as[] = {1.0f, 9.5367431640625e-07f /* 2^-20: fp16 subnormal */, 5.9604644775390625e-08f /* 2^-24: smallest subnormal */, 3.0e-5f};
    for (float a : as) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, 1024.0f, d);
        float h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("a = %.6e  cvt_pkrtz -> %.6e   cvt_pk (nearest) a, 1.5 a -> %.6e, %.6e   mfma(32 x a x 1024) = %.6e  (exact %.6e)\n", a, h[1], h[2], h[3], h[0], 32.0 * a * 1024.0);
    }
    return 0;
}
