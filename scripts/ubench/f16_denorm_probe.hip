// Does v_mfma_f32_32x32x16_f16 on gfx950 keep fp16 SUBNORMAL inputs, and does v_cvt_f16_f32 produce them?  (decides whether the
// two-plane fp16 split of gemm_x3.hip needs per-row scaling)   hipcc --offload-arch=gfx950 -O2 f16_denorm_probe.hip -o f16_denorm_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(float a_val, float b_val, float* out) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; }
}
int main() {
    float* d; hipMalloc(&d, 8);
    const float vals[4] = {1.0f, 6.103515625e-05f, 9.5367431640625e-07f /* 2^-20: subnormal */, 5.9604644775390625e-08f /* 2^-24: smallest */};
    for (int i = 0; i < 4; ++i) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, vals[i], 1024.0f, d);
        float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("a = %.10e (as fp16 -> %.10e)  b = 1024: mfma sum of 16 products = %.10e  expected %.10e  %s\n", vals[i], h[1], h[0], 16.0 * vals[i] * 1024.0,
               h[0] == (float)(16.0 * vals[i] * 1024.0) ? "KEPT" : "FLUSHED/DIFFERENT");
    }
    return 0;
}
