// Micro-benchmark (tuning evidence, not product code): can ONE wave overlap its own VALU work with its own MFMAs on gfx950 ?
// Each wave runs a loop of {1 v_mfma_f32_32x32x16_bf16, N VALU instructions} with everything independent (DEP = 0) or with the
// MFMAs chained on one accumulator (DEP = 1), one or two waves per SIMD; prints cycles per loop iteration (s_memtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int N, int DEP, int KIND>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = threadIdx.x * 0.5f + i;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (DEP) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            else if (u == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            else if (u == 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            else if (u == 2) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
            else acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc3, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j % 12]) : "v"(v[(j + 5) % 12]));
                else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j % 12]));
                else asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[j % 12]) : "v"(v[(j + 5) % 12]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    for (int i = 0; i < 12; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int N, int DEP, int KIND>
static void run(int threads, const char* kind) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<N, DEP, KIND>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<N, DEP, KIND>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double nm = iters * 4.0, waves = 256.0 * threads / 64;
    printf("waves/SIMD %d  %-10s dep %d  VALU per MFMA %2d : %6.1f s_memtime ticks per (MFMA + VALU group) per wave; kernel %.3f ms -> %.0f TF bf16, "
           "%.2f ns per tick\n", threads / 256, kind, DEP, N, (double)c / nm, ms, waves * nm * 32768.0 / (ms * 1e-3) / 1e12, ms * 1e6 / (double)c);
    hipFree(out); hipFree(cyc);
}
#define SWEEP(DEP, KIND, NAME) \
    run<0, DEP, KIND>(t, NAME); run<2, DEP, KIND>(t, NAME); run<4, DEP, KIND>(t, NAME); run<6, DEP, KIND>(t, NAME); \
    run<8, DEP, KIND>(t, NAME); run<12, DEP, KIND>(t, NAME);
int main() {
    for (int t = 256; t <= 512; t += 256) {
        SWEEP(0, 0, "v_fma")
        SWEEP(1, 0, "v_fma")
        SWEEP(0, 1, "v_exp")
        SWEEP(0, 2, "v_cvt_pk")
    }
    return 0;
}
