// Micro-benchmark (tuning evidence, not product code): what limits v_mfma_f32_32x32x16_bf16 throughput on a full MI355X when NOTHING
// but the matrix pipe works — no global / LDS traffic at all?  The split-plane GEMM stops at ~1.1-1.3 PF executed with the pipe ~50 %
// busy (profiles/history/r04_model_clock.txt); the round-2/3 notes call that "the power limit" on the strength of clock readings.  This probe
// separates the candidates: every wave holds NSETS operand register sets and walks them in a loop of MFMAs into four accumulators
//   mode 0  ONE operand set whose values are a smooth ramp (what scripts/ubench/mfma_valu_coissue.hip measured: 2.32 PF)
//   mode 1  ONE operand set of RANDOM bf16 bit patterns (exponents confined to 2^-4 .. 2^4)
//   mode 2  EIGHT random operand sets used round-robin: consecutive MFMAs read different registers (operand buses toggle)
//   mode 3  mode 2 + a raw s_barrier after every 12 MFMAs (the ping-pong GEMM's phase length), 8 waves per block
//   mode 4  (round 5) the operands the bf16x3 GEMM really multiplies: the THREE exact split planes (vn_split3: 8 + 8 + 8 significand bits)
//           of Gaussian activations (sigma 1) and weights (sigma 1 / sqrt(1280)), walked in the kernel's six-product order (A0 W2, A2 W0,
//           A1 W1, A0 W1, A1 W0, A0 W0) — plane 0 has a narrow exponent field, planes 1 / 2 are residuals 2^-8 / 2^-16 below it with
//           random mantissas: where between "smooth" (2.4 PF) and "uniformly random words" (1.65 PF) do real planes put the wall?
//   modes 5-7 (round 6, VERDICT r5 probe (c)): the SAME six products of the SAME planes in three other orders — does the wall move when
//           consecutive MFMAs share an operand register (the source operand does not toggle)?
//           5 grouped by A plane   A0W2 A0W1 A0W0 A1W1 A1W0 A2W0      (A changes twice per k-step, W every time)
//           6 grouped by W plane   A2W0 A1W0 A0W0 A1W1 A0W1 A0W2      (W changes twice per k-step)
//           7 both operands change on every instruction   A0W2 A1W1 A2W0 A0W1 A1W0 A0W0
//           (mode 4, the kernel's order A0W2 A2W0 A1W1 A0W1 A1W0 A0W0, changes both on 4 of 6)
//           Already known from modes 1 / 2: ONE random set reused by every MFMA (nothing toggles between instructions) 1746 TF, eight
//           sets round-robin 1686 TF — the order is worth <= 3 %; what costs is the bit activity INSIDE a product, not between products.
// Each mode runs ~0.3 s (DVFS settles in milliseconds); prints executed TF and the shader clock from s_memtime / wall time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
__device__ __forceinline__ bf16x8 random_set(unsigned& s) {
    u32x4 w;
    for (int i = 0; i < 4; ++i) {
        unsigned lo = rnd(s) >> 16, hi = rnd(s) >> 16;
        // sign | exponent 123..131 (2^-4 .. 2^4) | 7 random mantissa bits
        lo = (lo & 0x807fu) | ((123u + (lo >> 7) % 9u) << 7);
        hi = (hi & 0x807fu) | ((123u + (hi >> 7) % 9u) << 7);
        w[i] = lo | (hi << 16);
    }
    return __builtin_bit_cast(bf16x8, w);
}

__device__ __forceinline__ float gauss(unsigned& s) {          // Box-Muller from two LCG draws
    const float u1 = ((rnd(s) >> 8) + 1) * (1.0f / 16777217.0f), u2 = (rnd(s) >> 8) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.2831853f * u2);
}
__device__ __forceinline__ unsigned short bf16_rne(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ float bf16_f32(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }
// eight Gaussian values (sigma) -> their three exact split planes, as vn_common.h vn_split3
__device__ __forceinline__ void plane_sets(unsigned& s, float sigma, bf16x8 (&p)[3]) {
    unsigned short q[3][8];
    for (int i = 0; i < 8; ++i) {
        const float x = sigma * gauss(s);
        q[0][i] = bf16_rne(x);
        const float r1 = x - bf16_f32(q[0][i]);
        q[1][i] = bf16_rne(r1);
        q[2][i] = bf16_rne(r1 - bf16_f32(q[1][i]));
    }
    for (int t = 0; t < 3; ++t) {
        u32x4 w;
        for (int i = 0; i < 4; ++i) w[i] = q[t][2 * i] | ((unsigned)q[t][2 * i + 1] << 16);
        p[t] = __builtin_bit_cast(bf16x8, w);
    }
}

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    constexpr int NS = MODE >= 4 ? 3 : MODE >= 2 ? 8 : 1;
    f32x16 acc[4] = {};
    bf16x8 a[NS], b[NS];
    unsigned s = 12345u + threadIdx.x * 977u + blockIdx.x * 131071u;
    if (MODE >= 4) {
        bf16x8 pa[3], pb[3];
        plane_sets(s, 1.0f, pa);
        plane_sets(s, 0.02795f, pb);
        for (int q = 0; q < 3; ++q) { a[q % NS] = pa[q]; b[q % NS] = pb[q]; }
    }
    for (int q = 0; q < NS && MODE < 4; ++q) {
        if (MODE == 0) {
            for (int i = 0; i < 8; ++i) { a[q][i] = (__bf16)(threadIdx.x * 0.001f + i); b[q][i] = (__bf16)(1.0f + i * 0.01f); }
        } else {
            a[q] = random_set(s);
            b[q] = random_set(s);
        }
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            if (MODE >= 4) {                                  // the six plane products of a k-step; mode 4 = gemm_x3.hip's order (mac_prod)
                constexpr int QA[4][6] = {{0, 2, 1, 0, 1, 0}, {0, 0, 0, 1, 1, 2}, {2, 1, 0, 1, 0, 0}, {0, 1, 2, 0, 1, 0}};
                constexpr int QB[4][6] = {{2, 0, 1, 1, 0, 0}, {2, 1, 0, 1, 0, 0}, {0, 0, 0, 1, 1, 2}, {2, 1, 0, 1, 0, 0}};
                constexpr int O = MODE - 4;
                acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[QA[O][u % 6] % NS], b[QB[O][u % 6] % NS], acc[u & 3], 0, 0, 0);
            } else
            acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u % NS], b[(u / 2) % NS], acc[u & 3], 0, 0, 0);
            if (MODE == 3 && (u % 12) == 11) __builtin_amdgcn_s_barrier();
        }
        // keep the accumulators bounded without touching the operand registers: scale by 2^-k every so often (4 VALU per 24 MFMA x 16... rare)
        if ((it & 63) == 63) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] *= 1.0e-6f;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0.f;
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) r += acc[q][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
static void run(const char* what, int blocks) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)blocks * 512 * 4); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 2000;
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(512), 0, 0, out, cyc, iters);        // warm-up / calibration
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(512), 0, 0, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    iters = (int)(iters * 300.0f / ms);                                                   // ~0.3 s
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(512), 0, 0, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double nm = iters * 24.0, waves = blocks * 8.0;
    printf("mode %d  %-66s %6.1f ms  %7.0f TF executed  %5.1f ticks per MFMA per wave  %.3f GHz (s_memtime ticks / wall)\n", MODE, what, ms,
           waves * nm * 32768.0 / (ms * 1e-3) / 1e12, (double)c / nm, (double)c / (ms * 1e6));
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int blocks = 256; blocks <= 512; blocks += 256) {
        printf("-- %d blocks of 8 waves (%d waves per SIMD)\n", blocks, blocks / 128);
        run<0>("one smooth operand set", blocks);
        run<1>("one RANDOM operand set", blocks);
        run<2>("eight random operand sets, round-robin", blocks);
        run<3>("eight random sets + s_barrier every 12 MFMAs", blocks);
        run<4>("the three split planes of Gaussian activations x weights, six-product order", blocks);
        run<5>("same planes, products grouped by A plane (A0W2 A0W1 A0W0 A1W1 A1W0 A2W0)", blocks);
        run<6>("same planes, products grouped by W plane (A2W0 A1W0 A0W0 A1W1 A0W1 A0W2)", blocks);
        run<7>("same planes, both operands change every MFMA (A0W2 A1W1 A2W0 A0W1 A1W0 A0W0)", blocks);
    }
    return 0;
}
