// Micro-benchmark (tuning evidence, not product code; VERDICT r5 "move the headline or prove the wall", probes (a) and (b)): what does
// the BLOCK TILE of the split-plane GEMM buy?  gemm_x3.hip's column width is a hard 128 (X3_BN), so an A plane tile is fetched N / 128
// times; the review asked for a 256 x 256 tile (k-tile 16, three stages of 48 KiB, wave tile 128 x 64: a third fewer L2 -> LDS bytes per
// MFMA than 256 x 128) and for a four-wave, one-wave-per-SIMD form (wave tile 128 x 128 in a 512-register budget: half the fragment
// reads per MFMA).  This file is ONE k-loop template instantiated for those geometries and for the shipped ones, so that the arms
// differ in nothing but the tile:
//     arm 0   128 x 128, k-tile 32, 8 waves 4 x 2 (wave tile 32 x 64),   3 stages x 48 KiB   [gemm_x3.hip CFG 1]
//     arm 1   256 x 128, k-tile 32, 8 waves 4 x 2 (wave tile 64 x 64),   2 stages x 72 KiB   [gemm_x3.hip CFG 2]
//     arm 2   256 x 256, k-tile 16, 8 waves 2 x 4 (wave tile 128 x 64),  3 stages x 48 KiB   [probe (a)]
//     arm 3   256 x 256, k-tile 16, 4 waves 2 x 2 (wave tile 128 x 128), 3 stages x 48 KiB   [probe (b): one wave per SIMD, 512 VGPRs]
//     arm 4/5 288 x 256 / 192 x 256 "column strips": 8 waves 1 x 8, a wave owns ALL BM rows of 32 columns; only A goes through LDS
//             (k-tile 32, 2 stages x 54 / 36 KiB), a wave's W fragments are its own and come straight from L2 into registers.
//             288 rows make ONE round of the model's B = 8 shapes: M = 4600 -> 16 x 288: QKV 240 tiles, classifier 256 tiles on 256 CUs
//             (256 x 256 leaves 1.05 / 1.12 rounds); W1 (20 column tiles) would need 384 rows = 192 accumulator registers per lane, which
//             does not fit two waves per SIMD — its arm is 192 x 256: 480 tiles = 1.88 rounds, the fill 256 x 128 has today (2.81).
// Same arithmetic as the product (three exact bf16 planes per operand, six v_mfma_f32_32x32x16_bf16 products per 16-wide k-step in the
// kernel's order, fp32 accumulation), same transport (LDS-DMA of contiguous 1 KiB pieces from a tiled plane image, source-side bank
// swizzle, raw s_barrier + counted vmcnt, prefetch distance stages - 1), plain fp32 store epilogue.  The schedule is the simple
// lock-step one (one barrier per k-tile) in every arm — the product's ping-pong is worth 7-12 % on top in all of them alike.
// Operand image: [rows / RB][K / KT][plane][RB][KT] bf16 with RB x KT x 2 B = 1 KiB (RB = 16 at KT = 32, 32 at KT = 16), 16-byte slot s
// of row r stored at s ^ ((r >> 2) & 3) (KT = 32) / s ^ ((r >> 3) & 1) (KT = 16): every ds_read_b128 lane group then touches 16 slots.
// Output: per arm and shape the launch time, fp32-equivalent TF (2 M N K / t), tiles and rounds on 256 CUs; rocprofv3 --pmc FETCH_SIZE
// of the same binary gives the fabric bytes per launch (scripts/gpu_r6_tile_probe.sh).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define RAW_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

template <int BM_, int BN_, int KT_, int NW_, int WR_, int NST_>
struct geo {
    static constexpr int BM = BM_, BN = BN_, KT = KT_, NW = NW_, WR = WR_, WC = NW_ / WR_, NST = NST_;
    static constexpr int RI = BM / (32 * WR), CJ = BN / (32 * WC);            // 32 x 32 MFMA tiles of a wave: RI x CJ
    static constexpr int RB = 512 / KT;                                       // rows of a 1 KiB piece
    static constexpr int KS = KT / 16;                                        // 16-wide k-steps per k-tile
    static constexpr int APL = BM * KT * 2, WPL = BN * KT * 2;                // bytes of one plane tile
    static constexpr int STAGE = 3 * (APL + WPL);
    static constexpr int NPIECE = STAGE / 1024, PW = NPIECE / NW;             // DMA wave-instructions per stage / per wave
    static_assert(NPIECE % NW == 0, "pieces divide over the waves");
    static_assert(NST * STAGE <= 160 * 1024, "LDS");
};

// byte offset of (row r, 16-byte slot s) inside a plane tile image
template <int KT>
__device__ __forceinline__ int frag_off(int r, int s) {
    if constexpr (KT == 32) return (r >> 4) * 1024 + (r & 15) * 64 + ((s ^ ((r >> 2) & 3)) << 4);
    else return (r >> 5) * 1024 + (r & 31) * 32 + ((s ^ ((r >> 3) & 1)) << 4);
}

template <class G>
__global__ __launch_bounds__(G::NW * 64, G::NW == 4 ? 1 : 2) void gemm_probe(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W,
                                                                           float* __restrict__ C, int M, int N, int K, int store) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / G::WC, wn = wave % G::WC;
    const int tiles_m = M / G::BM, tiles_n = N / G::BN, nwg = tiles_m * tiles_n, nk = K / G::KT;
    // XCD-aware walk (block b runs on XCD b % 8): every XCD gets a contiguous eighth of the list, walked in 8-row groups
    int t;
    {
        const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3, q = nwg >> 3, r = nwg & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tm, tn;
    {
        const int GM = 8, per_group = GM * tiles_n, grp = t / per_group, first = grp * GM;
        const int gsz = tiles_m - first < GM ? tiles_m - first : GM, in = t - grp * per_group;
        tm = first + in % gsz; tn = in / gsz;
    }
    // DMA: piece pi of a stage = (operand, plane q, piece pp of the plane tile); source = [row piece][k-tile][plane] x 1 KiB
    const size_t kpieces = (size_t)nk * 3;
    auto issue = [&](int kt) {
        char* st = smem + (kt % G::NST) * G::STAGE;
#pragma unroll
        for (int j = 0; j < G::PW; ++j) {
            const int pi = wave + j * G::NW;
            constexpr int APIECES = 3 * G::APL / 1024;
            const bool isw = pi >= APIECES;
            const int pj = isw ? pi - APIECES : pi;
            const int per_plane = (isw ? G::WPL : G::APL) / 1024;
            const int q = pj / per_plane, pp = pj % per_plane;
            const size_t rp = (size_t)(isw ? tn * (G::BN / G::RB) : tm * (G::BM / G::RB)) + pp;
            const uint16_t* src = (isw ? W : A) + ((rp * kpieces + (size_t)kt * 3 + q) * 512) + lane * 8;
            char* dst = st + (isw ? 3 * G::APL : 0) + q * (isw ? G::WPL : G::APL) + pp * 1024;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                             (void __attribute__((address_space(3)))*)dst, 16, 0, 0);
        }
    };
    f32x16 acc[G::RI][G::CJ];
#pragma unroll
    for (int i = 0; i < G::RI; ++i)
#pragma unroll
        for (int j = 0; j < G::CJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int l31 = lane & 31, hh = lane >> 5;
    issue(0);
    if (G::NST == 3 && nk > 1) issue(1);
    for (int kt = 0; kt < nk; ++kt) {
        if (G::NST == 3 && kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::PW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RAW_BARRIER();                                   // tile kt landed for everybody; everybody is done with tile kt - 1
        if (kt + G::NST - 1 < nk) issue(kt + G::NST - 1);
        const char* st = smem + (kt % G::NST) * G::STAGE;
#pragma unroll
        for (int s = 0; s < G::KS; ++s) {
            bf16x8 af[G::RI][3], wf[G::CJ][3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
#pragma unroll
                for (int i = 0; i < G::RI; ++i)
                    af[i][q] = *(const bf16x8*)(st + q * G::APL + frag_off<G::KT>(wm * 32 * G::RI + 32 * i + l31, 2 * s + hh));
#pragma unroll
                for (int j = 0; j < G::CJ; ++j)
                    wf[j][q] = *(const bf16x8*)(st + 3 * G::APL + q * G::WPL + frag_off<G::KT>(wn * 32 * G::CJ + 32 * j + l31, 2 * s + hh));
            }
            constexpr int QA[6] = {0, 2, 1, 0, 1, 0}, QB[6] = {2, 0, 1, 1, 0, 0};       // smallest terms first (gemm_x3.hip)
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int i = 0; i < G::RI; ++i)
#pragma unroll
                    for (int j = 0; j < G::CJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][QA[p]], wf[j][QB[p]], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
    }
    // C/D map of the 32 x 32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int m0 = tm * G::BM + wm * 32 * G::RI, n0 = tn * G::BN + wn * 32 * G::CJ;
    if (store) {
#pragma unroll
        for (int i = 0; i < G::RI; ++i)
#pragma unroll
            for (int j = 0; j < G::CJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    C[(size_t)(m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hh) * N + n0 + 32 * j + l31] = acc[i][j][r];
    } else {                                             // k-loop only: one value per lane keeps the accumulators live
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < G::RI; ++i)
#pragma unroll
            for (int j = 0; j < G::CJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
        C[(size_t)(m0 + l31) * N + n0 + hh] = sacc;
    }
}

// ---- arm 4 / 5: column strips (see the header).  W image as in arms 0 / 1 (k-tile 32 layout); lane l of wave w holds W row
// n0 + 32 w + (l & 31), k = 16 s + 8 (l >> 5) .. + 7 of every plane: 6 x 16-byte loads per k-tile, fetched one k-tile ahead
template <int BM>
__global__ __launch_bounds__(512, 2) void gemm_strip(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W, float* __restrict__ C,
                                                     int M, int N, int K, int store) {
    constexpr int RI = BM / 32, APL = BM * 64, STAGE = 3 * APL, NPIECE = STAGE / 1024, PW = (NPIECE + 7) / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_m = M / BM, tiles_n = N / 256, nwg = tiles_m * tiles_n, nk = K / 32;
    int t;
    {
        const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3, q = nwg >> 3, r = nwg & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = t % tiles_m, tn = t / tiles_m;         // a column of tiles shares its W strip in the XCD's L2
    const size_t kpieces = (size_t)nk * 3;
    auto issue = [&](int kt) {
        char* st = smem + (kt & 1) * STAGE;
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            int pi = wave + j * 8;
            pi = pi < NPIECE ? pi : NPIECE - 1;          // BM = 288: 54 pieces over 8 waves — the last slots re-fetch the last piece
            const int q = pi / (APL / 1024), pp = pi % (APL / 1024);
            const uint16_t* src = A + (((size_t)(tm * (BM / 16) + pp) * kpieces + (size_t)kt * 3 + q) * 512) + lane * 8;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                             (void __attribute__((address_space(3)))*)(st + q * APL + pp * 1024), 16, 0, 0);
        }
    };
    // W fragments by buffer loads: descriptor + ONE 32-bit per-lane offset per k-step (row and swizzled slot) + a uniform offset for
    // (k-tile, plane) — six 64-bit per-lane pointers were what the register allocator spilled first
    const int wrow = tn * 256 + wave * 32 + l31;
    const int wsw = ((wrow & 15) >> 2) & 3;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)((size_t)N * K * 6), 0x00020000);
    const unsigned wv0 = (unsigned)(((size_t)(wrow >> 4) * kpieces) * 1024 + (wrow & 15) * 64 + (((0 + hh) ^ wsw) << 4));
    const unsigned wv1 = (unsigned)(((size_t)(wrow >> 4) * kpieces) * 1024 + (wrow & 15) * 64 + (((2 + hh) ^ wsw) << 4));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    auto load_w = [&](int kt, bf16x8 (&wf)[3][2]) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const unsigned so = (unsigned)((kt * 3 + q) * 1024);
            wf[q][0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, wv0, so, 0));
            wf[q][1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, wv1, so, 0));
        }
    };
    f32x16 acc[RI];
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 wa[3][2], wb[3][2];
    issue(0);
    load_w(0, wa);
    // A fragments one row block ahead (two register sets, names fixed by the full unroll); the sched_barrier after each block's six
    // products keeps hipcc from hoisting every ds_read of the k-tile to the top (27-36 fragments = 108-144 VGPRs: it spilled 118)
    auto compute = [&](int kt, const bf16x8 (&wf)[3][2]) {
        const char* st = smem + (kt & 1) * STAGE;
        constexpr int QA[6] = {0, 2, 1, 0, 1, 0}, QB[6] = {2, 0, 1, 1, 0, 0};
        bf16x8 af[2][3];
#pragma unroll
        for (int q = 0; q < 3; ++q) af[0][q] = *(const bf16x8*)(st + q * APL + frag_off<32>(l31, hh));
#pragma unroll
        for (int x = 0; x < 2 * RI; ++x) {
            const int s = x / RI, i = x % RI;
            if (x + 1 < 2 * RI) {
                const int s1 = (x + 1) / RI, i1 = (x + 1) % RI;
#pragma unroll
                for (int q = 0; q < 3; ++q) af[(x + 1) & 1][q] = *(const bf16x8*)(st + q * APL + frag_off<32>(32 * i1 + l31, 2 * s1 + hh));
            }
#pragma unroll
            for (int p = 0; p < 6; ++p) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[x & 1][QA[p]], wf[QB[p]][s], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // ONE copy of the MFMA phase in the loop body (two copies — one per W register set — made hipcc keep two images of the
    // accumulators: 118-174 spilled VGPRs); the W set of the next k-tile moves into place with 24 v_mov per 6 RI x 2 MFMAs
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RAW_BARRIER();
        if (kt + 1 < nk) { issue(kt + 1); load_w(kt + 1, wb); }
        compute(kt, wa);
#pragma unroll
        for (int q = 0; q < 3; ++q) { wa[q][0] = wb[q][0]; wa[q][1] = wb[q][1]; }
    }
    const int m0 = tm * BM, n0 = tn * 256 + wave * 32;
    if (store) {
#pragma unroll
        for (int i = 0; i < RI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) C[(size_t)(m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hh) * N + n0 + l31] = acc[i][r];
    } else {
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < RI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc += acc[i][r];
        C[(size_t)(m0 + l31) * N + n0 + hh] = sacc;
    }
}

// ---- host ---------------------------------------------------------------------------------------------------------------------------
static uint16_t bf16_rne(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf16_f32(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

// logical matrix X [rows][K] (fp32) -> the probe's tiled, swizzled three-plane image for k-tile KT
static void make_image(const std::vector<float>& X, int rows, int K, int KT, std::vector<uint16_t>& img) {
    const int RB = 512 / KT, nk = K / KT;
    img.assign((size_t)rows * K * 3, 0);
    for (int r = 0; r < rows; ++r)
        for (int k = 0; k < K; ++k) {
            const float x = X[(size_t)r * K + k];
            uint16_t p[3];
            p[0] = bf16_rne(x);
            const float r1 = x - bf16_f32(p[0]);
            p[1] = bf16_rne(r1);
            p[2] = bf16_rne(r1 - bf16_f32(p[1]));
            const int rp = r / RB, rr = r % RB, kt = k / KT, kk = k % KT, s = kk / 8, e = kk % 8;
            const int ss = KT == 32 ? (s ^ ((rr >> 2) & 3)) : (s ^ ((rr >> 3) & 1));
            for (int q = 0; q < 3; ++q)
                img[(((size_t)rp * nk + kt) * 3 + q) * 512 + (size_t)rr * KT + ss * 8 + e] = p[q];
        }
}

struct arm_info { const char* name; int BM, BN, KT; };

typedef void (*kern_t)(const uint16_t*, const uint16_t*, float*, int, int, int, int);
template <class G>
static double run_arm(const char* name, const uint16_t* dA, const uint16_t* dW, float* dC, int M, int N, int K, int store, int iters,
                      const std::vector<float>* hA, const std::vector<float>* hW);
static double run_kernel(const char* name, kern_t kern, int BM, int BN, int KT, int threads, size_t lds, const uint16_t* dA, const uint16_t* dW,
                         float* dC, int M, int N, int K, int store, int iters, const std::vector<float>* hA, const std::vector<float>* hW) {
    if (M % BM || N % BN || K % (2 * KT)) { printf("  %-44s skipped (shape not a multiple of the tile)\n", name); return 0; }
    const int nwg = (M / BM) * (N / BN);
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(threads), lds, 0, dA, dW, dC, M, N, K, 1);
    CHECK(hipDeviceSynchronize());
    if (hA) {                                            // sampled check against float64 of the logical operands
        std::vector<float> C((size_t)M * N);
        CHECK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int smp = 0; smp < 4000; ++smp) {
            const int m = (int)((smp * 2654435761u) % (unsigned)M), n = (int)((smp * 40503u + 17) % (unsigned)N);
            double ref = 0, mag = 0;
            for (int k = 0; k < K; ++k) { const double a = (*hA)[(size_t)m * K + k], w = (*hW)[(size_t)n * K + k]; ref += a * w; mag += fabs(a * w); }
            const double e = fabs(C[(size_t)m * N + n] - ref) / (mag + 1e-30);
            if (e > worst) worst = e;
        }
        if (worst > 2e-6) { printf("  %-44s WRONG: max |err| / sum|a w| = %.3e\n", name, worst); return 0; }
    }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(nwg), dim3(threads), lds, 0, dA, dW, dC, M, N, K, store);
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(nwg), dim3(threads), lds, 0, dA, dW, dC, M, N, K, store);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / iters, tf = 2.0 * M * N * (double)K / (us * 1e-6) / 1e12;
    printf("  %-44s %5d tiles = %4.2f rounds  %8.1f us  %6.1f TF-eq  %6.0f TF executed  (%s)\n", name, nwg, nwg / 256.0, us, tf, 6 * tf,
           store ? "with the fp32 store" : "k-loop only");
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return us;
}
template <class G>
static double run_arm(const char* name, const uint16_t* dA, const uint16_t* dW, float* dC, int M, int N, int K, int store, int iters,
                      const std::vector<float>* hA, const std::vector<float>* hW) {
    return run_kernel(name, gemm_probe<G>, G::BM, G::BN, G::KT, G::NW * 64, (size_t)G::NST * G::STAGE, dA, dW, dC, M, N, K, store, iters, hA, hW);
}

int main(int argc, char** argv) {
    const int only = argc > 1 ? atoi(argv[1]) : -1;     // run one arm only (the PMC passes)
    const int iters = argc > 2 ? atoi(argv[2]) : 30;
    struct shape { int M, N, K; const char* what; } shapes[] = {
        {4096, 4096, 1280, "every arm fills whole rounds of 256 CUs: the per-tile rate"},
        {4096, 4096, 5120, "the same with a 4 x longer k-loop (epilogue share / 4)"},
        {4608, 3840, 1280, "QKV at B = 8 (M = 4600 padded to 18 x 256)"},
        {4608, 5120, 1280, "W1 at B = 8"},
        {4608, 1280, 2560, "W2 at B = 8"},
        {4608, 4096, 1280, "classifier at B = 8"},
    };
    for (const shape& sh : shapes) {
        if (only >= 0 && &sh != &shapes[only >= 4 ? 5 : 0]) continue;
        const int M = sh.M, N = sh.N, K = sh.K;
        std::vector<float> hA((size_t)M * K), hW((size_t)N * K);
        uint32_t s = 12345u;
        auto gauss = [&]() {
            s = s * 1664525u + 1013904223u; const float u1 = ((s >> 8) + 1) * (1.0f / 16777217.0f);
            s = s * 1664525u + 1013904223u; const float u2 = (s >> 8) * (1.0f / 16777216.0f);
            return sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
        };
        for (auto& v : hA) v = gauss();                  // activations sigma 1, weights sigma 1 / sqrt(K): the model's operand statistics
        const float ws = 1.0f / sqrtf((float)K);
        for (auto& v : hW) v = ws * gauss();
        std::vector<uint16_t> a32, w32, a16, w16;
        make_image(hA, M, K, 32, a32); make_image(hW, N, K, 32, w32);
        make_image(hA, M, K, 16, a16); make_image(hW, N, K, 16, w16);
        uint16_t *dA32, *dW32, *dA16, *dW16; float* dC;
        CHECK(hipMalloc(&dA32, a32.size() * 2)); CHECK(hipMalloc(&dW32, w32.size() * 2));
        CHECK(hipMalloc(&dA16, a16.size() * 2)); CHECK(hipMalloc(&dW16, w16.size() * 2));
        CHECK(hipMalloc(&dC, (size_t)M * N * 4));
        CHECK(hipMemcpy(dA32, a32.data(), a32.size() * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dW32, w32.data(), w32.size() * 2, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dA16, a16.data(), a16.size() * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dW16, w16.data(), w16.size() * 2, hipMemcpyHostToDevice));
        printf("== M = %d, N = %d, K = %d: %s\n", M, N, K, sh.what);
        const bool chk = &sh == &shapes[0] || &sh == &shapes[2];
        for (int store = 1; store >= 0; --store) {
            const std::vector<float>*pa = chk && store ? &hA : nullptr, *pw = chk && store ? &hW : nullptr;
            if (only < 0 || only == 0) run_arm<geo<128, 128, 32, 8, 4, 3>>("arm 0  128 x 128  k32  8 waves (32 x 64)", dA32, dW32, dC, M, N, K, store, iters, pa, pw);
            if (only < 0 || only == 1) run_arm<geo<256, 128, 32, 8, 4, 2>>("arm 1  256 x 128  k32  8 waves (64 x 64)", dA32, dW32, dC, M, N, K, store, iters, pa, pw);
            if (only < 0 || only == 2) run_arm<geo<256, 256, 16, 8, 2, 3>>("arm 2  256 x 256  k16  8 waves (128 x 64)", dA16, dW16, dC, M, N, K, store, iters, pa, pw);
            if (only < 0 || only == 3) run_arm<geo<256, 256, 16, 4, 2, 3>>("arm 3  256 x 256  k16  4 waves (128 x 128)", dA16, dW16, dC, M, N, K, store, iters, pa, pw);
            if (only < 0 || only == 4) run_kernel("arm 4  288 x 256  k32  8 strips (288 x 32), W in regs", gemm_strip<288>, 288, 256, 32, 512, 2 * 3 * 288 * 64, dA32, dW32, dC, M, N, K, store, iters, pa, pw);
            if (only < 0 || only == 5) run_kernel("arm 5  192 x 256  k32  8 strips (192 x 32), W in regs", gemm_strip<192>, 192, 256, 32, 512, 2 * 3 * 192 * 64, dA32, dW32, dC, M, N, K, store, iters, pa, pw);
            if (only >= 0) break;
        }
        CHECK(hipFree(dA32)); CHECK(hipFree(dW32)); CHECK(hipFree(dA16)); CHECK(hipFree(dW16)); CHECK(hipFree(dC));
    }
    return 0;
}
