// STAGED FOR ROUND 2 — compiled, exported as single ops (vn_split2h_f32 / vn_gemm_f16x2), NOT used by any model path and NOT
// yet run on a GPU (the round-1 GPU budget ended with gemm_x3.hip).  tests/test_gpu_kernels.py holds its parity tests behind
// VN_EXPERIMENTAL=1.
//
// fp32-grade GEMM on the fp16 matrix cores of gfx950 ("f16x2"):  C[M][N] (op)= A[M][K] * W[N][K]^T
//
// Successor of the bf16x3 scheme (gemm_x3.hip) with HALF the matrix work and two thirds of the operand bytes.  Each operand
// is held as TWO fp16 planes
//        h0 = fp16(x),      h1 = fp16((x - h0) * 2^11)            (x - h0 is exact in fp32; |x - h0| <= 2^-11 |x|)
// so x = h0 + 2^-11 h1 up to 2^-22 |x| (fp16 carries 11 significand bits; the 2^11 pre-scale keeps h1 in fp16's normal
// range whenever h0 is).  The product keeps three terms in two accumulators,
//        main = A0 W0                    corr = A0 W1 + A1 W0                    C = main + 2^-11 corr,
// three v_mfma_f32_32x32x16_f16 per 16-wide k-step with fp32 accumulation; every fp16 x fp16 product is exact in fp32 and
// the dropped A1 W1 term is 2^-22 relative.  Operand error + dropped term give ~7e-7 per product with random sign, i.e.
// ~sqrt(K) * 7e-7 * rms|a w| on a dot product — an order of magnitude BELOW the rounding of the fp32 accumulation itself
// (tests/test_host_logic.py::test_f16x2_split_numerics_on_the_host: max error 2.7e-6 vs 2.9e-6 for an fp32 GEMM at
// K = 1280, both against float64).  Range: |x| must stay below 65 504 (GEMM operands here are normalised rows, attention
// outputs, GEGLU outputs and weights); magnitudes under 6e-5 lose relative, not absolute, precision.
// Ceiling: 2500 / 3 = 833 fp32-equivalent TFLOP/s (bf16x3: 417; fp32-input MFMA: 157).
//
// Kernel = gemm_x3.hip's structure with two planes and a tile height template: 512 threads, 8 waves as 4 x 2, wave tile
// (32 MI) x 64, block tile (128 MI) x 128, k-tile 32 fp16 = 64-byte rows, the same source-side swizzle
// slot ^ ((row >> 2) & 3), LDS-DMA double buffering, register-prefetched fragments, one barrier per k-tile between its two
// k-steps.  MI = 2 (256 x 128) halves the DMA bytes per MFMA of the A planes: 48 KiB per k-tile against 24 MFMAs per wave.
// Split-K over gridDim.y for the store epilogue as in gemm_x3.hip.
#include <stdlib.h>
#include "vn_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define H2_KT 32                                   // fp16 per k-tile
#define H2_BPLANE (128 * 16)                       // floats of one W plane tile (128 rows x 64 B)
#define H2_SCALE 2048.0f                           // 2^11: pre-scale of the low plane
#define H2_INV_SCALE (1.0f / 2048.0f)

template <int MI>
struct H2Cfg {
    static constexpr int BM = 128 * MI;
    static constexpr int APLANE = BM * 16;                          // floats of one A plane tile
    static constexpr int STAGE = 2 * APLANE + 2 * H2_BPLANE;        // floats: A0 | A1 | W0 | W1
    static constexpr int LDS_BYTES = 2 * STAGE * 4;                 // 64 KiB (MI 1) / 96 KiB (MI 2)
    static constexpr int NPW = 2 * MI + 2;                          // DMA wave-instructions per wave and stage
};

__device__ __forceinline__ int h2_xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ void h2_tile_coords(int t, int tiles_m, int tiles_n, int& tm, int& tn) {
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * tiles_n;
    const int grp = t / per_group;
    const int first_m = grp * GROUP_M;
    const int gsz = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
    const int in_grp = t - grp * per_group;
    tm = first_m + in_grp % gsz;
    tn = in_grp / gsz;
}

// fp32 -> (h0, h1) with x ~= h0 + 2^-11 h1 (see the header); v_cvt_f16_f32 rounds to nearest even
__device__ __forceinline__ void vn_split2h(float x, uint16_t& h0, uint16_t& h1) {
    const _Float16 a = (_Float16)x;
    const _Float16 b = (_Float16)((x - (float)a) * H2_SCALE);
    h0 = __builtin_bit_cast(uint16_t, a);
    h1 = __builtin_bit_cast(uint16_t, b);
}

template <int EPI, int MI>
__global__ __launch_bounds__(512, 2) void vn_gemm_h2_kernel(vn_gemm_args p, int tiles_m, int tiles_n) {
    using Cfg = H2Cfg<MI>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int tm, tn;
    h2_tile_coords(h2_xcd_remap(blockIdx.x, tiles_m * tiles_n), tiles_m, tiles_n, tm, tn);
    const int m0 = tm * Cfg::BM, n0 = tn * 128;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const uint16_t* A16 = (const uint16_t*)p.A;
    const uint16_t* W16 = (const uint16_t*)p.W;
    const int nk_all = p.K / H2_KT;
    const int kb = (int)((long)nk_all * blockIdx.y / gridDim.y), ke = (int)((long)nk_all * (blockIdx.y + 1) / gridDim.y);
    if (gridDim.y > 1) p.C += (size_t)blockIdx.y * p.M * p.N;           // split-K: raw image per split (store epilogue)

    // DMA: instruction q (1 KiB = 16 rows x 64 B) lands at float offset 256 q of the stage: A0 | A1 | W0 | W1
    const uint16_t* src[Cfg::NPW];
    const int drow = lane >> 2, dslot = (lane & 3) ^ ((lane >> 4) & 3);      // (row >> 2) & 3 == (lane >> 4) & 3
#pragma unroll
    for (int j = 0; j < Cfg::NPW; ++j) {
        const int q = wave * Cfg::NPW + j;
        if (q < 16 * MI) {
            const int plane = q / (8 * MI), row = (q % (8 * MI)) * 16 + drow;
            int g = m0 + row;
            g = g < p.M ? g : p.M - 1;
            src[j] = A16 + (size_t)plane * p.a_plane + (size_t)g * p.K + dslot * 8 + kb * H2_KT;
        } else {
            const int qb = q - 16 * MI;
            const int plane = qb >> 3, row = (qb & 7) * 16 + drow;
            int g = n0 + row;
            g = g < p.N ? g : p.N - 1;
            src[j] = W16 + (size_t)plane * p.w_plane + (size_t)g * p.K + dslot * 8 + kb * H2_KT;
        }
    }
    auto stage = [&](int buf, int k0) {
        float* base = lds + buf * Cfg::STAGE + wave * (Cfg::NPW * 256);
#pragma unroll
        for (int j = 0; j < Cfg::NPW; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + k0),
                                             (__attribute__((address_space(3))) void*)(base + j * 256), 16, 0, 0);
    };

    // fragment offsets (floats) inside a plane tile: row * 16 + ((2 s + h) ^ ((row >> 2) & 3)) * 4
    const int l31 = lane & 31, h = lane >> 5, sw = (lane >> 2) & 3;
    const int aRow = (wm * 32 * MI + l31) * 16;
    const int bRow = (wn * 64 + l31) * 16;

    f32x16 accm[MI][2], accc[MI][2];                   // main = A0 W0 ; corr = A0 W1 + A1 W0
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accm[i][j][r] = 0.0f; accc[i][j][r] = 0.0f; }

    struct Frags { f16x8 a[2][MI], b[2][2]; };
    auto load_frags = [&](Frags& f, int buf, int s) {
        const float* sA = lds + buf * Cfg::STAGE;
        const float* sB = sA + 2 * Cfg::APLANE;
        const int off = ((2 * s + h) ^ sw) * 4;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
                f.a[q][i] = __builtin_bit_cast(f16x8, *(const f32x4*)(sA + q * Cfg::APLANE + aRow + i * 32 * 16 + off));
#pragma unroll
            for (int j = 0; j < 2; ++j)
                f.b[q][j] = __builtin_bit_cast(f16x8, *(const f32x4*)(sB + q * H2_BPLANE + bRow + j * 32 * 16 + off));
        }
    };
    // head = the first pair of MFMAs of a k-step (the next step's fragment reads are issued behind it), tail = the rest
    auto mac_head = [&](const Frags& f) {
#pragma unroll
        for (int j = 0; j < 2; ++j) accc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[0][0], f.b[1][j], accc[0][j], 0, 0, 0);
    };
    auto mac_tail = [&](const Frags& f) {
#pragma unroll
        for (int i = 1; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) accc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[0][i], f.b[1][j], accc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) accc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[1][i], f.b[0][j], accc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) accm[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[0][i], f.b[0][j], accm[i][j], 0, 0, 0);
    };

    const int nk = ke - kb;
    Frags f0, f1;
    stage(0, 0);
    if (nk > 1) stage(1, H2_KT);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Cfg::NPW) : "memory");      // tile 0 landed (tile 1 may be in flight)
    if (nk == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    load_frags(f0, 0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        __builtin_amdgcn_s_setprio(1);
        mac_head(f0);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(f1, cur, 1);
        __builtin_amdgcn_sched_barrier(0);
        mac_tail(f0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of tile kt + 1 has landed
        __syncthreads();                                      // + lgkmcnt(0): f1 in registers; tile kt + 1 complete
        if (kt + 2 < nk) stage(cur, (kt + 2) * H2_KT);
        __builtin_amdgcn_s_setprio(1);
        mac_head(f1);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) load_frags(f0, cur ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mac_tail(f1);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int colw = n0 + wn * 64 + l31;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 * MI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row >= p.M) continue;
            const float v0 = accm[i][0][r] + accc[i][0][r] * H2_INV_SCALE;
            const float v1 = accm[i][1][r] + accc[i][1][r] * H2_INV_SCALE;
            if constexpr (EPI == VN_EPI_GEGLU) {
                // wave tile = 64 packed columns = 32 value (j = 0) + 32 gate (j = 1), interleaved at pack time
                const int ocol = (n0 + wn * 64) / 2 + l31;
                if (2 * ocol >= p.N) continue;
                const float o = v0 * vn_gelu_tanh(v1);
                if (p.C16) {
                    uint16_t t0, t1;
                    vn_split2h(o, t0, t1);
                    uint16_t* d = p.C16 + (size_t)row * p.ldc + ocol;
                    d[0] = t0; d[p.c_plane] = t1;
                } else {
                    p.C[(size_t)row * p.ldc + ocol] = o;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = colw + j * 32;
                    if (col >= p.N) continue;
                    const float v = j ? v1 : v0;
                    if constexpr (EPI == VN_EPI_STORE) {
                        p.C[(size_t)row * p.ldc + col] = v;
                    } else if constexpr (EPI == VN_EPI_BIAS) {
                        p.C[(size_t)row * p.ldc + col] = v + p.bias[col];
                    } else if constexpr (EPI == VN_EPI_RESIDUAL) {
                        float* c = p.C + (size_t)row * p.ldc + col;
                        *c = *c + v;
                    } else if constexpr (EPI == VN_EPI_QKV) {
                        const int D = p.H * VN_DHEAD;
                        const int which = col / D, rem = col - which * D;
                        const int hd = rem >> 6, d = rem & 63;
                        const int b = row / p.T, t = row - b * p.T;
                        p.C[which * p.qkv_plane + (((size_t)b * p.H + hd) * p.T + t) * VN_DHEAD + d] = v;
                    }
                }
            }
        }
    }
}

#define H2_WS_FLOATS (32L << 20)          // shares ctx->x3_ws (128 MiB, allocated once)

template <int EPI, int MI>
static int h2_launch(vn_ctx* ctx, const vn_gemm_args& a, int nsplit, hipStream_t s) {
    using Cfg = H2Cfg<MI>;
    static bool attr = false;             // staged code: one device per process is assumed until this joins vn_ctx::attr_mask
    if (!attr) {
        VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)vn_gemm_h2_kernel<EPI, MI>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
        attr = true;
    }
    const int tiles_m = vn_cdiv(a.M, Cfg::BM), tiles_n = vn_cdiv(a.N, 128);
    hipLaunchKernelGGL((vn_gemm_h2_kernel<EPI, MI>), dim3(tiles_m * tiles_n, nsplit), dim3(512), Cfg::LDS_BYTES, s, a, tiles_m, tiles_n);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

template <int MI>
static int h2_dispatch(vn_ctx* ctx, const vn_gemm_args& a, int epilogue, int nsplit, hipStream_t s) {
    if (nsplit > 1) {
        if (epilogue != VN_EPI_STORE && epilogue != VN_EPI_RESIDUAL) return vn_fail(ctx, VN_ERR_INVALID, "gemm_h2: split-K needs the store / residual epilogue%s", "");
        if ((double)nsplit * a.M * a.N > (double)H2_WS_FLOATS) return vn_fail(ctx, VN_ERR_INVALID, "gemm_h2: split-K workspace too small%s", "");
        if (!ctx->x3_ws) VN_HIP_CHECK(ctx, hipMalloc((void**)&ctx->x3_ws, (size_t)H2_WS_FLOATS * sizeof(float)));
        vn_gemm_args q = a;
        q.C = ctx->x3_ws;
        q.ldc = a.N;
        int rc = h2_launch<VN_EPI_STORE, MI>(ctx, q, nsplit, s);
        if (rc) return rc;
        return vn_launch_splitk_reduce(ctx, ctx->x3_ws, nsplit, a.C, a.M, a.N, a.ldc, epilogue == VN_EPI_RESIDUAL, s);
    }
    switch (epilogue) {
        case VN_EPI_STORE: return h2_launch<VN_EPI_STORE, MI>(ctx, a, 1, s);
        case VN_EPI_BIAS:
            if (!a.bias) return vn_fail(ctx, VN_ERR_INVALID, "gemm_h2: bias epilogue needs bias%s", "");
            return h2_launch<VN_EPI_BIAS, MI>(ctx, a, 1, s);
        case VN_EPI_RESIDUAL: return h2_launch<VN_EPI_RESIDUAL, MI>(ctx, a, 1, s);
        case VN_EPI_GEGLU: return h2_launch<VN_EPI_GEGLU, MI>(ctx, a, 1, s);
        case VN_EPI_QKV: return h2_launch<VN_EPI_QKV, MI>(ctx, a, 1, s);
    }
    return vn_fail(ctx, VN_ERR_INVALID, "gemm_h2: unknown epilogue %s%ld", "", epilogue);
}

// tile_m: 128 or 256 rows per block; nsplit: 1, or 2..8 k-splits (store / residual epilogues)
static int vn_launch_gemm_h2(vn_ctx* ctx, const vn_gemm_args& a, int epilogue, int tile_m, int nsplit, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return vn_fail(ctx, VN_ERR_INVALID, "gemm_h2: empty problem%s", "");
    if (a.K % H2_KT) return vn_fail(ctx, VN_ERR_INVALID, "gemm_h2: K=%s%ld must be a multiple of 32", "", a.K);
    if (a.N % 64) return vn_fail(ctx, VN_ERR_INVALID, "gemm_h2: N=%s%ld must be a multiple of 64", "", a.N);
    if (a.a_plane <= 0 || a.w_plane <= 0 || (a.a_plane & 7) || (a.w_plane & 7))
        return vn_fail(ctx, VN_ERR_INVALID, "gemm_h2: plane strides must be positive multiples of 8 elements%s", "");
    if (((uintptr_t)a.A | (uintptr_t)a.W) & 15) return vn_fail(ctx, VN_ERR_INVALID, "gemm_h2: operands must be 16-byte aligned%s", "");
    if (nsplit < 1 || nsplit > 8 || a.K / H2_KT < nsplit) return vn_fail(ctx, VN_ERR_INVALID, "gemm_h2: bad split count %s%ld", "", nsplit);
    if (tile_m == 256) return h2_dispatch<2>(ctx, a, epilogue, nsplit, s);
    if (tile_m == 128) return h2_dispatch<1>(ctx, a, epilogue, nsplit, s);
    return vn_fail(ctx, VN_ERR_INVALID, "gemm_h2: tile_m must be 128 or 256 (got %s%ld)", "", tile_m);
}

// model-path launcher (engine.hip, precision "f16x2"): tile height from VN_H2_TILE (128 | 256, default 128), split count for
// the store / residual epilogues from the same cost model as gemm_x3.hip with this kernel's (estimated) k-tile time
int vn_launch_gemm_h2_auto(vn_ctx* ctx, const vn_gemm_args& a, int epilogue, hipStream_t s) {
    static const int tile_m = [] { const char* e = getenv("VN_H2_TILE"); return e && atoi(e) == 256 ? 256 : 128; }();
    static const int forced = [] { const char* e = getenv("VN_H2_SPLITK"); return e ? atoi(e) : -1; }();
    int ns = 1;
    if ((epilogue == VN_EPI_STORE || epilogue == VN_EPI_RESIDUAL) && forced != 0 && forced != 1 && !(a.N & 3) && !(a.ldc & 3)) {
        const int tiles = vn_cdiv(a.M, tile_m) * vn_cdiv(a.N, 128), nk = a.K / H2_KT;
        const double t_ktile = tile_m == 256 ? 1.4 : 0.8;                     // us per k-tile and round: to be calibrated
        double best = 1e300;
        for (int c = 1; c <= 4; c *= 2) {
            if (c > 1 && (nk / c < 8 || (double)c * a.M * a.N > (double)H2_WS_FLOATS)) continue;
            double cost = ceil(tiles * c / 256.0) * (nk / (double)c) * t_ktile;
            if (c > 1) cost += (c + (epilogue == VN_EPI_RESIDUAL ? 2 : 1)) * 4.0 * a.M * (double)a.N / 3.5e6;
            if (forced > 1) cost = (c == forced) ? 0 : 1e299;
            if (cost < best) { best = cost; ns = c; }
        }
    }
    const double n_out = (epilogue == VN_EPI_GEGLU) ? a.N / 2 : a.N;
    const double bytes = 4.0 * ((double)a.M * a.K + (double)a.N * a.K) + 4.0 * (double)a.M * n_out * (epilogue == VN_EPI_RESIDUAL ? 2 : 1);
    const int pi = vn_prof_pre(ctx, 0, 2.0 * a.M * (double)a.N * a.K, s, bytes);       // algorithmic (fp32-equivalent) flops
    const int rc = vn_launch_gemm_h2(ctx, a, epilogue, tile_m, ns, s);
    vn_prof_post(ctx, pi, s);
    return rc;
}

// ---- RMSNorm writing the two fp16 planes directly (VN_H2_FUSE=1; same arithmetic as vn_rmsnorm_kernel in elementwise.hip:
// one wave per row, float4 loads, shuffle reduce, exact 1/sqrt) — saves the separate split pass over the normalised rows
template <int VEC>
__global__ __launch_bounds__(256) void vn_rmsnorm_h2_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            uint16_t* __restrict__ y2, long plane, int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const f32x4* xr = (const f32x4*)(x + (size_t)row * D);
    f32x4 v[VEC];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        v[i] = xr[lane + 64 * i];
        ss += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float rstd = 1.0f / sqrtf(ss / (float)D + eps);
    const f32x4* wr = (const f32x4*)w;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        const f32x4 ww = wr[lane + 64 * i];
        uint16_t t[2][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) vn_split2h(ww[e] * (v[i][e] * rstd), t[0][e], t[1][e]);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            uint2 pk;
            pk.x = t[q][0] | ((unsigned)t[q][1] << 16);
            pk.y = t[q][2] | ((unsigned)t[q][3] << 16);
            *(uint2*)(y2 + q * plane + (size_t)row * D + 4 * (lane + 64 * i)) = pk;
        }
    }
}

int vn_launch_rmsnorm_h2(vn_ctx* ctx, const float* x, const float* w, uint16_t* y2, long plane, int rows, int D, float eps,
                         hipStream_t s) {
    if (rows <= 0) return VN_OK;
    const dim3 grid(vn_cdiv(rows, 4)), block(256);
    if (D == 1280) hipLaunchKernelGGL(vn_rmsnorm_h2_kernel<5>, grid, block, 0, s, x, w, y2, plane, rows, D, eps);
    else if (D == 256) hipLaunchKernelGGL(vn_rmsnorm_h2_kernel<1>, grid, block, 0, s, x, w, y2, plane, rows, D, eps);
    else return vn_fail(ctx, VN_ERR_UNSUPPORTED, "rmsnorm_h2: D=%s%ld unsupported (256 or 1280)", "", D);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

// ---- plane builder: dst[0][i] = h0, dst[plane_stride + i] = h1
__global__ __launch_bounds__(256) void vn_split2h_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, long n4,
                                                         long plane) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256L) {
        const f32x4 v = ((const f32x4*)src)[i];
        uint16_t t[2][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) vn_split2h(v[e], t[0][e], t[1][e]);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            uint2 pk;
            pk.x = t[q][0] | ((unsigned)t[q][1] << 16);
            pk.y = t[q][2] | ((unsigned)t[q][3] << 16);
            *(uint2*)(dst + q * plane + 4 * i) = pk;
        }
    }
}

extern "C" int vn_split2h_f32(vn_ctx* ctx, const float* src, void* dst16, int64_t n, int64_t plane_stride, void* stream) {
    if (!ctx || !src || !dst16 || n <= 0 || (n & 3) || plane_stride < n || (plane_stride & 7)) return VN_ERR_INVALID;
    const long n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(vn_split2h_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (uint16_t*)dst16, n4, (long)plane_stride);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

int vn_launch_split2h(vn_ctx* ctx, const float* src, uint16_t* dst, long n, long plane_stride, hipStream_t s) {
    return vn_split2h_f32(ctx, src, dst, n, plane_stride, s);
}

// single-op entry (tests / tuning): A2 [2][M][K] and W2 [2][N][K] fp16 split planes -> fp32 C
extern "C" int vn_gemm_f16x2(vn_ctx* ctx, const void* A2, int64_t a_plane, const void* W2, int64_t w_plane, const float* bias,
                             float* C, int M, int N, int K, int epilogue, int tile_m, int nsplit, void* stream) {
    if (!ctx || !A2 || !W2 || !C) return VN_ERR_INVALID;
    if (epilogue < VN_EPI_STORE || epilogue > VN_EPI_GEGLU) return vn_fail(ctx, VN_ERR_INVALID, "bad epilogue%s", "");
    vn_gemm_args a{};
    a.A = (const float*)A2; a.W = (const float*)W2; a.bias = bias; a.C = C; a.M = M; a.N = N; a.K = K;
    a.ldc = epilogue == VN_EPI_GEGLU ? N / 2 : N;
    a.bf16 = 3; a.a_plane = a_plane; a.w_plane = w_plane;
    return vn_launch_gemm_h2(ctx, a, epilogue, tile_m, nsplit, (hipStream_t)stream);
}
