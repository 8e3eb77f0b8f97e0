// Exact-fp32 GEMM for gfx950:  C[M][N] (op)= A[M][K] * W[N][K]^T   (torch F.linear semantics)
//
// Replaces the reference's nn.Linear / lora.Linear / 1x1 WNConv1d calls on the hot path
// (vampnet/modules/transformer.py:81-84 FFN, :229-231 QKV, :255 fc, :632 classifier).
//
// Design (MI355X_MICROARCH.md / cdna_hip_programming.md §3 "FP32-input MFMA"):
//   * v_mfma_f32_32x32x2_f32: exact f32 (bitwise an fmaf chain), 64 cycles/instr/SIMD = the f32
//     vector peak (157.3 TF chip).  One wave per SIMD with >= 1 independent 32x32 accumulator already
//     issues back to back (dependent latency == issue interval == 64 cycles).
//   * block = 256 threads = 2x2 waves; block tile BM x BN (128x128 default), wave tile BM/2 x BN/2 made of
//     32x32 MFMA tiles; BK = 32 floats (one 128-byte line per row per k-tile).
//   * global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB = 8 rows x 128 B per wave instruction),
//     double-buffered: tile t+1 streams in while tile t is multiplied; one barrier per k-tile.
//   * LDS image is lane-linear (DMA constraint), so the bank-conflict swizzle is applied on the SOURCE
//     address: the 16-B slot s of row r is stored at slot s ^ (r & 7) (guide §5.4 rule 21), and
//     fragments are read with the same XOR by ds_read_b128.
//   * k-order inside a tile: lane half h = lane>>5 reads float4 at k = 8*s + 4*h .. +3 (s = 0..3) and
//     feeds element e to MFMA #e, i.e. MFMA (s,e) contracts k-pair {8s+e, 8s+4+e}.  Any bijection of k
//     is a valid fp32 summation order (the reference's own order is MKL's, not specified).
//   * epilogues fused: bias, residual add, GEGLU gate (value/gate columns interleaved at pack time so
//     they sit in the same lane of adjacent MFMA tiles), QKV head-major scatter.
//   * rows >= M are clamped on load (duplicate last row) and masked on store, so M needs no padding.
//   * blockIdx -> tile mapping is XCD-aware (block b runs on XCD b % 8): each XCD owns a contiguous
//     range of the tile grid so neighbouring tiles share A/W panels in one 4 MiB L2.
#include "vn_common.h"

#define BK 32

template <int BM, int BN>
struct GemmCfg {
    static constexpr int MI = BM / 64;            // 32x32 tiles per wave along M
    static constexpr int NI = BN / 64;            // along N
    static constexpr int A_FLOATS = BM * BK;
    static constexpr int B_FLOATS = BN * BK;
    static constexpr int STAGE_FLOATS = A_FLOATS + B_FLOATS;
    static constexpr int LDS_BYTES = 2 * STAGE_FLOATS * 4;
    static constexpr int A_INSTR = BM / 32;       // DMA wave-instructions per wave for A (BM/8 total / 4 waves)
    static constexpr int B_INSTR = BN / 32;
};

__device__ __forceinline__ float vn_gelu_tanh(float x) {
    // activations.py:16-26: 0.5*x*(1+tanh(sqrt(2/pi)*(x+0.044715*x^3)))
    const float c = 0.7978845608028654f;
    float x3 = x * x * x;
    return 0.5f * x * (1.0f + tanhf(c * (x + 0.044715f * x3)));
}

template <int BM, int BN, int EPI>
__global__ __launch_bounds__(256, 2) void vn_gemm_f32_kernel(vn_gemm_args p, int tiles_m, int tiles_n) {
    using Cfg = GemmCfg<BM, BN>;
    constexpr int MI = Cfg::MI, NI = Cfg::NI;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- XCD-aware tile mapping (bijective for any grid size; guide §5 "XCD swizzle must be bijective")
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, idx = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // walk tiles column-panel-major inside the XCD chunk: consecutive blocks share the W panel
    const int tn = bid / tiles_m;
    const int tm = bid - tn * tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- per-lane DMA source pointers (chunk p = 16-byte slot index inside the tile image)
    const float* srcA[Cfg::A_INSTR];
    const float* srcB[Cfg::B_INSTR];
#pragma unroll
    for (int q = 0; q < Cfg::A_INSTR; ++q) {
        const int pidx = (wave * Cfg::A_INSTR + q) * 64 + lane;
        const int row = pidx >> 3, slot = (pidx & 7) ^ (row & 7);
        int gm = m0 + row;
        gm = gm < p.M ? gm : p.M - 1;
        srcA[q] = p.A + (size_t)gm * p.K + slot * 4;
    }
#pragma unroll
    for (int q = 0; q < Cfg::B_INSTR; ++q) {
        const int pidx = (wave * Cfg::B_INSTR + q) * 64 + lane;
        const int row = pidx >> 3, slot = (pidx & 7) ^ (row & 7);
        int gn = n0 + row;
        gn = gn < p.N ? gn : p.N - 1;
        srcB[q] = p.W + (size_t)gn * p.K + slot * 4;
    }

    auto stage = [&](int buf, int k0) {
        float* dA = lds + buf * Cfg::STAGE_FLOATS;
        float* dB = dA + Cfg::A_FLOATS;
#pragma unroll
        for (int q = 0; q < Cfg::A_INSTR; ++q)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(srcA[q] + k0),
                (__attribute__((address_space(3))) void*)(dA + (wave * Cfg::A_INSTR + q) * 256), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < Cfg::B_INSTR; ++q)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(srcB[q] + k0),
                (__attribute__((address_space(3))) void*)(dB + (wave * Cfg::B_INSTR + q) * 256), 16, 0, 0);
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // fragment read offsets (floats) inside a stage: row*32 + ((2s + h) ^ (row&7))*4 ; row&7 == lane&7
    const int l31 = lane & 31, h = lane >> 5, sw = lane & 7;
    const int aRow = (wm * (BM / 2) + l31) * BK;
    const int bRow = (wn * (BN / 2) + l31) * BK;

    const int nk = p.K / BK;
    stage(0, 0);
    __syncthreads();   // glds in flight -> hipcc emits vmcnt(0) before the barrier (guide §5)

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, (kt + 1) * BK);
        const float* sA = lds + cur * Cfg::STAGE_FLOATS;
        const float* sB = sA + Cfg::A_FLOATS;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int off = ((2 * s + h) ^ sw) * 4;
            f32x4 a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = *(const f32x4*)(sA + aRow + i * 32 * BK + off);
#pragma unroll
            for (int j = 0; j < NI; ++j) b[j] = *(const f32x4*)(sB + bRow + j * 32 * BK + off);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
        __syncthreads();   // tile kt consumed by all waves; tile kt+1 landed (vmcnt(0) + barrier)
    }

    // ---- epilogue.  C/D map of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int colw = n0 + wn * (BN / 2) + l31;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row >= p.M) continue;
            if constexpr (EPI == VN_EPI_GEGLU) {
                static_assert(NI == 2 || EPI != VN_EPI_GEGLU, "GEGLU needs value+gate tiles in one wave");
                // wave tile = 64 packed columns = 32 value (tile j=0) + 32 gate (tile j=1)
                const int ocol = (n0 + wn * (BN / 2)) / 2 + l31;
                const float v = acc[i][0][r], g = acc[i][NI - 1][r];
                p.C[(size_t)row * p.ldc + ocol] = v * vn_gelu_tanh(g);
            } else {
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int col = colw + j * 32;
                    if (col >= p.N) continue;
                    float v = acc[i][j][r];
                    if constexpr (EPI == VN_EPI_STORE) {
                        p.C[(size_t)row * p.ldc + col] = v;
                    } else if constexpr (EPI == VN_EPI_BIAS) {
                        p.C[(size_t)row * p.ldc + col] = v + p.bias[col];
                    } else if constexpr (EPI == VN_EPI_RESIDUAL) {
                        float* c = p.C + (size_t)row * p.ldc + col;
                        *c = *c + v;
                    } else if constexpr (EPI == VN_EPI_QKV) {
                        // col in [0, 3D): which = col / D; head = (col % D) / 64; d = col % 64
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

template <int BM, int BN, int EPI>
static int launch_cfg(vn_ctx* ctx, const vn_gemm_args& a, hipStream_t s) {
    using Cfg = GemmCfg<BM, BN>;
    const int tiles_m = vn_cdiv(a.M, BM), tiles_n = vn_cdiv(a.N, BN);
    auto kern = vn_gemm_f32_kernel<BM, BN, EPI>;
    static bool attr_set = false;
    if (!attr_set) {
        VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                              Cfg::LDS_BYTES));
        attr_set = true;
    }
    const int pi = vn_prof_pre(ctx, 0, 2.0 * a.M * (double)a.N * a.K, s);
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), Cfg::LDS_BYTES, s, a, tiles_m, tiles_n);
    vn_prof_post(ctx, pi, s);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

template <int EPI>
static int launch_epi(vn_ctx* ctx, const vn_gemm_args& a, hipStream_t s) {
    // tile choice: fill 256 CUs.  128x128 once that yields >= 2 waves of blocks, else 64-row tiles.
    const long big = (long)vn_cdiv(a.M, 128) * vn_cdiv(a.N, 128);
    if constexpr (EPI == VN_EPI_GEGLU) {
        if (big >= 512) return launch_cfg<128, 128, EPI>(ctx, a, s);
        return launch_cfg<64, 128, EPI>(ctx, a, s);
    } else {
        if (big >= 512) return launch_cfg<128, 128, EPI>(ctx, a, s);
        if ((long)vn_cdiv(a.M, 64) * vn_cdiv(a.N, 128) >= 256) return launch_cfg<64, 128, EPI>(ctx, a, s);
        return launch_cfg<64, 64, EPI>(ctx, a, s);
    }
}

int vn_launch_gemm_f32(vn_ctx* ctx, const vn_gemm_args& a, int epilogue, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return vn_fail(ctx, VN_ERR_INVALID, "gemm: empty problem%s", "");
    if (a.K % BK != 0) return vn_fail(ctx, VN_ERR_INVALID, "gemm: K=%s%ld must be a multiple of 32", "", a.K);
    if (a.N % 64 != 0) return vn_fail(ctx, VN_ERR_INVALID, "gemm: N=%s%ld must be a multiple of 64", "", a.N);
    switch (epilogue) {
        case VN_EPI_STORE: return launch_epi<VN_EPI_STORE>(ctx, a, s);
        case VN_EPI_BIAS: return launch_epi<VN_EPI_BIAS>(ctx, a, s);
        case VN_EPI_RESIDUAL: return launch_epi<VN_EPI_RESIDUAL>(ctx, a, s);
        case VN_EPI_GEGLU:
            if (a.N % 128 != 0) return vn_fail(ctx, VN_ERR_INVALID, "gemm/geglu: N=%s%ld must be a multiple of 128", "", a.N);
            return launch_epi<VN_EPI_GEGLU>(ctx, a, s);
        case VN_EPI_QKV: return launch_epi<VN_EPI_QKV>(ctx, a, s);
    }
    return vn_fail(ctx, VN_ERR_INVALID, "gemm: unknown epilogue %s%ld", "", epilogue);
}
