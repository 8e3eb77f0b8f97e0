// fp32 1-D convolution stack of the DAC codec (Interface.encode / Interface.decode; SURVEY.md §8 rows a18/a19,
// App. D) for gfx950.  PARITY UNPINNED: the codec source (`lac`) is not in /root/reference; these kernels implement
// the published DAC layers and are checked against oracle/dac_oracle.py.
//
// Layout: activations are channels-last [B][T][C] fp32, so a convolution tap is a plain row shift:
//   y[b][t'][co] = bias[co] + sum_j sum_ci W[co][j][ci] * x[b][t'*in_stride + j*dil - pad][ci]
// is a GEMM with M = B*T' rows, N = C_out, K = taps*C_in whose A row pointer moves with the tap.  The kernel is the
// exact-f32 MFMA GEMM of gemm_f32.hip (v_mfma_f32_32x32x2_f32, LDS-DMA double buffering, source-side XOR swizzle)
// with the A source address recomputed per k-tile; rows that fall in the zero padding (or beyond M) DMA from a
// zero page instead.  Weights are packed [C_out][taps][C_in] (C_in % 32 == 0) by the host.
//   * WNConv1d k=7 dilated / k=3 / k=1 / strided k=2s (encoder down-sampling): in_stride = s.
//   * WNConvTranspose1d k=2s stride s pad ceil(s/2) (decoder up-sampling): output positions split into s phases
//     r = (t + pad) mod s; phase r is a 2-tap convolution (taps j = r, r + s reading x[t'], x[t' - 1]) writing rows
//     t = t'*s + r - pad: launched as s GEMMs (out_stride = s, out_off = r - pad, dil = -1).
//   * fused epilogue: + bias, + residual (ResidualUnit), tanh, and the NEXT layer's Snake1d pre-activation
//     snake(v) = v + sin(alpha v)^2 / (alpha + 1e-9)  (vampnet/modules/layers.py:12-18) written as a second tensor,
//     so the 170 MB/item audio-rate activations are never re-read by a stand-alone activation kernel.
// Algorithmic FLOPs 2*M*N*K; bytes: x read once per tap-group from L2, y written once (+ y2).
#include "vn_common.h"

#define BK 32

struct vn_conv_args {
    const float* x;       // [B][T_in][C_in]
    const float* w;       // [C_out][taps][C_in]
    const float* bias;    // [C_out] or null
    const float* resid;   // [B][T_out][C_out] or null
    const float* alpha;   // [C_out] (needed iff y2)
    float* y;             // [B][T_out][C_out] raw result or null
    float* y2;            // snake(result) or null
    uint16_t* y2_16;      // snake(result) as three split bf16 planes (y2_plane elements apart) for a consumer on the bf16x3 pipe, or null
    long y2_plane;
    const float* zeros;   // >= 128 B of zeros
    int B, T_in, T_rows, T_out, C_in, C_out, taps;
    int in_stride, dil, pad;       // t_in  = t' * in_stride + j * dil - pad
    int out_stride, out_off;       // t_out = t' * out_stride + out_off
    int act;                       // 1: tanh on the result
    unsigned* sat;                 // the context's saturation words (fp16 planes only; set by the launcher)
};

// WM = waves along M (4 waves per block, WN = 4 / WM along N): 2 x 2 wave grid for the 128/64-wide tiles, 4 x 1 for the
// 96-wide tile of the C_out = 96 audio-rate layers (each wave 32 rows x 96 columns: no padded MFMA columns).
template <int BM, int BN, int WM = 2>
__global__ __launch_bounds__(256, 2) void vn_conv1d_f32_kernel(vn_conv_args p, int tiles_m, int tiles_n) {
    constexpr int WN = 4 / WM, WTM = BM / WM, WTN = BN / WN;
    constexpr int MI = WTM / 32, NI = WTN / 32;
    static_assert(MI >= 1 && NI >= 1 && MI * 32 == WTM && NI * 32 == WTN, "wave tile must be a multiple of 32 x 32");
    constexpr int A_FLOATS = BM * BK, B_FLOATS = BN * BK, STAGE = A_FLOATS + B_FLOATS;
    constexpr int A_INSTR = BM / 32, B_INSTR = BN / 32;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int M = p.B * p.T_rows, K = p.taps * p.C_in;
    // XCD-contiguous, row-panel-major walk (consecutive blocks share the weight panel, which is tiny here)
    int bid = blockIdx.x;
    {
        const int nwg = tiles_m * tiles_n, xcd = bid & 7, idx = bid >> 3, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // per-lane DMA rows: (batch, t') of each A row this lane fetches, and its 16-byte slot
    int a_b[A_INSTR], a_t[A_INSTR], a_slot[A_INSTR];
    const float* srcB[B_INSTR];
#pragma unroll
    for (int q = 0; q < A_INSTR; ++q) {
        const int pidx = (wave * A_INSTR + q) * 64 + lane;
        const int row = pidx >> 3;
        a_slot[q] = ((pidx & 7) ^ (row & 7)) * 4;
        const int gm = m0 + row;
        if (gm < M) { a_b[q] = gm / p.T_rows; a_t[q] = gm - a_b[q] * p.T_rows; }
        else { a_b[q] = -1; a_t[q] = 0; }
    }
#pragma unroll
    for (int q = 0; q < B_INSTR; ++q) {
        const int pidx = (wave * B_INSTR + q) * 64 + lane;
        const int row = pidx >> 3, slot = (pidx & 7) ^ (row & 7);
        int gn = n0 + row;
        gn = gn < p.C_out ? gn : p.C_out - 1;
        srcB[q] = p.w + (size_t)gn * K + slot * 4;
    }
    const int cpt = p.C_in / BK;      // k-tiles per tap
    auto stage = [&](int buf, int kt) {
        float* dA = lds + buf * STAGE;
        float* dB = dA + A_FLOATS;
        const int j = kt / cpt, c0 = (kt - j * cpt) * BK;
#pragma unroll
        for (int q = 0; q < A_INSTR; ++q) {
            const int t_in = a_t[q] * p.in_stride + j * p.dil - p.pad;
            const bool ok = a_b[q] >= 0 && t_in >= 0 && t_in < p.T_in;
            const float* src = ok ? p.x + ((size_t)a_b[q] * p.T_in + t_in) * p.C_in + c0 + a_slot[q] : p.zeros + a_slot[q];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dA + (wave * A_INSTR + q) * 256),
                                             16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < B_INSTR; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcB[q] + kt * BK),
                                             (__attribute__((address_space(3))) void*)(dB + (wave * B_INSTR + q) * 256),
                                             16, 0, 0);
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jn = 0; jn < NI; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.0f;

    const int l31 = lane & 31, h = lane >> 5, sw = lane & 7;
    const int aRow = (wm * WTM + l31) * BK, bRow = (wn * WTN + l31) * BK;
    const int nk = K / BK;
    stage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const float* sA = lds + cur * STAGE;
        const float* sB = sA + A_FLOATS;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int off = ((2 * s + h) ^ sw) * 4;
            f32x4 a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = *(const f32x4*)(sA + aRow + i * 32 * BK + off);
#pragma unroll
            for (int jn = 0; jn < NI; ++jn) b[jn] = *(const f32x4*)(sB + bRow + jn * 32 * BK + off);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int jn = 0; jn < NI; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[jn][e], acc[i][jn], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue, two phases through LDS (free after the k-loop; BM*BN*4 <= the stage buffers):
    //  1. accumulators -> LDS tile [BM][BN] (C/D map: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5));
    //     fully unrolled with compile-time accumulator indices and nothing else in the loop body;
    //  2. a rolled loop over the tile, 256 consecutive columns-within-rows per pass: bias, residual, tanh / snake and
    //     row-contiguous (coalesced) global accesses.  Keeping the transcendental code out of the unrolled nest keeps
    //     the accumulators in registers (an inlined sinf x 64 made hipcc spill them to scratch).
    float* tile = lds;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jn = 0; jn < NI; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int cl = wn * WTN + jn * 32 + l31;
                tile[rl * BN + cl] = acc[i][jn][r];
            }
    __syncthreads();
    // phase 2: each thread owns 4 consecutive output channels (bias / alpha / 1/(alpha+1e-9) live in registers) and walks
    // down the tile rows; (batch, t') advance incrementally (no per-element integer division); 16-byte global accesses.
    constexpr int CG = BN / 4, RPP = 256 / CG;            // column groups per row, rows per pass (BN = 96: 240 threads work)
    const int cg = tid % CG, col = n0 + 4 * cg;
    if (col < p.C_out && tid < CG * RPP) {                 // C_out % 4 == 0: a float4 is valid as a whole
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, al4 = {1.f, 1.f, 1.f, 1.f}, inv4 = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bias4 = *(const f32x4*)(p.bias + col);
        if (p.y2 || p.y2_16) {
            al4 = *(const f32x4*)(p.alpha + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) inv4[e] = 1.0f / (al4[e] + 1e-9f);
        }
        int rl = tid / CG;
        int m = m0 + rl;
        int b = m / p.T_rows, tq = m - b * p.T_rows;
        bool bad = false;                                  // fp16 planes: saturation ledger (vn_common.h)
        for (; rl < BM && m < M; rl += RPP, m += RPP, tq += RPP) {
            while (tq >= p.T_rows) { tq -= p.T_rows; ++b; }
            const int t_out = tq * p.out_stride + p.out_off;
            if (t_out < 0 || t_out >= p.T_out) continue;
            const size_t o = ((size_t)b * p.T_out + t_out) * p.C_out + col;
            f32x4 v = *(const f32x4*)(tile + rl * BN + 4 * cg);
            v[0] += bias4[0]; v[1] += bias4[1]; v[2] += bias4[2]; v[3] += bias4[3];
            if (p.resid) {
                const f32x4 r4 = *(const f32x4*)(p.resid + o);
                v[0] += r4[0]; v[1] += r4[1]; v[2] += r4[2]; v[3] += r4[3];
            }
            if (p.act == 1) { v[0] = tanhf(v[0]); v[1] = tanhf(v[1]); v[2] = tanhf(v[2]); v[3] = tanhf(v[3]); }
            if (p.y) *(f32x4*)(p.y + o) = v;
            if (p.y2 || p.y2_16) {
                f32x4 w4;
#pragma unroll
                for (int e = 0; e < 4; ++e) w4[e] = vn_snake(v[e], al4[e], inv4[e]);
                if (p.y2) *(f32x4*)(p.y2 + o) = w4;
                if (p.y2_16) {                                  // y2_plane < 0: two fp16 planes (f16x2), -y2_plane apart
                    if (p.y2_plane < 0) vn_store_h2x4(p.y2_16 + o, -p.y2_plane, w4, bad);
                    else vn_store_bf16x4(p.y2_16 + o, p.y2_plane, w4);
                }
            }
        }
        vn_sat_report(p.sat, VN_SAT_OPERAND, bad);
    }
}

template <int BM, int BN, int WM = 2>
static int launch_conv(vn_ctx* ctx, const vn_conv_args& a_in, hipStream_t s) {
    vn_conv_args a = a_in;
    a.sat = ctx->sat;
    const int M = a.B * a.T_rows;
    const int tiles_m = vn_cdiv(M, BM), tiles_n = vn_cdiv(a.C_out, BN);
    constexpr int LDS = 2 * (BM + BN) * BK * 4;
    const double bytes = 4.0 * ((double)a.B * a.T_in * a.C_in + (double)a.C_out * a.taps * a.C_in +
                                (double)M * a.C_out * ((a.y ? 1 : 0) + (a.y2 ? 1 : 0) + (a.y2_16 ? 1.5 : 0) + (a.resid ? 1 : 0)));
    const double fl = 2.0 * M * (double)a.C_out * a.taps * a.C_in;
    const int pi = vn_prof_pre(ctx, vn_conv_class(fl, bytes, VN_PROF_CONV_F32, 157.3), fl, s, bytes);
    hipLaunchKernelGGL((vn_conv1d_f32_kernel<BM, BN, WM>), dim3(tiles_m * tiles_n), dim3(256), LDS, s, a, tiles_m, tiles_n);
    vn_prof_post(ctx, pi, s);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

static int zero_page(vn_ctx* ctx) {
    if (ctx->zero_page) return VN_OK;
    VN_HIP_CHECK(ctx, vn_dev_malloc((void**)&ctx->zero_page, 1024));
    VN_HIP_CHECK(ctx, hipMemset(ctx->zero_page, 0, 1024));
    return VN_OK;
}

extern "C" int vn_conv1d_f32(vn_ctx* ctx, const float* x, const float* w, const float* bias, const float* resid,
                             const float* alpha, float* y, float* y2, void* y2_16, int64_t y2_plane, int B, int T_in, int T_rows,
                             int T_out, int C_in, int C_out, int taps, int in_stride, int dil, int pad, int out_stride, int out_off,
                             int act, void* stream) {
    if (!ctx || !x || !w || (!y && !y2 && !y2_16)) return VN_ERR_INVALID;
    if (y2_16 && (y2_plane == 0 || ((y2_plane < 0 ? -y2_plane : y2_plane) & 3) || ((uintptr_t)y2_16 & 7)))
        return vn_fail(ctx, VN_ERR_INVALID, "conv1d: the plane output needs an 8-byte aligned base and a plane stride %% 4 == 0%s", "");
    if (C_in % BK) return vn_fail(ctx, VN_ERR_INVALID, "conv1d: C_in=%s%ld must be a multiple of 32", "", C_in);
    if (C_out % 4) return vn_fail(ctx, VN_ERR_INVALID, "conv1d: C_out=%s%ld must be a multiple of 4", "", C_out);
    if ((y2 || y2_16) && !alpha) return vn_fail(ctx, VN_ERR_INVALID, "conv1d: snake output needs alpha%s", "");
    if (B <= 0 || T_rows <= 0 || C_out <= 0 || taps <= 0) return vn_fail(ctx, VN_ERR_INVALID, "conv1d: empty problem%s", "");
    int rc = zero_page(ctx);
    if (rc) return rc;
    vn_conv_args a{x, w, bias, resid, alpha, y, y2, (uint16_t*)y2_16, (long)y2_plane, ctx->zero_page, B, T_in, T_rows, T_out, C_in, C_out,
                   taps, in_stride, dil, pad, out_stride, out_off, act};
    hipStream_t s = (hipStream_t)stream;
    const long M = (long)B * T_rows;
    // N tile: 128 wastes (128 - C_out % 128) columns of the last tile; 64 fits 64-multiples exactly (C_out = 192: 3 x 64
    // instead of 128 + a half-empty 128: +25 % useful MFMA work per launched tile)
    if (C_out % 96 == 0 && C_out % 64 != 0 && (long)vn_cdiv((int)M, 128) * (C_out / 96) >= 384)
        return launch_conv<128, 96, 4>(ctx, a, s);          // C_out = 96: exact 96-wide tiles
    const int pad128 = vn_cdiv(C_out, 128) * 128 - C_out, pad64 = vn_cdiv(C_out, 64) * 64 - C_out;
    const bool wide = C_out > 64 && pad128 <= pad64;
    if (wide) {
        if ((long)vn_cdiv((int)M, 128) * vn_cdiv(C_out, 128) >= 384) return launch_conv<128, 128>(ctx, a, s);
        return launch_conv<64, 128>(ctx, a, s);
    }
    if ((long)vn_cdiv((int)M, 128) * vn_cdiv(C_out, 64) >= 384) return launch_conv<128, 64>(ctx, a, s);
    return launch_conv<64, 64>(ctx, a, s);
}

// ---------------------------------------------------------------------------------------------
// Encoder stem: WNConv1d(1 -> C, k = 7, pad 3) on the raw waveform (K = 7: not MFMA-shaped, HBM-bound on the write).
//   y[b][t][c] = bias[c] + sum_j w[c][j] * x[b][t + j - 3] ; y2 = snake(y, alpha)
// ---------------------------------------------------------------------------------------------
// A thread owns FOUR consecutive channels of one sample (C % 4 == 0): the seven waveform taps are loaded once for the four, the weights as
// their 28 scalars, and both outputs go out as 16-byte stores (one channel per thread issued seven loads and a 64-bit division per 4-byte
// store: 1.5 ms for the 0.9 + 0.9 GB it writes at B = 8 — profiles/r05_codec_kernel_trace_bf16x3.txt).
__global__ __launch_bounds__(256) void vn_dac_conv_in_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ alpha, float* __restrict__ y,
                                                             float* __restrict__ y2, int B, int T, int C) {
    const int c4n = C >> 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // over B*T*(C/4), channel group fastest (coalesced writes)
    if (i >= (long)B * T * c4n) return;
    const int c = (int)(i % c4n) * 4;
    const long bt = i / c4n;
    const int t = (int)(bt % T);
    const float* xb = x + (bt - t);
    float xv[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int ti = t + j - 3;
        xv[j] = (ti >= 0 && ti < T) ? xb[ti] : 0.f;
    }
    f32x4 v = *(const f32x4*)(bias + c);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < 7; ++j) v[e] = fmaf(w[(c + e) * 7 + j], xv[j], v[e]);      // taps outside the clip multiply a zero: same sum
    if (y) *(f32x4*)(y + bt * C + c) = v;
    if (y2) {
        const f32x4 al = *(const f32x4*)(alpha + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = vn_snake(v[e], al[e], 1.0f / (al[e] + 1e-9f));
        *(f32x4*)(y2 + bt * C + c) = o;
    }
}

extern "C" int vn_dac_conv_in_f32(vn_ctx* ctx, const float* x, const float* w, const float* bias, const float* alpha,
                                  float* y, float* y2, int B, int T, int C, void* stream) {
    if (!ctx || !x || !w || !bias || (!y && !y2) || (y2 && !alpha)) return VN_ERR_INVALID;
    if ((C & 3) || (((uintptr_t)bias | (uintptr_t)alpha | (uintptr_t)y | (uintptr_t)y2) & 15))
        return vn_fail(ctx, VN_ERR_INVALID, "conv_in: C=%s%ld must be a multiple of 4 and bias / alpha / outputs 16-byte aligned", "", C);
    const long n = (long)B * T * (C >> 2);
    hipLaunchKernelGGL(vn_dac_conv_in_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, w,
                       bias, alpha, y, y2, B, T, C);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

// ---------------------------------------------------------------------------------------------
// Decoder head: WNConv1d(C -> 1, k = 7, pad 3) + tanh on the (already snake-activated) audio-rate tensor.
// N = 1: a dot product of 7*C values per sample, HBM/L2-bound (each input row is read by 7 outputs; a block stages
// 64 + 6 rows in LDS).  w packed [7][C].
// ---------------------------------------------------------------------------------------------
#define COUT_T 64
// LDS rows are C + 4 floats apart: lane (sample o, part p) reads 16 bytes at (o + j) (C + 4) + 16 i + 4 p — for C = 96 the sixteen samples
// of a wave start 36 banks apart (all distinct multiples of 4), so a wave's read covers the 64 banks once (row pitch C: 32 o mod 64 — the
// wave hit 8 banks, and the kernel took 1.9 ms for the 1.35 GB it reads at B = 8)
__global__ __launch_bounds__(256) void vn_dac_conv_out_kernel(const float* __restrict__ xs, const float* __restrict__ w,
                                                              float bias, float* __restrict__ y, int B, int T, int C) {
    extern __shared__ __attribute__((aligned(16))) float sm[];      // [(COUT_T + 6)][C + 4]
    const int b = blockIdx.y, t0 = blockIdx.x * COUT_T;
    const int rows = COUT_T + 6, c4n = C >> 2, CP = C + 4;
    for (int i = threadIdx.x; i < rows * c4n; i += 256) {
        const int r = i / c4n, c = (i - r * c4n) * 4;
        const int t = t0 + r - 3;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t >= 0 && t < T) v = *(const f32x4*)(xs + ((size_t)b * T + t) * C + c);
        *(f32x4*)(sm + r * CP + c) = v;
    }
    __syncthreads();
    // 4 lanes per output sample: each sums a quarter of the 7*C products (16-byte pieces 4 p, 4 p + 16, ...), then a 2-step shuffle reduce
    const int o = threadIdx.x >> 2, part = threadIdx.x & 3;
    float acc = 0.f;
    for (int j = 0; j < 7; ++j)
        for (int c = 4 * part; c < C; c += 16) {
            const f32x4 xv = *(const f32x4*)(sm + (o + j) * CP + c), wv = *(const f32x4*)(w + j * C + c);
            acc = fmaf(wv[0], xv[0], acc); acc = fmaf(wv[1], xv[1], acc); acc = fmaf(wv[2], xv[2], acc); acc = fmaf(wv[3], xv[3], acc);
        }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    if (part == 0 && t0 + o < T) y[(size_t)b * T + t0 + o] = tanhf(acc + bias);
}

extern "C" int vn_dac_conv_out_f32(vn_ctx* ctx, const float* xs, const float* w, float bias, float* y, int B, int T,
                                   int C, void* stream) {
    if (!ctx || !xs || !w || !y) return VN_ERR_INVALID;
    if ((C & 3) || (((uintptr_t)xs | (uintptr_t)w) & 15)) return vn_fail(ctx, VN_ERR_INVALID, "conv_out: C=%s%ld must be a multiple of 4, operands 16-byte aligned", "", C);
    const size_t lds = (size_t)(COUT_T + 6) * (C + 4) * sizeof(float);
    if (lds > 64 * 1024) return vn_fail(ctx, VN_ERR_INVALID, "conv_out: C=%s%ld too wide", "", C);
    hipLaunchKernelGGL(vn_dac_conv_out_kernel, dim3(vn_cdiv(T, COUT_T), B), dim3(256), lds, (hipStream_t)stream, xs, w,
                       bias, y, B, T, C);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

// ---------------------------------------------------------------------------------------------
// Residual vector quantiser (dac/nn/quantize.py), one 256-thread block per frame, all n levels in sequence:
//   z_e = in_proj(residual) (L -> 8) ; idx = argmax_k cos(z_e, codebook[k]) via -(|e|^2 - 2 e.c + |c|^2) on
//   L2-normalised vectors ; z_q = z_e + (codebook[idx] - z_e) ; z_q = out_proj(z_q) (8 -> L) ; residual -= z_q.
// Weights: win [n][8][L], bin [n][8], cb [n][Kc][8], wout [n][L][8], bout [n][L].  z: [frames][L].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vn_rvq_encode_kernel(const float* __restrict__ z, const float* __restrict__ win,
                                                            const float* __restrict__ bin, const float* __restrict__ cb,
                                                            const float* __restrict__ wout,
                                                            const float* __restrict__ bout, int64_t* __restrict__ codes,
                                                            int B, int T, int L, int n_levels, int Kc) {
    extern __shared__ __attribute__((aligned(16))) float res[];     // [L] residual
    __shared__ float s_e[8], s_en[8], s_zq[8];
    __shared__ float s_best[4];
    __shared__ int s_bidx[4];
    const int frame = blockIdx.x, b = frame / T, t = frame - b * T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < L; i += 256) res[i] = z[(size_t)frame * L + i];
    __syncthreads();
    for (int lv = 0; lv < n_levels; ++lv) {
        // in_proj: 8 outputs, 32 threads each
        {
            const int o = tid >> 5, part = tid & 31;
            const float* wr = win + ((size_t)lv * 8 + o) * L;
            float acc = 0.f;
            for (int k = part; k < L; k += 32) acc = fmaf(wr[k], res[k], acc);
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
            if (part == 0) s_e[o] = acc + bin[lv * 8 + o];
        }
        __syncthreads();
        if (tid == 0) {
            float nn = 0.f;
            for (int d = 0; d < 8; ++d) nn += s_e[d] * s_e[d];
            const float inv = 1.0f / fmaxf(sqrtf(nn), 1e-12f);       // F.normalize eps
            for (int d = 0; d < 8; ++d) s_en[d] = s_e[d] * inv;
        }
        __syncthreads();
        // nearest codebook row: score = -(|e|^2 - 2 e.c_n + |c_n|^2), first max wins
        float best = -INFINITY;
        int bidx = 0;
        float e2 = 0.f;
        for (int d = 0; d < 8; ++d) e2 += s_en[d] * s_en[d];
        for (int k = tid; k < Kc; k += 256) {
            const float* c = cb + ((size_t)lv * Kc + k) * 8;
            float cn = 0.f;
            for (int d = 0; d < 8; ++d) cn += c[d] * c[d];
            const float inv = 1.0f / fmaxf(sqrtf(cn), 1e-12f);
            float dot = 0.f, c2 = 0.f;
            for (int d = 0; d < 8; ++d) { const float v = c[d] * inv; dot += s_en[d] * v; c2 += v * v; }
            const float score = -(e2 - 2.0f * dot + c2);
            if (score > best) { best = score; bidx = k; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ob = __shfl_xor(best, off);
            const int oi = __shfl_xor(bidx, off);
            if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
        }
        if (lane == 0) { s_best[wave] = best; s_bidx[wave] = bidx; }
        __syncthreads();
        if (tid == 0) {
            float bb = s_best[0];
            int bi = s_bidx[0];
            for (int wv = 1; wv < 4; ++wv)
                if (s_best[wv] > bb || (s_best[wv] == bb && s_bidx[wv] < bi)) { bb = s_best[wv]; bi = s_bidx[wv]; }
            codes[((size_t)b * n_levels + lv) * T + t] = bi;
            const float* c = cb + ((size_t)lv * Kc + bi) * 8;
            for (int d = 0; d < 8; ++d) s_zq[d] = s_e[d] + (c[d] - s_e[d]);       // straight-through arithmetic
        }
        __syncthreads();
        for (int i = tid; i < L; i += 256) {
            const float* wr = wout + ((size_t)lv * L + i) * 8;
            float acc = 0.f;
#pragma unroll
            for (int d = 0; d < 8; ++d) acc = fmaf(wr[d], s_zq[d], acc);
            res[i] -= acc + bout[(size_t)lv * L + i];
        }
        __syncthreads();
    }
}

extern "C" int vn_rvq_encode_f32(vn_ctx* ctx, const float* z, const float* win, const float* bin, const float* cb,
                                 const float* wout, const float* bout, int64_t* codes, int B, int T, int L, int n_levels,
                                 int codebook_size, void* stream) {
    if (!ctx || !z || !win || !bin || !cb || !wout || !bout || !codes) return VN_ERR_INVALID;
    hipLaunchKernelGGL(vn_rvq_encode_kernel, dim3(B * T), dim3(256), (size_t)L * sizeof(float), (hipStream_t)stream, z,
                       win, bin, cb, wout, bout, codes, B, T, L, n_levels, codebook_size);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

// codes -> z_q = sum_lv out_proj_lv(codebook_lv[code]) + bias_lv   (quantizer.from_latents on exact codebook rows,
// transformer.py:671-672).  One thread per (frame, channel).
__global__ __launch_bounds__(256) void vn_rvq_decode_kernel(const int64_t* __restrict__ codes, const float* __restrict__ cb,
                                                            const float* __restrict__ wout,
                                                            const float* __restrict__ bout, float* __restrict__ zq,
                                                            int B, int T, int L, int n_levels, int Kc) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * T * L) return;
    const int ch = (int)(i % L);
    const long frame = i / L;
    const int b = (int)(frame / T), t = (int)(frame - (long)b * T);
    float total = 0.f;
    for (int lv = 0; lv < n_levels; ++lv) {
        const int64_t code = codes[((size_t)b * n_levels + lv) * T + t];
        const float* c = cb + ((size_t)lv * Kc + code) * 8;
        const float* wr = wout + ((size_t)lv * L + ch) * 8;
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) acc = fmaf(wr[d], c[d], acc);
        total += acc + bout[(size_t)lv * L + ch];
    }
    zq[i] = total;
}

extern "C" int vn_rvq_decode_f32(vn_ctx* ctx, const int64_t* codes, const float* cb, const float* wout, const float* bout,
                                 float* zq, int B, int T, int L, int n_levels, int codebook_size, void* stream) {
    if (!ctx || !codes || !cb || !wout || !bout || !zq) return VN_ERR_INVALID;
    const long n = (long)B * T * L;
    hipLaunchKernelGGL(vn_rvq_decode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, codes,
                       cb, wout, bout, zq, B, T, L, n_levels, codebook_size);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}
