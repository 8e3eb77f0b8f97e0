// HBM-bound kernels of the VampNet TRAINING step (scripts/exp/train.py:237-304) for gfx950: dropout-aware
// residual / GEGLU passes, RMSNorm backward, layout transposes that feed the backward GEMMs, the label-smoothed
// cross-entropy (forward + gradient in one pass), weight-norm fold / backward, embedding gradients, gradient-norm
// and the AdamW update.  All fp32; every kernel streams its operands once with 16-byte accesses.
//
// Determinism: column sums (norm weights, biases, loss) go through fixed-shape partial buffers + a second pass,
// never through floating-point atomics, so a step is bitwise reproducible run to run (attention_train.hip reduces the
// shared relative-position-bias gradient the same way).
#include "vn_common.h"
#include "vn_train.h"

// ---------------------------------------------------------------------------------------------
// x_out = x_in + dropout(y)        (transformer.py:347, :367)      M x N, N % 4 == 0
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vn_resid_dropout_kernel(const float* __restrict__ x_in, const float* __restrict__ y,
                                                               float* __restrict__ x_out, int M, int N4, vn_drop d) {
    const long total = (long)M * N4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        const int row = (int)(i / N4), c4 = (int)(i - (long)row * N4);
        f32x4 a = ((const f32x4*)x_in)[i];
        const f32x4 b = ((const f32x4*)y)[i];
        if (d.thresh16) {
            const uint32_t rk = vn_drop_rowkey(d, row);
            const uint32_t b0 = vn_drop_bits(rk, 4 * c4), b1 = vn_drop_bits(rk, 4 * c4 + 2);
            a[0] += b[0] * vn_drop_mul(d, b0, 0);
            a[1] += b[1] * vn_drop_mul(d, b0, 1);
            a[2] += b[2] * vn_drop_mul(d, b1, 0);
            a[3] += b[3] * vn_drop_mul(d, b1, 1);
        } else {
            a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
        }
        ((f32x4*)x_out)[i] = a;
    }
}

int vn_launch_resid_dropout(vn_ctx* ctx, const float* x_in, const float* y, float* x_out, int M, int N, const vn_drop& d,
                            hipStream_t s) {
    if (M <= 0) return VN_OK;
    const long total = (long)M * (N / 4);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(vn_resid_dropout_kernel, dim3(blocks), dim3(256), 0, s, x_in, y, x_out, M, N / 4, d);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

// dy_out = dy_in * keep * scale   (backward of the residual-branch dropout)
__global__ __launch_bounds__(256) void vn_dropout_bwd_kernel(const float* __restrict__ dy, float* __restrict__ out, int M,
                                                             int N4, vn_drop d, uint16_t* __restrict__ out16) {
    const long total = (long)M * N4;
    bool bad = false;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        const int row = (int)(i / N4), c4 = (int)(i - (long)row * N4);
        f32x4 a = ((const f32x4*)dy)[i];
        const uint32_t rk = vn_drop_rowkey(d, row);
        const uint32_t b0 = vn_drop_bits(rk, 4 * c4), b1 = vn_drop_bits(rk, 4 * c4 + 2);
        a[0] *= vn_drop_mul(d, b0, 0);
        a[1] *= vn_drop_mul(d, b0, 1);
        a[2] *= vn_drop_mul(d, b1, 0);
        a[3] *= vn_drop_mul(d, b1, 1);
        ((f32x4*)out)[i] = a;
        if (out16) vn_store_planes4(out16, VN_PLANES_TILED, row, 4 * c4, 4 * N4, a, bad);
    }
}

int vn_launch_dropout_bwd(vn_ctx* ctx, const float* dy, float* out, int M, int N, const vn_drop& d, hipStream_t s, uint16_t* out16) {
    if (M <= 0) return VN_OK;
    if (out16 && (!d.thresh16 || (N & 31))) return vn_fail(ctx, VN_ERR_INVALID, "dropout_bwd: plane output needs the mask on and N %% 32 == 0 (N=%s%ld)", "", N);
    if (!d.thresh16) {
        if (dy != out) VN_HIP_CHECK(ctx, hipMemcpyAsync(out, dy, (size_t)M * N * sizeof(float), hipMemcpyDeviceToDevice, s));
        return VN_OK;
    }
    const long total = (long)M * (N / 4);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(vn_dropout_bwd_kernel, dim3(blocks), dim3(256), 0, s, dy, out, M, N / 4, d, out16);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

// keep-mask export for the parity tests: out[row][col] = 1 if kept
__global__ __launch_bounds__(256) void vn_dropout_mask_kernel(uint8_t* __restrict__ out, long rows, int cols, vn_drop d) {
    const long total = rows * cols;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        const long row = i / cols;
        const int col = (int)(i - row * cols);
        const uint32_t bits = vn_drop_bits(vn_drop_rowkey(d, row), col);
        out[i] = d.thresh16 == 0 || vn_drop_mul(d, bits, col) != 0.0f;
    }
}

int vn_launch_dropout_mask(vn_ctx* ctx, uint8_t* out, long rows, int cols, const vn_drop& d, hipStream_t s) {
    if (rows <= 0) return VN_OK;
    hipLaunchKernelGGL(vn_dropout_mask_kernel, dim3(4096), dim3(256), 0, s, out, rows, cols, d);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

// ---------------------------------------------------------------------------------------------
// GEGLU on the packed w_1 output u [M][4D] (value/gate interleaved in 32-column blocks, see VN_W_W1):
//   g[m][o] = dropout( u_val * gelu_tanh(u_gate) )          (transformer.py:80-82, activations.py:33-35)
//   backward: du_val = dg' * gelu(gate) ; du_gate = dg' * val * gelu'(gate),  dg' = dg * keep * scale
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float vn_gelu_tanh_grad(float x) {
    const float c = 0.7978845608028654f;
    const float x2 = x * x;
    const float t = tanhf(c * (x + 0.044715f * x2 * x));
    return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * c * (1.0f + 3.0f * 0.044715f * x2);
}

// out16 (optional): the result once more as TILED bf16x3 planes — the A operand of the GEMM that consumes it on the split-plane pipe
// (forward: g [M][D2] -> W2; backward: du [M][2 D2] -> dX of W1), written from the registers that hold it instead of by a split pass
template <bool BWD>
__global__ __launch_bounds__(256) void vn_geglu_train_kernel(const float* __restrict__ u, const float* __restrict__ dg,
                                                             float* __restrict__ out, uint16_t* __restrict__ out16, int M, int D2, vn_drop d) {
    const int n4 = D2 / 4;
    const long total = (long)M * n4;
    bool bad = false;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        const int row = (int)(i / n4), o = 4 * (int)(i - (long)row * n4);
        const int uc = 64 * (o >> 5) + (o & 31);
        const size_t ub = (size_t)row * 2 * D2 + uc;
        const f32x4 val = *(const f32x4*)(u + ub);
        const f32x4 gate = *(const f32x4*)(u + ub + 32);
        f32x4 m = {1.f, 1.f, 1.f, 1.f};
        if (d.thresh16) {
            const uint32_t rk = vn_drop_rowkey(d, row);
            const uint32_t b0 = vn_drop_bits(rk, o), b1 = vn_drop_bits(rk, o + 2);
            m[0] = vn_drop_mul(d, b0, 0); m[1] = vn_drop_mul(d, b0, 1);
            m[2] = vn_drop_mul(d, b1, 0); m[3] = vn_drop_mul(d, b1, 1);
        }
        if constexpr (!BWD) {
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = val[e] * vn_gelu_tanh(gate[e]) * m[e];
            *(f32x4*)(out + (size_t)row * D2 + o) = r;
            if (out16) vn_store_planes4(out16, VN_PLANES_TILED, row, o, D2, r, bad);
        } else {
            const f32x4 g = *(const f32x4*)(dg + (size_t)row * D2 + o);
            f32x4 dv, dgt;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gg = g[e] * m[e];
                dv[e] = gg * vn_gelu_tanh(gate[e]);
                dgt[e] = gg * val[e] * vn_gelu_tanh_grad(gate[e]);
            }
            *(f32x4*)(out + ub) = dv;
            *(f32x4*)(out + ub + 32) = dgt;
            if (out16) {
                vn_store_planes4(out16, VN_PLANES_TILED, row, uc, 2 * D2, dv, bad);
                vn_store_planes4(out16, VN_PLANES_TILED, row, uc + 32, 2 * D2, dgt, bad);
            }
        }
    }
}

int vn_launch_geglu_train(vn_ctx* ctx, const float* u, const float* dg, float* out, int M, int D2, const vn_drop& d,
                          bool bwd, hipStream_t s, uint16_t* out16) {
    if (M <= 0) return VN_OK;
    if (D2 % 32) return vn_fail(ctx, VN_ERR_INVALID, "geglu: width %s%ld must be a multiple of 32", "", D2);
    const long total = (long)M * (D2 / 4);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (bwd) hipLaunchKernelGGL(vn_geglu_train_kernel<true>, dim3(blocks), dim3(256), 0, s, u, dg, out, out16, M, D2, d);
    else hipLaunchKernelGGL(vn_geglu_train_kernel<false>, dim3(blocks), dim3(256), 0, s, u, dg, out, out16, M, D2, d);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

// ---------------------------------------------------------------------------------------------
// RMSNorm backward (transformer.py:55-58):  y = w * x * r,  r = rsqrt(mean(x^2) + eps)
//   dx = r * (w o dy) - x * r^3 * mean(x o w o dy)      (+ dres, the gradient arriving on the residual path)
//   dw = sum_rows dy o x * r                              -> partial[blockIdx][D], reduced by vn_reduce_rows
// One wave per row, VN_RB rows per wave, 4 waves per block.
// ---------------------------------------------------------------------------------------------
#define VN_RB 8     // rows per wave
template <int VEC>  // float4 per lane: D <= VEC * 256
__global__ __launch_bounds__(256) void vn_rmsnorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ dy, const float* __restrict__ dres,
                                                             float* __restrict__ dx, float* __restrict__ dw_partial,
                                                             int rows, int D, float eps) {
    __shared__ f32x4 red[3][64 * VEC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = D >> 2;
    f32x4 wv[VEC], dwacc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        const int c = lane + 64 * i;
        wv[i] = c < nv ? ((const f32x4*)w)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
        dwacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int row0 = (blockIdx.x * 4 + wave) * VN_RB;
    for (int rr = 0; rr < VN_RB; ++rr) {
        const int row = row0 + rr;
        if (row >= rows) break;
        const f32x4* xr = (const f32x4*)(x + (size_t)row * D);
        const f32x4* gr = (const f32x4*)(dy + (size_t)row * D);
        f32x4 xv[VEC], gv[VEC];
        float ss = 0.f, dot = 0.f;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) { xv[i] = xr[c]; gv[i] = gr[c]; }
            else { xv[i] = f32x4{0.f, 0.f, 0.f, 0.f}; gv[i] = xv[i]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ss += xv[i][e] * xv[i][e];
                dot += xv[i][e] * wv[i][e] * gv[i][e];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { ss += __shfl_xor(ss, o); dot += __shfl_xor(dot, o); }
        const float r = 1.0f / sqrtf(ss / (float)D + eps);
        const float k = dot / (float)D * r * r * r;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const int c = lane + 64 * i;
            if (c >= nv) continue;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = r * (wv[i][e] * gv[i][e]) - xv[i][e] * k;
                dwacc[i][e] += gv[i][e] * (xv[i][e] * r);
            }
            if (dres) {
                const f32x4 a = ((const f32x4*)(dres + (size_t)row * D))[c];
                o[0] += a[0]; o[1] += a[1]; o[2] += a[2]; o[3] += a[3];
            }
            ((f32x4*)(dx + (size_t)row * D))[c] = o;
        }
    }
    // block-level sum of the 4 waves' dw accumulators (fixed order: wave 0 + 1 + 2 + 3)
    if (wave > 0)
#pragma unroll
        for (int i = 0; i < VEC; ++i) red[wave - 1][lane + 64 * i] = dwacc[i];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const int c = lane + 64 * i;
            if (c >= nv) continue;
            f32x4 a = dwacc[i];
#pragma unroll
            for (int k2 = 0; k2 < 3; ++k2) {
                const f32x4 b = red[k2][c];
                a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
            }
            ((f32x4*)(dw_partial + (size_t)blockIdx.x * D))[c] = a;
        }
    }
}

// out[c] = sum_b partial[b][c]   (fixed order: 4 interleaved row groups, then group 0 + 1 + 2 + 3)
__global__ __launch_bounds__(256) void vn_reduce_rows_kernel(const float* __restrict__ partial, int nb, int C,
                                                             float* __restrict__ out) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    float a = 0.f;
    if (c < C)
        for (int b = part; b < nb; b += 4) a += partial[(size_t)b * C + c];
    red[part][threadIdx.x & 63] = a;
    __syncthreads();
    if (part == 0 && c < C) out[c] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

int vn_launch_reduce_rows(vn_ctx* ctx, const float* partial, int nb, int C, float* out, hipStream_t s) {
    hipLaunchKernelGGL(vn_reduce_rows_kernel, dim3(vn_cdiv(C, 64)), dim3(256), 0, s, partial, nb, C, out);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

int vn_rmsnorm_bwd_blocks(int rows) { return vn_cdiv(rows, 4 * VN_RB); }

int vn_launch_rmsnorm_bwd(vn_ctx* ctx, const float* x, const float* w, const float* dy, const float* dres, float* dx,
                          float* dw, float* partial, int rows, int D, float eps, hipStream_t s) {
    if (rows <= 0) return VN_OK;
    if (D % 4 || D > 8 * 256) return vn_fail(ctx, VN_ERR_UNSUPPORTED, "rmsnorm_bwd: D=%s%ld unsupported", "", D);
    const int nb = vn_rmsnorm_bwd_blocks(rows);
    const int vec = vn_cdiv(D, 256);
#define VN_RB_CASE(V)                                                                                              \
    case V:                                                                                                        \
        hipLaunchKernelGGL(vn_rmsnorm_bwd_kernel<V>, dim3(nb), dim3(256), 0, s, x, w, dy, dres, dx, partial, rows, D, eps); \
        break;
    switch (vec) {
        VN_RB_CASE(1) VN_RB_CASE(2) VN_RB_CASE(3) VN_RB_CASE(4) VN_RB_CASE(5) VN_RB_CASE(6) VN_RB_CASE(7) VN_RB_CASE(8)
    }
#undef VN_RB_CASE
    VN_LAUNCH_CHECK(ctx);
    return vn_launch_reduce_rows(ctx, partial, nb, D, dw, s);
}

// ---------------------------------------------------------------------------------------------
// dst[c][r] = src[r][c]   src [R][C] (row stride lds), dst [C][ldd] with columns r in [R, ldd) zero-filled:
// the backward GEMMs contract over the token axis, and the GEMM kernel wants the contraction axis contiguous
// and a multiple of 32.  64x64 tiles through LDS (padded rows: conflict-free both ways), 16-byte global accesses.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vn_transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int R,
                                                           int C, int lds_, int ldd) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;     // 16 x 16 threads, float4 each, 4 passes
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = r0 + ty + 16 * p, c = c0 + 4 * tx;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < R) {
            if (c + 3 < C) v = *(const f32x4*)(src + (size_t)r * lds_ + c);
            else
                for (int e = 0; e < 4; ++e)
                    if (c + e < C) v[e] = src[(size_t)r * lds_ + c + e];
        }
        tile[ty + 16 * p][4 * tx + 0] = v[0];
        tile[ty + 16 * p][4 * tx + 1] = v[1];
        tile[ty + 16 * p][4 * tx + 2] = v[2];
        tile[ty + 16 * p][4 * tx + 3] = v[3];
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int c = c0 + ty + 16 * p, r = r0 + 4 * tx;
        if (c >= C || r >= ldd) continue;
        f32x4 v;
        v[0] = tile[4 * tx + 0][ty + 16 * p];
        v[1] = tile[4 * tx + 1][ty + 16 * p];
        v[2] = tile[4 * tx + 2][ty + 16 * p];
        v[3] = tile[4 * tx + 3][ty + 16 * p];
        *(f32x4*)(dst + (size_t)c * ldd + r) = v;      // ldd % 4 == 0 and r % 4 == 0: whole float4 is inside the row
    }
}

int vn_launch_transpose(vn_ctx* ctx, const float* src, float* dst, int R, int C, int lds_, int ldd, hipStream_t s) {
    if (R <= 0 || C <= 0) return VN_OK;
    if (ldd % 4 || ldd < R) return vn_fail(ctx, VN_ERR_INVALID, "transpose: bad destination stride %s%ld", "", ldd);
    hipLaunchKernelGGL(vn_transpose_kernel, dim3(vn_cdiv(ldd, 64), vn_cdiv(C, 64)), dim3(256), 0, s, src, dst, R, C, lds_, ldd);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

// ---------------------------------------------------------------------------------------------
// Operands of the training GEMMs on the split-plane pipe (gemm_x3.hip, train.hip): fp32 matrices -> TILED bf16x3 planes
// ([row / 16][k / 32][plane][16][32], the layout the kernel's LDS-DMA fetches in whole 1 KiB pieces).
//   vn_split3_tiled_kernel            src [R][K] (row stride lds_) -> planes of the same matrix; one wave per 16 x 32 piece: lane
//                                     (row r = lane / 4, 8 k's) reads 32 contiguous bytes and writes 16 bytes per plane, the wave's
//                                     stores are the piece's contiguous 1 KiB.  Rows >= R of the last row block are written as zeros.
//   vn_transpose_split3_tiled_kernel  src [R][C] -> planes of its TRANSPOSE [C][Rp] (Rp = R rounded up to 32, zero filled): the dW
//                                     GEMMs contract over the token axis.  64 x 64 tiles through LDS (as vn_transpose_kernel), then
//                                     each wave writes two 16 x 32 pieces.  C % 16 == 0.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vn_split3_tiled_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int R, int K,
                                                              int lds_) {
    const int lane = threadIdx.x & 63;
    const long piece = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int kblocks = K >> 5;
    const long n_pieces = (long)((R + 15) >> 4) * kblocks;
    if (piece >= n_pieces) return;
    const int rb = (int)(piece / kblocks), kb = (int)(piece - (long)rb * kblocks);
    const int row = rb * 16 + (lane >> 2), col = kb * 32 + (lane & 3) * 8;
    f32x8 v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (row < R) {
        const f32x4 a = *(const f32x4*)(src + (size_t)row * lds_ + col), b = *(const f32x4*)(src + (size_t)row * lds_ + col + 4);
        v = f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    }
    bool bad = false;
    vn_store_planes8_tiled(dst, VN_PLANES_TILED, row, col, K, v, bad);
}

int vn_launch_split3_tiled(vn_ctx* ctx, const float* src, uint16_t* dst, int R, int K, int lds_, hipStream_t s) {
    if (R <= 0 || K <= 0) return VN_OK;
    if ((K & 31) || (lds_ & 3) || (((uintptr_t)src | (uintptr_t)dst) & 15))
        return vn_fail(ctx, VN_ERR_INVALID, "split3_tiled: K %% 32, row stride %% 4 and 16-byte aligned operands (K=%s%ld, ld=%ld)", "", K, lds_);
    const long n_pieces = (long)((R + 15) >> 4) * (K >> 5);
    hipLaunchKernelGGL(vn_split3_tiled_kernel, dim3((unsigned)((n_pieces + 3) / 4)), dim3(256), 0, s, src, dst, R, K, lds_);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

// rows16 != nullptr: ALSO the tiled planes of the matrix itself ([R rounded up to 16][C], C % 32 == 0) from the same LDS tile — the dY of a
// layer is the A operand of its dX GEMM (row-major planes) and of its dW GEMM (transposed planes), a weight is needed both ways after
// every update: one read of the fp32 matrix instead of two
__global__ __launch_bounds__(256) void vn_transpose_split3_tiled_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int R,
                                                                        int C, int lds_, int Rp, uint16_t* __restrict__ rows16) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;     // 16 x 16 threads, float4 each, 4 passes
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = r0 + ty + 16 * p, c = c0 + 4 * tx;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < R) {
            if (c + 3 < C) v = *(const f32x4*)(src + (size_t)r * lds_ + c);
            else
                for (int e = 0; e < 4; ++e)
                    if (c + e < C) v[e] = src[(size_t)r * lds_ + c + e];
        }
        tile[ty + 16 * p][4 * tx + 0] = v[0];
        tile[ty + 16 * p][4 * tx + 1] = v[1];
        tile[ty + 16 * p][4 * tx + 2] = v[2];
        tile[ty + 16 * p][4 * tx + 3] = v[3];
    }
    __syncthreads();
    // output rows = c (64 of them: four blocks of 16), output k = r (64: two blocks of 32); wave w writes the pieces (cb = w, kb = 0 / 1)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cl = wave * 16 + (lane >> 2);                     // column of the tile = output row
    if (c0 + wave * 16 < C) {                                   // C % 16 == 0: whole row blocks
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int rl = kb * 32 + (lane & 3) * 8;            // first of this lane's eight tile rows = output k
            if (r0 + kb * 32 >= Rp) continue;
            f32x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tile[rl + e][cl];    // rows >= R were loaded as zeros
            bool bad = false;
            vn_store_planes8_tiled(dst, VN_PLANES_TILED, c0 + cl, r0 + rl, Rp, v, bad);
        }
    }
    if (rows16 && r0 + wave * 16 < ((R + 15) & ~15)) {         // the matrix itself: wave w writes row block w, k blocks 0 / 1
        const int rl = wave * 16 + (lane >> 2);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int cc = kb * 32 + (lane & 3) * 8;
            if (c0 + kb * 32 >= C) continue;
            f32x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tile[rl][cc + e];
            bool bad = false;
            vn_store_planes8_tiled(rows16, VN_PLANES_TILED, r0 + rl, c0 + cc, C, v, bad);
        }
    }
}

int vn_launch_transpose_split3_tiled(vn_ctx* ctx, const float* src, uint16_t* dst, int R, int C, int lds_, int Rp, hipStream_t s,
                                     uint16_t* rows16) {
    if (R <= 0 || C <= 0) return VN_OK;
    if ((Rp & 31) || Rp < R || (C & 15) || ((uintptr_t)dst & 15))
        return vn_fail(ctx, VN_ERR_INVALID, "transpose_split3_tiled: Rp %% 32, Rp >= R, C %% 16 (Rp=%s%ld, C=%ld)", "", Rp, C);
    if (rows16 && ((C & 31) || ((uintptr_t)rows16 & 15)))
        return vn_fail(ctx, VN_ERR_INVALID, "transpose_split3_tiled: the row-major planes need C %% 32 == 0 (C=%s%ld)", "", C);
    hipLaunchKernelGGL(vn_transpose_split3_tiled_kernel, dim3(vn_cdiv(Rp, 64), vn_cdiv(C, 64)), dim3(256), 0, s, src, dst, R, C, lds_, Rp, rows16);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

// ---------------------------------------------------------------------------------------------
// column sums of src [R][C] (bias gradients): partial[b][c] over 64-row chunks, then vn_reduce_rows.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vn_colsum_partial_kernel(const float* __restrict__ src, int R, int C,
                                                                float* __restrict__ partial) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int r0 = blockIdx.y * 64, r1 = r0 + 64 < R ? r0 + 64 : R;
    float a = 0.f;
    for (int r = r0; r < r1; ++r) a += src[(size_t)r * C + c];
    partial[(size_t)blockIdx.y * C + c] = a;
}

int vn_launch_colsum(vn_ctx* ctx, const float* src, int R, int C, float* partial, float* out, hipStream_t s) {
    const int nb = vn_cdiv(R, 64);
    hipLaunchKernelGGL(vn_colsum_partial_kernel, dim3(vn_cdiv(C, 256), nb), dim3(256), 0, s, src, R, C, partial);
    VN_LAUNCH_CHECK(ctx);
    return vn_launch_reduce_rows(ctx, partial, nb, C, out, s);
}

// ---------------------------------------------------------------------------------------------
// Label-smoothed cross-entropy over rows of V logits, mean over rows whose target != ignore (-100)
// (train.py:267-278, conf/vampnet.yml:17; torch.nn.CrossEntropyLoss semantics):
//   loss_row = (1-e) * (logZ - x_t) + e * (logZ - mean_c x_c)
//   dlogits  = (softmax - (1-e) * onehot_t - e/V) / n_valid         written IN PLACE over the logits
// One wave per row (V = VEC*256).  n_valid is counted on the device first (no host sync).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vn_ce_prepare_kernel(const int64_t* __restrict__ target, int32_t* __restrict__ t32,
                                                            long n, int V, int32_t* __restrict__ n_valid) {
    __shared__ int cnt[4];
    int local = 0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        const int64_t t = target[i];
        const bool ok = t >= 0 && t < V;
        t32[i] = ok ? (int32_t)t : -1;
        local += ok;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(n_valid, cnt[0] + cnt[1] + cnt[2] + cnt[3]);      // integer: order independent
}

template <int VEC>
__global__ __launch_bounds__(256) void vn_ce_kernel(float* __restrict__ logits, const int32_t* __restrict__ t32, long rows,
                                                    float ls, const int32_t* __restrict__ n_valid,
                                                    float* __restrict__ row_loss) {
    const int lane = threadIdx.x & 63;
    const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (row >= rows) return;
    constexpr int V = VEC * 256;
    f32x4* xr = (f32x4*)(logits + row * V);
    const int t = t32[row];
    if (t < 0) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) xr[lane + 64 * i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (lane == 0) row_loss[row] = 0.f;
        return;
    }
    f32x4 x[VEC];
    float mx = -INFINITY, sx = 0.f, xt = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        x[i] = xr[lane + 64 * i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mx = fmaxf(mx, x[i][e]);
            sx += x[i][e];
            if ((lane + 64 * i) * 4 + e == t) xt = x[i][e];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, o));
        sx += __shfl_xor(sx, o);
        xt += __shfl_xor(xt, o);
    }
    float se = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            x[i][e] = expf(x[i][e] - mx);
            se += x[i][e];
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
    const float logZ = mx + logf(se);
    const float inv_n = 1.0f / (float)(*n_valid);
    if (lane == 0) row_loss[row] = (1.0f - ls) * (logZ - xt) + ls * (logZ - sx / (float)V);
    const float inv_se = 1.0f / se, sm = ls / (float)V;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        f32x4 g;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float p = x[i][e] * inv_se - sm;
            if ((lane + 64 * i) * 4 + e == t) p -= (1.0f - ls);
            g[e] = p * inv_n;
        }
        xr[lane + 64 * i] = g;
    }
}

// loss = sum(row_loss) / n_valid  in double, one block, fixed order
__global__ __launch_bounds__(1024) void vn_ce_finish_kernel(const float* __restrict__ row_loss, long rows,
                                                            const int32_t* __restrict__ n_valid, float* __restrict__ loss) {
    __shared__ double red[1024];
    double a = 0.0;
    for (long i = threadIdx.x; i < rows; i += 1024) a += (double)row_loss[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = (float)(red[0] / (double)(*n_valid > 0 ? *n_valid : 1));
}

int vn_launch_cross_entropy(vn_ctx* ctx, float* logits, const int64_t* target, int32_t* t32, long rows, int V, float ls,
                            int32_t* n_valid, float* row_loss, float* loss, hipStream_t s) {
    if (V != 1024 && V != 256) return vn_fail(ctx, VN_ERR_UNSUPPORTED, "cross-entropy: vocab=%s%ld unsupported", "", V);
    VN_HIP_CHECK(ctx, hipMemsetAsync(n_valid, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL(vn_ce_prepare_kernel, dim3(256), dim3(256), 0, s, target, t32, rows, V, n_valid);
    const int blocks = (int)((rows + 3) / 4);
    if (V == 1024) hipLaunchKernelGGL(vn_ce_kernel<4>, dim3(blocks), dim3(256), 0, s, logits, t32, rows, ls, n_valid, row_loss);
    else hipLaunchKernelGGL(vn_ce_kernel<1>, dim3(blocks), dim3(256), 0, s, logits, t32, rows, ls, n_valid, row_loss);
    hipLaunchKernelGGL(vn_ce_finish_kernel, dim3(1), dim3(1024), 0, s, row_loss, rows, n_valid, loss);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

// ---------------------------------------------------------------------------------------------
// Weight norm of the classifier (layers.py:47-48, old-style torch weight_norm, dim 0):  W[r] = g[r] * v[r] / ||v[r]||
//   fold:     W <- g, v
//   backward: dg[r] = (dW[r] . v[r]) / n ;  dv[r] = g/n * dW[r] - g * (dW[r] . v[r]) / n^3 * v[r]
// One wave per row.
// ---------------------------------------------------------------------------------------------
template <bool BWD>
__global__ __launch_bounds__(256) void vn_weight_norm_kernel(const float* __restrict__ g, const float* __restrict__ v,
                                                             const float* __restrict__ dW, float* __restrict__ outW,
                                                             float* __restrict__ dg, float* __restrict__ dv, int rows, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = D >> 2;
    const f32x4* vr = (const f32x4*)(v + (size_t)row * D);
    float ss = 0.f, dot = 0.f;
    for (int c = lane; c < nv; c += 64) {
        const f32x4 a = vr[c];
        ss += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
        if constexpr (BWD) {
            const f32x4 b = ((const f32x4*)(dW + (size_t)row * D))[c];
            dot += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ss += __shfl_xor(ss, o); dot += __shfl_xor(dot, o); }
    const float n = sqrtf(ss), gg = g[row];
    if constexpr (!BWD) {
        const float k = gg / n;
        for (int c = lane; c < nv; c += 64) {
            f32x4 a = vr[c];
            a[0] *= k; a[1] *= k; a[2] *= k; a[3] *= k;
            ((f32x4*)(outW + (size_t)row * D))[c] = a;
        }
    } else {
        if (lane == 0) dg[row] = dot / n;
        const float k1 = gg / n, k2 = gg * dot / (n * n * n);
        for (int c = lane; c < nv; c += 64) {
            const f32x4 a = vr[c];
            const f32x4 b = ((const f32x4*)(dW + (size_t)row * D))[c];
            f32x4 o;
            o[0] = k1 * b[0] - k2 * a[0]; o[1] = k1 * b[1] - k2 * a[1];
            o[2] = k1 * b[2] - k2 * a[2]; o[3] = k1 * b[3] - k2 * a[3];
            ((f32x4*)(dv + (size_t)row * D))[c] = o;
        }
    }
}

int vn_launch_weight_norm_fold(vn_ctx* ctx, const float* g, const float* v, float* W, int rows, int D, hipStream_t s) {
    hipLaunchKernelGGL(vn_weight_norm_kernel<false>, dim3(vn_cdiv(rows, 4)), dim3(256), 0, s, g, v, nullptr, W, nullptr, nullptr, rows, D);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}
int vn_launch_weight_norm_bwd(vn_ctx* ctx, const float* g, const float* v, const float* dW, float* dg, float* dv, int rows,
                              int D, hipStream_t s) {
    hipLaunchKernelGGL(vn_weight_norm_kernel<true>, dim3(vn_cdiv(rows, 4)), dim3(256), 0, s, g, v, dW, nullptr, dg, dv, rows, D);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

// ---------------------------------------------------------------------------------------------
// Embedding backward (layers.py:134-163).  Forward: x[m][:] = b + sum_j lat[m][j] * Wt[j][:],  lat[m][c*ld + e] =
// tables[c][z[m,c]][e]  (row `vocab` of a table = the trainable MASK special, layers.py:146-150).
//   dWt[j][d]  = sum_m lat[m][j] * dx[m][d]      -> partial over 64-row chunks, then vn_reduce_rows
//   dMASK[c][e] = sum_{m : z[m,c] == vocab} sum_d dx[m][d] * Wt[c*ld+e][d]
//   db = colsum(dx)                               (vn_launch_colsum)
// z is the int32 token buffer [B][C][T]; m = b*T + t.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vn_embed_dwt_partial_kernel(const float* __restrict__ dx, const int32_t* __restrict__ z,
                                                                   const float* __restrict__ tables, float* __restrict__ partial,
                                                                   int B, int C, int T, int V1, int ld, int D) {
    // grid: (D/256, row chunks of 64, J/16)   J = C*ld; thread owns one d column and 16 j's
    __shared__ float lat[64][16];
    const int d = blockIdx.x * 256 + threadIdx.x;
    const int m0 = blockIdx.y * 64, j0 = blockIdx.z * 16;
    const int M = B * T, J = C * ld;
    for (int i = threadIdx.x; i < 64 * 16; i += 256) {
        const int mm = i >> 4, jj = i & 15;
        const int m = m0 + mm, j = j0 + jj;
        float v = 0.f;
        if (m < M && j < J) {
            const int c = j / ld, e = j - c * ld;
            const int b = m / T, t = m - b * T;
            const int tok = z[((size_t)b * C + c) * T + t];
            v = tables[((size_t)c * V1 + tok) * ld + e];
        }
        lat[mm][jj] = v;
    }
    __syncthreads();
    if (d >= D) return;
    float acc[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) acc[jj] = 0.f;
    const int mend = m0 + 64 < M ? 64 : M - m0;
    for (int mm = 0; mm < mend; ++mm) {
        const float g = dx[(size_t)(m0 + mm) * D + d];
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) acc[jj] = fmaf(lat[mm][jj], g, acc[jj]);
    }
#pragma unroll
    for (int jj = 0; jj < 16; ++jj)
        if (j0 + jj < J) partial[((size_t)blockIdx.y * J + j0 + jj) * D + d] = acc[jj];
}

// dlat[m][c*ld+e] = (z[m,c] == MASK) ? dx[m] . Wt[c*ld+e] : 0     one wave per row; column sums of dlat = dMASK
__global__ __launch_bounds__(256) void vn_embed_dlat_kernel(const float* __restrict__ dx, const int32_t* __restrict__ z,
                                                            const float* __restrict__ wt, float* __restrict__ dlat, int B,
                                                            int C, int T, int V1, int ld, int D) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int M = B * T, J = C * ld;
    if (m >= M) return;
    const int b = m / T, t = m - b * T;
    const f32x4* g = (const f32x4*)(dx + (size_t)m * D);
    const int nv = D >> 2;
    for (int c = 0; c < C; ++c) {
        const bool masked = z[((size_t)b * C + c) * T + t] == V1 - 1;      // wave-uniform
        for (int e = 0; e < ld; ++e) {
            float acc = 0.f;
            if (masked) {
                const f32x4* w = (const f32x4*)(wt + (size_t)(c * ld + e) * D);
                for (int i = lane; i < nv; i += 64) {
                    const f32x4 a = g[i], bb = w[i];
                    acc += a[0] * bb[0] + a[1] * bb[1] + a[2] * bb[2] + a[3] * bb[3];
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            }
            if (lane == 0) dlat[(size_t)m * J + c * ld + e] = acc;
        }
    }
}

// dtables[c][V1-1][e] = colsum[c*ld+e]
__global__ void vn_embed_dmask_scatter_kernel(const float* __restrict__ colsum, float* __restrict__ dtables, int C, int V1,
                                              int ld) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= C * ld) return;
    const int c = j / ld, e = j - c * ld;
    dtables[((size_t)c * V1 + (V1 - 1)) * ld + e] = colsum[j];
}

int vn_embed_bwd_partial_floats(int B, int T, int C, int ld, int D) { return vn_cdiv(B * T, 64) * C * ld * D; }

int vn_launch_embed_bwd(vn_ctx* ctx, const float* dx, const int32_t* z, const float* tables, const float* wt, float* dtables,
                        float* dwt, float* db, float* partial, float* dlat /* [B*T*C*ld + C*ld] scratch */, int B, int C, int T,
                        int V1, int ld, int D, hipStream_t s) {
    const int M = B * T, J = C * ld, nb = vn_cdiv(M, 64);
    hipLaunchKernelGGL(vn_embed_dwt_partial_kernel, dim3(vn_cdiv(D, 256), nb, vn_cdiv(J, 16)), dim3(256), 0, s, dx, z, tables,
                       partial, B, C, T, V1, ld, D);
    VN_LAUNCH_CHECK(ctx);
    int rc = vn_launch_reduce_rows(ctx, partial, nb, J * D, dwt, s);
    if (rc) return rc;
    hipLaunchKernelGGL(vn_embed_dlat_kernel, dim3(vn_cdiv(M, 4)), dim3(256), 0, s, dx, z, wt, dlat, B, C, T, V1, ld, D);
    VN_LAUNCH_CHECK(ctx);
    float* colsum = dlat + (size_t)M * J;
    if ((rc = vn_launch_colsum(ctx, dlat, M, J, partial, colsum, s))) return rc;
    hipLaunchKernelGGL(vn_embed_dmask_scatter_kernel, dim3(vn_cdiv(J, 64)), dim3(64), 0, s, colsum, dtables, C, V1, ld);
    VN_LAUNCH_CHECK(ctx);
    return vn_launch_colsum(ctx, dx, M, D, partial, db, s);
}

// ---------------------------------------------------------------------------------------------
// Gradient norm + AdamW (train.py:296-299; torch.optim.AdamW defaults + clip_grad_norm_).
//   norm  = ||gscale * g||_2 over the whole gradient vector (non-trainable / padding entries are zero)
//   coef  = min(1, clip / (norm + 1e-6))
//   p    <- p * (1 - lr*wd) ; m <- b1 m + (1-b1) g' ; v <- b2 v + (1-b2) g'^2 ; p <- p - lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
// The squared norm is reduced in double through fixed-size partials (deterministic), read by the update from
// device memory: no host round trip between backward and update.
// ---------------------------------------------------------------------------------------------
#define VN_NORM_BLOCKS 1024
__global__ __launch_bounds__(256) void vn_sumsq_partial_kernel(const float* __restrict__ g, long n4, double* __restrict__ partial) {
    __shared__ double red[4];
    double a = 0.0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256L) {
        const f32x4 v = ((const f32x4*)g)[i];
        a += (double)(v[0] * v[0] + v[1] * v[1]) + (double)(v[2] * v[2] + v[3] * v[3]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(1024) void vn_sumsq_finish_kernel(const double* __restrict__ partial, int n, float gscale,
                                                               float* __restrict__ norm_out) {
    __shared__ double red[1024];
    red[threadIdx.x] = (int)threadIdx.x < n ? partial[threadIdx.x] : 0.0;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *norm_out = (float)(sqrt(red[0]) * (double)gscale);
}

__global__ __launch_bounds__(1024) void vn_sumsq_finish_f64_kernel(const double* __restrict__ partial, int n, double* __restrict__ out) {
    __shared__ double red[1024];
    red[threadIdx.x] = (int)threadIdx.x < n ? partial[threadIdx.x] : 0.0;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
}

// sum of squares in double (the ranks of a ZeRO-1 job add these before taking the root)
int vn_launch_grad_sumsq(vn_ctx* ctx, const float* g, long n, double* partial, double* out, hipStream_t s) {
    if (n % 4) return vn_fail(ctx, VN_ERR_INVALID, "grad_sumsq: length %s%ld must be a multiple of 4", "", n);
    hipLaunchKernelGGL(vn_sumsq_partial_kernel, dim3(VN_NORM_BLOCKS), dim3(256), 0, s, g, n / 4, partial);
    hipLaunchKernelGGL(vn_sumsq_finish_f64_kernel, dim3(1), dim3(1024), 0, s, partial, VN_NORM_BLOCKS, out);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

int vn_launch_grad_norm(vn_ctx* ctx, const float* g, long n, float gscale, double* partial, float* norm_out, hipStream_t s) {
    if (n % 4) return vn_fail(ctx, VN_ERR_INVALID, "grad_norm: length %s%ld must be a multiple of 4", "", n);
    hipLaunchKernelGGL(vn_sumsq_partial_kernel, dim3(VN_NORM_BLOCKS), dim3(256), 0, s, g, n / 4, partial);
    hipLaunchKernelGGL(vn_sumsq_finish_kernel, dim3(1), dim3(1024), 0, s, partial, VN_NORM_BLOCKS, gscale, norm_out);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

__global__ __launch_bounds__(256) void vn_adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, long n, vn_adamw_args a,
                                                       const float* __restrict__ norm) {
    float coef = a.gscale;
    if (a.clip > 0.f) {
        const float c = a.clip / (*norm + 1e-6f);
        coef *= c < 1.0f ? c : 1.0f;
    }
    const float decay = 1.0f - a.lr * a.weight_decay;
    const float step = a.lr / a.bc1, s2 = sqrtf(a.bc2);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        const float gg = g[i] * coef;
        const float mm = a.beta1 * m[i] + (1.0f - a.beta1) * gg;
        const float vv = a.beta2 * v[i] + (1.0f - a.beta2) * gg * gg;
        m[i] = mm;
        v[i] = vv;
        p[i] = p[i] * decay - step * (mm / (sqrtf(vv) / s2 + a.eps));
    }
}

int vn_launch_adamw(vn_ctx* ctx, float* p, const float* g, float* m, float* v, long n, const vn_adamw_args& a,
                    const float* norm, hipStream_t s) {
    if (n <= 0) return VN_OK;
    const long nb = (n + 255) / 256;
    hipLaunchKernelGGL(vn_adamw_kernel, dim3((int)(nb < 16384 ? nb : 16384)), dim3(256), 0, s, p, g, m, v, n, a, norm);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

// packed-layout helper: i32 tokens <- i64 (B,C,T) is vn_launch_i64_to_i32 (elementwise.hip)

// =============================================================================================
// LoRA fine-tuning (train.py:696 `lora.mark_only_lora_as_trainable`; loralib Linear(r=8, lora_alpha=1)):
//   y = x W^T + s * (x A^T) B^T,  s = alpha / r;  W frozen, A (r x K) and B (N x r) trained.
// The engine keeps W_eff = W + s B A in the blob (forward and dX need nothing else) and computes the two small
// gradients from rank-r projections.  A is stored TRANSPOSED (At [K][r]) so every rank-r operand is [C][r] row-major:
//   down:  H[m][j]  = scale * sum_c Y[m][c] * P[c][j]          (h = x At ;  dh = s * dY B)          one wave per row
//   up:    G[c][j]  = scale * sum_m Y[m][c] * H[m][j]          (dB = s * dY^T h ;  dAt = x^T dh)     partials + reduce
//   merge: W_eff[n][k] = W[n][k] + s * sum_j B[n][j] * At[k][j]
// All three stream Y / W once (HBM-bound); r is fixed to 8 (transformer.py:22 LORA_R).
// =============================================================================================
#define VN_LORA_R 8

__global__ __launch_bounds__(256) void vn_lora_down_kernel(const float* __restrict__ Y, int ldy, const float* __restrict__ P,
                                                           float* __restrict__ H, int M, int Cn, float scale) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    float acc[VN_LORA_R];
#pragma unroll
    for (int j = 0; j < VN_LORA_R; ++j) acc[j] = 0.f;
    const float* y = Y + (size_t)m * ldy;
    for (int c = lane; c < Cn; c += 64) {
        const float v = y[c];
        const f32x4 p0 = *(const f32x4*)(P + (size_t)c * VN_LORA_R), p1 = *(const f32x4*)(P + (size_t)c * VN_LORA_R + 4);
        acc[0] = fmaf(v, p0[0], acc[0]); acc[1] = fmaf(v, p0[1], acc[1]); acc[2] = fmaf(v, p0[2], acc[2]); acc[3] = fmaf(v, p0[3], acc[3]);
        acc[4] = fmaf(v, p1[0], acc[4]); acc[5] = fmaf(v, p1[1], acc[5]); acc[6] = fmaf(v, p1[2], acc[6]); acc[7] = fmaf(v, p1[3], acc[7]);
    }
#pragma unroll
    for (int j = 0; j < VN_LORA_R; ++j)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[j] += __shfl_xor(acc[j], o);
    if (lane == 0) {
        f32x4 a = {acc[0] * scale, acc[1] * scale, acc[2] * scale, acc[3] * scale};
        f32x4 b = {acc[4] * scale, acc[5] * scale, acc[6] * scale, acc[7] * scale};
        *(f32x4*)(H + (size_t)m * VN_LORA_R) = a;
        *(f32x4*)(H + (size_t)m * VN_LORA_R + 4) = b;
    }
}

int vn_launch_lora_down(vn_ctx* ctx, const float* Y, int ldy, const float* P, float* H, int M, int Cn, float scale,
                        hipStream_t s) {
    hipLaunchKernelGGL(vn_lora_down_kernel, dim3(vn_cdiv(M, 4)), dim3(256), 0, s, Y, ldy, P, H, M, Cn, scale);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

// partial[chunk][c][j] over 64-row chunks; thread owns column c
__global__ __launch_bounds__(256) void vn_lora_up_partial_kernel(const float* __restrict__ Y, int ldy, const float* __restrict__ H,
                                                                 float* __restrict__ partial, int M, int Cn) {
    __shared__ float hs[64][VN_LORA_R];
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int m0 = blockIdx.y * 64;
    const int mend = m0 + 64 < M ? 64 : M - m0;
    for (int i = threadIdx.x; i < 64 * VN_LORA_R; i += 256) {
        const int mm = i / VN_LORA_R;
        hs[mm][i - mm * VN_LORA_R] = mm < mend ? H[(size_t)(m0 + mm) * VN_LORA_R + (i - mm * VN_LORA_R)] : 0.f;
    }
    __syncthreads();
    if (c >= Cn) return;
    float acc[VN_LORA_R];
#pragma unroll
    for (int j = 0; j < VN_LORA_R; ++j) acc[j] = 0.f;
    for (int mm = 0; mm < mend; ++mm) {
        const float v = Y[(size_t)(m0 + mm) * ldy + c];
#pragma unroll
        for (int j = 0; j < VN_LORA_R; ++j) acc[j] = fmaf(v, hs[mm][j], acc[j]);
    }
    float* dst = partial + ((size_t)blockIdx.y * Cn + c) * VN_LORA_R;
    *(f32x4*)dst = f32x4{acc[0], acc[1], acc[2], acc[3]};
    *(f32x4*)(dst + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
}

__global__ __launch_bounds__(256) void vn_scale_kernel(float* __restrict__ x, long n, float s) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) x[i] *= s;
}

int vn_lora_up_partial_floats(int M, int Cn) { return vn_cdiv(M, 64) * Cn * VN_LORA_R; }

int vn_launch_lora_up(vn_ctx* ctx, const float* Y, int ldy, const float* H, float* G, float* partial, int M, int Cn, float scale,
                      hipStream_t s) {
    const int nb = vn_cdiv(M, 64);
    hipLaunchKernelGGL(vn_lora_up_partial_kernel, dim3(vn_cdiv(Cn, 256), nb), dim3(256), 0, s, Y, ldy, H, partial, M, Cn);
    VN_LAUNCH_CHECK(ctx);
    int rc = vn_launch_reduce_rows(ctx, partial, nb, Cn * VN_LORA_R, G, s);
    if (rc) return rc;
    if (scale != 1.0f) {
        hipLaunchKernelGGL(vn_scale_kernel, dim3(vn_cdiv(Cn * VN_LORA_R, 256)), dim3(256), 0, s, G, (long)Cn * VN_LORA_R, scale);
        VN_LAUNCH_CHECK(ctx);
    }
    return VN_OK;
}

__global__ __launch_bounds__(256) void vn_lora_merge_kernel(const float* __restrict__ W, const float* __restrict__ Bm,
                                                            const float* __restrict__ At, float* __restrict__ Weff, int N, int K,
                                                            float scale) {
    const int k4 = K >> 2;
    const long total = (long)N * k4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        const int n = (int)(i / k4), k = 4 * (int)(i - (long)n * k4);
        f32x4 w = ((const f32x4*)W)[i];
        const f32x4 b0 = *(const f32x4*)(Bm + (size_t)n * VN_LORA_R), b1 = *(const f32x4*)(Bm + (size_t)n * VN_LORA_R + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x4 a0 = *(const f32x4*)(At + (size_t)(k + e) * VN_LORA_R), a1 = *(const f32x4*)(At + (size_t)(k + e) * VN_LORA_R + 4);
            const float d = b0[0] * a0[0] + b0[1] * a0[1] + b0[2] * a0[2] + b0[3] * a0[3] + b1[0] * a1[0] + b1[1] * a1[1] +
                            b1[2] * a1[2] + b1[3] * a1[3];
            w[e] = fmaf(scale, d, w[e]);
        }
        ((f32x4*)Weff)[i] = w;
    }
}

int vn_launch_lora_merge(vn_ctx* ctx, const float* W, const float* Bm, const float* At, float* Weff, int N, int K, float scale,
                         hipStream_t s) {
    const long total = (long)N * (K / 4);
    hipLaunchKernelGGL(vn_lora_merge_kernel, dim3((int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192)), dim3(256), 0, s, W,
                       Bm, At, Weff, N, K, scale);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

// ---------------------------------------------------------------------------------------------
// Validation statistics (train.py:327-377 `val_loop` + :155-215 `accuracy` / `_metrics`): for every row of V logits
//   row_loss = label-smoothed CE against target[row]           (every row has a target here; the caller masks)
//   rank     = number of classes with a logit strictly greater than the target's  (target in top-k  <=>  rank < k)
// One wave per row.
// ---------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void vn_eval_rows_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                           long rows, float ls, float* __restrict__ row_loss,
                                                           int32_t* __restrict__ rank) {
    const int lane = threadIdx.x & 63;
    const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (row >= rows) return;
    constexpr int V = VEC * 256;
    const f32x4* xr = (const f32x4*)(logits + row * V);
    int t = (int)target[row];
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    f32x4 x[VEC];
    float mx = -INFINITY, sx = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        x[i] = xr[lane + 64 * i];
#pragma unroll
        for (int e = 0; e < 4; ++e) { mx = fmaxf(mx, x[i][e]); sx += x[i][e]; }
    }
    const float xt = logits[row * V + t];
    int gt = 0;
    float se = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) gt += x[i][e] > xt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, o));
        sx += __shfl_xor(sx, o);
        gt += __shfl_xor(gt, o);
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) se += expf(x[i][e] - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
    if (lane == 0) {
        const float logZ = mx + logf(se);
        row_loss[row] = (1.0f - ls) * (logZ - xt) + ls * (logZ - sx / (float)V);
        rank[row] = gt;
    }
}

int vn_launch_eval_rows(vn_ctx* ctx, const float* logits, const int64_t* target, long rows, int V, float ls, float* row_loss,
                        int32_t* rank, hipStream_t s) {
    if (V != 1024 && V != 256) return vn_fail(ctx, VN_ERR_UNSUPPORTED, "eval: vocab=%s%ld unsupported", "", V);
    const int blocks = (int)((rows + 3) / 4);
    if (V == 1024) hipLaunchKernelGGL(vn_eval_rows_kernel<4>, dim3(blocks), dim3(256), 0, s, logits, target, rows, ls, row_loss, rank);
    else hipLaunchKernelGGL(vn_eval_rows_kernel<1>, dim3(blocks), dim3(256), 0, s, logits, target, rows, ls, row_loss, rank);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}
