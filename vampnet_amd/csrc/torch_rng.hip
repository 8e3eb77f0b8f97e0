// torch's CPU random stream on the device — seeded PARITY mode at device speed.
//
// The reference reseeds torch's global CPU generator (`seed=`, transformer.py:711-712) and then draws, per sampling step,
// `multinomial(1)` (= one `exponential_` over (B*N, 1024)) and the Gumbel `uniform_(1e-20, 1)` over (B, N)
// (transformer.py:28-30, :1024-1028, :1038-1074).  On the host those 2.4 M exponentials per item per step cost ≈ 57 ms
// (0.7 s per clip).  Both distributions are plain functions of the sequential mt19937 stream of at::CPUGeneratorImpl:
//   exponential_  (float tensor): r64 = (mt[2i] << 32) | mt[2i+1];  u = (r64 & (2^53-1)) * 2^-53;  x = float(-log1p(-u))   (double)
//   uniform_(a,b) (float tensor): u = float(mt[i] & (2^24-1)) * 2^-24;  x = u * (b - a) + a                                   (float)
// (pinned against torch on the host by tests/test_host_logic.py::test_torch_rng_stream_formulas and on the device by
// tests/test_gpu_kernels.py::test_torch_rng_on_device).  So the engine can produce the SAME numbers from the generator's state:
// one workgroup walks the mt19937 recurrence (624-word blocks; thread t owns words t, 227+t, 454+t, whose updates chain through
// its own registers: one barrier per block, double buffered in LDS), tempering and the distribution transforms run at full width.  The host hands over the generator state before a
// generate() call and takes the advanced state back afterwards, so torch's global generator ends where the reference's would.
#include "vn_common.h"

#define MT_N 624
#define MT_M 397

__device__ __forceinline__ uint32_t mt_twist(uint32_t u, uint32_t v) {
    const uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
    return (y >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// state[624] + *pos (index of the next word to emit; 624 = block exhausted) -> n tempered outputs; state / pos advanced.
// out may be NULL (advance only: another rank's rows of a sharded batch).
// Chunked use (vn_mt19937_generate_chunks): block c starts from states[c] at position 0 and writes words [c*chunk, (c+1)*chunk) of
// the stream (pos == NULL, nothing written back).
__global__ __launch_bounds__(256) void vn_mt19937_kernel(uint32_t* __restrict__ state, int32_t* __restrict__ pos,
                                                         uint32_t* __restrict__ out, long n, long chunk) {
    __shared__ uint32_t buf[2][MT_N + 8];
    const int tid = threadIdx.x;
    if (chunk > 0) {
        state += (size_t)blockIdx.x * MT_N;
        const long first = (long)blockIdx.x * chunk;
        if (out) out += first;
        n = n - first < chunk ? n - first : chunk;
    }
    for (int i = tid; i < MT_N; i += 256) buf[0][i] = state[i];
    __syncthreads();
    int cur = 0;
    int p = pos ? *pos : 0;
    long done = 0;
    if (p < MT_N && n > 0) {             // rest of the block the generator was in
        const long take = (long)(MT_N - p) < n ? (MT_N - p) : n;
        if (out)
            for (int i = tid; i < take; i += 256) out[i] = mt_temper(buf[0][p + i]);
        p += (int)take;
        done = take;
    }
    // whole blocks: next_state().  new[i] = X[(i + 397) mod 624] ^ twist(old[i], Y[i + 1]) where X / Y are NEW words once the index
    // wraps.  Thread t < 227 owns words t, 227 + t, 454 + t: new[227 + t] needs new[t] and new[454 + t] needs new[227 + t] — its OWN
    // results — so the three "phases" chain through registers; the only foreign new word is new[0] for word 623, which its owner
    // (thread 169) recomputes.  One barrier per 624 words; every word is tempered and emitted by the thread that made it.
    while (done < n) {
        const int take = (n - done) < MT_N ? (int)(n - done) : MT_N;
        const uint32_t* o = buf[cur];
        uint32_t* w = buf[cur ^ 1];
        uint32_t* dst = out ? out + done : nullptr;
        if (tid < 227) {
            const uint32_t a0 = o[tid], a1 = o[tid + 1], am = o[tid + MT_M];
            const uint32_t b0 = o[227 + tid], b1 = o[228 + tid];
            const uint32_t v0 = am ^ mt_twist(a0, a1);
            const uint32_t v1 = v0 ^ mt_twist(b0, b1);
            w[tid] = v0;
            w[227 + tid] = v1;
            if (dst && tid < take) dst[tid] = mt_temper(v0);
            if (dst && 227 + tid < take) dst[227 + tid] = mt_temper(v1);
            if (tid < 170) {                               // words 454 .. 623
                const uint32_t c0 = o[454 + tid];
                uint32_t c1;
                if (tid < 169) c1 = o[455 + tid];
                else c1 = o[MT_M] ^ mt_twist(o[0], o[1]);  // new[0], recomputed by the owner of word 623
                const uint32_t v2 = v1 ^ mt_twist(c0, c1);
                w[454 + tid] = v2;
                if (dst && 454 + tid < take) dst[454 + tid] = mt_temper(v2);
            }
        }
        __syncthreads();
        cur ^= 1;
        p = take;
        done += take;
    }
    if (pos == nullptr) return;
    for (int i = tid; i < MT_N; i += 256) state[i] = buf[cur][i];
    if (tid == 0) *pos = p;
}

extern "C" int vn_mt19937_generate(vn_ctx* ctx, uint32_t* state624, int32_t* pos, uint32_t* out_raw, int64_t n, void* stream) {
    if (!ctx || !state624 || !pos || n < 0) return VN_ERR_INVALID;
    if (n == 0) return VN_OK;
    hipLaunchKernelGGL(vn_mt19937_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, state624, pos, out_raw, (long)n, 0L);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

extern "C" int vn_mt19937_generate_chunks(vn_ctx* ctx, const uint32_t* states, int n_chunks, uint32_t* out_raw, int64_t chunk_words,
                                          int64_t total_words, void* stream) {
    if (!ctx || !states || !out_raw || n_chunks <= 0 || chunk_words <= 0 || total_words <= (int64_t)(n_chunks - 1) * chunk_words ||
        total_words > (int64_t)n_chunks * chunk_words)
        return VN_ERR_INVALID;
    hipLaunchKernelGGL(vn_mt19937_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, (uint32_t*)states, (int32_t*)nullptr,
                       out_raw, (long)total_words, (long)chunk_words);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Jump-ahead (vampnet_amd/mt_jump.py): with g = x^J mod phi as a 624-word bit vector, the state J steps ahead is
//   x_{J+k} = XOR_{i : g_i} x_{k+i},  k = 0 .. 623,   over the UNTEMPERED stream x_0 = state[pos], x_1, ...
// One workgroup per target: 34 consecutive blocks (block 0 = the state array) are laid out in LDS (84 KiB), then every thread
// slides the polynomial's ~10 k set bits over its three output words (uniform loop, conflict-free consecutive LDS reads).
// The result is a state array at position 0.
// ---------------------------------------------------------------------------------------------------------------------
#define MT_JUMP_BLOCKS 34
// base_index / poly_index (both or neither): target b starts from the state array base_index[b] of `state` AT POSITION 0 and applies
// polynomial poly_index[b] — the second level of a two-level plan (vn_mt19937_jump_indexed: first the states at the starts of all
// sampling steps of a call from the generator's state, then every step's chunk starts from ITS state with the polynomials of the
// chunk offsets, which are the same for every step)
__global__ __launch_bounds__(256) void vn_mt19937_jump_kernel(const uint32_t* __restrict__ state, const int32_t* __restrict__ pos,
                                                              const uint32_t* __restrict__ polys, uint32_t* __restrict__ out_states,
                                                              const int32_t* __restrict__ base_index, const int32_t* __restrict__ poly_index) {
    extern __shared__ uint32_t Y[];                       // [MT_JUMP_BLOCKS][624]
    const int tid = threadIdx.x;
    if (base_index) state += (size_t)base_index[blockIdx.x] * MT_N;
    for (int i = tid; i < MT_N; i += 256) Y[i] = state[i];
    __syncthreads();
    for (int blk = 0; blk + 1 < MT_JUMP_BLOCKS; ++blk) {
        const uint32_t* o = Y + blk * MT_N;
        uint32_t* w = Y + (blk + 1) * MT_N;
        if (tid < 227) {
            const uint32_t v0 = o[tid + MT_M] ^ mt_twist(o[tid], o[tid + 1]);
            const uint32_t v1 = v0 ^ mt_twist(o[227 + tid], o[228 + tid]);
            w[tid] = v0;
            w[227 + tid] = v1;
            if (tid < 170) {
                const uint32_t c1 = tid < 169 ? o[455 + tid] : (o[MT_M] ^ mt_twist(o[0], o[1]));
                w[454 + tid] = v1 ^ mt_twist(o[454 + tid], c1);
            }
        }
        __syncthreads();
    }
    const uint32_t* poly = polys + (size_t)(poly_index ? poly_index[blockIdx.x] : blockIdx.x) * MT_N;
    const uint32_t* x = Y + (base_index ? 0 : *pos);      // x_0 = state[pos]; pos == 624 starts at block 1
    uint32_t a0 = 0, a1 = 0, a2 = 0;
    const bool has2 = tid + 512 < MT_N;
    for (int wd = 0; wd < MT_N; ++wd) {
        uint32_t bits = __builtin_amdgcn_readfirstlane(poly[wd]);
        while (bits) {
            const int i = 32 * wd + __builtin_ctz(bits);
            bits &= bits - 1;
            a0 ^= x[tid + i];
            a1 ^= x[tid + 256 + i];
            if (has2) a2 ^= x[tid + 512 + i];
        }
    }
    uint32_t* dst = out_states + (size_t)blockIdx.x * MT_N;
    dst[tid] = a0;
    dst[tid + 256] = a1;
    if (has2) dst[tid + 512] = a2;
}

extern "C" int vn_mt19937_jump(vn_ctx* ctx, const uint32_t* state624, const int32_t* pos, const uint32_t* polys, int n_targets,
                               uint32_t* out_states, void* stream) {
    if (!ctx || !state624 || !pos || !polys || !out_states || n_targets <= 0) return VN_ERR_INVALID;
    const size_t lds = (size_t)MT_JUMP_BLOCKS * MT_N * sizeof(uint32_t);
    if (!(ctx->attr_mask & VN_ATTR_MT_JUMP)) {
        VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)vn_mt19937_jump_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ctx->attr_mask |= VN_ATTR_MT_JUMP;
    }
    hipLaunchKernelGGL(vn_mt19937_jump_kernel, dim3(n_targets), dim3(256), lds, (hipStream_t)stream, state624, pos, polys, out_states,
                       (const int32_t*)nullptr, (const int32_t*)nullptr);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

extern "C" int vn_mt19937_jump_indexed(vn_ctx* ctx, const uint32_t* base_states, const int32_t* base_index, const uint32_t* polys,
                                       const int32_t* poly_index, int n_targets, uint32_t* out_states, void* stream) {
    if (!ctx || !base_states || !base_index || !polys || !poly_index || !out_states || n_targets <= 0) return VN_ERR_INVALID;
    const size_t lds = (size_t)MT_JUMP_BLOCKS * MT_N * sizeof(uint32_t);
    if (!(ctx->attr_mask & VN_ATTR_MT_JUMP)) {
        VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)vn_mt19937_jump_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ctx->attr_mask |= VN_ATTR_MT_JUMP;
    }
    hipLaunchKernelGGL(vn_mt19937_jump_kernel, dim3(n_targets), dim3(256), lds, (hipStream_t)stream, base_states, (const int32_t*)nullptr, polys,
                       out_states, base_index, poly_index);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

// x[i] = float(-log1p(-u53(raw[2i], raw[2i+1])))          at::exponential_distribution<double> on a float tensor, lambda = 1
__global__ __launch_bounds__(256) void vn_torch_exponential_kernel(const uint32_t* __restrict__ raw, float* __restrict__ out, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        const uint2 r = ((const uint2*)raw)[i];
        const unsigned long long r64 = ((unsigned long long)r.x << 32) | r.y;
        const double u = (double)(r64 & ((1ULL << 53) - 1)) * 0x1.0p-53;
        out[i] = (float)(-log1p(-u));
    }
}

extern "C" int vn_torch_exponential_f32(vn_ctx* ctx, const uint32_t* raw, float* out, int64_t n, void* stream) {
    if (!ctx || !raw || !out || n < 0) return VN_ERR_INVALID;
    if (n == 0) return VN_OK;
    const long nb = (n + 255) / 256;
    hipLaunchKernelGGL(vn_torch_exponential_kernel, dim3((unsigned)(nb < 8192 ? nb : 8192)), dim3(256), 0, (hipStream_t)stream, raw,
                       out, (long)n);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

// x[i] = float(raw[i] & (2^24 - 1)) * 2^-24 * (hi - lo) + lo        at::uniform_real_distribution<float>
__global__ __launch_bounds__(256) void vn_torch_uniform_kernel(const uint32_t* __restrict__ raw, float* __restrict__ out, long n,
                                                               float lo, float hi) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        const float u = (float)(raw[i] & ((1u << 24) - 1)) * 0x1.0p-24f;
        out[i] = u * (hi - lo) + lo;
    }
}

extern "C" int vn_torch_uniform_f32(vn_ctx* ctx, const uint32_t* raw, float* out, int64_t n, float lo, float hi, void* stream) {
    if (!ctx || !raw || !out || n < 0) return VN_ERR_INVALID;
    if (n == 0) return VN_OK;
    const long nb = (n + 255) / 256;
    hipLaunchKernelGGL(vn_torch_uniform_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, (hipStream_t)stream, raw, out,
                       (long)n, lo, hi);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}
