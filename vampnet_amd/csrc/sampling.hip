// Sampling kernels of the iterative parallel decoder (HBM-bound; one pass over the logits).
//
//  vn_sample_kernel  = sample_from_logits (vampnet/modules/transformer.py:952-1034) fused with the
//                      where()/+inf bookkeeping of generate (:879-900)
//  vn_remask_kernel  = mask_by_random_topk (:1038-1074) + num_to_mask clamp (:903-913) + re-mask and
//                      unflatten (:922-932)
//
// RNG: in parity mode the caller supplies the exact torch-CPU draws (Exp(1) for the multinomial race,
// U(1e-20,1) for the Gumbel noise); with a NULL noise pointer a counter-based
// Philox4x32-10 stream keyed by (seed, step) and indexed by (row, element) is used instead, so results
// do not depend on launch geometry or on how a batch is sharded over GPUs (row = GLOBAL row index is the
// caller's business: it offsets `row0`).
#include "vn_common.h"

__device__ __forceinline__ float vn_u01_open(uint32_t x) {   // (0, 1]
    return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f);
}

// ---------------------------------------------------------------------------------------------
// One 64-lane wave per (b, t, c) row of V logits (V = 16 * 64: 4 float4 per lane, coalesced).
//   p      = softmax(logits / temperature)            (exp(x - max) / sum, like torch's CPU softmax)
//   token  = argmax_i p_i / E_i   (E ~ Exp(1))  == torch.multinomial(p, 1)   if do_sample
//          = argmax_i logits_i                                                otherwise
//   ties -> lowest index (torch CPU argmax).  Rows whose token is not MASK are skipped entirely
//   (their logits are never read): sampled = old token, psel = +inf.
// Algorithmic bytes per masked row: 4*V logits (+ 4*V noise in parity mode) in, 8 out.
// ---------------------------------------------------------------------------------------------
template <int VEC>   // V = VEC * 256
__global__ __launch_bounds__(256) void vn_sample_kernel(vn_sample_args a) {
    const int lane = threadIdx.x & 63;
    const int Cp = a.C - a.n_cond;
    const int N = a.T * Cp;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)a.B * N) return;
    const int b = (int)(row / N), n = (int)(row - (long)b * N);
    const int t = n / Cp, c = n - t * Cp;
    const int32_t cur = a.z[((size_t)b * a.C + a.n_cond + c) * a.T + t];
    if (cur != a.V) {                    // wave-uniform
        if (lane == 0) {
            a.sampled[row] = cur;
            a.psel[row] = INFINITY;
        }
        return;
    }
    const f32x4* lrow = (const f32x4*)(a.logits + (size_t)row * a.V);
    f32x4 x[VEC];
    float mx = -INFINITY;
    float best_l = -INFINITY;
    int best_li = 0;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        const f32x4 l = lrow[lane + 64 * i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (l[e] > best_l) { best_l = l[e]; best_li = (lane + 64 * i) * 4 + e; }   // ascending index: strict >
            x[i][e] = l[e] / a.temperature;
            mx = fmaxf(mx, x[i][e]);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            x[i][e] = expf(x[i][e] - mx);
            sum += x[i][e];
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);

    float best_s, best_p;
    int best_i;
    if (a.do_sample) {
        best_s = -INFINITY; best_p = 0.f; best_i = 0;
        const f32x4* nrow = a.exp_noise ? (const f32x4*)(a.exp_noise + (size_t)row * a.V) : nullptr;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            f32x4 e4;
            if (nrow) {
                e4 = nrow[lane + 64 * i];
            } else {
                const long grow = ((long)(b / a.call_batch) * a.global_batch + a.batch_offset + b % a.call_batch) * N + n;   // global row
                const uint4 r = vn_philox4x32_10(make_uint4((uint32_t)grow, (uint32_t)(grow >> 32), lane + 64 * i, 0x53414d50u),
                                                 make_uint2((uint32_t)a.seed ^ (a.step * 0x9E3779B9u), (uint32_t)(a.seed >> 32)));
                e4[0] = fmaxf(-logf(vn_u01_open(r.x)), 1e-30f);
                e4[1] = fmaxf(-logf(vn_u01_open(r.y)), 1e-30f);
                e4[2] = fmaxf(-logf(vn_u01_open(r.z)), 1e-30f);
                e4[3] = fmaxf(-logf(vn_u01_open(r.w)), 1e-30f);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float p = x[i][e] / sum;
                const float sc = p / e4[e];
                if (sc > best_s) { best_s = sc; best_p = p; best_i = (lane + 64 * i) * 4 + e; }
            }
        }
    } else {
        best_s = best_l; best_i = best_li;
        // p at the argmax index: that element lives in this lane
        const int ii = (best_li >> 2) - lane;      // = 64*i
        best_p = 0.f;
#pragma unroll
        for (int i = 0; i < VEC; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (ii == 64 * i && (best_li & 3) == e) best_p = x[i][e] / sum;
    }
    // wave arg-max with lowest-index tie break
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float os = __shfl_xor(best_s, o);
        const float op = __shfl_xor(best_p, o);
        const int oi = __shfl_xor(best_i, o);
        if (os > best_s || (os == best_s && oi < best_i)) { best_s = os; best_p = op; best_i = oi; }
    }
    if (lane == 0) {
        a.sampled[row] = best_i;
        a.psel[row] = best_p;
    }
}

// ---------------------------------------------------------------------------------------------
// Nucleus (top-p) filtering, transformer.py:1001-1016, in place on the logits of the currently-masked rows:
//   v, idx = logits.sort(descending); cum = softmax(v).cumsum(-1)  (fp64 running sum rounded to fp32 per element,
//   as torch's CPU cumsum does); remove[j] = cum[j-1] > top_p (the first token over the threshold is kept);
//   logits[removed] = -inf.
// One wave per row.  The descending rank of every logit is found by counting (1024 broadcast LDS reads x 16
// compares per lane; ties broken by index), probabilities are scattered to their sorted position in LDS, an
// exclusive fp64 prefix sum over the sorted order (16 sequential adds per lane + a 64-lane shuffle scan) gives the
// mass strictly before each position, and the decision is gathered back by rank.  HBM: 4*V read + 4*V written per row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vn_top_p_kernel(float* __restrict__ logits, const int32_t* __restrict__ z,
                                                       int B, int T, int C, int n_cond, int V, float top_p) {
    __shared__ float s_log[4][1024];
    __shared__ float s_p[4][1024];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int Cp = C - n_cond, N = T * Cp;
    const long row = (long)blockIdx.x * 4 + w;
    if (row >= (long)B * N) return;
    const int b = (int)(row / N), n = (int)(row - (long)b * N);
    const int t = n / Cp, c = n - t * Cp;
    if (z[((size_t)b * C + n_cond + c) * T + t] != V) return;          // wave-uniform: only rows being sampled
    f32x4* lrow = (f32x4*)(logits + (size_t)row * V);
    float x[16];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 l = lrow[lane + 64 * i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            x[4 * i + e] = l[e];
            s_log[w][(lane + 64 * i) * 4 + e] = l[e];
            mx = fmaxf(mx, l[e]);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float p[16], sum = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { p[k] = expf(x[k] - mx); sum += p[k]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    __builtin_amdgcn_wave_barrier();
    // descending rank of each of this lane's 16 logits: #(l_j > l_i) + #(l_j == l_i and j < i)
    int rank[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) rank[k] = 0;
    for (int jj = 0; jj < 1024; ++jj) {
        const float lj = s_log[w][jj];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int idx = (lane + 64 * (k >> 2)) * 4 + (k & 3);
            rank[k] += (lj > x[k]) || (lj == x[k] && jj < idx);
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) s_p[w][rank[k]] = p[k] / sum;
    __builtin_amdgcn_wave_barrier();
    // exclusive fp64 prefix over the sorted order; lane owns sorted positions [16*lane, 16*lane + 16)
    double run = 0.0, pre[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { pre[k] = run; run += (double)s_p[w][16 * lane + k]; }
    double incl = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    const double base = incl - run;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int pos = 16 * lane + k;
        // cum[pos-1] = fp32(fp64 sum of positions 0..pos-1); position 0 is never removed (F.pad(..., value=False))
        s_log[w][pos] = (pos > 0 && (float)(base + pre[k]) > top_p) ? 1.0f : 0.0f;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = s_log[w][rank[4 * i + e]] != 0.0f ? -INFINITY : x[4 * i + e];
        lrow[lane + 64 * i] = o;
    }
}

int vn_launch_top_p(vn_ctx* ctx, float* logits, const int32_t* z, int B, int T, int C, int n_cond, int V, float top_p,
                    hipStream_t s) {
    if (V != 1024) return vn_fail(ctx, VN_ERR_UNSUPPORTED, "top_p: vocab=%s%ld unsupported (1024)", "", V);
    const long rows = (long)B * T * (C - n_cond);
    if (rows <= 0) return VN_OK;
    hipLaunchKernelGGL(vn_top_p_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, logits, z, B, T, C, n_cond, V,
                       top_p);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

int vn_launch_sample(vn_ctx* ctx, const vn_sample_args& a, hipStream_t s) {
    const long rows = (long)a.B * a.T * (a.C - a.n_cond);
    if (rows <= 0) return VN_OK;
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (a.V == 1024) hipLaunchKernelGGL(vn_sample_kernel<4>, grid, block, 0, s, a);
    else if (a.V == 256) hipLaunchKernelGGL(vn_sample_kernel<1>, grid, block, 0, s, a);
    else return vn_fail(ctx, VN_ERR_UNSUPPORTED, "sample: vocab=%s%ld unsupported (1024 or 256)", "", a.V);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

// ---------------------------------------------------------------------------------------------
// Re-masking: one 1024-thread workgroup per batch row, N = T*Cp confidences staged in LDS.
//   conf_n = log(psel_n) + mask_temp * gumbel(u_n)          (psel = +inf on known tokens -> conf = +inf)
//   k      = last_step ? k_sched : max(1, min(count_masked - 1, k_sched))
//   mask_n = conf_n < sorted(conf)[k]   <=>   #{ j : conf_j <= conf_n } <= k      (strict '<', ties stay)
// The count form needs no sort and reproduces torch's tie semantics exactly; only currently-masked
// positions can be re-masked (conf = +inf elsewhere), so only those are ranked: O(N * N_masked) LDS
// broadcast reads, N <= 2300 (coarse) / 1730 (c2f); up to four blocks per item, each deciding a slice of the positions.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void vn_remask_kernel(vn_remask_args a) {
    extern __shared__ __attribute__((aligned(16))) float conf[];
    __shared__ int s_count;
    const int b = blockIdx.x;
    const int Cp = a.C - a.n_cond;
    const int N = a.T * Cp;
    const int tid = threadIdx.x;
    if (tid == 0) s_count = 0;
    __syncthreads();
    int local = 0;
    for (int n = tid; n < N; n += 1024) {
        const float ps = a.psel[(size_t)b * N + n];
        float u;
        if (a.unif_noise) {
            u = a.unif_noise[(size_t)b * N + n];
        } else {
            const long gb = (long)(b / a.call_batch) * a.global_batch + a.batch_offset + b % a.call_batch;
            const uint4 r = vn_philox4x32_10(make_uint4((uint32_t)gb, (uint32_t)(gb >> 32), (uint32_t)n, 0x4d41534bu),
                                             make_uint2((uint32_t)a.seed ^ (a.step * 0x9E3779B9u), (uint32_t)(a.seed >> 32)));
            u = fmaxf(((float)(r.x >> 8) + 0.5f) * (1.0f / 16777216.0f), 1e-20f);
        }
        const float gmb = -logf(-logf(u));
        conf[n] = logf(ps) + a.mask_temp * gmb;
        local += (ps != INFINITY);
    }
    // block-wide count of masked positions
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if ((tid & 63) == 0 && local) atomicAdd(&s_count, local);
    __syncthreads();
    const int count = s_count;
    long k = (long)a.k_sched[b];
    if (!a.last_step) {
        long lim = (long)count - 1;
        k = k < lim ? k : lim;
        k = k > 1 ? k : 1;
    }
    if (k > N - 1) k = N - 1;
    if (k < 0) k = 0;

    // gridDim.y blocks share an item: every block ranks against the whole conf[] (recomputed per block: N cheap evaluations) but
    // decides only its own slice of positions — the O(N^2) counting is what takes the time and one block per item left 248 CUs idle
    const int chunk = (N + gridDim.y - 1) / gridDim.y;
    const int n_lo = blockIdx.y * chunk, n_hi = n_lo + chunk < N ? n_lo + chunk : N;
    for (int n = n_lo + tid; n < n_hi; n += 1024) {
        const float cn = conf[n];
        const int t = n / Cp, c = n - t * Cp;
        const size_t zi = ((size_t)b * a.C + a.n_cond + c) * a.T + t;
        const int32_t tok = a.sampled[(size_t)b * N + n];
        bool remask = false;
        if (cn != INFINITY) {
            int le = 0;
            for (int jj = 0; jj < N; ++jj) le += (conf[jj] <= cn);
            remask = (long)le <= k;
        }
        a.z[zi] = remask ? a.V : tok;
        if (a.out_sampled) a.out_sampled[zi] = tok;
    }
    if (a.out_sampled && a.n_cond > 0 && blockIdx.y == 0) {
        const int nc = a.n_cond * a.T;
        for (int i = tid; i < nc; i += 1024) {
            const size_t zi = (size_t)b * a.C * a.T + i;
            a.out_sampled[zi] = a.z[zi];
        }
    }
}

int vn_launch_remask(vn_ctx* ctx, const vn_remask_args& a, hipStream_t s) {
    const int N = a.T * (a.C - a.n_cond);
    if (a.B <= 0 || N <= 0) return VN_OK;
    const size_t lds = (size_t)N * sizeof(float);
    if (lds > 120 * 1024) return vn_fail(ctx, VN_ERR_INVALID, "remask: T*Cp=%s%ld too large", "", N);
    if (!(ctx->attr_mask & VN_ATTR_REMASK)) {
        VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)vn_remask_kernel,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
        ctx->attr_mask |= VN_ATTR_REMASK;
    }
    const int slices = N >= 2048 ? 4 : (N >= 512 ? 2 : 1);          // <= 1024 positions per block: one per thread
    hipLaunchKernelGGL(vn_remask_kernel, dim3(a.B, slices), dim3(1024), lds, s, a);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}
