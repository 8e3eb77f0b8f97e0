// Training-mode self-attention for gfx950: forward with probability dropout + log-sum-exp stash, and the flash-style
// backward (no T x T tensor is ever materialised), exact fp32 on v_mfma_f32_16x16x4_f32.
//
// Reference semantics (vampnet/modules/transformer.py:234-254):
//   S = q k^T / 8 + bias[h, bucket(key - query)] ;  P = softmax_keys(S) ;  Pd = dropout(P) (:250) ;  O = Pd v
// Backward, per (b, h), with delta[q] = sum_d dO[q][d] O[q][d]  (= sum_k Pd[q][k] dPd[q][k]):
//   dPd = dO v^T ; dP = keep*scale*dPd ; dS = P o (dP - delta) ;
//   dv = Pd^T dO ; dq = dS k / 8 ; dk = dS^T q / 8 ; dbias[h][bucket(key-query)] += dS
// Two kernels, both built like the inference kernel (attention_f32.hip): every product is arranged so that the
// softmax-side operand of the second GEMM is the very register the first GEMM produced.
//   dq kernel : block = (64 queries, h, b), wave owns 16 queries (lane-local j = query), loops over key tiles;
//               S^T[key][q] = K.Q^T and dP^T[key][q] = V.dO^T (A = K / V tile in LDS, B = Q / dO fragments in registers),
//               dQ^T[d][q] += K^T[d][key] . dS^T[key][q]; also delta (written for the dk/dv kernel) and the bias gradient
//               (per-wave LDS tables over key-query, bucketed once per block into the block's own slot: no atomics).
//   dkv kernel: block = (64 keys, h, b), wave owns 16 keys (lane-local j = key), loops over query tiles;
//               S[q][key] = Q.K^T and dP[q][key] = dO.V^T (A = Q / dO tile in LDS, B = K / V fragments in registers),
//               dV^T[d][key] += dO^T[d][q] . Pd[q][key] ; dK^T[d][key] += Q^T[d][q] . dS[q][key].
// P is recomputed from the stashed log-sum-exp (exp(S - lse)); the dropout keep-mask is recomputed from the
// counter-based hash of vn_common.h (row = global (b, h, query), column = key), so nothing T x T is stored.
// Outputs are written token-major [B*T][3D] (dq | dk | dv), the A-operand layout of the w_qkv backward GEMMs.
//
// Algorithmic FLOPs per (b, h): forward 4*T*T*64; backward 10*T*T*64 (5 products; S and dP are computed twice:
// 14*T*T*64 executed).
#include "vn_common.h"
#include "vn_train.h"

#define ATT_KT 64
#define ATT_LD 68

// ---- forward (training) ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void vn_attention_train_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                     const float* __restrict__ v,
                                                                     const float* __restrict__ bias_full,
                                                                     float* __restrict__ out, float* __restrict__ lse,
                                                                     int B, int H, int T, vn_drop d) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = Ks + ATT_KT * ATT_LD;
    float* bt = Vs + ATT_KT * ATT_LD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const size_t headoff = ((size_t)b * H + h) * (size_t)T * VN_DHEAD;
    const float* Q = q + headoff;
    const float* K = k + headoff;
    const float* V = v + headoff;
    const int qrow = qb * 64 + wave * 16 + j;
    const int qrow_c = qrow < T ? qrow : T - 1;
    const int nb = 2 * T - 1;
    for (int i = tid; i < nb; i += 256) bt[i] = bias_full[(size_t)h * nb + i];

    f32x4 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const f32x4*)(Q + (size_t)qrow_c * VN_DHEAD + 16 * s + 4 * g);
    const uint32_t rk = vn_drop_rowkey(d, ((long)b * H + h) * T + qrow_c);

    f32x4 kreg[4], vreg[4];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx >> 4, c4 = idx & 15;
            const int key = kt * ATT_KT + row;
            if (key < T) {
                kreg[i] = *(const f32x4*)(K + (size_t)key * VN_DHEAD + c4 * 4);
                vreg[i] = *(const f32x4*)(V + (size_t)key * VN_DHEAD + c4 * 4);
            } else {
                kreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                vreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto write_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx >> 4, c4 = idx & 15;
            *(f32x4*)(Ks + row * ATT_LD + c4 * 4) = kreg[i];
            *(f32x4*)(Vs + row * ATT_LD + c4 * 4) = vreg[i];
        }
    };

    float m_run = -INFINITY, l_run = 0.f;
    f32x4 o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nkt = (T + ATT_KT - 1) / ATT_KT;
    load_tile(0);
    write_tile();
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) load_tile(kt + 1);
        f32x4 sacc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) sacc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 kf[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) kf[u] = *(const f32x4*)(Ks + (u * 16 + j) * ATT_LD + 16 * s + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    sacc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[u][e], qf[s][e], sacc[u], 0, 0, 0);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * ATT_KT + u * 16 + 4 * g + r;
                const int key_c = key < T ? key : T - 1;
                float x = sacc[u][r] * 0.125f + bt[key_c - qrow_c + (T - 1)];
                x = key < T ? x : -INFINITY;
                sacc[u][r] = x;
                mx = fmaxf(mx, x);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = vn_exp_neg(m_run - m_new);
        float lsum = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int key0 = kt * ATT_KT + u * 16 + 4 * g;
            uint32_t b0 = 0, b1 = 0;
            if (d.thresh16) { b0 = vn_drop_bits(rk, key0); b1 = vn_drop_bits(rk, key0 + 2); }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pexp = vn_exp_neg(sacc[u][r] - m_new);
                lsum += pexp;                                           // the softmax denominator ignores dropout
                sacc[u][r] = d.thresh16 ? pexp * vn_drop_mul(d, r < 2 ? b0 : b1, r) : pexp;
            }
        }
        l_run = l_run * alpha + lsum;
        m_run = m_new;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e][0] *= alpha; o[e][1] *= alpha; o[e][2] *= alpha; o[e][3] *= alpha;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x4 vf = *(const f32x4*)(Vs + (u * 16 + 4 * g + r) * ATT_LD + 4 * j);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[e], sacc[u][r], o[e], 0, 0, 0);
            }
        __syncthreads();
        if (kt + 1 < nkt) {
            write_tile();
            __syncthreads();
        }
    }
    float l_tot = l_run;
    l_tot += __shfl_xor(l_tot, 16);
    l_tot += __shfl_xor(l_tot, 32);
    if (qrow < T) {
        const size_t ooff = ((size_t)b * T + qrow) * ((size_t)H * VN_DHEAD) + h * VN_DHEAD + 16 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f32x4 ov;
            ov[0] = o[0][r] / l_tot; ov[1] = o[1][r] / l_tot; ov[2] = o[2][r] / l_tot; ov[3] = o[3][r] / l_tot;
            *(f32x4*)(out + ooff + 4 * r) = ov;
        }
        if (g == 0) lse[((size_t)b * H + h) * T + qrow] = m_run + logf(l_tot);
    }
}

// ---- backward: dq (+ delta, + bias gradient) ---------------------------------------------------
__global__ __launch_bounds__(256) void vn_attention_bwd_dq_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, const float* __restrict__ bias_full,
    const int32_t* __restrict__ lut, const float* __restrict__ out, const float* __restrict__ dout,
    const float* __restrict__ lse, float* __restrict__ delta, float* __restrict__ dqkv, float* __restrict__ dbias_partial, int B,
    int H, int T, int nbuckets, vn_drop d) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = Ks + ATT_KT * ATT_LD;
    const int nb = 2 * T - 1;
    float* bt = Vs + ATT_KT * ATT_LD;
    float* dbt_all = bt + nb;       // [4 waves][nb]: PER-WAVE d(bias) tables over key - query (plain += : run-to-run bitwise)
    float* bk_all = dbt_all + 4 * nb;   // [4 waves][64]: per-wave sums of the single-bucket ("far") tiles
    float* wscr_all = bk_all + 256; // [4 waves][16 queries][80]: per-wave dS scratch for the diagonal sums

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int Dm = H * VN_DHEAD;
    const size_t headoff = ((size_t)b * H + h) * (size_t)T * VN_DHEAD;
    const float* Q = q + headoff;
    const float* K = k + headoff;
    const float* V = v + headoff;
    const int qrow = qb * 64 + wave * 16 + j;
    const int qrow_c = qrow < T ? qrow : T - 1;
    for (int i = tid; i < nb; i += 256) bt[i] = bias_full[(size_t)h * nb + i];
    for (int i = tid; i < 4 * nb + 256; i += 256) dbt_all[i] = 0.f;      // the four tables and, contiguous, the far sums
    for (int i = tid; i < 4 * 16 * 80; i += 256) wscr_all[i] = 0.f;      // slots outside the tile's diagonals stay 0
    float* wscr = wscr_all + wave * (16 * 80);
    float* dbt = dbt_all + wave * nb;
    float* bk = bk_all + wave * 64;

    f32x4 qf[4], dof[4];
    float dl = 0.f;
    {
        const size_t tok = ((size_t)b * T + qrow_c) * Dm + h * VN_DHEAD;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qf[s] = *(const f32x4*)(Q + (size_t)qrow_c * VN_DHEAD + 16 * s + 4 * g);
            dof[s] = *(const f32x4*)(dout + tok + 16 * s + 4 * g);
            const f32x4 of = *(const f32x4*)(out + tok + 16 * s + 4 * g);
            dl += dof[s][0] * of[0] + dof[s][1] * of[1] + dof[s][2] * of[2] + dof[s][3] * of[3];
        }
    }
    dl += __shfl_xor(dl, 16);
    dl += __shfl_xor(dl, 32);
    const float my_lse = lse[((size_t)b * H + h) * T + qrow_c];
    if (g == 0 && qrow < T) delta[((size_t)b * H + h) * T + qrow] = dl;
    const uint32_t rk = vn_drop_rowkey(d, ((long)b * H + h) * T + qrow_c);

    f32x4 kreg[4], vreg[4];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx >> 4, c4 = idx & 15;
            const int key = kt * ATT_KT + row;
            if (key < T) {
                kreg[i] = *(const f32x4*)(K + (size_t)key * VN_DHEAD + c4 * 4);
                vreg[i] = *(const f32x4*)(V + (size_t)key * VN_DHEAD + c4 * 4);
            } else {
                kreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                vreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto write_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx >> 4, c4 = idx & 15;
            *(f32x4*)(Ks + row * ATT_LD + c4 * 4) = kreg[i];
            *(f32x4*)(Vs + row * ATT_LD + c4 * 4) = vreg[i];
        }
    };

    f32x4 o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nkt = (T + ATT_KT - 1) / ATT_KT;
    load_tile(0);
    write_tile();
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) load_tile(kt + 1);
        f32x4 sacc[4], dpacc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { sacc[u] = f32x4{0.f, 0.f, 0.f, 0.f}; dpacc[u] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 kf[4], vfa[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                kf[u] = *(const f32x4*)(Ks + (u * 16 + j) * ATT_LD + 16 * s + 4 * g);
                vfa[u] = *(const f32x4*)(Vs + (u * 16 + j) * ATT_LD + 16 * s + 4 * g);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    sacc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[u][e], qf[s][e], sacc[u], 0, 0, 0);
                    dpacc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(vfa[u][e], dof[s][e], dpacc[u], 0, 0, 0);
                }
        }
        // dS^T for this lane's query and keys kt*64 + 16u + 4g + r.
        // Bias gradient, without any floating-point atomic (deterministic): a tile whose whole key-query range falls into ONE
        // bucket (every tile 3+ tiles off the diagonal at max_distance 128) is summed in registers into the wave's far-sum
        // slot; near-diagonal tiles add their 79 diagonal sums to the WAVE's own key-query table.
        const int rel_lo = kt * ATT_KT - (qb * 64 + 63), rel_hi = kt * ATT_KT + 63 - qb * 64;
        const int lo_c = rel_lo < -(T - 1) ? -(T - 1) : rel_lo, hi_c = rel_hi > T - 1 ? T - 1 : rel_hi;
        const int far_bucket = lut[lo_c + T - 1];
        const bool far = (rel_lo > 0 || rel_hi < 0) && far_bucket == lut[hi_c + T - 1];     // block-uniform
        float far_sum = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int key0 = kt * ATT_KT + u * 16 + 4 * g;
            uint32_t b0 = 0, b1 = 0;
            if (d.thresh16) { b0 = vn_drop_bits(rk, key0); b1 = vn_drop_bits(rk, key0 + 2); }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = key0 + r;
                const int key_c = key < T ? key : T - 1;
                const float x = sacc[u][r] * 0.125f + bt[key_c - qrow_c + (T - 1)];
                const float p = (key < T && qrow < T) ? vn_exp_neg(x - my_lse) : 0.f;
                const float mul = d.thresh16 ? vn_drop_mul(d, r < 2 ? b0 : b1, r) : 1.0f;
                const float ds = p * (dpacc[u][r] * mul - dl);
                sacc[u][r] = ds;
                far_sum += ds;
            }
        }
        if (far) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) far_sum += __shfl_xor(far_sum, o);
            if (lane == 0) bk[far_bucket] += far_sum;
        } else {
            // near-diagonal tile: the wave's 64 keys x 16 queries of dS go through a private LDS scratch laid out
            // [query j][diagonal i - j + 15] (row stride 80: conflict-free both ways), each lane then sums one (two)
            // of the 79 diagonals with plain reads and makes ONE table update for it (16 LDS float atomics per lane cost
            // ~175 cycles per wave instruction on gfx950, measured, and are order-dependent).
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) wscr[j * 80 + (u * 16 + 4 * g + r) - j + 15] = sacc[u][r];
            const int rel0 = kt * ATT_KT - (qb * 64 + wave * 16) - 15 + (T - 1);     // table index of diagonal 0
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int dg = lane + 64 * pass;
                if (dg < 79) {
                    float a = 0.f;
#pragma unroll
                    for (int jj = 0; jj < 16; ++jj) a += wscr[jj * 80 + dg];
                    const int idx = rel0 + dg;
                    if (idx >= 0 && idx < nb) dbt[idx] += a;       // lanes of a wave own distinct diagonals
                }
            }
        }
        // dQ^T[d][q] += K^T[d][key] . dS^T[key][q]
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x4 kf2 = *(const f32x4*)(Ks + (u * 16 + 4 * g + r) * ATT_LD + 4 * j);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf2[e], sacc[u][r], o[e], 0, 0, 0);
            }
        __syncthreads();
        if (kt + 1 < nkt) {
            write_tile();
            __syncthreads();
        }
    }
    if (qrow < T) {
        const size_t ooff = ((size_t)b * T + qrow) * (size_t)(3 * Dm) + h * VN_DHEAD + 16 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f32x4 ov;
            ov[0] = o[0][r] * 0.125f; ov[1] = o[1][r] * 0.125f; ov[2] = o[2][r] * 0.125f; ov[3] = o[3][r] * 0.125f;
            *(f32x4*)(dqkv + ooff + 4 * r) = ov;
        }
    }
    // bias gradient of this block, in a fixed order: (1) the four per-wave tables are summed 0+1+2+3; (2) every bucket is one
    // contiguous run of key - query offsets (relative_position_bucket is monotone on each side of 0), found from the LUT;
    // (3) eight threads per bucket sum strided slices of its run, combined 0..7, plus the waves' far sums; (4) the block's
    // nbuckets values go to ITS slot of dbias_partial (vn_dbias_reduce_kernel adds the slots up in order).
    if (dbias_partial == nullptr) return;
    __syncthreads();
    int* run_lo = (int*)Ks;          // K / V tiles are dead: reuse
    int* run_hi = run_lo + 64;
    float* red = (float*)(run_hi + 64);       // [64 buckets][8]
    if (tid < 64) { run_lo[tid] = 0; run_hi[tid] = 0; }
    for (int i = tid; i < nb; i += 256) dbt_all[i] = ((dbt_all[i] + dbt_all[nb + i]) + dbt_all[2 * nb + i]) + dbt_all[3 * nb + i];
    __syncthreads();
    for (int i = tid; i < nb; i += 256) {
        const int bkt = lut[i];
        if (i == 0 || lut[i - 1] != bkt) run_lo[bkt] = i;
        if (i == nb - 1 || lut[i + 1] != bkt) run_hi[bkt] = i + 1;
    }
    __syncthreads();
    for (int bkt = tid >> 3; bkt < nbuckets; bkt += 32) {
        float a = 0.f;
        for (int i = run_lo[bkt] + (tid & 7); i < run_hi[bkt]; i += 8) a += dbt_all[i];
        red[bkt * 8 + (tid & 7)] = a;
    }
    __syncthreads();
    if (tid < nbuckets) {
        float a = 0.f;
#pragma unroll
        for (int pth = 0; pth < 8; ++pth) a += red[tid * 8 + pth];
        a += ((bk_all[tid] + bk_all[64 + tid]) + bk_all[128 + tid]) + bk_all[192 + tid];
        const size_t slot = ((size_t)b * H + h) * gridDim.x + qb;
        dbias_partial[slot * 64 + tid] = a;
    }
}

// dbias[bucket][h] (+)= sum over the slots (b, h, q-block) of n_slabs consecutive slabs (layers).  One wave per output: lane i adds the
// terms i, i + 64, ... (term index = (slab, b, q-block) in that order), then the 64 lane sums meet in a fixed xor tree — deterministic
// (run to run and for any launch geometry), and ~50x shorter than the one-thread-per-output walk of 1440 strided terms it replaces
// (0.69 ms of a 76 ms step at B = 8)
__global__ __launch_bounds__(64) void vn_dbias_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dbias, int n_slabs,
                                                            long slab_floats, int B, int H, int nqb, int nbuckets, int accumulate) {
    const int i = blockIdx.x;
    if (i >= nbuckets * H) return;
    const int bkt = i / H, h = i - bkt * H;
    const int per_slab = B * nqb, total = n_slabs * per_slab;
    float a = 0.f;
    for (int n = threadIdx.x; n < total; n += 64) {
        const int l = n / per_slab, r = n - l * per_slab;
        const int b = r / nqb, qb = r - b * nqb;
        a += partial[l * slab_floats + (((size_t)b * H + h) * nqb + qb) * 64 + bkt];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (threadIdx.x == 0) dbias[i] = accumulate ? dbias[i] + a : a;
}

int vn_launch_dbias_reduce(vn_ctx* ctx, const float* partial, float* dbias, int n_slabs, long slab_floats, int B, int H, int nqb,
                           int nbuckets, bool accumulate, hipStream_t s) {
    hipLaunchKernelGGL(vn_dbias_reduce_kernel, dim3(nbuckets * H), dim3(64), 0, s, partial, dbias, n_slabs, slab_floats, B, H,
                       nqb, nbuckets, accumulate ? 1 : 0);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

// ---- backward: dk, dv -------------------------------------------------------------------------
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void vn_attention_bwd_dkv_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, const float* __restrict__ bias_full,
    const float* __restrict__ dout, const float* __restrict__ lse, const float* __restrict__ delta, float* __restrict__ dqkv,
    int B, int H, int T, vn_drop d) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem;
    float* dOs = Qs + ATT_KT * ATT_LD;
    float* lse_s = dOs + ATT_KT * ATT_LD;
    float* del_s = lse_s + 64;
    uint32_t* rk_s = (uint32_t*)(del_s + 64);
    float* bt = del_s + 128;
    const int nb = 2 * T - 1;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int kb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int Dm = H * VN_DHEAD;
    const size_t headoff = ((size_t)b * H + h) * (size_t)T * VN_DHEAD;
    const float* Q = q + headoff;
    const float* K = k + headoff;
    const float* V = v + headoff;
    const int krow = kb * 64 + wave * 16 + j;
    const int krow_c = krow < T ? krow : T - 1;
    for (int i = tid; i < nb; i += 256) bt[i] = bias_full[(size_t)h * nb + i];

    f32x4 kf[4], vf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        kf[s] = *(const f32x4*)(K + (size_t)krow_c * VN_DHEAD + 16 * s + 4 * g);
        vf[s] = *(const f32x4*)(V + (size_t)krow_c * VN_DHEAD + 16 * s + 4 * g);
    }

    f32x4 qreg[4], doreg[4];
    float lreg = 0.f, dreg = 0.f;
    uint32_t rreg = 0;
    auto load_tile = [&](int qt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx >> 4, c4 = idx & 15;
            const int qq = qt * ATT_KT + row;
            if (qq < T) {
                qreg[i] = *(const f32x4*)(Q + (size_t)qq * VN_DHEAD + c4 * 4);
                doreg[i] = *(const f32x4*)(dout + ((size_t)b * T + qq) * Dm + h * VN_DHEAD + c4 * 4);
            } else {
                qreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                doreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        if (tid < 64) {
            const int qq = qt * ATT_KT + tid;
            const int qc = qq < T ? qq : T - 1;
            lreg = lse[((size_t)b * H + h) * T + qc];
            dreg = delta[((size_t)b * H + h) * T + qc];
            rreg = vn_drop_rowkey(d, ((long)b * H + h) * T + qc);
        }
    };
    auto write_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx >> 4, c4 = idx & 15;
            *(f32x4*)(Qs + row * ATT_LD + c4 * 4) = qreg[i];
            *(f32x4*)(dOs + row * ATT_LD + c4 * 4) = doreg[i];
        }
        if (tid < 64) { lse_s[tid] = lreg; del_s[tid] = dreg; rk_s[tid] = rreg; }
    };

    f32x4 accK[4], accV[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { accK[e] = f32x4{0.f, 0.f, 0.f, 0.f}; accV[e] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int nqt = (T + ATT_KT - 1) / ATT_KT;
    load_tile(0);
    write_tile();
    __syncthreads();

    for (int qt = 0; qt < nqt; ++qt) {
        if (qt + 1 < nqt) load_tile(qt + 1);
        f32x4 sacc[4], dpacc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { sacc[u] = f32x4{0.f, 0.f, 0.f, 0.f}; dpacc[u] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 qa[4], da[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                qa[u] = *(const f32x4*)(Qs + (u * 16 + j) * ATT_LD + 16 * s + 4 * g);
                da[u] = *(const f32x4*)(dOs + (u * 16 + j) * ATT_LD + 16 * s + 4 * g);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    sacc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[u][e], kf[s][e], sacc[u], 0, 0, 0);
                    dpacc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(da[u][e], vf[s][e], dpacc[u], 0, 0, 0);
                }
        }
        // lane (j = key, g) holds S / dP for queries qt*64 + 16u + 4g + r
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ql = u * 16 + 4 * g + r;
                const int qq = qt * ATT_KT + ql;
                const int qc = qq < T ? qq : T - 1;
                const float x = sacc[u][r] * 0.125f + bt[krow_c - qc + (T - 1)];
                const float p = (qq < T && krow < T) ? vn_exp_neg(x - lse_s[ql]) : 0.f;
                const float mul = d.thresh16 ? vn_drop_mul(d, vn_drop_bits(rk_s[ql], krow_c), krow_c) : 1.0f;
                sacc[u][r] = p * mul;                                   // Pd
                dpacc[u][r] = p * (dpacc[u][r] * mul - del_s[ql]);      // dS
            }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x4 d4 = *(const f32x4*)(dOs + (u * 16 + 4 * g + r) * ATT_LD + 4 * j);
                const f32x4 q4 = *(const f32x4*)(Qs + (u * 16 + 4 * g + r) * ATT_LD + 4 * j);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    accV[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(d4[e], sacc[u][r], accV[e], 0, 0, 0);
                    accK[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(q4[e], dpacc[u][r], accK[e], 0, 0, 0);
                }
            }
        __syncthreads();
        if (qt + 1 < nqt) {
            write_tile();
            __syncthreads();
        }
    }
    if (krow < T) {
        const size_t ooff = ((size_t)b * T + krow) * (size_t)(3 * Dm) + h * VN_DHEAD + 16 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f32x4 kv, vv;
#pragma unroll
            for (int e = 0; e < 4; ++e) { kv[e] = accK[e][r] * 0.125f; vv[e] = accV[e][r]; }
            *(f32x4*)(dqkv + ooff + Dm + 4 * r) = kv;
            *(f32x4*)(dqkv + ooff + 2 * Dm + 4 * r) = vv;
        }
    }
}

// ---- launchers ---------------------------------------------------------------------------------
static int att_train_attrs(vn_ctx* ctx) {
    if (ctx->attr_mask & VN_ATTR_ATTN_TRAIN) return VN_OK;
    VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)vn_attention_train_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)vn_attention_bwd_dq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)vn_attention_bwd_dkv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ctx->attr_mask |= VN_ATTR_ATTN_TRAIN;
    return VN_OK;
}

int vn_launch_attention_train_fwd(vn_ctx* ctx, const float* q, const float* k, const float* v, const float* relbias_full,
                                  float* out, float* lse, int B, int H, int T, const vn_drop& d, hipStream_t s) {
    if (B <= 0 || T <= 0) return VN_OK;
    const size_t lds = (size_t)(2 * ATT_KT * ATT_LD + 2 * T - 1 + 3) * sizeof(float);
    if (lds > 160 * 1024) return vn_fail(ctx, VN_ERR_INVALID, "attention: T=%s%ld too long for the LDS bias table", "", T);
    int rc = att_train_attrs(ctx);
    if (rc) return rc;
    const int pi = vn_prof_pre(ctx, 1, 4.0 * T * (double)T * VN_DHEAD * H * B, s, 16.0 * T * VN_DHEAD * (double)H * B);
    hipLaunchKernelGGL(vn_attention_train_fwd_kernel, dim3(vn_cdiv(T, 64), H, B), dim3(256), lds, s, q, k, v, relbias_full, out,
                       lse, B, H, T, d);
    vn_prof_post(ctx, pi, s);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

int vn_launch_attention_bwd(vn_ctx* ctx, const float* q, const float* k, const float* v, const float* relbias_full,
                            const int32_t* lut_dev, const float* out, const float* dout, const float* lse, float* delta,
                            float* dqkv, float* dbias_partial, int B, int H, int T, int nbuckets, const vn_drop& d, hipStream_t s) {
    if (B <= 0 || T <= 0) return VN_OK;
    if (nbuckets > 64) return vn_fail(ctx, VN_ERR_UNSUPPORTED, "attention backward: num_buckets=%s%ld > 64", "", nbuckets);
    const int nb = 2 * T - 1;
    const size_t lds_dq = (size_t)(2 * ATT_KT * ATT_LD + 5 * nb + 256 + 4 * 16 * 80 + 4) * sizeof(float);
    const size_t lds_kv = (size_t)(2 * ATT_KT * ATT_LD + 192 + nb + 4) * sizeof(float);
    if (lds_dq > 160 * 1024) return vn_fail(ctx, VN_ERR_INVALID, "attention backward: T=%s%ld too long", "", T);
    int rc = att_train_attrs(ctx);
    if (rc) return rc;
    const dim3 grid(vn_cdiv(T, 64), H, B);
    const double fl = 2.0 * T * (double)T * VN_DHEAD * H * B;     // one T x T x 64 product
    int pi = vn_prof_pre(ctx, 1, 3.0 * fl, s, 24.0 * T * VN_DHEAD * (double)H * B);
    hipLaunchKernelGGL(vn_attention_bwd_dq_kernel, grid, dim3(256), lds_dq, s, q, k, v, relbias_full, lut_dev, out, dout, lse,
                       delta, dqkv, dbias_partial, B, H, T, nbuckets, d);
    vn_prof_post(ctx, pi, s);
    pi = vn_prof_pre(ctx, 1, 4.0 * fl, s, 28.0 * T * VN_DHEAD * (double)H * B);
    hipLaunchKernelGGL(vn_attention_bwd_dkv_kernel, grid, dim3(256), lds_kv, s, q, k, v, relbias_full, dout, lse, delta, dqkv, B, H,
                       T, d);
    vn_prof_post(ctx, pi, s);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}
