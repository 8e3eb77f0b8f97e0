// Fused fp32 self-attention with the shared T5 relative-position bias, for gfx950.
//
// Replaces MultiHeadRelativeAttention.forward's core (vampnet/modules/transformer.py:234-254):
//   attn = einsum(q,k)/sqrt(64) + bias[h, bucket(k - q)] ; softmax(dim=keys) ; einsum(attn, v)
// (mask is all-ones at inference, transformer.py:619; dropout is eval-mode identity.)
//
// The (H,B,T,T) score tensor (26.5 MB per layer per item in the reference) is never materialised:
// flash-style online softmax, exact fp32 throughout (v_mfma_f32_16x16x4_f32 = bitwise fmaf chains).
//
// Work decomposition: grid = (ceil(T/64) q-blocks, H, B); block = 4 waves; wave w owns 16 query rows.
// Both products are computed TRANSPOSED so that every softmax quantity is lane-local
// (cdna_hip_programming.md App. B "swapped QK^T"):
//   S^T[key][q]  = K[key][:] . Q[q][:]    A = K tile (LDS),  B = Q (registers)   -> lane (j=lane&15, g=lane>>4)
//                                          holds S for its query j and keys 16u + 4g + r  (u,r = 0..3)
//   O^T[d][q]    = V^T[d][key] . P^T[key][q]   A = V tile (LDS, read as float4 over d), B = P (the very
//                                          registers the softmax produced) -> no data movement between the GEMMs.
// Row max needs two xor-shuffles (lanes j, j+16, j+32, j+48 share a query); row sum is reduced once at the end.
// The bias is a 2T-1 entry per-head table in LDS indexed by key - query (see vn_bias_expand_kernel).
// K/V tiles (64 keys x 64 d) are register-prefetched one tile ahead (issue-early / write-late, guide T14).
// LDS rows are padded to 68 floats so the float4 fragment reads are (almost) bank-conflict free.
//
// Algorithmic FLOPs: 4*T*T*64 per (b,h)  (2 GEMMs); HBM bytes: Q,K,V read + O written = 4*T*64*4 per (b,h)
// (K/V are re-read by the ceil(T/64) q-blocks of a head, from L2).
#include <stdlib.h>
#include "vn_common.h"

#define ATT_KT 64        // keys per tile
#define ATT_LD 68        // padded LDS row (floats)

template <int VARIANT>
__global__ __launch_bounds__(256) void vn_attention_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v,
                                                           const float* __restrict__ bias_full,  // [H][2T-1]
                                                           float* __restrict__ out, uint16_t* __restrict__ out16, long plane16,
                                                           int B, int H, int T, unsigned* sat) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = Ks + ATT_KT * ATT_LD;
    float* bt = Vs + ATT_KT * ATT_LD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    // 1-D XCD-aware grid: every XCD (blockIdx % 8) walks a contiguous eighth of the (b, h, q-block) list, so the q-blocks of a
    // head share its K / V in ONE L2 instead of fetching them on all eight
    int lid;
    {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int qq = nwg >> 3, rr = nwg & 7;
        lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
    const int nqb = (T + 63) / 64;
    const int qb = lid % nqb, h = (lid / nqb) % H, b = lid / (nqb * H);
    const size_t headoff = ((size_t)b * H + h) * (size_t)T * VN_DHEAD;
    const float* Q = q + headoff;
    const float* K = k + headoff;
    const float* V = v + headoff;

    const int qrow = qb * 64 + wave * 16 + j;
    const int qrow_c = qrow < T ? qrow : T - 1;

    // bias table for this head -> LDS
    const int nb = 2 * T - 1;
    for (int i = tid; i < nb; i += 256) bt[i] = bias_full[(size_t)h * nb + i];

    // Q fragment: lane (j,g) holds Q[q_j][16s + 4g + e]
    f32x4 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const f32x4*)(Q + (size_t)qrow_c * VN_DHEAD + 16 * s + 4 * g);

    // tile staging: thread handles float4 idx = tid + 256*i  (row = idx>>4, c4 = idx&15)
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

        // ---- S^T = K . Q^T  (4 key sub-tiles u, contraction over d in 16 steps of 4)
        f32x4 sacc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) sacc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_setprio(1);
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

        __builtin_amdgcn_s_setprio(0);
        // ---- online softmax over this tile's 64 keys; lane holds keys kt*64 + 16u + 4g + r for query j
        float mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * ATT_KT + u * 16 + 4 * g + r;
                const int key_c = key < T ? key : T - 1;
                float x = sacc[u][r] * 0.125f + bt[key_c - qrow_c + (T - 1)];   // /sqrt(64) exact; += bias
                x = key < T ? x : -INFINITY;
                sacc[u][r] = x;
                mx = fmaxf(mx, x);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);          // finite: every tile has >= 1 valid key
        const float alpha = (VARIANT & 1) ? vn_exp_neg(m_run - m_new) : expf(m_run - m_new);   // first tile: exp(-inf) = 0
        float lsum = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pexp = (VARIANT & 1) ? vn_exp_neg(sacc[u][r] - m_new) : expf(sacc[u][r] - m_new);
                sacc[u][r] = pexp;
                lsum += pexp;
            }
        l_run = l_run * alpha + lsum;
        m_run = m_new;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e][0] *= alpha; o[e][1] *= alpha; o[e][2] *= alpha; o[e][3] *= alpha;
        }

        // ---- O^T += V^T . P^T : lane (j,g) reads V[key(u,g,r)][4j .. 4j+3]; tile e covers d = 4i + e
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x4 vf = *(const f32x4*)(Vs + (u * 16 + 4 * g + r) * ATT_LD + 4 * j);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[e], sacc[u][r], o[e], 0, 0, 0);
            }

        __builtin_amdgcn_s_setprio(0);
        __syncthreads();                       // every wave finished reading Ks/Vs
        if (kt + 1 < nkt) {
            write_tile();
            __syncthreads();
        }
    }

    // ---- finish: row sum across the 4 lanes of a query, normalise, store.
    // accumulator o[e][r] = O[q_j][d = 16g + 4r + e]  (C/D map of 16x16 MFMA: row i = 4*(lane>>4) + r, d = 4i + e)
    float l_tot = l_run;
    l_tot += __shfl_xor(l_tot, 16);
    l_tot += __shfl_xor(l_tot, 32);
    if (qrow < T) {
        const size_t ooff = ((size_t)b * T + qrow) * ((size_t)H * VN_DHEAD) + h * VN_DHEAD + 16 * g;
        bool bad = false;                      // fp16 planes: saturation ledger (vn_common.h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f32x4 ov;
            ov[0] = o[0][r] / l_tot;
            ov[1] = o[1][r] / l_tot;
            ov[2] = o[2][r] / l_tot;
            ov[3] = o[3][r] / l_tot;
            if (out16) {   // bf16 / bf16x3 modes: the attention output is only the A operand of the fc GEMM
                vn_store_planes4(out16, plane16, (long)b * T + qrow, h * VN_DHEAD + 16 * g + 4 * r, H * VN_DHEAD, ov, bad);
            } else {
                *(f32x4*)(out + ooff + 4 * r) = ov;
            }
        }
        vn_sat_report(sat, VN_SAT_OPERAND, bad);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16 FAST MODE attention (vn_model_set_bf16; not bit-exact — the reference's own GPU path runs this op under
// torch.autocast(bfloat16), interface.py:364,428).  Same decomposition and transposed-product trick as the fp32 kernel,
// on v_mfma_f32_16x16x32_bf16: K is staged as a bf16 [key][d] tile, V as a TRANSPOSED bf16 [d][key] tile (so both MFMA A
// operands are 16-byte LDS reads), Q (pre-scaled by 1/8, exact) and the probabilities P are bf16 B operands held in
// registers; scores, softmax state and the output accumulators stay fp32.  16 MFMAs per 64-key tile per wave (256 matrix
// cycles) against ~1 k VALU cycles of softmax: the kernel is VALU-bound, exp is a bare v_exp_f32 (bf16 keeps 8 bits).
// MFMA k-slot mapping of the P.V product: lane group kk carries keys {16u + 4kk + r} and {16(u+1) + 4kk + r}, r = 0..3 —
// exactly the keys whose probabilities the lane computed — the V^T operand reads the same key sets.
// ---------------------------------------------------------------------------------------------------------------------
#define ATB_LDK 72        // bf16 per padded row of the K / V^T tiles (64 + 8: 144-byte rows, conflict-free 16-byte reads)

__device__ __forceinline__ unsigned vn_pack_bf16x2(float a, float b) {
    return (unsigned)vn_f32_to_bf16(a) | ((unsigned)vn_f32_to_bf16(b) << 16);
}

__global__ __launch_bounds__(256) void vn_attention_bf16_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                const float* __restrict__ v,
                                                                const float* __restrict__ bias_full,
                                                                uint16_t* __restrict__ out16, int B, int H, int T) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint16_t* Ks = (uint16_t*)smem;                       // [64 keys][ATB_LDK]
    uint16_t* Vt = Ks + ATT_KT * ATB_LDK;                 // [64 d][ATB_LDK] (key along the row)
    float* bt = (float*)(Vt + VN_DHEAD * ATB_LDK);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, g = lane >> 4;
    // 1-D XCD-aware grid: every XCD (blockIdx % 8) walks a contiguous eighth of the (b, h, q-block) list, so the q-blocks of a
    // head share its K / V in ONE L2 instead of fetching them on all eight
    int lid;
    {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int qq = nwg >> 3, rr = nwg & 7;
        lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
    const int nqb = (T + 63) / 64;
    const int qb = lid % nqb, h = (lid / nqb) % H, b = lid / (nqb * H);
    const size_t headoff = ((size_t)b * H + h) * (size_t)T * VN_DHEAD;
    const float* Q = q + headoff;
    const float* K = k + headoff;
    const float* V = v + headoff;
    const int qrow = qb * 64 + wave * 16 + j;
    const int qrow_c = qrow < T ? qrow : T - 1;
    const int nb = 2 * T - 1;
    for (int i = tid; i < nb; i += 256) bt[i] = bias_full[(size_t)h * nb + i];

    // Q fragments (B operand of S^T = K Q^T): lane (j, g) holds Q[q_j][32s + 8g .. +7] * 1/8 as bf16x8
    bf16x8 qf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const f32x4 a = *(const f32x4*)(Q + (size_t)qrow_c * VN_DHEAD + 32 * s + 8 * g);
        const f32x4 c = *(const f32x4*)(Q + (size_t)qrow_c * VN_DHEAD + 32 * s + 8 * g + 4);
        u32x4 pk = {vn_pack_bf16x2(a[0] * 0.125f, a[1] * 0.125f), vn_pack_bf16x2(a[2] * 0.125f, a[3] * 0.125f),
                    vn_pack_bf16x2(c[0] * 0.125f, c[1] * 0.125f), vn_pack_bf16x2(c[2] * 0.125f, c[3] * 0.125f)};
        qf[s] = __builtin_bit_cast(bf16x8, pk);
    }

    // staging: thread handles float4 idx = tid + 256*i (row = idx>>4 = key, c4 = idx&15 -> d = 4*c4 .. +3)
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
            uint2 pk = {vn_pack_bf16x2(kreg[i][0], kreg[i][1]), vn_pack_bf16x2(kreg[i][2], kreg[i][3])};
            *(uint2*)(Ks + row * ATB_LDK + c4 * 4) = pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) Vt[(c4 * 4 + e) * ATB_LDK + row] = vn_f32_to_bf16(vreg[i][e]);
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
    const float L2E = 1.4426950408889634f;

    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) load_tile(kt + 1);
        // ---- S^T[key][q]: 4 key sub-tiles x 2 k-steps of 32
        f32x4 sacc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) sacc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bf16x8 kf = *(const bf16x8*)(Ks + (u * 16 + j) * ATB_LDK + 32 * s + 8 * g);
                sacc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[s], sacc[u], 0, 0, 0);
            }
        // ---- online softmax (fp32 state); lane holds keys kt*64 + 16u + 4g + r for query j
        float mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * ATT_KT + u * 16 + 4 * g + r;
                const int key_c = key < T ? key : T - 1;
                float x = sacc[u][r] + bt[key_c - qrow_c + (T - 1)];
                x = key < T ? x : -INFINITY;
                sacc[u][r] = x;
                mx = fmaxf(mx, x);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(fmaxf(m_run - m_new, -126.0f) * L2E);
        const float mb = m_new * L2E;
        float lsum = 0.f;
        bf16x8 pf[2];
#pragma unroll
        for (int up = 0; up < 2; ++up) {
            float pv[8];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pexp = __builtin_amdgcn_exp2f(fmaxf(fmaf(sacc[2 * up + hh][r], L2E, -mb), -126.0f));
                    lsum += pexp;
                    pv[4 * hh + r] = pexp;
                }
            u32x4 pk = {vn_pack_bf16x2(pv[0], pv[1]), vn_pack_bf16x2(pv[2], pv[3]), vn_pack_bf16x2(pv[4], pv[5]),
                        vn_pack_bf16x2(pv[6], pv[7])};
            pf[up] = __builtin_bit_cast(bf16x8, pk);
        }
        l_run = l_run * alpha + lsum;
        m_run = m_new;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e][0] *= alpha; o[e][1] *= alpha; o[e][2] *= alpha; o[e][3] *= alpha;
        }
        // ---- O^T[d][q] += V^T[d][keys] . P^T[keys][q]; lane (i = j, kk = g) reads V^T[16e + j][{16u+4g+r}, {16(u+1)+4g+r}]
#pragma unroll
        for (int up = 0; up < 2; ++up)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint2 lo = *(const uint2*)(Vt + (16 * e + j) * ATB_LDK + 32 * up + 4 * g);
                const uint2 hi = *(const uint2*)(Vt + (16 * e + j) * ATB_LDK + 32 * up + 16 + 4 * g);
                u32x4 pk = {lo.x, lo.y, hi.x, hi.y};
                o[e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, pk), pf[up], o[e], 0, 0, 0);
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
        // accumulator o[e][r] = O[q_j][d = 16e + 4g + r]
        const float inv = 1.0f / l_tot;
        const size_t ooff = ((size_t)b * T + qrow) * ((size_t)H * VN_DHEAD) + h * VN_DHEAD + 4 * g;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint2 pk = {vn_pack_bf16x2(o[e][0] * inv, o[e][1] * inv), vn_pack_bf16x2(o[e][2] * inv, o[e][3] * inv)};
            *(uint2*)(out16 + ooff + 16 * e) = pk;
        }
    }
}

int vn_launch_attention(vn_ctx* ctx, const float* q, const float* k, const float* v, const float* relbias_full,
                        float* out, int B, int H, int T, hipStream_t s, uint16_t* out16, long plane16) {
    if (B <= 0 || T <= 0) return VN_OK;
    const size_t lds = (size_t)(2 * ATT_KT * ATT_LD + 2 * T - 1 + 3) * sizeof(float);
    if (lds > 160 * 1024) return vn_fail(ctx, VN_ERR_INVALID, "attention: T=%s%ld too long for the LDS bias table", "", T);
    // 1: split-product exp2 (default); 0: ocml expf (A/B reference, VN_ATTN_VARIANT=0)
    static const int variant = [] { const char* e = getenv("VN_ATTN_VARIANT"); return e ? atoi(e) & 1 : 1; }();
    if (!(ctx->attr_mask & VN_ATTR_ATTN)) {
        VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)vn_attention_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)vn_attention_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)vn_attention_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ctx->attr_mask |= VN_ATTR_ATTN;
    }
    const int pi = vn_prof_pre(ctx, 1, 4.0 * T * (double)T * VN_DHEAD * H * B, s, 16.0 * T * VN_DHEAD * (double)H * B);
    const dim3 grid(vn_cdiv(T, 64) * H * B);
    static const bool bf16_attn = [] { const char* e = getenv("VN_ATTN_BF16"); return !(e && e[0] == '0'); }();
    if (out16 && bf16_attn && plane16 == 0) {        // fast mode: bf16 MFMA attention (VN_ATTN_BF16=0 keeps the fp32 kernel for A/B runs)
        const size_t lds16 = (size_t)(2 * ATT_KT * ATB_LDK) * sizeof(uint16_t) + (size_t)(2 * T - 1 + 3) * sizeof(float);
        hipLaunchKernelGGL(vn_attention_bf16_kernel, grid, dim3(256), lds16, s, q, k, v, relbias_full, out16, B, H, T);
        vn_prof_post(ctx, pi, s);
        VN_LAUNCH_CHECK(ctx);
        return VN_OK;
    }
    if (variant == 0) hipLaunchKernelGGL(vn_attention_kernel<0>, grid, dim3(256), lds, s, q, k, v, relbias_full, out, out16, plane16, B, H, T, ctx->sat);
    else hipLaunchKernelGGL(vn_attention_kernel<1>, grid, dim3(256), lds, s, q, k, v, relbias_full, out, out16, plane16, B, H, T, ctx->sat);
    vn_prof_post(ctx, pi, s);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}
