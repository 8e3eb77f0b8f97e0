// Device helpers of the split-plane attention kernels (attention_x3.hip: inference; attention_train_x3.hip: training forward with
// dropout + log-sum-exp, and the flash-style backward): plane-product order, exp, K / V^T fragment reads, online softmax, LDS-DMA.
// The layout conventions are described at the top of attention_x3.hip.
#pragma once
#include <type_traits>
#include "vn_common.h"

#define AX_KT 32                          // keys per tile
#define AX_PLANE_FLOATS 1024              // one plane tile of K (32 x 128 B) or V^T (64 x 64 B): 4 KiB
// NP = planes per operand: 3 = bf16x3 (exact three-way bf16 split, six products), 2 = the f16x2 precision's attention format (two fp16
// planes, second one unscaled, three products into the same accumulator; P and V carry a factor 16 each — vn_common.h vn_split2u).
// A stage = NP K plane tiles then NP V^T plane tiles.
#define AX_STAGE_FLOATS_NP(NP) (2 * (NP) * AX_PLANE_FLOATS)
template <int NP>
__device__ __forceinline__ f32x16 ax_mfma(const f32x4& a, const f32x4& b, const f32x16& c) {
    if constexpr (NP == 3) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// the kept plane products (A-operand plane, B-operand plane), smallest terms first
template <int NP> __device__ __forceinline__ constexpr int ax_nprod() { return NP == 3 ? 6 : 3; }
template <int NP> __device__ __forceinline__ constexpr int ax_pa(int t) { return NP == 3 ? (t == 1 ? 2 : (t == 2 || t == 4) ? 1 : 0) : (t == 1 ? 1 : 0); }
template <int NP> __device__ __forceinline__ constexpr int ax_pb(int t) { return NP == 3 ? (t == 0 ? 2 : (t == 2 || t == 3) ? 1 : 0) : (t == 0 ? 1 : 0); }
// raw barrier (no implied vmcnt(0): LDS-DMA stays in flight across it); the asm fences keep hipcc from moving LDS accesses over it
#define AX_RAW_BARRIER()                          \
    do {                                          \
        asm volatile("" ::: "memory");            \
        __builtin_amdgcn_s_barrier();             \
        asm volatile("" ::: "memory");            \
    } while (0)
#define AX_THR 6.0f                       // the softmax reference max is raised when a tile's max exceeds it by more than this

__device__ __forceinline__ int ax_swap23(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

// exp(t) at fp32 grade for t <= AX_THR:  hi = fl(t log2 e), e = 2^hi (v_exp_f32, 1 ulp), d = t - hi ln 2 exactly enough (two
// fmas against ln 2 = LN2_HI + LN2_LO; |d| < 1e-5, so exp(d) = 1 + d to 1e-10).  !FINITE: t may be -inf (masked key): clamped
// to -104, where 2^hi flushes to 0.
template <bool FINITE>
__device__ __forceinline__ float ax_exp(float t) {
    if constexpr (!FINITE) t = fmaxf(t, -104.0f);
    const float hi = t * 1.44269502162933349609375f;
    const float e = __builtin_amdgcn_exp2f(hi);
    float d = fmaf(hi, -0.693147182464599609375f, t);
    d = fmaf(hi, 1.904654323148236e-09f, d);               // ln 2 = 0.69314718246... - 1.9047e-9
    return fmaf(e, d, e);
}

// per-wave constants of the fragment reads
struct ax_lane {
    int l31, hh;
    int kOff, kSw;          // K row feeding MFMA row l31 (floats) and its swizzle key; 128-byte rows
    int vSw;                // rows d = 32 dt + l31: (d >> 2) & 3 == (l31 >> 2) & 3
};
__device__ __forceinline__ ax_lane ax_lane_init(int lane) {
    ax_lane L;
    L.l31 = lane & 31; L.hh = lane >> 5;
    const int kr = ax_swap23(L.l31);
    L.kOff = kr * 32; L.kSw = (kr >> 1) & 7;
    L.vSw = (L.l31 >> 2) & 3;
    return L;
}

// S^T accumulator start: the bias of (key, query) for this lane's 16 keys of the tile whose first key index is key0.
// FULL: every key of the tile belongs to the item; else keys outside [0, T) read a clamped (finite) entry and are masked later.
template <bool FULL>
__device__ __forceinline__ void ax_bias_init(f32x16& sacc, const float* bt, int key0, int hh, int qrow_c, int T) {
    if constexpr (FULL) {
        const float* brow = bt + (key0 + 8 * hh - qrow_c + (T - 1));
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = brow[16 * (r >> 3) + (r & 7)];
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + 16 * (r >> 3) + 8 * hh + (r & 7);
            const int key_c = key < 0 ? 0 : (key < T ? key : T - 1);
            sacc[r] = bt[key_c - qrow_c + (T - 1)];
        }
    }
}

// S^T += K . Q^T for the K tile at Ks: four 16-wide d steps x six plane products (smallest terms first).
// DUAL: the 24 MFMAs of a tile all accumulate into sacc — one dependent chain, each link waiting for the previous result.  With
// three waves per SIMD (the shared-tile kernel) other waves fill those gaps; a key-split block runs one or two waves per SIMD, so it
// accumulates odd d steps in a second accumulator (two interleaved chains) and adds the two once (16 VALU adds, +16 VGPRs).
template <int NP, bool DUAL = false>
__device__ __forceinline__ void ax_qk(f32x16& sacc, const float* Ks, const f32x4 (&qf)[NP][4], const ax_lane& L) {
    f32x16 sacc2;
    if constexpr (DUAL) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc2[r] = 0.f;
    }
    if constexpr (!DUAL) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 kf[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) kf[p] = *(const f32x4*)(Ks + p * AX_PLANE_FLOATS + L.kOff + ((2 * s + L.hh) ^ L.kSw) * 4);
#pragma unroll
            for (int t = 0; t < ax_nprod<NP>(); ++t) sacc = ax_mfma<NP>(kf[ax_pa<NP>(t)], qf[ax_pb<NP>(t)][s], sacc);
        }
    } else {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {                    // d steps 2 s2 (-> sacc) and 2 s2 + 1 (-> sacc2), MFMAs alternating
            f32x4 ka[NP], kb[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                ka[p] = *(const f32x4*)(Ks + p * AX_PLANE_FLOATS + L.kOff + ((4 * s2 + L.hh) ^ L.kSw) * 4);
                kb[p] = *(const f32x4*)(Ks + p * AX_PLANE_FLOATS + L.kOff + ((4 * s2 + 2 + L.hh) ^ L.kSw) * 4);
            }
            const int sa = 2 * s2, sb = 2 * s2 + 1;
#pragma unroll
            for (int t = 0; t < ax_nprod<NP>(); ++t) {
                sacc = ax_mfma<NP>(ka[ax_pa<NP>(t)], qf[ax_pb<NP>(t)][sa], sacc);
                sacc2 = ax_mfma<NP>(kb[ax_pa<NP>(t)], qf[ax_pb<NP>(t)][sb], sacc2);
            }
        }
        sacc += sacc2;
    }
}

// online softmax of one tile (scores incl. bias in sacc): P planes -> pf, reference max / sum, O rescaled when the reference moves.
// `full` (uniform): every key of the tile belongs to the item.  Only the masking and the clamp inside exp differ between the two
// cases; the rescale branch and the MFMA phases around this function exist ONCE (two whole copies of the tile code made the
// register allocator keep two images of O: +32 VGPRs and 32 moves per tile).
template <int NP, bool FULL>
__device__ __forceinline__ float ax_probs(const f32x16& sacc, f32x4 (&pf)[NP][2], float m_run) {
    float lsum = 0.f;
    if constexpr (NP == 2) {
        // fp16 planes: the weights carry a factor 16 (vn_common.h vn_split2u), which costs nothing when it rides in the exponent:
        // P' = 16 exp(s - m) = 2^(s log2 e + c), c = 4 - m log2 e once per tile — ONE fma and one v_exp_f32 per score instead of the
        // six instructions of ax_exp (PMC: this kernel issues 9.9 VALU per MFMA, profiles/history/r03_h2_pmc_attention.txt).  The two
        // roundings (c and the fma) move the exponent by <= ulp(|m log2 e|) / 2 + ulp(|s log2 e + c|) / 2, i.e. a weight by a few
        // 1e-7 relative at |scores| ~ 10 — the size of the rounding the scores themselves carry out of their fp32 accumulation; c is
        // common to every weight formed under the same reference max, so it cancels between O and l.  A masked score (-inf) gives
        // 2^-inf = 0 without a clamp.  l is summed in the same scaled units (the kernels divide by 16 l' at the end).
        const float c = fmaf(m_run, -1.44269502162933349609375f, 4.0f);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x8 pe;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                pe[e] = __builtin_amdgcn_exp2f(fmaf(sacc[8 * s + e], 1.44269502162933349609375f, c));
                lsum += pe[e];
            }
            f16x8 p0, p1;
            vn_split2u_x8(pe, p0, p1);                      // <= 16 e^AX_THR < 6.5e3
            pf[0][s] = __builtin_bit_cast(f32x4, p0); pf[1][s] = __builtin_bit_cast(f32x4, p1);
        }
        return lsum;
    } else {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x8 pe;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                pe[e] = ax_exp<FULL>(sacc[8 * s + e] - m_run);
                lsum += pe[e];
            }
            bf16x8 p0, p1, p2;
            vn_split3_x8(pe, p0, p1, p2);
            pf[0][s] = __builtin_bit_cast(f32x4, p0); pf[1][s] = __builtin_bit_cast(f32x4, p1); pf[2][s] = __builtin_bit_cast(f32x4, p2);
        }
        return lsum;
    }
}
template <int NP>
__device__ __forceinline__ void ax_softmax(f32x16& sacc, f32x4 (&pf)[NP][2], float& m_run, float& l_run, f32x16 (&o)[2], int key0,
                                           int hh, int T, bool full) {
    if (!full) {                                            // first / last tile: the neighbours' tokens are masked out
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + 16 * (r >> 3) + 8 * hh + (r & 7);
            sacc[r] = (key >= 0 && key < T) ? sacc[r] : -INFINITY;
        }
    }
    float mx = sacc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));                     // finite: every tile has >= 1 valid key
    if (!__all(mx - m_run <= AX_THR)) {                     // first tile: m_run = -inf -> taken (alpha = 0 on zeros)
        const float m_new = fmaxf(m_run, mx);
        const float alpha = vn_exp_neg(m_run - m_new);      // lanes whose reference stays: exp(0) = 1 exactly
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        l_run *= alpha;
        m_run = m_new;
    }
    l_run += full ? ax_probs<NP, true>(sacc, pf, m_run) : ax_probs<NP, false>(sacc, pf, m_run);
}

// ---- training (attention_train_x3.hip; the TRAIN instantiation of the shared-tile kernel) ----------------------------------------
// keep-multipliers (0 or 1 / (1 - p)) of the EIGHT consecutive keys kb .. kb + 7 of the row whose key is rk.  vn_drop_bits hashes one
// PAIR of columns (col >> 1): eight keys touch four pairs when kb is even, five when it is odd (kb's parity is wave-uniform) —
// the same values as vn_drop_mul(d, vn_drop_bits(rk, kb + e), kb + e), with half the hashes.
__device__ __forceinline__ void ax_drop8(const vn_drop& d, uint32_t rk, int kb, float (&mul)[8]) {
    const int base = kb >> 1;
    uint32_t hs[5];
#pragma unroll
    for (int i = 0; i < 4; ++i) hs[i] = vn_mix32(rk ^ (uint32_t)(base + i));
    if (kb & 1) {
        hs[4] = vn_mix32(rk ^ (uint32_t)(base + 4));
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t bits = hs[(e + 1) >> 1], v = (e & 1) ? (bits & 0xFFFFu) : (bits >> 16);
            mul[e] = v >= d.thresh16 ? d.scale : 0.0f;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t bits = hs[e >> 1], v = (e & 1) ? (bits >> 16) : (bits & 0xFFFFu);
            mul[e] = v >= d.thresh16 ? d.scale : 0.0f;
        }
    }
}
// ax_probs with probability dropout (transformer.py:250): the row sum l counts every weight, the planes that feed P V carry the kept ones
template <bool FULL>
__device__ __forceinline__ float ax_probs_train(const f32x16& sacc, f32x4 (&pf)[3][2], float m_run, const vn_drop& d, uint32_t rk, int kb0) {
    float lsum = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        f32x8 pe;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            pe[e] = ax_exp<FULL>(sacc[8 * s + e] - m_run);
            lsum += pe[e];
        }
        if (d.thresh16) {
            float mul[8];
            ax_drop8(d, rk, kb0 + 16 * s, mul);
#pragma unroll
            for (int e = 0; e < 8; ++e) pe[e] *= mul[e];
        }
        bf16x8 p0, p1, p2;
        vn_split3_x8(pe, p0, p1, p2);
        pf[0][s] = __builtin_bit_cast(f32x4, p0); pf[1][s] = __builtin_bit_cast(f32x4, p1); pf[2][s] = __builtin_bit_cast(f32x4, p2);
    }
    return lsum;
}
// ax_softmax for training: d / rk = the dropout stream of this lane's query row
__device__ __forceinline__ void ax_softmax_train(f32x16& sacc, f32x4 (&pf)[3][2], float& m_run, float& l_run, f32x16 (&o)[2], int key0,
                                                 int hh, int T, bool full, const vn_drop& d, uint32_t rk) {
    if (!full) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + 16 * (r >> 3) + 8 * hh + (r & 7);
            sacc[r] = (key >= 0 && key < T) ? sacc[r] : -INFINITY;
        }
    }
    float mx = sacc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (!__all(mx - m_run <= AX_THR)) {
        const float m_new = fmaxf(m_run, mx);
        const float alpha = vn_exp_neg(m_run - m_new);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        l_run *= alpha;
        m_run = m_new;
    }
    l_run += full ? ax_probs_train<true>(sacc, pf, m_run, d, rk, key0 + 8 * hh) : ax_probs_train<false>(sacc, pf, m_run, d, rk, key0 + 8 * hh);
}

// O^T += V^T . P^T for the V^T tile at Vs: two 32-row d tiles x two 16-key steps x six plane products.
// ILV: the MFMAs of the two d tiles alternate (two independent chains in flight instead of six dependent MFMAs in a row) —
// for the key-split blocks, as above; costs the second tile's V^T fragments live at the same time (+12 VGPRs).
template <int NP, bool ILV = false>
__device__ __forceinline__ void ax_pv(f32x16 (&o)[2], const float* Vs, const f32x4 (&pf)[NP][2], const ax_lane& L) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if constexpr (!ILV) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                f32x4 vf[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) vf[p] = *(const f32x4*)(Vs + p * AX_PLANE_FLOATS + (32 * dt + L.l31) * 16 + ((2 * s + L.hh) ^ L.vSw) * 4);
#pragma unroll
                for (int t = 0; t < ax_nprod<NP>(); ++t) o[dt] = ax_mfma<NP>(vf[ax_pa<NP>(t)], pf[ax_pb<NP>(t)][s], o[dt]);
            }
        } else {
            f32x4 va[NP], vb[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                va[p] = *(const f32x4*)(Vs + p * AX_PLANE_FLOATS + L.l31 * 16 + ((2 * s + L.hh) ^ L.vSw) * 4);
                vb[p] = *(const f32x4*)(Vs + p * AX_PLANE_FLOATS + (32 + L.l31) * 16 + ((2 * s + L.hh) ^ L.vSw) * 4);
            }
#pragma unroll
            for (int t = 0; t < ax_nprod<NP>(); ++t) {
                o[0] = ax_mfma<NP>(va[ax_pa<NP>(t)], pf[ax_pb<NP>(t)][s], o[0]);
                o[1] = ax_mfma<NP>(vb[ax_pa<NP>(t)], pf[ax_pb<NP>(t)][s], o[1]);
            }
        }
    }
}

// one tile of the main loop from LDS images of K and V^T
template <int NP, bool TRAIN = false>
__device__ __forceinline__ void ax_tile(const float* Ks, const float* Vs, const float* bt, const f32x4 (&qf)[NP][4], const ax_lane& L,
                                        float& m_run, float& l_run, f32x16 (&o)[2], int key0, int qrow_c, int T, const vn_drop& d = vn_drop{},
                                        uint32_t rk = 0) {
    const bool full = key0 >= 0 && key0 + AX_KT <= T;      // uniform
    f32x16 sacc;
    f32x4 pf[NP][2];
    if (full) ax_bias_init<true>(sacc, bt, key0, L.hh, qrow_c, T);
    else ax_bias_init<false>(sacc, bt, key0, L.hh, qrow_c, T);
    ax_qk<NP>(sacc, Ks, qf, L);
    if constexpr (TRAIN && NP == 3) ax_softmax_train(sacc, pf, m_run, l_run, o, key0, L.hh, T, full, d, rk);
    else ax_softmax<NP>(sacc, pf, m_run, l_run, o, key0, L.hh, T, full);
    ax_pv<NP>(o, Vs, pf, L);
}

// LDS-DMA sources as buffer loads: descriptors over the k planes and the V^T planes, byte offsets of this block's tile 0 in plane 0
// and the plane strides (all uniform -> SGPRs).  Tile 0 of an item starts at global token row 32 g_lo <= b T: for b > 0 that is
// inside the previous item's rows of the same head-major image, never in front of the k planes (offset >= 0 for every b, h).
// The descriptors carry the TRUE extents of the two buffers (k_bytes from k16 to the end of the q / k allocation including its
// 32-row pad, v_bytes = the three V^T planes): a last tile's deliberate over-read of up to 31 rows stays inside the pad the
// allocation owns (ax_k_extent below), and anything beyond would read 0 instead of faulting (round 6; before, num_records was
// 0x7fffffff, i.e. the hardware's range check was off).
struct ax_src {
    __amdgpu_buffer_rsrc_t krs, vrs;
    unsigned k0, v0, kplane, vplane;
};
__device__ __forceinline__ ax_src ax_src_init(const uint16_t* k16, const uint16_t* vt16, long plane_qk, long plane_vt, int b, int h, int H,
                                              int T, int m_lo, int g_lo, int MT, unsigned k_bytes, unsigned v_bytes) {
    ax_src s;
    s.krs = __builtin_amdgcn_make_buffer_rsrc((void*)k16, 0, (int)k_bytes, 0x00020000);
    s.vrs = __builtin_amdgcn_make_buffer_rsrc((void*)vt16, 0, (int)v_bytes, 0x00020000);
    s.k0 = (unsigned)((((long)b * H + h) * T + (g_lo * AX_KT - m_lo)) * (VN_DHEAD * 2));
    s.v0 = (unsigned)(((long)h * MT + g_lo) * (VN_DHEAD * AX_KT * 2));
    s.kplane = (unsigned)(plane_qk * 2);
    s.vplane = (unsigned)(plane_vt * 2);
    return s;
}
// one LDS-DMA wave-instruction: 64 lanes x 16 B from rsrc[voff (per lane) + soff (uniform) + IMM] to lds + IMM .. + 1 KiB — the
// instruction offset of a buffer load to LDS is added on BOTH sides (MUBUF: LDS_ADDR = M0 base + inst_offset + lane * 16), so the
// four 1 KiB pieces of a plane tile are one M0 setting and IMM = 0 / 1024 / 2048 / 3072
template <int IMM = 0>
__device__ __forceinline__ void ax_dma(__amdgpu_buffer_rsrc_t rsrc, float* lds, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, IMM, 0);
}

// XCD-aware 1-D walk: blocks with blockIdx % 8 == x run on XCD x; every XCD gets a contiguous eighth of the (b, h, q-block) list,
// so the q-blocks of a head run on ONE XCD and share its K / V^T tiles in that L2 (with the natural order a head's q-blocks went
// round-robin over the XCDs and every one of them fetched the head's planes from the fabric: 235 MB per launch at B = 8)
__device__ __forceinline__ int ax_walk(int bid, int nwg) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int qq = nwg >> 3, rr = nwg & 7;
    return (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
}

// Q fragments (B operand of S^T = K Q^T): lane (j, hh) holds Q[q_j][16 step + 8 hh .. + 7] of every plane
template <int NP>
__device__ __forceinline__ void ax_load_q(f32x4 (&qf)[NP][4], const uint16_t* Qp, long plane_qk, int qrow_c, int hh) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[p][s] = *(const f32x4*)(Qp + (size_t)p * plane_qk + (size_t)qrow_c * VN_DHEAD + 16 * s + 8 * hh);
}

// normalise and store the 4-column group g of d tile dt of query row qrow:  o[dt][4 g ..] = O[q][32 dt + 8 g + 4 hh ..]
// (C/D map of the 32x32 MFMA).  The reference divides (softmax), it does not multiply by a reciprocal: keep the division per element
__device__ __forceinline__ void ax_store4(const f32x4& acc, float l_tot, float* out, uint16_t* out16, long plane16, long row, int h,
                                          int H, int hh, int dt, int g) {
    const f32x4 ov = {acc[0] / l_tot, acc[1] / l_tot, acc[2] / l_tot, acc[3] / l_tot};
    const int col = h * VN_DHEAD + 4 * hh + 32 * dt + 8 * g;
    // fp16 planes: O is a convex combination of the values, and a value beyond 4094 (16 v >= 65504) is already on the saturation
    // ledger from the QKV epilogue that wrote V^T — nothing new can saturate here, so the flag is not reported
    bool bad = false;
    // inference passes ONE of the two; the training forward both (fp32 for the backward, planes for the Wo GEMM that follows)
    if (out16) vn_store_planes4(out16, plane16, row, col, H * VN_DHEAD, ov, bad);
    if (out) *(f32x4*)(out + (size_t)row * ((size_t)H * VN_DHEAD) + col) = ov;
}
