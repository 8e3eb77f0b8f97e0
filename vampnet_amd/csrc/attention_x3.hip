// Self-attention with the shared T5 relative-position bias on the bf16 matrix cores at fp32 grade ("bf16x3"), for gfx950.
//
// Replaces MultiHeadRelativeAttention.forward's core (vampnet/modules/transformer.py:234-254) when the model runs in the
// bf16x3 precision:  attn = einsum(q,k)/sqrt(64) + bias[h, bucket(k - q)] ; softmax(dim=keys) ; einsum(attn, v).
// Same decomposition as attention_f32.hip (flash-style online softmax, both products computed transposed so that every
// softmax quantity is lane-local, bias from a 2T-1 table in LDS), but both GEMMs run as SIX v_mfma_f32_32x32x16_bf16
// products of exact three-way operand splits (gemm_x3.hip: x = x0 + x1 + x2 exactly, terms down to 2^-16 of the leading one
// kept, fp32 accumulation): 6/16 of the fp32-input MFMA's matrix time at the same error class.
//
// Operands arrive as planes, written by the QKV GEMM's epilogues (gemm_x3.hip):
//   q16, k16 : [3 planes][B][H][T][64] bf16 (q pre-multiplied by 1/8 = 1/sqrt(64): a power of two commutes with the split)
//   vt16     : [3 planes][H][ceil(B T / 32)][64 d][32 tokens] bf16 — V TRANSPOSED and blocked by tiles of 32 GLOBAL token rows
//              m = b T + t (the row index of the activations), so that a tile is one contiguous 4 KiB block (whole cache lines
//              for the LDS-DMA) whose rows are the MFMA A operand of O^T = V^T P^T, and the producing GEMM writes 16-byte
//              aligned runs whatever T is.  Key tiles are therefore aligned to m, not to t: an item's first and last tile
//              also hold the neighbouring items' tokens, which are masked (P = 0; the buffer is zero-filled once, so every
//              masked value is finite).
// Softmax probabilities are split into their three planes in registers (P in [0, 1]: exact).
//
// Work decomposition: one block per (b, h, q-block of 32 NW queries), XCD-aware 1-D walk; block = NW waves; wave w owns 32 query columns.
//   S^T[key][q] = K[key][:] . Q[q][:]      A = K tile rows (LDS), B = Q (registers, loaded once)
//   O^T[d][q]   = V^T[d][key] . P^T[key][q]  A = V^T tile rows (LDS), B = P (the registers the softmax produced)
// MFMA row rho of S^T is mapped to key swap_bits_2_3(rho) (the A operand simply reads that K row): a lane (query j = lane & 31,
// half hh = lane >> 5) then holds keys 16 (r >> 3) + 8 hh + (r & 7), r = 0..15 — for each 16-key step of the second product
// EIGHT CONSECUTIVE keys, i.e. exactly its k-slots of the B operand, and the V^T operand is one 16-byte LDS read.
// K/V^T tiles (32 keys: 12 + 12 KiB for the three planes) are double-buffered by LDS-DMA (global_load_lds_dwordx4), one
// barrier per tile.  LDS images are lane-linear, so the bank swizzles sit on the DMA source address:
//   K rows are 128 B: slot s of row r lives at s ^ ((r >> 1) & 7);  V^T rows are 64 B: s ^ ((r >> 2) & 3)  (both give every
//   16-lane ds_read_b128 group 16 distinct 16-byte slots of the 256-byte bank row).
// Algorithmic FLOPs: 4*T*T*64 per (b,h); executed on the bf16 pipe: 6x that.
#include <type_traits>
#include "vn_common.h"

#define AX_KT 32                          // keys per tile
#define AX_PLANE_FLOATS 1024              // one plane tile of K (32 x 128 B) or V^T (64 x 64 B): 4 KiB
#define AX_STAGE_FLOATS (6 * AX_PLANE_FLOATS)

__device__ __forceinline__ int ax_swap23(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

// ABL (tuning only): bit 0 = no exp / split (P = bf16(S) in all planes), bit 1 = no S^T MFMAs, bit 2 = no O^T MFMAs (results invalid
// with bits 0-2); bit 3 = s_setprio 1 around the MFMA phases (off by default: 110.7 vs 112.5 us at B = 8); bit 4 = phase trace: wave 0 of every 16th block accumulates
// s_memtime deltas per phase over its tiles into trace[block / 16][8] = {wait, barrier, dma issue, qk, softmax, pv, total, hw_id}.
// stagger (any variant): a block whose waves sit in SIMD wave slot w starts (w % 3) * stagger * 64 cycles late, so that the blocks
// sharing a CU do not run the same phase at the same time.
// (Tried and dropped, profiles/r02_attention_x3_phase_order_ab.txt: one VALU phase + one 48-MFMA phase per tile with K staged a
// tile ahead of V^T — 125 vs 119 us; six- and two-wave blocks; see also r02_attention_x3_kernel_times.txt.
// Round 2, second half (profiles/r02_attention_x3_probe_*.txt, r02_ubench_mfma_valu_coissue.txt): the phase trace shows a tile
// costing one wave ~4500 cycles alone (MFMA 1536, ~280 VALU instructions ~1400, LDS / DMA / barrier waits the rest) and ~7400 with
// three blocks per CU, i.e. the matrix pipe and the VALU are about equally loaded and overlap only partly — the micro-benchmark
// puts the overlap the hardware gives at <= 4 plain VALU instructions per MFMA and wave for free, ~5 cycles each beyond that.
// A software-pipelined main loop (S of the next tile and half of PV in the same basic block as the softmax, bitwise equal to
// this loop, commit 491c2d5) ran 115.5 vs 112.5 us at two blocks per CU (256 VGPRs) and was removed; start staggers by SIMD wave
// slot change nothing (117.8-118.9 vs 119.9 us); 2 blocks per CU 114.8, 1 block 145 us.  What did pay: DMA sources as a uniform tile
// base + a fixed per-lane offset (-5 %), no s_setprio (-1.6 %).
// The tail round (800 blocks of 128 queries on 768 slots: 110.7 us against 88.1 us for the 760 blocks of 19 heads,
// profiles/r02_attention_x3_tail_round_probe.txt) was attacked with a 96-query block shape (three waves, V^T single-buffered + the
// T+95-entry window of the bias table = 38.6 KiB, FOUR blocks per CU, 960 blocks = one round, no idle wave; bitwise equal to this
// shape, commit "96-query block shape"): 104.7 vs 109.6 us kernel-only, but a tile costs a wave 8200 instead of 7200 cycles (second
// barrier per tile, a third more K / V^T traffic per query) and inside the model it came out even to slower (271.0 vs 269.3 ms per
// step, profiles/r02_attention_x3_probe_96_vs_128_query_blocks.txt) — removed.)
template <int NW, int ABL = 0>
__global__ __launch_bounds__(NW * 64, 3) void vn_attention_x3_kernel(const uint16_t* __restrict__ q16, const uint16_t* __restrict__ k16,
                                                                    long plane_qk, const uint16_t* __restrict__ vt16, long plane_vt,
                                                                    const float* __restrict__ bias_full, float* __restrict__ out,
                                                                    uint16_t* __restrict__ out16, long plane16, int B, int H, int T,
                                                                    int stagger, unsigned* __restrict__ trace) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* bt = smem + 2 * AX_STAGE_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    // 1-D grid, XCD-aware: blocks with blockIdx % 8 == x run on XCD x; give every XCD a contiguous eighth of the (b, h, q-block)
    // walk so that the q-blocks of a head run on ONE XCD and share its K / V^T tiles in that L2 (with the natural order a head's
    // q-blocks went round-robin over the XCDs and every one of them fetched the head's planes from the fabric: 235 MB per launch)
    const int nqb = (T + NW * 32 - 1) / (NW * 32);
    int lid;
    {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int qq = nwg >> 3, rr = nwg & 7;
        lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
    const int qb = lid % nqb, h = (lid / nqb) % H, b = lid / (nqb * H);
    // key tiles of this item in global token rows: tile g holds tokens 32 g .. 32 g + 31, key index t = m - b T
    const int m_lo = b * T;
    const int g_lo = m_lo / AX_KT, NT = (m_lo + T - 1) / AX_KT - g_lo + 1;
    const int MT = (B * T + AX_KT - 1) / AX_KT;
    const size_t head = (size_t)b * H + h;
    const uint16_t* Qp = q16 + head * (size_t)T * VN_DHEAD;
    const uint16_t* Kp = k16 + head * (size_t)T * VN_DHEAD;
    const uint16_t* Vp = vt16 + ((size_t)h * MT + g_lo) * (VN_DHEAD * AX_KT);
    const int q0 = qb * (NW * 32) + wave * 32;
    const bool active = q0 < T;                         // waves past the end only help with the DMA and the barriers
    const int qrow = q0 + l31;
    const int qrow_c = qrow < T ? qrow : T - 1;

    const int nb = 2 * T - 1;
    for (int i = tid; i < nb; i += NW * 64) bt[i] = bias_full[(size_t)h * nb + i];

    // Q fragments (B operand of S^T = K Q^T): lane (j, hh) holds Q[q_j][16 step + 8 hh .. + 7] of every plane
    bf16x8 qf[3][4];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int s = 0; s < 4; ++s)
            qf[p][s] = __builtin_bit_cast(bf16x8, *(const f32x4*)(Qp + (size_t)p * plane_qk + (size_t)qrow_c * VN_DHEAD + 16 * s + 8 * hh));

    // LDS-DMA: 24 wave-instructions of 1 KiB per stage (K: 3 planes x 4, each 8 rows x 128 B; V^T: 3 planes x 4, each 16 rows x 64 B);
    // wave w issues piece w of every plane tile.  A piece's source is (uniform plane / tile base) + (a per-lane offset that is the
    // same for every plane and tile), so a tile costs six scalar adds and no vector address arithmetic.  K rows are NOT clamped to
    // the item: the rows of a first / last tile that belong to the neighbouring items (or to the q planes in front of / the
    // 32-row padding behind the k planes) are read as they are and masked to -inf before the softmax.
    static_assert(NW == 4, "one piece of every plane tile per wave");
    const int krow = 8 * wave + (lane >> 3), vrow = 16 * wave + (lane >> 2);
    const unsigned kvoff = (unsigned)(krow * VN_DHEAD + ((lane & 7) ^ ((krow >> 1) & 7)) * 8) * 2u;      // bytes inside a K tile
    const unsigned vvoff = (unsigned)(vrow * AX_KT + ((lane & 3) ^ ((vrow >> 2) & 3)) * 8) * 2u;         // bytes inside a V^T tile
    const char* kbase = (const char*)(Kp + (long)(g_lo * AX_KT - m_lo) * VN_DHEAD);                       // tile 0, plane 0 (key0 <= 0)
    const char* vbase = (const char*)Vp;
    auto stage = [&](int buf, int kt) {                       // tile kt -> stage buf
        if (kt >= NT) return;
        float* base = smem + buf * AX_STAGE_FLOATS + wave * 256;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const char* ks = kbase + ((size_t)p * plane_qk * 2 + (size_t)kt * (AX_KT * VN_DHEAD * 2));
            const char* vs = vbase + ((size_t)p * plane_vt * 2 + (size_t)kt * (VN_DHEAD * AX_KT * 2));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ks + kvoff),
                                             (__attribute__((address_space(3))) void*)(base + (4 * p) * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vs + vvoff),
                                             (__attribute__((address_space(3))) void*)(base + (12 + 4 * p) * 256), 16, 0, 0);
        }
    };

    const int kr = ax_swap23(l31);                      // K row feeding MFMA row l31
    const int kOff = kr * 32, kSw = (kr >> 1) & 7;      // floats; 128-byte rows
    const int vSw = (l31 >> 2) & 3;                     // rows d = 32 dt + l31: (d >> 2) & 3 == (l31 >> 2) & 3

    float m_run = -INFINITY, l_run = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;

    f32x16 sacc;
    bf16x8 pf[3][2];
    // ---- S^T = K . Q^T for the tile in stage `kb`: four 16-wide d steps x six plane products (smallest terms first)
    auto qk_phase = [&](int kb) {
        const float* Ks = smem + kb * AX_STAGE_FLOATS;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
        if constexpr (ABL & 8) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 kf[3];
#pragma unroll
            for (int p = 0; p < 3; ++p)
                kf[p] = __builtin_bit_cast(bf16x8, *(const f32x4*)(Ks + p * AX_PLANE_FLOATS + kOff + ((2 * s + hh) ^ kSw) * 4));
            if constexpr (ABL & 2) {
                sacc[s] += (float)kf[0][0] + (float)kf[1][1] + (float)kf[2][2];
            } else {
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[2][s], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2], qf[0][s], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1], qf[1][s], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[1][s], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1], qf[0][s], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0][s], sacc, 0, 0, 0);
            }
        }
        if constexpr (ABL & 8) __builtin_amdgcn_s_setprio(0);
    };
    // ---- online softmax of tile kt (scores in sacc): P planes -> pf, running max / sum, O rescaled
    auto softmax_phase = [&](int kt) {
        float mx = -INFINITY;
        const int key0 = (g_lo + kt) * AX_KT - m_lo;                   // key index of the tile's first row
        const float* brow = bt + (key0 + 8 * hh - qrow_c + (T - 1));   // bias of key key0 + 8 hh for this query
        const bool full = key0 >= 0 && key0 + AX_KT <= T;              // every key of the tile belongs to this item (uniform)
        if (full) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float x = sacc[r] + brow[16 * (r >> 3) + (r & 7)];   // q was pre-scaled by 1/sqrt(64); += bias
                sacc[r] = x;
                mx = fmaxf(mx, x);
            }
        } else {                                                       // first / last tile: the neighbours' tokens are masked out
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key0 + 16 * (r >> 3) + 8 * hh + (r & 7);
                const int key_c = key < 0 ? 0 : (key < T ? key : T - 1);
                float x = sacc[r] + bt[key_c - qrow_c + (T - 1)];
                x = (key >= 0 && key < T) ? x : -INFINITY;
                sacc[r] = x;
                mx = fmaxf(mx, x);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);                          // finite: every tile has >= 1 valid key
        const float alpha = vn_exp_neg(m_run - m_new);                 // first tile: exp(-inf) = 0
        float lsum = 0.f;
        auto probs = [&](auto finite) {        // a full tile has no -inf score: exp without the clamp (same value for finite arguments)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f32x8 pe;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    pe[e] = (ABL & 1) ? sacc[8 * s + e] - m_new : vn_exp_neg<decltype(finite)::value>(sacc[8 * s + e] - m_new);
                    lsum += pe[e];
                }
                if constexpr (ABL & 1) pf[0][s] = pf[1][s] = pf[2][s] = __builtin_convertvector(pe, bf16x8);
                else vn_split3_x8(pe, pf[0][s], pf[1][s], pf[2][s]);
            }
        };
        if (full) probs(std::true_type{});
        else probs(std::false_type{});
        l_run = l_run * alpha + lsum;
        m_run = m_new;
        if (!__all(alpha == 1.0f)) {                                   // exp(0) = 1 exactly: skipping the multiply changes no bit
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
    };
    // ---- O^T += V^T . P^T for the tile in stage `vb`: two 32-row d tiles x two 16-key steps x six plane products
    auto pv_phase = [&](int vb) {
        const float* Vs = smem + vb * AX_STAGE_FLOATS + 3 * AX_PLANE_FLOATS;
        if constexpr (ABL & 8) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                bf16x8 vf[3];
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    vf[p] = __builtin_bit_cast(bf16x8, *(const f32x4*)(Vs + p * AX_PLANE_FLOATS + (32 * dt + l31) * 16 + ((2 * s + hh) ^ vSw) * 4));
                if constexpr (ABL & 4) {
                    o[dt][s] += (float)vf[0][0] * (float)pf[0][s][0] + (float)vf[1][1] * (float)pf[1][s][1] + (float)vf[2][2] * (float)pf[2][s][2];
                } else {
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0], pf[2][s], o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[2], pf[0][s], o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1], pf[1][s], o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0], pf[1][s], o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1], pf[0][s], o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0], pf[0][s], o[dt], 0, 0, 0);
                }
            }
        if constexpr (ABL & 8) __builtin_amdgcn_s_setprio(0);
    };

    if (stagger > 0) {                                      // de-phase the blocks that share this CU
        const int slot = __builtin_amdgcn_s_getreg((3 << 11) | 4) % 3;         // HW_REG_HW_ID bits [3:0]: wave slot in the SIMD
        for (int i = 0; i < slot * stagger; ++i) __builtin_amdgcn_s_sleep(1);
    }
    unsigned long long tr_t = 0;
    unsigned tr_acc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tr_start = 0;
    const bool tracing = (ABL & 16) && trace && wave == 0 && (lid & 15) == 0;
    auto tick = [&](int slot) {
        if constexpr (ABL & 16) {
            if (tracing) {
                const unsigned long long now = __builtin_readcyclecounter();
                tr_acc[slot] += (unsigned)(now - tr_t);
                tr_t = now;
            }
        }
    };
    if constexpr (ABL & 16) { tr_t = tr_start = __builtin_readcyclecounter(); }
    stage(0, 0);
    for (int kt = 0; kt < NT; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pieces of tile kt have landed
        tick(0);
        __syncthreads();                                        // all pieces landed; everybody is done with tile kt - 1
        tick(1);
        stage((kt + 1) & 1, kt + 1);
        tick(2);
        if (!active) continue;
        qk_phase(kt & 1);
        tick(3);
        softmax_phase(kt);
        tick(4);
        pv_phase(kt & 1);
        tick(5);
    }
    if constexpr (ABL & 16) {
        if (tracing && lane == 0) {
            unsigned* t = trace + (lid >> 4) * 8;
#pragma unroll
            for (int i = 0; i < 6; ++i) t[i] = tr_acc[i];
            t[6] = (unsigned)(__builtin_readcyclecounter() - tr_start);
            t[7] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        }
    }

    // ---- finish: the two lanes of a query add their row sums; normalise; store.
    // accumulator o[dt][r] = O[q_j][d = 32 dt + (r & 3) + 8 (r >> 2) + 4 hh]  (C/D map of the 32x32 MFMA)
    float l_tot = l_run + __shfl_xor(l_run, 32);
    if (active && qrow < T) {
        const float inv = 1.0f / l_tot;
        const size_t ooff = ((size_t)b * T + qrow) * ((size_t)H * VN_DHEAD) + h * VN_DHEAD + 4 * hh;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // the reference divides (softmax), it does not multiply by a reciprocal: keep the division per element
                const f32x4 ov = {o[dt][4 * g] / l_tot, o[dt][4 * g + 1] / l_tot, o[dt][4 * g + 2] / l_tot, o[dt][4 * g + 3] / l_tot};
                (void)inv;
                const size_t off = ooff + 32 * dt + 8 * g;
                if (out16) vn_store_planes4(out16, plane16, (long)b * T + qrow, h * VN_DHEAD + 4 * hh + 32 * dt + 8 * g, H * VN_DHEAD, ov);
                else *(f32x4*)(out + off) = ov;
            }
    }
}

// tuning hooks (process-global; scripts/attn_probe.py): ablation / variant bits, dynamic-LDS override (occupancy: > 80 KiB = one block
// per CU, > 53.3 KiB = two), start stagger, phase-trace buffer
static int g_ax_abl = -1, g_ax_lds = 0, g_ax_stagger = -1;
static unsigned* g_ax_trace = nullptr;
extern "C" int vn_debug_attention_x3_config(int abl, int lds_bytes, int stagger, void* trace_dev) {
    g_ax_abl = abl; g_ax_lds = lds_bytes; g_ax_stagger = stagger; g_ax_trace = (unsigned*)trace_dev;
    return VN_OK;
}

int vn_launch_attention_x3(vn_ctx* ctx, const uint16_t* q16, const uint16_t* k16, long plane_qk, const uint16_t* vt16, long plane_vt,
                           const float* relbias_full, float* out, uint16_t* out16, long plane16, int B, int H, int T, hipStream_t s) {
    if (B <= 0 || T <= 0) return VN_OK;
    size_t lds = (size_t)(2 * AX_STAGE_FLOATS + 2 * T - 1 + 3) * sizeof(float);
    if (lds > 80 * 1024) return vn_fail(ctx, VN_ERR_INVALID, "attention_x3: T=%s%ld too long for two blocks per CU", "", T);
    if (g_ax_lds > (int)lds && g_ax_lds <= 160 * 1024) lds = g_ax_lds;
    if (!(ctx->attr_mask & VN_ATTR_ATTN_X3)) {
        VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)(vn_attention_x3_kernel<4, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)(vn_attention_x3_kernel<4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)(vn_attention_x3_kernel<4, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)(vn_attention_x3_kernel<4, 7>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)(vn_attention_x3_kernel<4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)(vn_attention_x3_kernel<4, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ctx->attr_mask |= VN_ATTR_ATTN_X3;
    }
    const int pi = vn_prof_pre(ctx, 1, 4.0 * T * (double)T * VN_DHEAD * H * B, s, 16.0 * T * VN_DHEAD * (double)H * B);
    // four waves (128 queries) per block, three blocks per CU (LDS: 2 x 24 KiB stages + the bias table; 168 VGPRs).  Measured
    // alternatives (profiles/r02_attention_x3_kernel_times.txt): six waves (one round of 480 blocks at B = 8) 119 vs 112 us,
    // two waves no better at any batch size.
    static const int abl_env = [] { const char* e = getenv("VN_ATTN_X3_ABL"); return e ? atoi(e) & 31 : 0; }();   // tuning only
    static const int stagger_env = [] { const char* e = getenv("VN_ATTN_X3_STAGGER"); return e ? atoi(e) : 0; }();
    const int abl = g_ax_abl >= 0 ? g_ax_abl : abl_env;
    const int stagger = g_ax_stagger >= 0 ? g_ax_stagger : stagger_env;
#define AX_GO(A) hipLaunchKernelGGL((vn_attention_x3_kernel<4, A>), dim3(vn_cdiv(T, 128) * H * B), dim3(256), lds, s, q16, k16, plane_qk, vt16, \
                                    plane_vt, relbias_full, out, out16, plane16, B, H, T, stagger, g_ax_trace)
    switch (abl) {
        case 1: AX_GO(1); break;
        case 6: AX_GO(6); break;
        case 7: AX_GO(7); break;
        case 8: AX_GO(8); break;
        case 16: AX_GO(16); break;
        default: AX_GO(0); break;
    }
#undef AX_GO
    vn_prof_post(ctx, pi, s);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}
