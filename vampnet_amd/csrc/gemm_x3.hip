// fp32-grade GEMM on the bf16 matrix cores of gfx950 ("bf16x3"):  C[M][N] (op)= A[M][K] * W[N][K]^T
//
// Both operands arrive as THREE bf16 planes with x = x0 + x1 + x2 exactly (vn_split3: 8 + 8 + 8 significand bits; bf16
// shares fp32's exponent range, so nothing under- or overflows).  The product keeps every term down to 2^-16 of the
// leading one,
//     A W^T  ~=  A0 W0 + (A0 W1 + A1 W0) + (A0 W2 + A1 W1 + A2 W0),
// six v_mfma_f32_32x32x16_bf16 per 16-wide k-step with fp32 accumulation: each bf16 x bf16 product is exact in fp32, and
// the dropped terms are <= 3 * 2^-24 relative — the same size as the rounding of one fp32 multiply (measured on the
// model's shapes: max error 1.4e-6 vs 2.9e-6 for an fp32-accumulating fp32 GEMM, both against float64).  The matrix
// cores run bf16 at 16x the fp32-input MFMA rate (MI355X_MICROARCH.md: 2.5 PF vs 157 TF dense), so six passes cost
// 6/16 of the exact-fp32 kernel's matrix time: the ceiling moves from 157 to ~417 fp32-equivalent TFLOP/s.
//
// SECOND OPERAND FORMAT, "f16x2" (round 3; FMT = 1 below, the model's default precision): TWO fp16 planes per operand,
//     h0 = fp16(x),  h1 = fp16((x - h0) * 2^11)          (vn_common.h vn_split2h)
// fp16 carries 11 significand bits, the remainder of the round-to-nearest conversion is <= 2^-11 |x| and exact in fp32, and the
// 2^11 puts it into the same binades as h0, so x = h0 + 2^-11 h1 to within 2^-22 |x| over fp16's whole normal range (6.1e-5 ..
// 65504, below it to 2^-25 absolute: gfx950 produces and multiplies fp16 subnormals, scripts/ubench/f16_denorm_probe.hip; beyond
// it the split saturates).  The product keeps THREE terms,
//     A W^T  ~=  A0 W0 + 2^-11 (A0 W1 + A1 W0),
// as three v_mfma_f32_32x32x16_f16 per k-step into TWO fp32 accumulators (the second one joins times 2^-11 before the epilogue);
// the dropped A1 W1 is 2^-22 of the leading term.  Operand error 2^-22 is four times the representation error of fp32 itself and
// random in sign: on the model's shapes the result is 2.4e-7 rms from the float64 product — what an fp32-accumulating GEMM of the
// UNSPLIT operands gives (2.5e-7), because the fp32 accumulation is the larger error in both (tests/test_gpu_f16x2.py,
// tests/test_host_logic.py::test_f16x2_split_numerics_on_the_host) — at HALF the matrix work and 2/3 of the operand bytes of bf16x3:
// the ceiling moves from 417 to 833 fp32-equivalent TFLOP/s, measured 1.5-1.7x on the model's shapes
// (profiles/history/r03_gemm_f16x2_vs_bf16x3.txt).  No per-tensor scaling is needed (both planes keep 11 bits wherever the value sits in
// fp16's range); what fp16 cannot hold is |x| > 65504 — no activation or weight of this model family comes near it (the reference
// itself runs bf16 autocast on a GPU), and the split clamps instead of producing inf.  The backward GEMMs of training stay on
// fp32 / bf16x3: gradients live far below fp16's range.
// Geometry: a stage holds 2 x (BM + 128) rows = 32 / 48 / 40 KiB, THREE stages are resident for every tile height; schedule: one
// phase per k-tile and wave group (fragments of both k-steps, 2 x 3 products; two barriers per k-tile) — see the k-loop.
// Ablations (scripts/gemm_h2_ablation.py, profiles/history/r03_gemm_f16x2_ablation.txt): without the DMA -15 %, without the fragment reads
// -2.5 %, MFMAs + barriers alone 428 TF-eq = 1283 TF executed on the QKV shape — the same executed rate the bf16x3 kernel stops at;
// moving the DMA issue between the phases or never waiting for it changes nothing: this format too runs at the chip's power limit,
// and what is left to gain is bytes moved per flop, not schedule.
//
// Kernel: one 512-thread block (8 waves) per output tile of BM x 128, one block per CU, two waves per SIMD; three tile
// configurations (x3_geo below): 128 x 128 (waves 4 x 2, wave tile 32 x 64), 256 x 128 (4 x 2, 64 x 64) and 192 x 128 (2 x 4,
// 96 x 32: the height that fills whole rounds of 256 CUs on the model's B = 8 shapes); x3_choose picks height and k-split per launch.
//   * k-tile = 32 bf16 = 64 bytes per row.  A stage holds the six plane tiles (3 x BM rows of A, 3 x 128 of W) = 48 / 72 / 60
//     KiB, filled by LDS-DMA (global_load_lds_dwordx4: one wave instruction = 16 rows x 64 B of one plane); three (128 rows) or
//     two stages are resident.
//   * Operand layouts in global memory: PLANAR (three [rows][K] images; single-op tests, training splits) or TILED
//     ([row / 16][k / 32][plane][16][32]: the 16 rows x 64 B of an instruction are ONE contiguous 1 KiB piece, i.e. eight whole
//     cache lines instead of sixteen half lines from sixteen rows — weights re-laid at load, activations written so by their
//     producers; +2.6 % / +4.2 % end to end).  The LDS image and the swizzle below are the same for both.
//   * Loading each plane tile ONCE and using it in two or three of the six products is what separates this kernel
//     from running a K' = 6K bf16 GEMM over concatenated planes: 6 plane-tile loads per 6 MFMA groups instead of 12.
//   * LDS image is lane-linear (DMA constraint), so the bank swizzle sits on the SOURCE address: 16-byte slot s of row r
//     is stored at slot s ^ ((r >> 2) & 3).  A ds_read_b128 is serviced in 16-lane groups whose rows are
//     {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (+32): with 64-byte rows the four rows that share r % 4 (hence a
//     256-byte bank-row quarter) have four different (r >> 2) & 3, so every group touches 16 distinct slots
//     (SQ_LDS_BANK_CONFLICT = 0, profiles/history/r02_pmc_gemm_x3_pipe4.txt).
//   * Fragment of v_mfma_f32_32x32x16_bf16: lane l holds row l & 31, k = 8 (l >> 5) .. + 7 of the 16-wide step, i.e.
//     slot 2 s + (l >> 5) of the row for sub-step s in {0, 1}.
//   * "Ping-pong" schedule: the eight waves form two groups (waves 0-3 / 4-7: one wave of each group on every SIMD) that
//     run the same step sequence ONE PHASE APART — while a group issues the 12 / 24 / 18 MFMAs of a k-step (s_setprio 1) the other
//     reads its next fragments and issues LDS-DMA; two raw s_barriers per k-step keep the alternation, counted vmcnt keeps
//     the DMA in flight across them (guide section 5, 8-phase template).  Round-1's lock-step schedule ran 7-12 % slower
//     (profiles/history/r02_gemm_x3_schedules.txt).
//   * Epilogues as in gemm_f32.hip (store / bias / residual / GEGLU gate / QKV head-major scatter); the GEGLU result is
//     written as three planes again (it is only ever the A operand of the next GEMM).
//   * rows >= M and columns >= N are clamped on load and masked on store.
// Work distribution: data-parallel, one block per output tile, XCD-aware tile walk (8-row tile groups, one contiguous eighth
// of the walk per XCD); the store / residual epilogues of badly filling shapes are split along K in two passes (split images
// to a workspace, vn_splitk_reduce_kernel adds them in fixed order).  A persistent stream-K distribution (per-XCD round-robin
// whole tiles + k-split tail, deterministic two-pass fix-up) was built, verified and measured in round 2 and removed again:
// with every CU busy the kernel sits at the chip's power limit (~1.75 GHz, GRBM_GUI_ACTIVE / duration) while a partly filled
// last round runs at a higher clock, so the evenly balanced launch gained <= 2.5 % on the QKV shape, lost to its fix-up pass on
// Wo / W1 / W2 and lost 3 ms of 55 on the B = 1 path (profiles/history/r02_gemm_x3_streamk_v2_vs_dp.txt, r02_c10_3_cfg1_*.json).
// On the GPU the result is within 7.5e-6 of float64 at 4600 x 3840 x 1280 (the fp32-input MFMA kernel: 9e-6).
#include <stdlib.h>
#include "vn_common.h"

#define X3_BN 128
#define X3_KT 32                                  // bf16 per k-tile
// staged epilogues: float offset of the folded norm's scale table = right behind the largest LDS image (QKV3's transposed V^T planes,
// [3][128][RP + 8] 16-bit values; the fp32 image [RP][128] is smaller)
#define X3_RS_OFF(RP) (3 * 128 * ((RP) + 8) / 2)
#define X3_BPLANE (128 * 16)                      // floats of one W plane tile: 128 rows x 64 B

// Geometry by tile configuration CFG: BM x 128 output tile, eight waves as WR x WC, a wave owns RI x CJ MFMA tiles of 32 x 32.
//   CFG 1: 128 x 128, waves 4 x 2, wave tile 32 x 64     CFG 2: 256 x 128, waves 4 x 2, wave tile 64 x 64
//   CFG 3: 192 x 128, waves 2 x 4, wave tile 96 x 32 — the height that fills whole rounds of 256 CUs on the model's shapes at
//          B = 8 (M = 4600 = 23.96 x 192: QKV 720 tiles = 2.8 rounds, Wo / W2 240 tiles = one round WITHOUT a k-split, classifier
//          768 tiles = 3.0 rounds) where 128 rows leave the last of 5 / 2 rounds mostly empty and 256 rows quantise worse still.
// A stage = three A plane tiles (BM rows x 64 B) then three W plane tiles (128 rows x 64 B).
// NP = planes per operand: 3 (bf16x3, six products per k-step) or 2 (f16x2, three products; vn_common.h vn_split2h).
template <int CFG, int NP = 3>
struct x3_geo {
    static constexpr int RI = (CFG == 3 || CFG == 4) ? 3 : CFG, CJ = (CFG == 3 || CFG == 4) ? 1 : 2;
    static constexpr int WR = CFG == 4 ? 1 : CFG == 3 ? 2 : 4, WC = (CFG == 3 || CFG == 4) ? 4 : 2;
    static constexpr int KG = CFG == 4 ? 2 : 1;                     // wave groups that split the k-steps of a k-tile between them (CFG 4)
    static constexpr int BM = 32 * RI * WR;
    static constexpr int APLANE = BM * 16;                          // floats
    static constexpr int STAGE = NP * (APLANE + X3_BPLANE);         // floats: 48 / 72 / 60 / 42 KiB (NP = 2: 32 / 48 / 40)
    static constexpr int NQ = NP * (BM + 128) / 16;                 // 1 KiB DMA wave-instructions per stage: 48 / 72 / 60 / 42 (32 / 48 / 40)
    static constexpr int NPW = (NQ + 7) / 8;                        // per wave: 6 / 9 / 8 (CFG 3, NP 3: the last one only in waves 0-3) / 6 (CFG 4: the last one only in waves 0-1); 4 / 6 / 5
    static constexpr int NBUF = (CFG == 1 || CFG == 4 || NP == 2) ? 3 : 2;      // resident stages (f16x2: 96 / 144 / 120 KiB)
    static constexpr int RP = 32 * WR;                              // rows of one pass of the staged epilogue's LDS image
    static constexpr int NPROD = NP == 3 ? 6 : 3;                   // matrix-core products per 16-wide k-step
    // folded RMSNorm: float offset of the per-row scale table (BM floats) — behind the stages AND behind the largest epilogue image
    // (QKV3's transposed V^T planes), so it survives from the kernel's prologue, where it is filled, to the epilogue that reads it
    static constexpr int RS_OFF = NBUF * STAGE > X3_RS_OFF(RP) ? NBUF * STAGE : X3_RS_OFF(RP);
    static constexpr int NI = NP == 3 ? 4 : 2;                      // two-buffer schedule: DMA pieces issued between the products of k-step 0
};
static_assert(x3_geo<1>::BM == 128 && x3_geo<2>::BM == 256 && x3_geo<3>::BM == 192 && x3_geo<4>::BM == 96, "tile heights");
static_assert(x3_geo<4>::KG * x3_geo<4>::BM * 128 <= x3_geo<4>::RS_OFF && x3_geo<3>::BM * 128 <= x3_geo<3>::RS_OFF, "the image epilogue's fp32 tile images end in front of the scale table");

__device__ __forceinline__ int x3_xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ void x3_tile_coords(int t, int tiles_m, int tiles_n, int& tm, int& tn, int GROUP_M) {
    // GROUP_M (8): 8-row groups of tiles, so that the tiles an XCD runs together share A / W panels in its L2
    const int per_group = GROUP_M * tiles_n;
    const int grp = t / per_group;
    const int first_m = grp * GROUP_M;
    const int gsz = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
    const int in_grp = t - grp * per_group;
    tm = first_m + in_grp % gsz;
    tn = in_grp / gsz;
}

// Operand mode bit of the kernel's ABL parameter (the other bits are tuning ablations): TN — C[M][N] = A^T W with BOTH operands given
// token-major, A [K][M] and W [K][N], as the tiled planes [k / 16][row / 32][plane][16][32] — which is, byte for byte, the tiled layout
// [token / 16][feature / 32][plane][16][32] of an activation matrix [tokens][features] that the NT form reads as its A operand.  The dW
// GEMMs of training (dW = dY^T X, contraction over the tokens) therefore read the planes that dX's GEMM and the forward pass already
// made — no transposing pass (round 6; train.hip grad_weight).  The DMA copies the same 1 KiB pieces; only the fragment read differs:
// ds_read_b64_tr_b16, the LDS transposing read of gfx950 (two per fragment instead of one ds_read_b128).
#define X3_MODE_TN 8
typedef short x3_s16x4 __attribute__((ext_vector_type(4)));
typedef float x3_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 x3_tr_frag(const float* at) {
    const auto* q = (const __attribute__((address_space(3))) x3_s16x4*)at;
    const x3_f32x2 lo = __builtin_bit_cast(x3_f32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) x3_s16x4*)q));
    const x3_f32x2 hi = __builtin_bit_cast(x3_f32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) x3_s16x4*)(q + 32)));
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}

#define X3_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define X3_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// raw barrier (no implied vmcnt(0): LDS-DMA stays in flight across it); the asm fences keep hipcc from moving LDS accesses over it
#define X3_BARRIER()                              \
    do {                                          \
        asm volatile("" ::: "memory");            \
        __builtin_amdgcn_s_barrier();             \
        asm volatile("" ::: "memory");            \
        __builtin_amdgcn_sched_barrier(0);        \
    } while (0)

// epilogue of one output tile from the accumulators in MFMA layout.
// C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
template <int EPI, int CFG, int FMT = 0>
__device__ __forceinline__ void x3_epilogue(const vn_gemm_args& p, const f32x16 (&acc)[x3_geo<CFG>::RI][x3_geo<CFG>::CJ], int m0, int n0,
                                            int wm, int wn, int lane) {
    using G = x3_geo<CFG>;
    static_assert(EPI != VN_EPI_GEGLU || G::CJ == 2, "GEGLU pairs the wave's two column tiles (value, gate)");
    const int l31 = lane & 31, h = lane >> 5;
    const int colw = n0 + wn * 32 * G::CJ + l31;
    bool bad = false;                     // f16x2 plane epilogues: a value that did not fit fp16 (vn_common.h, saturation ledger)
#pragma unroll
    for (int i = 0; i < G::RI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 * G::RI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row >= p.M) continue;
            if constexpr (EPI == VN_EPI_GEGLU) {
                // wave tile = 64 packed columns = 32 value (j = 0) + 32 gate (j = 1), interleaved at pack time
                const int ocol = (n0 + wn * 64) / 2 + l31;
                if (2 * ocol >= p.N) continue;
                const float o = acc[i][0][r] * vn_gelu_tanh(acc[i][G::CJ - 1][r]);
                if (p.C16) {
                    const bool til = vn_planes_tiled(p.c_plane);
                    const long cp = til ? 512 : (p.c_plane < 0 ? -p.c_plane : p.c_plane);
                    if constexpr (FMT) {
                        uint16_t t0, t1;
                        vn_split2h(o, t0, t1, bad);
                        uint16_t* d = p.C16 + (til ? vn_tiled_off_np(row, ocol, p.ldc, 2) : (size_t)row * p.ldc + ocol);
                        d[0] = t0; d[cp] = t1;
                    } else {
                        uint16_t t0, t1, t2;
                        vn_split3(o, t0, t1, t2);
                        uint16_t* d = p.C16 + (til ? vn_tiled_off(row, ocol, p.ldc) : (size_t)row * p.ldc + ocol);
                        d[0] = t0; d[cp] = t1; d[2 * cp] = t2;
                    }
                } else {
                    p.C[(size_t)row * p.ldc + ocol] = o;
                }
            } else {
#pragma unroll
                for (int j = 0; j < G::CJ; ++j) {
                    const int col = colw + j * 32;
                    if (col >= p.N) continue;
                    const float v = acc[i][j][r];
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
                    } else if constexpr (EPI == VN_EPI_QKV3) {
                        // columns [0, D) = q (x 1/sqrt(64), exact), [D, 2D) = k: split planes, head-major [which][b][h][t][64];
                        // [2D, 3D) = v -> V^T blocked by tiles of 32 token rows: [h][row / 32][d][row % 32]
                        const int D = p.H * VN_DHEAD;
                        const int which = col / D, rem = col - which * D;
                        const int hd = rem >> 6, d = rem & 63;
                        uint16_t t0, t1, t2 = 0;
                        // FMT 1: the attention operands of the f16x2 precision — fp16 two-plane, second plane unscaled, V times 16
                        if constexpr (FMT) vn_split2u(which == 0 ? v * 0.125f : which == 1 ? v : v * 16.0f, t0, t1, bad);
                        else vn_split3(which ? v : v * 0.125f, t0, t1, t2);
                        if (which < 2) {
                            const int b = row / p.T, t = row - b * p.T;
                            uint16_t* dst = p.C16 + which * p.qkv_plane + (((size_t)b * p.H + hd) * p.T + t) * VN_DHEAD + d;
                            dst[0] = t0; dst[p.c_plane] = t1;
                            if constexpr (!FMT) dst[2 * p.c_plane] = t2;
                        } else {
                            uint16_t* dst = p.V16 + (((size_t)hd * ((p.M + 31) >> 5) + (row >> 5)) * VN_DHEAD + d) * 32 + (row & 31);
                            dst[0] = t0; dst[p.v_plane] = t1;
                            if constexpr (!FMT) dst[2 * p.v_plane] = t2;
                        }
                    }
                }
            }
        }
    }
    if constexpr (FMT && (EPI == VN_EPI_GEGLU || EPI == VN_EPI_QKV3)) vn_sat_report(p.sat, EPI == VN_EPI_QKV3 ? VN_SAT_ATTN : VN_SAT_OPERAND, bad);
}

// The same epilogues staged through LDS (free after the k-loop): every wave drops its accumulators into a row-major image
// of RP = 32 WR tile rows (the i-th 32-row block of every wave row), then all 512 threads read the image back in 16-byte pieces
// and issue 16-byte global accesses.  The direct epilogue above issues one 2- or 4-byte store per accumulator register (32 per
// MFMA tile, x3 for split planes), and the store tail of a tile is ISSUE-bound (guide T21): the in-model GEMMs with plane
// epilogues ran 10-14 % below the fp32-store shapes (round-2 kernel stats).  RI passes; every
// thread of the block must call it.
//   fp32 kinds (store / bias / residual / QKV scatter): image [RP][128] fp32 (64 / 32 KiB)
//   GEGLU planes: image [3][RP][64] bf16;  QKV3 planes: [3][RP][128] bf16 for the q / k tiles, TRANSPOSED [3][128 columns][RP + 8]
//   bf16 for the v tiles (a tile is all q, all k or all v: D % 128 == 0)
template <int EPI, int CFG, int FMT = 0>
__device__ __forceinline__ void x3_epilogue_staged(const vn_gemm_args& p, const f32x16 (&acc)[x3_geo<CFG>::RI][x3_geo<CFG>::CJ], int m0, int n0,
                                                   int wave, int lane, float* lds) {
    using G = x3_geo<CFG>;
    constexpr int RI = G::RI, CJ = G::CJ, RP = G::RP, VP = RP + 8;          // VP: pitch of the transposed v image
    static_assert(EPI != VN_EPI_GEGLU || CJ == 2, "GEGLU pairs the wave's two column tiles (value, gate)");
    const int wm = wave / G::WC, wn = wave % G::WC, l31 = lane & 31, h = lane >> 5, tid = wave * 64 + lane;
    uint16_t* L16 = (uint16_t*)lds;
    bool bad = false;                     // saturation ledger (vn_common.h)
    // Folded RMSNorm, consumer side (vn_common.h, vn_gemm_args::ssq_in): r = rsqrt(mean(x^2) + eps) of this tile's BM rows from the
    // producers' group sums, once per tile, kept behind the largest image (X3_RS_OFF); 1.0 when nothing is folded (x * 1.0f == x
    // exactly, so the unfolded results do not change)
    constexpr bool FOLD_IN = EPI == VN_EPI_QKV3 || EPI == VN_EPI_QKV || EPI == VN_EPI_GEGLU || EPI == VN_EPI_BIAS;
    const float* rs = lds + x3_geo<CFG, FMT ? 2 : 3>::RS_OFF;      // filled by the kernel's prologue (x3_fold_load / x3_fold_table)
    const bool fold_in = FOLD_IN && p.ssq_in != nullptr;
#pragma unroll
    for (int i = 0; i < RI; ++i) {
        __syncthreads();                  // k-loop reads / the previous pass's read-out are done in every wave (and rs is complete)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int R = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;                 // image row; tile row = (R >> 5) * 32 RI + 32 i + (R & 31)
            // folded norm: the scale of accumulator row r (the plane kinds apply it here, before the split; fp32 kinds on read-out)
            const float sc = (fold_in && (EPI == VN_EPI_GEGLU || EPI == VN_EPI_QKV3)) ? rs[wm * 32 * RI + 32 * i + (R & 31)] : 1.0f;
            if constexpr (EPI == VN_EPI_GEGLU) {
                const float o = (acc[i][0][r] * sc) * vn_gelu_tanh(acc[i][CJ - 1][r] * sc);
                uint16_t* d = L16 + R * 64 + wn * 32 + l31;
                if constexpr (FMT) {
                    uint16_t t0, t1;
                    vn_split2h(o, t0, t1, bad);
                    d[0] = t0; d[RP * 64] = t1;
                } else {
                    uint16_t t0, t1, t2;
                    vn_split3(o, t0, t1, t2);
                    d[0] = t0; d[RP * 64] = t1; d[2 * RP * 64] = t2;
                }
            } else if constexpr (EPI == VN_EPI_QKV3) {
                if (n0 < 2 * p.H * VN_DHEAD) {                                      // q / k tile: row-major image
#pragma unroll
                    for (int j = 0; j < CJ; ++j) {
                        const int c = wn * 32 * CJ + j * 32 + l31;
                        uint16_t t0, t1, t2 = 0;
                        const float qv = n0 < p.H * VN_DHEAD ? (acc[i][j][r] * sc) * 0.125f : acc[i][j][r] * sc;    // q: x 1/sqrt(64)
                        if constexpr (FMT) vn_split2u(qv, t0, t1, bad);
                        else vn_split3(qv, t0, t1, t2);
                        uint16_t* d = L16 + R * 128 + c;
                        d[0] = t0; d[RP * 128] = t1;
                        if constexpr (!FMT) d[2 * RP * 128] = t2;
                    }
                } else if ((r & 3) == 0) {                                          // v tile: transposed image [column][row], pitch VP
#pragma unroll
                    for (int j = 0; j < CJ; ++j) {
                        const int c = wn * 32 * CJ + j * 32 + l31;
                        uint16_t t[3][4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float vv = acc[i][j][r + e] * (fold_in ? rs[wm * 32 * RI + 32 * i + (R & 31) + e] : 1.0f);     // rows R .. R + 3
                            if constexpr (FMT) vn_split2u(vv * 16.0f, t[0][e], t[1][e], bad);
                            else vn_split3(vv, t[0][e], t[1][e], t[2][e]);
                        }
#pragma unroll
                        for (int q = 0; q < (FMT ? 2 : 3); ++q) {
                            uint2 pk = {t[q][0] | ((unsigned)t[q][1] << 16), t[q][2] | ((unsigned)t[q][3] << 16)};
                            *(uint2*)(L16 + q * (128 * VP) + c * VP + R) = pk;      // rows R .. R + 3 (r & 3 = 0 .. 3)
                        }
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < CJ; ++j) lds[R * 128 + wn * 32 * CJ + j * 32 + l31] = acc[i][j][r];
            }
        }
        __syncthreads();
        if constexpr (EPI == VN_EPI_GEGLU) {
            constexpr int ONP = FMT ? 2 : 3;
#pragma unroll
            for (int k = 0; k < ONP * RP * 8 / 512; ++k) {                          // planes x RP rows x 8 pieces of 8 columns
                const int idx = tid + 512 * k;
                const int q = idx / (RP * 8), R = (idx >> 3) % RP, c8 = (idx & 7) * 8;
                const int row = m0 + (R >> 5) * 32 * RI + 32 * i + (R & 31), ocol = n0 / 2 + c8;
                if (row < p.M && 2 * ocol < p.N) {
                    uint16_t* dst = vn_planes_tiled(p.c_plane) ? p.C16 + vn_tiled_off_np(row, ocol, p.ldc, ONP) + 512 * q
                                                               : p.C16 + (size_t)q * (p.c_plane < 0 ? -p.c_plane : p.c_plane) + (size_t)row * p.ldc + ocol;
                    *(u32x4*)dst = *(const u32x4*)(L16 + q * (RP * 64) + R * 64 + c8);
                }
            }
        } else if constexpr (EPI == VN_EPI_QKV3) {
            const int D = p.H * VN_DHEAD;
            if (n0 < 2 * D) {
#pragma unroll
                for (int k = 0; k < (FMT ? 2 : 3) * RP * 16 / 512; ++k) {           // planes x RP rows x 16 pieces of 8 columns
                    const int idx = tid + 512 * k;
                    const int q = idx / (RP * 16), a = (idx >> 4) % RP, b8 = (idx & 15) * 8;
                    const int row = m0 + (a >> 5) * 32 * RI + 32 * i + (a & 31), col = n0 + b8;
                    if (row >= p.M || col >= p.N) continue;
                    const int which = col >= D ? 1 : 0, rem = col - which * D;
                    const int b = row / p.T, t = row - b * p.T;
                    *(u32x4*)(p.C16 + (size_t)q * p.c_plane + which * p.qkv_plane + (((size_t)b * p.H + (rem >> 6)) * p.T + t) * VN_DHEAD + (rem & 63)) =
                        *(const u32x4*)(L16 + q * (RP * 128) + a * 128 + b8);
                }
            } else {
#pragma unroll
                for (int k = 0; k < (FMT ? 2 : 3) * 128 * (RP / 8) / 512; ++k) {    // planes x 128 columns x RP / 8 pieces of 8 rows
                    const int idx = tid + 512 * k;
                    const int q = idx / (128 * (RP / 8)), a = (idx / (RP / 8)) & 127, b8 = (idx % (RP / 8)) * 8;     // a = column (feature)
                    const int row = m0 + (b8 >> 5) * 32 * RI + 32 * i + (b8 & 31), f = n0 + a - 2 * D;
                    if (row >= p.M || n0 + a >= p.N) continue;
                    *(u32x4*)(p.V16 + (size_t)q * p.v_plane + (((size_t)(f >> 6) * ((p.M + 31) >> 5) + (row >> 5)) * VN_DHEAD + (f & 63)) * 32 + (row & 31)) =
                        *(const u32x4*)(L16 + q * (128 * VP) + a * VP + b8);
                }
            }
        } else if constexpr (EPI == VN_EPI_CONV) {
            // conv1d_f32.hip's epilogue: bias, residual, tanh, and the NEXT layer's Snake1d written beside the raw result.  A thread
            // keeps ONE group of four output channels for the whole tile (idx & 31 does not depend on k or i): bias, alpha and
            // 1 / (alpha + 1e-9) live in registers.  The loop is kept ROLLED: with the transcendental code unrolled into the pass loop
            // hipcc gives up unrolling the loop over i, and the accumulators (indexed by i) go to scratch
            const int c4 = (tid & 31) * 4, col = n0 + c4;
            if (col < p.N) {
                f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, al = {1.f, 1.f, 1.f, 1.f}, inv = {0.f, 0.f, 0.f, 0.f};
                if (p.bias) bias4 = *(const f32x4*)(p.bias + col);
                if (p.Y2 || p.C16) {
                    al = *(const f32x4*)(p.alpha + col);
#pragma unroll
                    for (int e = 0; e < 4; ++e) inv[e] = 1.0f / (al[e] + 1e-9f);
                }
#pragma nounroll
                for (int k = 0; k < RP * 32 / 512; ++k) {
                    const int R = (tid >> 5) + 16 * k;
                    const int row = m0 + (R >> 5) * 32 * RI + 32 * i + (R & 31);
                    if (row >= p.M) continue;
                    f32x4 v = *(const f32x4*)(lds + R * 128 + c4) + bias4;
                    const int b = row / p.conv_trows, tq = row - b * p.conv_trows;
                    const int t_out = tq * p.conv_out_stride + p.conv_out_off;
                    if (t_out < 0 || t_out >= p.conv_tout) continue;
                    const long orow = (long)b * p.conv_tout + t_out;
                    const size_t o = (size_t)orow * p.N + col;
                    if (p.resid) v += *(const f32x4*)(p.resid + o);
                    if (p.conv_act == 1) { v[0] = tanhf(v[0]); v[1] = tanhf(v[1]); v[2] = tanhf(v[2]); v[3] = tanhf(v[3]); }
                    if (p.C) *(f32x4*)(p.C + o) = v;
                    if (p.Y2 || p.C16) {
                        const f32x4 w4 = {vn_snake(v[0], al[0], inv[0]), vn_snake(v[1], al[1], inv[1]), vn_snake(v[2], al[2], inv[2]),
                                          vn_snake(v[3], al[3], inv[3])};
                        if (p.Y2) *(f32x4*)(p.Y2 + o) = w4;
                        if (p.C16) vn_store_planes4(p.C16, p.c_plane, orow, col, p.N, w4, bad);
                    }
                }
            }
        } else if (EPI == VN_EPI_RESIDUAL && p.X16 != nullptr) {
            // folded norm, producer side: x += acc, the split planes of the new rows and their sums of squares.  A thread owns EIGHT
            // columns of a row (16-byte plane stores: the tile is 3 x 2 bytes per element on top of the fp32 row), the 16 threads of an
            // image row are 16 consecutive lanes (DPP sum, no LDS round trip); N % 128 == 0 (launcher), so only rows are masked
#pragma unroll
            for (int k = 0; k < RP * 16 / 512; ++k) {                               // RP rows x 16 pieces of 8 columns
                const int idx = tid + 512 * k;
                const int R = idx >> 4, c8 = (idx & 15) * 8;
                const int row = m0 + (R >> 5) * 32 * RI + 32 * i + (R & 31), col = n0 + c8;
                f32x8 v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (row < p.M) {
                    const f32x4 l0 = *(const f32x4*)(lds + R * 128 + c8), l1 = *(const f32x4*)(lds + R * 128 + c8 + 4);
                    const f32x8 acc8 = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                    if (!FMT && p.x16_only) {
                        // bf16x3: the planes ARE the residual stream (their sum is the fp32 value exactly): read the old row from them,
                        // write the new one to them — 6 + 6 bytes per element instead of 4 + 4 + 6, and no fp32 image of x at all
                        v = acc8 + vn_load_planes8_bf16x3_tiled(p.X16, row, col, p.N);
                    } else {
                        float* c = p.C + (size_t)row * p.ldc + col;
                        const f32x4 a0 = l0 + *(const f32x4*)c;
                        const f32x4 a1 = l1 + *(const f32x4*)(c + 4);
                        *(f32x4*)c = a0;
                        *(f32x4*)(c + 4) = a1;
                        v = f32x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                    }
                    vn_store_planes8_tiled(p.X16, p.x16_plane, row, col, p.N, v, bad);
                }
                const float s2 = vn_sum16(vn_ssq4(f32x4{v[4], v[5], v[6], v[7]}, vn_ssq4(f32x4{v[0], v[1], v[2], v[3]})));
                if ((tid & 15) == 0 && row < p.M) p.ssq_out[(size_t)(n0 >> 7) * p.M + row] = s2;
            }
        } else {
#pragma unroll
            for (int k = 0; k < RP * 32 / 512; ++k) {                               // RP rows x 32 pieces of 4 columns
                const int idx = tid + 512 * k;
                const int R = idx >> 5, c4 = (idx & 31) * 4;
                const int trow = (R >> 5) * 32 * RI + 32 * i + (R & 31);
                const int row = m0 + trow, col = n0 + c4;
                const bool ok = row < p.M && col < p.N;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok) {
                    v = *(const f32x4*)(lds + R * 128 + c4);
                    if constexpr (FOLD_IN) {
                        if (fold_in) { const float sc = rs[trow]; v[0] *= sc; v[1] *= sc; v[2] *= sc; v[3] *= sc; }
                    }
                    if constexpr (EPI == VN_EPI_QKV) {
                        const int D = p.H * VN_DHEAD;
                        const int which = col / D, rem = col - which * D;
                        const int b = row / p.T, t = row - b * p.T;
                        *(f32x4*)(p.C + which * p.qkv_plane + (((size_t)b * p.H + (rem >> 6)) * p.T + t) * VN_DHEAD + (rem & 63)) = v;
                    } else {
                        float* c = p.C + (size_t)row * p.ldc + col;
                        if constexpr (EPI == VN_EPI_BIAS) v += *(const f32x4*)(p.bias + col);
                        if constexpr (EPI == VN_EPI_RESIDUAL) v += *(const f32x4*)c;
                        *(f32x4*)c = v;
                    }
                }
            }
        }
    }
    if constexpr (FMT && (EPI == VN_EPI_GEGLU || EPI == VN_EPI_QKV3 || EPI == VN_EPI_CONV || EPI == VN_EPI_RESIDUAL))
        vn_sat_report(p.sat, EPI == VN_EPI_QKV3 ? VN_SAT_ATTN : VN_SAT_OPERAND, bad);
}

// Epilogues of the k-split tile (CFG 4) through ONE fp32 image of the whole tile per wave group: every wave drops its accumulators at
// their tile coordinates into its group's image [BM][128] (KG images: the stages are free after the k-loop), one barrier, then all 512
// threads read the images back in 16-byte pieces, ADD the groups' partial sums in group order (deterministic) and run the epilogue on
// the sums — so the k-split's reduction costs one more LDS read per element and no second pass.  The arithmetic of every epilogue is
// the expression of x3_epilogue_staged (results of a KG = 1 geometry would be bitwise the staged ones); plane kinds form the planes
// from the fp32 image on the way out (16-byte global stores), so no second image and no second barrier exist.  bf16x3 operands only.
template <int EPI, int CFG, int FMT = 0>
__device__ __forceinline__ void x3_epilogue_image(const vn_gemm_args& p, const f32x16 (&acc)[x3_geo<CFG>::RI][x3_geo<CFG>::CJ], int m0, int n0,
                                                  int wave, int lane, float* lds) {
    using G = x3_geo<CFG>;
    constexpr int RI = G::RI, CJ = G::CJ, BM = G::BM, KG = G::KG, IMG = BM * 128;
    static_assert(!FMT && EPI != VN_EPI_CONV, "image epilogue: bf16x3 operands, no convolution epilogue");
    const int wq = wave % (G::WR * G::WC), kg = wave / (G::WR * G::WC);
    const int wm = wq / G::WC, wn = wq % G::WC, l31 = lane & 31, h = lane >> 5, tid = wave * 64 + lane;
    {
        float* img = lds + kg * IMG;
#pragma unroll
        for (int i = 0; i < RI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int R = wm * 32 * RI + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;       // tile row
#pragma unroll
                for (int j = 0; j < CJ; ++j) img[R * 128 + wn * 32 * CJ + j * 32 + l31] = acc[i][j][r];
            }
    }
    __syncthreads();
    auto ld4 = [&](int R, int c) {
        f32x4 v = *(const f32x4*)(lds + R * 128 + c);
#pragma unroll
        for (int g = 1; g < KG; ++g) v += *(const f32x4*)(lds + g * IMG + R * 128 + c);
        return v;
    };
    auto ld1 = [&](int R, int c) {
        float v = lds[R * 128 + c];
#pragma unroll
        for (int g = 1; g < KG; ++g) v += lds[g * IMG + R * 128 + c];
        return v;
    };
    constexpr bool FOLD_IN = EPI == VN_EPI_QKV3 || EPI == VN_EPI_QKV || EPI == VN_EPI_GEGLU || EPI == VN_EPI_BIAS;
    const float* rs = lds + G::RS_OFF;                   // folded norm: per-row scales (kernel prologue)
    const bool fold_in = FOLD_IN && p.ssq_in != nullptr;
    bool bad = false;
    if constexpr (EPI == VN_EPI_GEGLU) {
        // the tile's 128 packed columns = 2 x (32 value, 32 gate); a thread forms four consecutive outputs of a row
#pragma unroll
        for (int k = 0; k < BM * 16 / 512; ++k) {
            const int idx = tid + 512 * k;
            const int R = idx >> 4, c4 = (idx & 15) * 4;
            const int vcol = (c4 >> 5) * 64 + (c4 & 31);
            const int row = m0 + R, ocol = n0 / 2 + c4;
            if (row >= p.M || 2 * ocol >= p.N) continue;
            const f32x4 val = ld4(R, vcol), gate = ld4(R, vcol + 32);
            const float sc = fold_in ? rs[R] : 1.0f;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (val[e] * sc) * vn_gelu_tanh(gate[e] * sc);
            if (p.C16) {
                vn_store_planes4(p.C16, p.c_plane, row, ocol, p.ldc, o, bad);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) p.C[(size_t)row * p.ldc + ocol + e] = o[e];
            }
        }
    } else if constexpr (EPI == VN_EPI_QKV3) {
        const int D = p.H * VN_DHEAD;
        if (n0 < 2 * D) {                                                       // q / k tile: eight consecutive head features of a token
#pragma unroll
            for (int k = 0; k < BM * 16 / 512; ++k) {
                const int idx = tid + 512 * k;
                const int R = idx >> 4, b8 = (idx & 15) * 8;
                const int row = m0 + R, col = n0 + b8;
                if (row >= p.M || col >= p.N) continue;
                const f32x4 l0 = ld4(R, b8), l1 = ld4(R, b8 + 4);
                const float sc = fold_in ? rs[R] : 1.0f;
                f32x8 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = l0[e] * sc; v[4 + e] = l1[e] * sc; }
                if (n0 < D) {                                                   // q: x 1/sqrt(64)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] * 0.125f;
                }
                bf16x8 p0, p1, p2;
                vn_split3_x8(v, p0, p1, p2);
                const int which = col >= D ? 1 : 0, rem = col - which * D;
                const int b = row / p.T, t = row - b * p.T;
                uint16_t* dst = p.C16 + which * p.qkv_plane + (((size_t)b * p.H + (rem >> 6)) * p.T + t) * VN_DHEAD + (rem & 63);
                *(u32x4*)dst = __builtin_bit_cast(u32x4, p0);
                *(u32x4*)(dst + p.c_plane) = __builtin_bit_cast(u32x4, p1);
                *(u32x4*)(dst + 2 * p.c_plane) = __builtin_bit_cast(u32x4, p2);
            }
        } else {                                                                // v tile: eight consecutive tokens of one head feature (V^T)
#pragma unroll
            for (int k = 0; k < 128 * (BM / 8) / 512; ++k) {
                const int idx = tid + 512 * k;
                const int a = idx & 127, b8 = (idx >> 7) * 8;
                const int row = m0 + b8, f = n0 + a - 2 * D;
                if (row >= p.M || n0 + a >= p.N) continue;
                f32x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = ld1(b8 + e, a) * (fold_in ? rs[b8 + e] : 1.0f);
                bf16x8 p0, p1, p2;
                vn_split3_x8(v, p0, p1, p2);
                uint16_t* dst = p.V16 + (((size_t)(f >> 6) * ((p.M + 31) >> 5) + (row >> 5)) * VN_DHEAD + (f & 63)) * 32 + (row & 31);
                *(u32x4*)dst = __builtin_bit_cast(u32x4, p0);
                *(u32x4*)(dst + p.v_plane) = __builtin_bit_cast(u32x4, p1);
                *(u32x4*)(dst + 2 * p.v_plane) = __builtin_bit_cast(u32x4, p2);
            }
        }
    } else if (EPI == VN_EPI_RESIDUAL && p.X16 != nullptr) {
        // folded norm, producer side (as x3_epilogue_staged): the 16 threads of a row are 16 consecutive lanes
#pragma unroll
        for (int k = 0; k < BM * 16 / 512; ++k) {
            const int idx = tid + 512 * k;
            const int R = idx >> 4, c8 = (idx & 15) * 8;
            const int row = m0 + R, col = n0 + c8;
            f32x8 v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (row < p.M) {
                const f32x4 l0 = ld4(R, c8), l1 = ld4(R, c8 + 4);
                const f32x8 acc8 = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                if (p.x16_only) {
                    v = acc8 + vn_load_planes8_bf16x3_tiled(p.X16, row, col, p.N);
                } else {
                    float* c = p.C + (size_t)row * p.ldc + col;
                    const f32x4 a0 = l0 + *(const f32x4*)c;
                    const f32x4 a1 = l1 + *(const f32x4*)(c + 4);
                    *(f32x4*)c = a0;
                    *(f32x4*)(c + 4) = a1;
                    v = f32x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                }
                vn_store_planes8_tiled(p.X16, p.x16_plane, row, col, p.N, v, bad);
            }
            const float s2 = vn_sum16(vn_ssq4(f32x4{v[4], v[5], v[6], v[7]}, vn_ssq4(f32x4{v[0], v[1], v[2], v[3]})));
            if ((tid & 15) == 0 && row < p.M) p.ssq_out[(size_t)(n0 >> 7) * p.M + row] = s2;
        }
    } else {
#pragma unroll
        for (int k = 0; k < BM * 32 / 512; ++k) {                                   // BM rows x 32 pieces of 4 columns
            const int idx = tid + 512 * k;
            const int R = idx >> 5, c4 = (idx & 31) * 4;
            const int row = m0 + R, col = n0 + c4;
            if (row >= p.M || col >= p.N) continue;
            f32x4 v = ld4(R, c4);
            if constexpr (FOLD_IN) {
                if (fold_in) { const float sc = rs[R]; v[0] *= sc; v[1] *= sc; v[2] *= sc; v[3] *= sc; }
            }
            if constexpr (EPI == VN_EPI_QKV) {
                const int D = p.H * VN_DHEAD;
                const int which = col / D, rem = col - which * D;
                const int b = row / p.T, t = row - b * p.T;
                *(f32x4*)(p.C + which * p.qkv_plane + (((size_t)b * p.H + (rem >> 6)) * p.T + t) * VN_DHEAD + (rem & 63)) = v;
            } else {
                float* c = p.C + (size_t)row * p.ldc + col;
                if constexpr (EPI == VN_EPI_BIAS) v += *(const f32x4*)(p.bias + col);
                if constexpr (EPI == VN_EPI_RESIDUAL) v += *(const f32x4*)c;
                *(f32x4*)c = v;
            }
        }
    }
}

// Epilogue of the CONVT mode (rows = output channels, columns = positions; CFG 3: 192 channels, CFG 4: 96 channels with the k-steps
// split between the two wave groups).  Every wave drops its accumulators TRANSPOSED into its group's image [128 positions][BM + 4]
// (a lane holds four consecutive channels of one position per register quad: 16-byte LDS stores, pitch BM + 4 floats = conflict-free),
// one barrier, then the threads read whole position rows back — a thread keeps ONE group of four channels (bias, alpha, 1 / alpha in
// registers), consecutive threads consecutive channel groups of one position: 16-byte global accesses that cover a position's channels
// contiguously — add the groups in group order and run conv1d_f32.hip's epilogue expression (bias, residual, tanh, the next layer's Snake).
template <int CFG>
__device__ __forceinline__ void x3_epilogue_convt(const vn_gemm_args& p, const f32x16 (&acc)[x3_geo<CFG>::RI][x3_geo<CFG>::CJ], int n0,
                                                  int wave, int lane, float* lds) {
    using G = x3_geo<CFG>;
    constexpr int RI = G::RI, CJ = G::CJ, BM = G::BM, KG = G::KG, PITCH = BM + 4, IMG = 128 * PITCH;
    static_assert(KG * IMG <= G::NBUF * G::STAGE, "the transposed images fit in the stages");
    const int wq = wave % (G::WR * G::WC), kg = wave / (G::WR * G::WC);
    const int wm = wq / G::WC, wn = wq % G::WC, l31 = lane & 31, h = lane >> 5, tid = wave * 64 + lane;
    __syncthreads();                                          // the k-loop's fragment reads are done in every wave
    {
        float* img = lds + kg * IMG;
#pragma unroll
        for (int i = 0; i < RI; ++i)
#pragma unroll
            for (int j = 0; j < CJ; ++j)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int R0 = wm * 32 * RI + 32 * i + 8 * g4 + 4 * h;          // channels R0 .. R0 + 3 = registers 4 g4 .. 4 g4 + 3
                    const f32x4 v = {acc[i][j][4 * g4], acc[i][j][4 * g4 + 1], acc[i][j][4 * g4 + 2], acc[i][j][4 * g4 + 3]};
                    *(f32x4*)(img + (wn * 32 * CJ + 32 * j + l31) * PITCH + R0) = v;
                }
    }
    __syncthreads();
    constexpr int NG = BM / 4, PPP = 512 / NG;                // channel groups of a position, positions per pass
    if (tid >= NG * PPP) return;
    const int c4 = (tid % NG) * 4;                            // this thread's four channels, for the whole tile
    if (c4 >= p.M) return;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, al = {1.f, 1.f, 1.f, 1.f}, inv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias4 = *(const f32x4*)(p.bias + c4);
    if (p.Y2 || p.C16) {
        al = *(const f32x4*)(p.alpha + c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) inv[e] = 1.0f / (al[e] + 1e-9f);
    }
    bool bad = false;
#pragma nounroll
    for (int pos = tid / NG; pos < 128; pos += PPP) {
        const int row = n0 + pos;                             // GEMM column = output position (b, t')
        if (row >= p.N) break;
        f32x4 v = *(const f32x4*)(lds + pos * PITCH + c4);
#pragma unroll
        for (int g = 1; g < KG; ++g) v += *(const f32x4*)(lds + g * IMG + pos * PITCH + c4);
        v += bias4;
        const int b = row / p.conv_trows, tq = row - b * p.conv_trows;
        const int t_out = tq * p.conv_out_stride + p.conv_out_off;
        if (t_out < 0 || t_out >= p.conv_tout) continue;
        const long orow = (long)b * p.conv_tout + t_out;
        const size_t o = (size_t)orow * p.M + c4;
        if (p.resid) v += *(const f32x4*)(p.resid + o);
        if (p.conv_act == 1) { v[0] = tanhf(v[0]); v[1] = tanhf(v[1]); v[2] = tanhf(v[2]); v[3] = tanhf(v[3]); }
        if (p.C) *(f32x4*)(p.C + o) = v;
        if (p.Y2 || p.C16) {
            const f32x4 w4 = {vn_snake(v[0], al[0], inv[0]), vn_snake(v[1], al[1], inv[1]), vn_snake(v[2], al[2], inv[2]),
                              vn_snake(v[3], al[3], inv[3])};
            if (p.Y2) *(f32x4*)(p.Y2 + o) = w4;
            if (p.C16) vn_store_planes4(p.C16, p.c_plane, orow, c4, p.M, w4, bad);
        }
    }
}

// one block per output tile (gridDim.y > 1: split-K images, store epilogue only)
// ABL (tuning only, results invalid): bit 0 = no DMA inside the k-loop, bit 1 = no fragment reads inside the k-loop, bit 2 =
// every DMA instruction fetches 8 rows x 128 B (whole cache lines, same volume) instead of 16 rows x 64 B
// FMT: 0 = bf16x3 operands (three planes, six products), 1 = f16x2 operands (two planes, three products, second accumulator)
template <int EPI, int CFG, int ABL = 0, int FMT = 0>
__global__ __launch_bounds__(512, 2) void vn_gemm_x3_kernel(vn_gemm_args p, int tiles_m, int tiles_n) {
    constexpr int NP = FMT ? 2 : 3;
    using G = x3_geo<CFG, NP>;
    constexpr int RI = G::RI, CJ = G::CJ;
    // FMT = 1: ABL bit 2 = DMA never waited for
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave % (G::WR * G::WC)) / G::WC, wn = wave % G::WC;      // CFG 4: waves w and w + 4 own the same 96 x 32 sub-tile
    const int grp = wave >> 2;                        // waves w and w + 4 share a SIMD: one of each group per SIMD
    const uint16_t* A16 = (const uint16_t*)p.A;
    const uint16_t* W16 = (const uint16_t*)p.W;
    const int nk_all = p.K / X3_KT;
    const int drow = (ABL & 4) && !FMT ? lane >> 3 : lane >> 2;
    const int dslot = (ABL & 4) && !FMT ? lane & 7 : (lane & 3) ^ ((lane >> 4) & 3);     // (row >> 2) & 3 == (lane >> 4) & 3
    // fragment offsets (floats) inside a plane tile: row * 16 + ((2 s + h) ^ ((row >> 2) & 3)) * 4
    const int l31 = lane & 31, h = lane >> 5, sw = (lane >> 2) & 3;
    const int aRow = (wm * 32 * RI + l31) * 16;
    const int bRow = (wn * 32 * CJ + l31) * 16;

    // ---- this block's tile (and, for split-K launches, its k-range)
    if (gridDim.y > 1) p.C += (size_t)blockIdx.y * p.M * p.N;
    {
        const int t = x3_xcd_remap(blockIdx.x, gridDim.x);
        const int kb = (int)((long)nk_all * blockIdx.y / gridDim.y), ke = (int)((long)nk_all * (blockIdx.y + 1) / gridDim.y);
        int tm, tn;
        x3_tile_coords(t, tiles_m, tiles_n, tm, tn, p.group_m > 0 ? p.group_m : 8);
        const int m0 = tm * G::BM, n0 = tn * X3_BN;

        // per-lane DMA sources: instruction q fills 16 rows of one plane tile at LDS offset q KiB of the stage (A planes first: q <
        // 3 BM / 16, plane q / (BM / 16)).  CFG 1 / 2: a wave owns NPW consecutive instructions; CFG 3 (60 instructions): instruction
        // 8 j + wave, so waves 0-3 issue eight and waves 4-7 seven
        auto piece_q = [&](int j) { return (CFG == 3 || CFG == 4) ? 8 * j + wave : wave * G::NPW + j; };
        constexpr int NA_PIECES = NP * (G::BM / 16);                // DMA instructions of a stage that fetch A
        constexpr bool CONV = EPI == VN_EPI_CONV;
        constexpr bool CONVT = EPI == VN_EPI_CONVT;         // the gather on the W side (rows of the W operand = output positions)
        constexpr bool TN = (ABL & X3_MODE_TN) != 0;        // both operands token-major (the dW GEMMs of training): see X3_MODE_TN
        const uint16_t* src[G::NPW];
        int kadv[G::NPW];                                   // elements per k-tile: 32 along a planar row, 3 x 512 between tiled pieces
        int t0v[(CONV || CONVT) ? G::NPW : 1];              // CONV / CONVT: input row of tap 0 for this lane's gathered row (may be < 0 / >= T_in)
#pragma unroll
        for (int j = 0; j < G::NPW; ++j) {
            const int q = piece_q(j);
            kadv[j] = X3_KT;
            if constexpr (TN) {
                // piece q = (plane pt, 32-row block r >> 1 of the tile, k-step r & 1 = q & 1): the contiguous 1 KiB [16 tokens][32 rows] of the
                // token-major tiled planes, copied lane-linearly (16 bytes per lane)
                int pt, r, fb, nfb;
                const uint16_t* base;
                if (q < NA_PIECES) { pt = q / (G::BM / 16); r = q % (G::BM / 16); fb = (m0 >> 5) + (r >> 1); nfb = p.M >> 5; base = A16; }
                else { const int qb = (q - NA_PIECES) % (8 * NP); pt = qb >> 3; r = qb & 7; fb = (n0 >> 5) + (r >> 1); nfb = p.N >> 5; base = W16; }
                fb = fb < nfb ? fb : nfb - 1;
                src[j] = base + (((size_t)(2 * kb + (r & 1)) * nfb + fb) * NP + pt) * 512 + lane * 8;
                kadv[j] = 2 * nfb * NP * 512;
            } else
            if (q < NA_PIECES) {
                const int pt = q / (G::BM / 16), row = (q % (G::BM / 16)) * 16 + drow;
                int g = m0 + row;
                g = g < p.M ? g : p.M - 1;
                if constexpr (CONV) {
                    // implicit GEMM: row g = (b, t') starts at input row t0 = t' in_stride - pad; tap j / channel block c0 of k-tile kt add
                    // the UNIFORM offset (j dil C_in + c0) — only whether that row exists differs per lane (stage_piece)
                    const int b = g / p.conv_trows, tq = g - b * p.conv_trows;
                    t0v[j] = tq * p.conv_in_stride - p.conv_pad;
                    src[j] = A16 + (size_t)pt * p.a_plane + ((long)b * p.conv_tin + t0v[j]) * (long)p.conv_cin + dslot * 8;
                } else if (vn_planes_tiled(p.a_plane) && !((ABL & 4) && !FMT)) {
                    src[j] = A16 + (((size_t)(g >> 4) * nk_all + kb) * NP + pt) * 512 + (g & 15) * 32 + dslot * 8;
                    kadv[j] = NP * 512;
                } else {
                    src[j] = A16 + (size_t)pt * p.a_plane + (size_t)g * p.K + dslot * 8 + kb * X3_KT;
                }
            } else {
                const int qb = (q - NA_PIECES) % (8 * NP);              // (CFG 3, NP 3, j = 7, waves 4-7: q >= NQ — never issued)
                const int pt = qb >> 3, row = (qb & 7) * 16 + drow;
                int g = n0 + row;
                g = g < p.N ? g : p.N - 1;
                if constexpr (CONVT) {
                    // W row g = output position (b, t'): the activation planes [B T_in][C_in] (planar, w_plane apart), gathered per tap
                    const int b = g / p.conv_trows, tq = g - b * p.conv_trows;
                    t0v[j] = tq * p.conv_in_stride - p.conv_pad;
                    src[j] = W16 + (size_t)pt * p.w_plane + ((long)b * p.conv_tin + t0v[j]) * (long)p.conv_cin + dslot * 8;
                } else
                if (p.w_tiled && !((ABL & 4) && !FMT)) {          // piece = the contiguous 1 KiB of (row block g / 16, k-tile, plane pt)
                    src[j] = W16 + (((size_t)(g >> 4) * nk_all + kb) * NP + pt) * 512 + (g & 15) * 32 + dslot * 8;
                    kadv[j] = NP * 512;
                } else {
                    src[j] = W16 + (size_t)pt * p.w_plane + (size_t)g * p.K + dslot * 8 + kb * X3_KT;
                }
            }
        }
        // CONV: (tap, channel block) of the k-tile being staged, kept incrementally (k-tiles are staged in increasing order; a jump
        // falls back to the division) — all uniform
        int cv_kt = -1, cv_tap = 0, cv_c0 = 0, cv_dt = 0;
        long cv_off = 0;
        auto conv_tile = [&](int kt) {
            if (kt == cv_kt) return;
            if (kt == cv_kt + 1 && cv_kt >= 0) {
                cv_c0 += X3_KT;
                if (cv_c0 == p.conv_cin) { cv_c0 = 0; ++cv_tap; }
            } else {
                const int cpt = p.conv_cin / X3_KT;
                cv_tap = kt / cpt;
                cv_c0 = (kt - cv_tap * cpt) * X3_KT;
            }
            cv_kt = kt;
            cv_dt = cv_tap * p.conv_dil;
            cv_off = (long)cv_dt * p.conv_cin + cv_c0;
        };
        auto stage_piece = [&](int buf, int k0, int j) {       // k0 = 32 x the k-tile index relative to kb
            if constexpr (G::NQ % 8 != 0) {
                if (j == G::NPW - 1 && wave >= G::NQ % 8) return;   // CFG 3, NP 3: 60 = 4 x 8 + 4 x 7 instructions
            }
            float* base = lds + buf * G::STAGE + piece_q(j) * 256;
            const uint16_t* from;
            if ((CONV && piece_q(j) < NA_PIECES) || (CONVT && piece_q(j) >= NA_PIECES)) {    // a gathered piece of the implicit GEMM: the tap's row, or zeros outside the signal
                conv_tile(kb + k0 / X3_KT);
                const bool ok = (unsigned)(t0v[(CONV || CONVT) ? j : 0] + cv_dt) < (unsigned)p.conv_tin;
                from = ok ? src[j] + cv_off : p.zeros16 + dslot * 8;
            } else if constexpr (TN) {
                // token block 2 (kb + k-tile) + k-step of the piece; blocks past the last token come from the zero page
                const bool ok = 2 * (kb + k0 / X3_KT) + (piece_q(j) & 1) < p.tn_blocks;
                from = ok ? src[j] + (size_t)(k0 / X3_KT) * kadv[j] : p.zeros16 + lane * 8;
            } else {
                if constexpr ((ABL & 4) && !FMT) k0 = (2 * k0) % p.K;     // a fresh 128-byte line per k-tile (probe: K % 64 == 0, data-parallel form)
                else k0 = (k0 / X3_KT) * kadv[j];
                from = src[j] + k0;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)from,
                                             (__attribute__((address_space(3))) void*)base, 16, 0, 0);
        };
        auto stage = [&](int buf, int k0) {
#pragma unroll
            for (int j = 0; j < G::NPW; ++j) stage_piece(buf, k0, j);
        };

        // Folded RMSNorm, consumer side: thread r < BM fetches the K / 128 group sums of squares of tile row r BEFORE the first DMA is
        // issued (VMEM returns in order, so they have landed when the first stage has) and turns them into the row's scale right
        // after the prologue's barrier; the table sits behind stages and images (x3_geo::RS_OFF) until the epilogue reads it.  Left to
        // the epilogue these were K / 128 dependent L2 round trips per tile with nothing to hide them (measured: the whole gain of
        // the fold).  Up to 16 groups (K <= 2048) live in registers; zeros pad the sum (s + 0 == s).
        constexpr bool FOLD_IN = EPI == VN_EPI_QKV3 || EPI == VN_EPI_QKV || EPI == VN_EPI_GEGLU || EPI == VN_EPI_BIAS;
        constexpr int FOLD_NT = 16;
        float fold_sq[FOLD_IN ? FOLD_NT : 1];
        const bool fold_in = FOLD_IN && p.ssq_in != nullptr;
        if constexpr (FOLD_IN) {
            if (fold_in && tid < G::BM) {
                const int nt = p.K >> 7;
                const int row = m0 + tid < p.M ? m0 + tid : p.M - 1;
                const float* q = p.ssq_in + row;                      // [group][row]: the lanes of a load read consecutive rows
#pragma unroll
                for (int t = 0; t < FOLD_NT; ++t) fold_sq[t] = t < nt ? q[(size_t)t * p.M] : 0.0f;
            }
        }
        auto fold_table = [&]() {
            if constexpr (FOLD_IN) {
                if (fold_in && tid < G::BM) {
                    float s2 = fold_sq[0];
#pragma unroll
                    for (int t = 1; t < FOLD_NT; ++t) s2 += fold_sq[t];
                    lds[G::RS_OFF + tid] = 1.0f / sqrtf(s2 / (float)p.K + p.fold_eps);       // exact div + sqrt, as vn_rmsnorm_row
                }
            }
        };
        // f16x2: acc takes a0 b0, acc_lo takes a0 b1 + a1 b0 (both second planes carry 2^11) and joins acc times 2^-11 at the end
        f32x16 acc[RI][CJ], acc_lo[FMT ? RI : 1][FMT ? CJ : 1];
#pragma unroll
        for (int i = 0; i < RI; ++i)
#pragma unroll
            for (int j = 0; j < CJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[i][j][r] = 0.0f;
                    if constexpr (FMT) acc_lo[i][j][r] = 0.0f;
                }

        struct Frags { f32x4 a[NP][RI], b[NP][CJ]; };
        auto load_frags = [&](Frags& f, int buf, int s) {
            const float* sA = lds + buf * G::STAGE;
            const float* sB = sA + NP * G::APLANE;
            if constexpr (TN) {
                // a piece is [16 k][32 rows] with 64-byte k-rows: the transposing read hands lane l the k's 8 (l >> 5) + 4 c .. + 3 (c = 0, 1:
                // two reads 256 bytes apart) of row l & 31 — exactly the MFMA fragment (scripts/ubench/tr_read_probe.hip,
                // profiles/r06_tr_read_probe.txt); each 16-lane group reads [4 k][16 rows], lane (l & 15) supplying k-row (l & 15) >> 2,
                // rows 4 (l & 3) .. + 3 of its half
                const int trl = (8 * h + ((lane & 15) >> 2)) * 16 + ((lane >> 4) & 1) * 8 + (lane & 3) * 2;     // floats
#pragma unroll
                for (int q = 0; q < NP; ++q) {
#pragma unroll
                    for (int i = 0; i < RI; ++i) f.a[q][i] = x3_tr_frag(sA + q * G::APLANE + ((wm * RI + i) * 2 + s) * 256 + trl);
#pragma unroll
                    for (int j = 0; j < CJ; ++j) f.b[q][j] = x3_tr_frag(sB + q * X3_BPLANE + ((wn * CJ + j) * 2 + s) * 256 + trl);
                }
                return;
            }
            const int off = ((2 * s + h) ^ sw) * 4;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
#pragma unroll
                for (int i = 0; i < RI; ++i) f.a[q][i] = *(const f32x4*)(sA + q * G::APLANE + aRow + i * 32 * 16 + off);
#pragma unroll
                for (int j = 0; j < CJ; ++j) f.b[q][j] = *(const f32x4*)(sB + q * X3_BPLANE + bRow + j * 32 * 16 + off);
            }
        };
        // the six plane products of one 16-wide k-step, smallest terms first (t = 0: A0 W2, then A2 W0, A1 W1, A0 W1, A1 W0,
        // A0 W0); consecutive MFMAs go to different accumulators
        // f16x2: t = 0: A1 W0, t = 1: A0 W1 (-> acc_lo), t = 2: A0 W0 (-> acc)
        auto mac_prod = [&](const Frags& f, int t) {
            if constexpr (FMT) {
                const int qa = t == 0 ? 1 : 0, qb = t == 1 ? 1 : 0;
#pragma unroll
                for (int i = 0; i < RI; ++i)
#pragma unroll
                    for (int j = 0; j < CJ; ++j) {
                        const f16x8 av = __builtin_bit_cast(f16x8, f.a[qa][i]), bv = __builtin_bit_cast(f16x8, f.b[qb][j]);
                        if (t == 2) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[i][j], 0, 0, 0);
                        else acc_lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc_lo[i][j], 0, 0, 0);
                    }
            } else {
                // CONVT swaps the operands' roles (A = weights, W = activations): the plane PAIRS are taken in the mirrored order, so that
                // every output element sees the same products in the same sequence as in the CONV form — bitwise the same sums
                const int qa0 = t == 1 ? 2 : (t == 2 || t == 4) ? 1 : 0, qb0 = t == 0 ? 2 : (t == 2 || t == 3) ? 1 : 0;
                const int qa = EPI == VN_EPI_CONVT ? qb0 : qa0, qb = EPI == VN_EPI_CONVT ? qa0 : qb0;
#pragma unroll
                for (int i = 0; i < RI; ++i)
#pragma unroll
                    for (int j = 0; j < CJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.a[qa][i]), __builtin_bit_cast(bf16x8, f.b[qb][j]),
                                                                            acc[i][j], 0, 0, 0);
            }
        };
        Frags f;
        auto compute = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int t = 0; t < G::NPROD; ++t) mac_prod(f, t);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        };

        const int nk = ke - kb;
        if constexpr (FMT) {
            // f16x2: ONE phase per k-tile and group.  A k-step has half the matrix work of bf16x3's (three products), so the bf16x3
            // schedules' phases of one k-step (12 / 9 MFMAs) are too short against their barrier and LDS latency; here a load phase
            // reads the fragments of BOTH k-steps of a tile (16 KiB per wave) and a compute phase issues both steps' products
            // (2 x 3 x RI x CJ MFMAs): two barriers per k-tile instead of four.  Three buffers, tile kt in buffer kt % 3: phase
            // 2 kt = group 0 loads tile kt, 2 kt + 1 = group 0 computes it while group 1 loads it, 2 kt + 2 = group 1 computes it.
            // Tile kt + 2 goes into the buffer of tile kt - 1 (last read in phase 2 kt - 1) and is first read in phase 2 kt + 4:
            // every wave issues its pieces after the fragment reads of its own load phase and waits for them at the end of its NEXT
            // load phase with the pieces of tile kt + 3 in flight (counted vmcnt) — two full phases between issue and wait.  (Issuing
            // group 0's pieces between the MFMAs of its compute phase instead measured the same: profiles/history/r03_gemm_f16x2_ablation.txt.)
            Frags f1;
            stage(0, 0);
            if (nk > 1) stage(1, X3_KT);
            if (nk > 1) X3_VMCNT(G::NPW); else X3_VMCNT(0);
            X3_BARRIER();                                   // tile 0 complete
            fold_table();
            if (grp) X3_BARRIER();                          // group 1 runs one phase behind
            if constexpr (ABL & 2) { load_frags(f, 0, 0); load_frags(f1, 0, 1); }
            int b = 0;
            for (int kt = 0; kt < nk; ++kt) {
                const int b2 = b == 0 ? 2 : b - 1;          // buffer of tile kt + 2
                const bool more = (kt + 2 < nk) && !(ABL & 1);
                const bool last = kt + 1 == nk;
                // ---- load phase
                if constexpr (!(ABL & 2)) {
                    load_frags(f, b, 0);
                    load_frags(f1, b, 1);
                }
                if (more) {
                    stage(b2, (kt + 2) * X3_KT);
                    if constexpr (!(ABL & 4)) X3_VMCNT(G::NPW);         // tile kt + 1 landed (this wave's pieces); tile kt + 2 in flight
                } else {
                    if constexpr (!(ABL & 4)) X3_VMCNT(0);
                }
                X3_LGKM0();
                X3_BARRIER();
                // ---- compute phase
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    mac_prod(t < 3 ? f : f1, t < 3 ? t : t - 3);
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_s_setprio(0);
                if (!(last && grp)) X3_BARRIER();           // group 1's last compute phase has no partner phase
                b = b == 2 ? 0 : b + 1;
            }
        } else if constexpr (CFG == 4) {
            // 96 x 128 tile, the K SPLIT INSIDE THE BLOCK: wave group g (waves 4 g .. 4 g + 3, a 96 x 32 sub-tile each) owns k-step g of
            // every k-tile, so a k-tile is ONE load phase and ONE compute phase (18 MFMAs) per group, the groups one phase apart as in the
            // other schedules (two barriers per k-tile instead of four), and the two partial sums meet in the epilogue's LDS images
            // (x3_epilogue_image).  What it buys: M = 575 / 576 rows are 6 x 96 exactly (4.5 x 128: a tenth of the matrix work of the
            // 128-row tiles multiplies padding), the one-sequence launches become 180 / 240 tiles of 0.75 of the k-loop of a 128-row
            // tile, and 12 MFMA tiles of 32 x 32 do not divide over eight waves any other way.
            // Three buffers, tile kt in buffer kt % 3, interval I_j between consecutive barriers: group 0 reads tile kt in I_2kt,
            // group 1 in I_2kt+1 (both end with lgkmcnt(0) before their barrier).  Tile kt + 2 goes into the buffer of tile kt - 1 (last
            // read in I_2kt-1): every wave issues its pieces in its own load phase of tile kt (I_2kt / I_2kt+1) and waits for them at
            // the end of its load phase of tile kt + 1 with the pieces of tile kt + 3 in flight (counted vmcnt: waves 0-1 issue six
            // pieces per stage, the others five) — one barrier later, at the earliest, group 0 reads the tile in I_2kt+4.
            const bool six = wave < G::NQ % 8;
            auto wait_for_all_but_one_stage = [&]() {
                if (six) X3_VMCNT(G::NPW);
                else X3_VMCNT(G::NPW - 1);
            };
            stage(0, 0);
            if (nk > 1) { stage(1, X3_KT); wait_for_all_but_one_stage(); }
            else X3_VMCNT(0);
            X3_BARRIER();                                   // tile 0 complete
            fold_table();
            if (grp) X3_BARRIER();                          // group 1 runs one phase behind
            int b = 0;
            for (int kt = 0; kt < nk; ++kt) {
                const int b2 = b == 0 ? 2 : b - 1;          // buffer of tile kt + 2 (= the one tile kt - 1 used)
                const bool more = kt + 2 < nk;
                const bool last = kt + 1 == nk;
                load_frags(f, b, grp);
                if (more) { stage(b2, (kt + 2) * X3_KT); wait_for_all_but_one_stage(); }   // tile kt + 1 landed (this wave's pieces)
                else X3_VMCNT(0);
                X3_LGKM0();
                X3_BARRIER();
                compute();
                if (!(last && grp)) X3_BARRIER();           // group 1's last compute phase has no partner phase
                b = b == 2 ? 0 : b + 1;
            }
        } else if constexpr (CFG == 1) {
            // three buffers; tile kt lives in buffer kt % 3.  Phase intervals I_j between consecutive barriers: group 0 loads
            // step i in I_2i and computes it in I_2i+1, group 1 one interval later.  Tile kt is read in I_4kt .. I_4kt+3 (every
            // load phase ends with lgkmcnt(0) BEFORE its barrier, so those reads have retired when I_4kt+4 starts), tile
            // kt + 3 may therefore be DMA'd into its buffer from I_4kt+4 on: it is issued in the load phases of tile kt + 1
            // (I_4kt+4 .. I_4kt+7) and every wave waits for its pieces at the end of the second load phase of tile kt + 2
            // (vmcnt leaves only the six younger pieces of the following tile in flight) — two barriers before the first read.
            stage(0, 0);
            if (nk > 1) stage(1, X3_KT);
            if (nk > 1) X3_VMCNT(G::NPW); else X3_VMCNT(0);
            X3_BARRIER();                                   // tile 0 complete
            fold_table();
            if (grp) X3_BARRIER();                          // group 1 runs one phase behind
            if constexpr (ABL & 2) load_frags(f, 0, 0);
            int b = 0;
            for (int kt = 0; kt < nk; ++kt) {
                const int b2 = b == 0 ? 2 : b - 1;          // buffer of tile kt + 2 (= the one tile kt - 1 used)
                const bool more = (kt + 2 < nk) && !(ABL & 1);
                const bool last = kt + 1 == nk;
                // ---- k-step 0
                if constexpr (!(ABL & 2)) load_frags(f, b, 0);
                if (more) {
#pragma unroll
                    for (int j = 0; j < G::NPW / 2; ++j) stage_piece(b2, (kt + 2) * X3_KT, j);
                }
                X3_LGKM0();
                X3_BARRIER();
                compute();
                X3_BARRIER();
                // ---- k-step 1
                if constexpr (!(ABL & 2)) load_frags(f, b, 1);
                if (more) {
#pragma unroll
                    for (int j = G::NPW / 2; j < G::NPW; ++j) stage_piece(b2, (kt + 2) * X3_KT, j);
                    X3_VMCNT(G::NPW);                       // tile kt + 1 landed (this wave's pieces); tile kt + 2 in flight
                } else {
                    X3_VMCNT(0);
                }
                X3_LGKM0();
                X3_BARRIER();
                compute();
                if (!(last && grp)) X3_BARRIER();           // group 1's last compute phase has no partner phase
                b = b == 2 ? 0 : b + 1;
            }
        } else {
            // two buffers (72 / 60 KiB); tile kt lives in buffer kt & 1.  Tile kt + 1 goes into the buffer tile kt - 1 used (last
            // read in I_4kt-1, retired before that interval's barrier) and is first read in I_4kt+4: each wave issues NA of its
            // pieces in the first load phase of tile kt and the other four between the products of its first compute phase
            // (group 0: I_4kt, I_4kt+1; group 1: I_4kt+1, I_4kt+2) and waits for them at the last point that still has a
            // barrier between it and the first read: group 0 at the end of its second compute phase (I_4kt+3), group 1 at
            // the end of its second load phase (I_4kt+3).
            constexpr int NA = G::NPW - G::NI;              // 5 (CFG 2) / 4 (CFG 3); f16x2: 4 / 3
            stage(0, 0);
            X3_VMCNT(0);
            X3_BARRIER();
            fold_table();
            if (grp) X3_BARRIER();
            if constexpr (ABL & 2) load_frags(f, 0, 0);
            for (int kt = 0; kt < nk; ++kt) {
                const int b = kt & 1;
                const bool more = (kt + 1 < nk) && !(ABL & 1);
                const bool last = kt + 1 == nk;
                const int k1 = (kt + 1) * X3_KT;
                // ---- k-step 0
                if constexpr (!(ABL & 2)) load_frags(f, b, 0);
                if (more) {
#pragma unroll
                    for (int j = 0; j < NA; ++j) stage_piece(b ^ 1, k1, j);
                }
                X3_LGKM0();
                X3_BARRIER();
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int t = 0; t < G::NPROD; ++t) {
                    mac_prod(f, t);
                    if (t < G::NI && more) stage_piece(b ^ 1, k1, NA + t);
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_s_setprio(0);
                X3_BARRIER();
                // ---- k-step 1
                if constexpr (!(ABL & 2)) load_frags(f, b, 1);
                if (grp) X3_VMCNT(0);
                X3_LGKM0();
                X3_BARRIER();
                compute();
                if (!grp) X3_VMCNT(0);
                if (!(last && grp)) X3_BARRIER();
            }
        }

        if constexpr (FMT) {
#pragma unroll
            for (int i = 0; i < RI; ++i)
#pragma unroll
                for (int j = 0; j < CJ; ++j) acc[i][j] += acc_lo[i][j] * VN_H2_INV_SCALE;
        }
        if constexpr (EPI == VN_EPI_CONVT) {
            x3_epilogue_convt<CFG>(p, acc, n0, wave, lane, lds);
        } else if constexpr (CFG == 4 || (CFG == 3 && EPI == VN_EPI_GEGLU && !FMT)) {
            // (CFG 3 + GEGLU: the 96 x 32 wave tile holds value and gate columns in different waves — they meet in the tile image)
            x3_epilogue_image<EPI, CFG, FMT>(p, acc, m0, n0, wave, lane, lds);       // the launcher guarantees the staged forms' alignment
        } else if constexpr (EPI == VN_EPI_CONV) {
            x3_epilogue_staged<EPI, CFG, FMT>(p, acc, m0, n0, wave, lane, lds);      // the launcher guarantees the alignment it needs
        } else if constexpr (!(EPI == VN_EPI_GEGLU && x3_geo<CFG>::CJ != 2)) {
            if (p.staged) x3_epilogue_staged<EPI, CFG, FMT>(p, acc, m0, n0, wave, lane, lds);
            else x3_epilogue<EPI, CFG, FMT>(p, acc, m0, n0, wm, wn, lane);
        }
    }
}

#define X3_WS_FLOATS (32L << 20)          // 128 MiB of split-K partial images, allocated once (graph-safe: never re-allocated)

// tuning / test hook of ONE context: tile height 96 / 128 / 192 / 256 (0 = by shape; 96 = the k-split tile, bf16x3 operands), forced split-K (0 / 1 off, 2 / 4 forced, < 0 =
// cost model), ablation bits (tuning only: results invalid; < 0 or 0 = none)
extern "C" int vn_debug_x3_config(vn_ctx* ctx, int bm, int splitk, int abl) {
    if (!ctx) return VN_ERR_INVALID;
    if (bm != 0 && bm != 96 && bm != 128 && bm != 192 && bm != 256) return vn_fail(ctx, VN_ERR_INVALID, "x3 tile height %s%ld is not 0 / 96 / 128 / 192 / 256", "", bm);
    ctx->tune.x3_bm = bm;
    ctx->tune.x3_split = splitk < 0 ? -2 : splitk;
    ctx->tune.x3_abl = abl < 0 ? 0 : (abl & 7);
    ++ctx->tune.epoch;
    return VN_OK;
}

// the stages, or the largest image of a staged epilogue (the transposed V^T planes of QKV3: [3][128][RP + 8] bf16) followed by the folded
// norm's per-row scale table (BM <= 256 floats at X3_RS_OFF) if that is bigger
template <int CFG, int NP = 3>
static constexpr size_t x3_lds_bytes() {
    return ((size_t)x3_geo<CFG, NP>::RS_OFF + 256) * 4;
}

// may the epilogue go through LDS with 16-byte global accesses ?  (VN_X3_STAGED=0: never — A/B runs)
template <int EPI>
static int x3_staged_ok(const vn_ctx* ctx, const vn_gemm_args& a) {
    if (!ctx->tune.x3_staged) return 0;
    auto al = [](const void* p, uintptr_t m) { return ((uintptr_t)p & (m - 1)) == 0; };
    if (EPI == VN_EPI_GEGLU)
        return a.C16 && al(a.C16, 16) && !(a.N & 15) && (vn_planes_tiled(a.c_plane) ? !(a.ldc & 31) : (!(a.ldc & 7) && !(a.c_plane & 7)));
    if (EPI == VN_EPI_QKV3)
        return al(a.C16, 16) && al(a.V16, 16) && !(a.c_plane & 7) && !(a.v_plane & 7) && !(a.qkv_plane & 7) && !((a.H * VN_DHEAD) & 127);
    if (!al(a.C, 16) || (a.N & 3)) return 0;
    if (EPI == VN_EPI_QKV) return !(a.qkv_plane & 3);
    if (EPI == VN_EPI_BIAS && !al(a.bias, 16)) return 0;
    return !(a.ldc & 3);
}

template <int EPI, int CFG, int ABL = 0, int FMT = 0>
static int x3_go(vn_ctx* ctx, const vn_gemm_args& a_in, int nsplit, hipStream_t s) {
    vn_gemm_args a = a_in;
    a.staged = x3_staged_ok<EPI>(ctx, a);
    a.sat = ctx->sat;
    if ((a.ssq_in || a.ssq_out || a.X16) && !a.staged && EPI != VN_EPI_CONV)
        return vn_fail(ctx, VN_ERR_UNSUPPORTED, "gemm_x3: the folded-norm epilogues exist in the LDS-staged form only (alignment / VN_X3_STAGED)%s", "");
    a.group_m = ctx->tune.x3_group_m;                                  // tuning: rows of tiles per walk group (0 = 8)
    const int tiles_m = vn_cdiv(a.M, x3_geo<CFG>::BM), tiles_n = vn_cdiv(a.N, X3_BN);
    constexpr int NP = FMT ? 2 : 3;
    const size_t lds_bytes = x3_lds_bytes<CFG, NP>();
    hipLaunchKernelGGL((vn_gemm_x3_kernel<EPI, CFG, ABL, FMT>), dim3(tiles_m * tiles_n, nsplit), dim3(512), lds_bytes, s, a, tiles_m, tiles_n);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}
template <int EPI, int FMT = 0>
static int x3_go_bm(vn_ctx* ctx, const vn_gemm_args& a, int nsplit, int bm, hipStream_t s) {
    if constexpr (EPI != VN_EPI_GEGLU || !FMT) {
        if (bm == 192) return x3_go<EPI, 3, 0, FMT>(ctx, a, nsplit, s);      // (GEGLU: bf16x3 operands + plane output only, see x3_choose)
    }
    if constexpr (EPI != VN_EPI_CONV && !FMT) {
        if (bm == 96) return x3_go<EPI, 4, 0, FMT>(ctx, a, nsplit, s);       // x3_choose offers it only where x3_tile96_ok holds
    }
    return bm == 256 ? x3_go<EPI, 2, 0, FMT>(ctx, a, nsplit, s) : x3_go<EPI, 1, 0, FMT>(ctx, a, nsplit, s);
}

// test hook: 0 = keep the reduce pass of a split RESIDUAL GEMM and the RMSNorm that follows it as two kernels, 1 = fuse, -1 = the
// context's default (VN_X3_FUSE_NORM, on)
extern "C" int vn_debug_x3_fuse_norm(vn_ctx* ctx, int on) {
    if (!ctx) return VN_ERR_INVALID;
    if (on < 0) { vn_tune t; vn_tune_init(&t); ctx->tune.x3_fuse_norm = t.x3_fuse_norm; }
    else ctx->tune.x3_fuse_norm = on != 0;
    ++ctx->tune.epoch;
    return VN_OK;
}
static bool x3_norm_fusable(const vn_ctx* ctx, const vn_gemm_args& a) {
    return ctx->tune.x3_fuse_norm && a.norm_w && a.norm_done && a.ldc == a.N && (a.N == 1280 || a.N == 256);
}
// folded norm: the reduce pass of a split RESIDUAL launch writes the planes and the group sums of squares (vn_launch_rowprep)
static bool x3_fold_out(const vn_gemm_args& a) { return a.X16 != nullptr; }

// Tile height and k-split of a launch, by a cost model in microseconds calibrated on the model's shapes (scripts/gemm_x3_plan_sweep.py,
// profiles/history/r02_gemm_x3_plan_sweep.txt): a launch runs ceil(tiles ns / CUs) rounds of K / ns k-tiles; a k-tile of a 128-row tile
// costs 1.45 us when every CU is busy, taller tiles proportionally more minus what their smaller operand traffic per flop gives
// back (192 rows: -17 % DMA bytes per flop, 256 rows: -25 %); a split launch adds its reduce pass over (ns + 1 or 2) images of C
// at ~3.5 TB/s, minus the RMSNorm read it absorbs when the norm is fused into that pass.  The 192-row tile is what fills whole
// rounds at B = 8 (M = 4600): QKV 720 tiles = 2.8 rounds (128 rows: 4.2 -> 5), Wo / W2 240 tiles = ONE round without a split,
// classifier 768 = 3.0 rounds; 256 rows stay best for W1 + GEGLU (720 tiles = 2.8 rounds; the GEGLU epilogue needs the 64-wide
// wave tile); one or two sequences keep 128 rows (more tiles) and split the N = 1280 projections.
struct x3_plan { int bm, ns; };
// may a launch use the 96-row k-split tile (CFG 4) ?  bf16x3 operands, no convolution, and the alignment of the LDS-staged epilogues
// (its image epilogue issues the same 16-byte global accesses); VN_X3_TILE96=0 / vn_tune::x3_tile96 switches it off (A/B runs)
template <int EPI, int FMT>
static bool x3_tile96_ok(const vn_ctx* ctx, const vn_gemm_args& a) {
    if constexpr (FMT || EPI == VN_EPI_CONV) return false;
    else return ctx->tune.x3_tile96 && x3_staged_ok<EPI>(ctx, a) != 0;
}
template <int EPI>
static x3_plan x3_choose(const vn_ctx* ctx, const vn_gemm_args& a, int cus, double kt_us = 1.45, double partial = 1.0, bool allow96 = false,
                          double rel128 = 1.0) {
    int bm_forced = ctx->tune.x3_bm;                         // 0 = by shape
    if (bm_forced == 96 && !allow96) bm_forced = 128;
    const int split_forced = ctx->tune.x3_split == -2 ? -1 : ctx->tune.x3_split;      // 0 / 1 off, 2 / 4 forced, -1 cost model
    constexpr bool can_split = EPI == VN_EPI_STORE || EPI == VN_EPI_RESIDUAL;
    constexpr bool residual = EPI == VN_EPI_RESIDUAL;
    const int nk = a.K / X3_KT;
    x3_plan best{128, 1};
    double best_cost = 1e300;
    static const int heights[4] = {128, 192, 256, 96};
    // 96 rows (k-split inside the block): 0.75 of a 128-row tile's matrix work per k-tile, + 17 % DMA bytes per flop; its two-image
    // epilogue adds ~1 us per launch (fixed[])
    static const double rel[4] = {1.0, 1.5 * 0.97, 2.0 * 0.95, 0.75 * 1.03};
    static const double fixed[4] = {0.0, 0.0, 0.0, 1.0};
    for (int hi = 0; hi < 4; ++hi) {
        const int bm = heights[hi];
        // GEGLU on the 192-row tile goes through the tile-image epilogue (value and gate columns sit in different waves): the same
        // conditions as the 96-row tile (bf16x3 operands, staged alignment = plane output)
        const bool geglu192 = EPI == VN_EPI_GEGLU && allow96;
        if (bm == 192 && EPI == VN_EPI_GEGLU && !geglu192) continue;
        if (bm == 96 && !allow96) continue;
        if (bm_forced && bm != bm_forced && !(bm_forced == 192 && EPI == VN_EPI_GEGLU && !geglu192 && bm == 128)) continue;
        const long tiles = (long)vn_cdiv(a.M, bm) * vn_cdiv(a.N, X3_BN);
        for (int ns = 1; ns <= (can_split ? 4 : 1); ns *= 2) {
            if (ns > 1) {
                if (split_forced == 0 || split_forced == 1 || (a.N & 3) || (a.ldc & 3)) continue;
                if (x3_fold_out(a) && !(a.ldc == a.N && (a.N == 1280 || a.N == 256))) continue;       // vn_launch_rowprep's widths
                if (nk / ns < 8 || (double)ns * a.M * a.N > (double)X3_WS_FLOATS) continue;
            }
            if (split_forced > 1 && ns != split_forced && ns != 1) continue;
            // a ONE-ROUND launch that leaves CUs idle runs its k-tiles faster — measured on this kernel (profiles/r05_gemm_underfill_ab.txt,
            // relative to 240 of 256 CUs busy): 0.72 at 30 tiles, 0.80 at 120, 0.83 at 150, 0.86 at 180, 0.96 at 210.  Piecewise linear
            // in the fill; the f16x2 calibration keeps its own single factor (`partial` < 1)
            const long blocks = tiles * ns, rounds = (blocks + cus - 1) / cus;
            // (the 96-row tile moves 17 % more operand bytes per flop: with every CU busy — several rounds — a k-tile costs 0.85 of a
            // 128-row tile's, measured at M = 4600 (profiles/r05_gemm_tile96_check_and_sweep.txt: 243 vs 239 us for 6 vs 5 rounds); the
            // 0.77 of the table is the one-round figure)
            const double relh = (bm == 96 && rounds > 1) ? 0.85 : bm == 128 ? rel[hi] * rel128 : rel[hi];
            double cost = (double)rounds * (nk / (double)ns) * kt_us * relh + fixed[hi];
            if (blocks < cus) {
                const double fill = (double)blocks / cus;
                const double under = fill <= 0.6 ? 0.70 + 0.20 * fill : 0.82 + 0.53 * (fill - 0.6);
                cost *= partial < 1.0 ? partial : (under < 1.0 ? under : 1.0);
            }
            if (ns > 1) {
                cost += (ns + (residual ? 2 : 1)) * 4.0 * a.M * (double)a.N / 3.5e6;
                if (residual && (x3_norm_fusable(ctx, a) || x3_fold_out(a))) cost -= 4.0 * a.M * (double)a.N / 3.5e6 + 1.5;
                if (split_forced > 1 && ns == split_forced) cost = -1.0 + 1e-3 * hi;       // forced split: keep the height order
            }
            if (cost < best_cost) { best_cost = cost; best = x3_plan{bm, ns}; }
        }
    }
    return best;
}

template <int EPI, int FMT = 0>
static int x3_launch(vn_ctx* ctx, const vn_gemm_args& a, hipStream_t s) {
    const double n_out = (EPI == VN_EPI_GEGLU) ? a.N / 2 : a.N;
    const double bytes = (FMT ? 4.0 : 6.0) * ((double)a.M * a.K + (double)a.N * a.K) +
                         ((EPI == VN_EPI_GEGLU && a.C16) ? (FMT ? 4.0 : 6.0) : 4.0) * (double)a.M * n_out * (EPI == VN_EPI_RESIDUAL ? 2 : 1);
    // algorithmic (fp32-equivalent) flops.  The codec's convolutions are booked by what BOUNDS them (vn_conv_class): their own operand
    // bytes — the activation planes are read once, not once per tap — against the ridge of this pipe (2500 / 6 TF over 8 TB/s)
    int cls = 0;
    double pbytes = bytes;
    if constexpr (EPI == VN_EPI_CONV) {
        const double rows_in = (double)(a.M / (a.conv_trows > 0 ? a.conv_trows : 1)) * a.conv_tin;
        pbytes = (FMT ? 4.0 : 6.0) * (rows_in * a.conv_cin + (double)a.N * a.K) +
                 (double)a.M * a.N * ((a.C ? 4.0 : 0.0) + (a.Y2 ? 4.0 : 0.0) + (a.C16 ? (FMT ? 4.0 : 6.0) : 0.0) + (a.resid ? 4.0 : 0.0));
        cls = vn_conv_class(2.0 * a.M * (double)a.N * a.K, pbytes, VN_PROF_CONV_X3, 2500.0 / (FMT ? 3.0 : 6.0));
    }
    const int pi = vn_prof_pre(ctx, cls, 2.0 * a.M * (double)a.N * a.K, s, pbytes);
    int rc = VN_OK;
    // f16x2: half the matrix work per k-tile (profiles/history/r03_gemm_f16x2_plan_sweep.txt)
    const x3_plan plan = FMT ? x3_choose<EPI>(ctx, a, vn_num_cus(ctx), 0.85, 0.8)
                             : x3_choose<EPI>(ctx, a, vn_num_cus(ctx), 1.45, 1.0, x3_tile96_ok<EPI, FMT>(ctx, a));
    const int bm = plan.bm;
    bool done = false;
    if constexpr (EPI == VN_EPI_STORE) {
        int abl = ctx->tune.x3_abl;                                        // ablations (tuning only; results invalid): vn_debug_x3_config
        if (abl == 4 && !FMT && (a.a_plane == VN_PLANES_TILED || a.w_tiled)) abl = 0;     // the full-line probe re-addresses PLANAR bf16x3 planes only
        if (abl) {
            const bool big = bm == 256;
            if (bm == 192) {
                if (abl == 1) rc = x3_go<VN_EPI_STORE, 3, 1, FMT>(ctx, a, 1, s);
                else if (abl == 2) rc = x3_go<VN_EPI_STORE, 3, 2, FMT>(ctx, a, 1, s);
                else if (abl == 3) rc = x3_go<VN_EPI_STORE, 3, 3, FMT>(ctx, a, 1, s);
                else rc = x3_go<VN_EPI_STORE, 3, 4, FMT>(ctx, a, 1, s);
            }
            else if (abl == 1) rc = big ? x3_go<VN_EPI_STORE, 2, 1, FMT>(ctx, a, 1, s) : x3_go<VN_EPI_STORE, 1, 1, FMT>(ctx, a, 1, s);
            else if (abl == 2) rc = big ? x3_go<VN_EPI_STORE, 2, 2, FMT>(ctx, a, 1, s) : x3_go<VN_EPI_STORE, 1, 2, FMT>(ctx, a, 1, s);
            else if (abl == 3) rc = big ? x3_go<VN_EPI_STORE, 2, 3, FMT>(ctx, a, 1, s) : x3_go<VN_EPI_STORE, 1, 3, FMT>(ctx, a, 1, s);
            else rc = big ? x3_go<VN_EPI_STORE, 2, 4, FMT>(ctx, a, 1, s) : x3_go<VN_EPI_STORE, 1, 4, FMT>(ctx, a, 1, s);
            done = true;
        }
    }
    if constexpr (EPI == VN_EPI_STORE || EPI == VN_EPI_RESIDUAL) {
        const int ns = done ? 1 : plan.ns;
        if (ns > 1) {
            if (!ctx->x3_ws) VN_HIP_CHECK(ctx, vn_dev_malloc((void**)&ctx->x3_ws, (size_t)X3_WS_FLOATS * sizeof(float)));
            vn_gemm_args q = a;                       // every split stores its raw image [M][N] into the workspace
            q.C = ctx->x3_ws;
            q.ldc = a.N;
            q.X16 = nullptr; q.ssq_out = nullptr; q.ssq_in = nullptr;
            rc = x3_go_bm<VN_EPI_STORE, FMT>(ctx, q, ns, bm, s);
            if (EPI == VN_EPI_RESIDUAL && x3_fold_out(a)) {
                // folded norm: reduce + the planes and group sums of the new residual rows in one pass (part of THIS GEMM's work)
                if (rc == VN_OK) rc = vn_launch_rowprep(ctx, ctx->x3_ws, ns, a.C, a.X16, a.x16_plane, a.ssq_out, a.M, a.N, s, !FMT && a.x16_only);
            } else
            // while launches are being event-bracketed (pi >= 0) the two-kernel form runs, so that the GEMM's bracket holds the GEMM's
            // own work (split images + reduce) and nothing of the norm — the bitwise same result (tests/test_gpu_kernels.py)
            if (EPI == VN_EPI_RESIDUAL && x3_norm_fusable(ctx, a) && pi < 0) {
                if (rc == VN_OK)
                    rc = vn_launch_splitk_reduce_rmsnorm(ctx, ctx->x3_ws, ns, a.C, a.norm_w, a.norm_y, a.norm_y16, a.norm_plane, a.M, a.N,
                                                         a.norm_eps, s);
                if (rc == VN_OK) *a.norm_done = 1;
            } else if (rc == VN_OK) {
                rc = vn_launch_splitk_reduce(ctx, ctx->x3_ws, ns, a.C, a.M, a.N, a.ldc, EPI == VN_EPI_RESIDUAL, s);
            }
            done = true;
        }
    }
    if (!done) rc = x3_go_bm<EPI, FMT>(ctx, a, 1, bm, s);
    vn_prof_post(ctx, pi, s);
    return rc;
}

// TN operand mode (X3_MODE_TN): the planner's tile height and k-split, the token-major kernels
static int x3_launch_tn(vn_ctx* ctx, const vn_gemm_args& a_in, hipStream_t s) {
    vn_gemm_args a = a_in;
    if (!ctx->zero_page) {
        VN_HIP_CHECK(ctx, vn_dev_malloc((void**)&ctx->zero_page, 1024));
        VN_HIP_CHECK(ctx, hipMemset(ctx->zero_page, 0, 1024));
    }
    a.zeros16 = (const uint16_t*)ctx->zero_page;
    const double bytes = 6.0 * ((double)a.M * a.K + (double)a.N * a.K) + 4.0 * (double)a.M * a.N;
    const int pi = vn_prof_pre(ctx, 0, 2.0 * a.M * (double)a.N * a.K, s, bytes);
    // (the 128-row tile's short phases — 12 MFMAs per k-step — hide the 18 transposing reads of a step worse than the 9 wide ones of the NT
    // form: + 18 % per launch where the taller tiles measure the same as NT, profiles/r06_train_tn_vs_transposes.txt)
    const x3_plan plan = x3_choose<VN_EPI_STORE>(ctx, a, vn_num_cus(ctx), 1.45, 1.0, false, 1.18);
    auto go = [&](const vn_gemm_args& q, int ns) {
        return plan.bm == 192 ? x3_go<VN_EPI_STORE, 3, X3_MODE_TN>(ctx, q, ns, s)
             : plan.bm == 256 ? x3_go<VN_EPI_STORE, 2, X3_MODE_TN>(ctx, q, ns, s) : x3_go<VN_EPI_STORE, 1, X3_MODE_TN>(ctx, q, ns, s);
    };
    int rc;
    if (plan.ns > 1) {
        if (!ctx->x3_ws) VN_HIP_CHECK(ctx, vn_dev_malloc((void**)&ctx->x3_ws, (size_t)X3_WS_FLOATS * sizeof(float)));
        vn_gemm_args q = a;
        q.C = ctx->x3_ws;
        q.ldc = a.N;
        rc = go(q, plan.ns);
        if (rc == VN_OK) rc = vn_launch_splitk_reduce(ctx, ctx->x3_ws, plan.ns, a.C, a.M, a.N, a.ldc, false, s);
    } else {
        rc = go(a, 1);
    }
    vn_prof_post(ctx, pi, s);
    return rc;
}

template <typename K>
static int x3_attr(vn_ctx* ctx, K kernel, size_t bytes) {
    VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return VN_OK;
}
template <int EPI, int FMT = 0>
static int x3_attrs(vn_ctx* ctx) {
    constexpr int NP = FMT ? 2 : 3;
    int rc = x3_attr(ctx, vn_gemm_x3_kernel<EPI, 1, 0, FMT>, x3_lds_bytes<1, NP>());
    if (!rc) rc = x3_attr(ctx, vn_gemm_x3_kernel<EPI, 2, 0, FMT>, x3_lds_bytes<2, NP>());
    if constexpr (EPI != VN_EPI_GEGLU || !FMT) {
        if (!rc) rc = x3_attr(ctx, vn_gemm_x3_kernel<EPI, 3, 0, FMT>, x3_lds_bytes<3, NP>());
    }
    if constexpr (EPI != VN_EPI_CONV && !FMT) {
        if (!rc) rc = x3_attr(ctx, vn_gemm_x3_kernel<EPI, 4, 0, FMT>, x3_lds_bytes<4, NP>());
    }
    return rc;
}
template <int MI, int ABL, int FMT = 0>
static int x3_attrs_abl(vn_ctx* ctx) {
    return x3_attr(ctx, vn_gemm_x3_kernel<VN_EPI_STORE, MI, ABL, FMT>, x3_lds_bytes<MI, FMT ? 2 : 3>());
}

int vn_launch_gemm_x3(vn_ctx* ctx, const vn_gemm_args& a, int epilogue, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return vn_fail(ctx, VN_ERR_INVALID, "gemm_x3: empty problem%s", "");
    if (a.K % X3_KT) return vn_fail(ctx, VN_ERR_INVALID, "gemm_x3: K=%s%ld must be a multiple of 32", "", a.K);
    if (epilogue == VN_EPI_CONV ? (a.N % 16) : (a.N % 64))
        return vn_fail(ctx, VN_ERR_INVALID, "gemm_x3: N=%s%ld must be a multiple of 64 (16 for the convolution epilogue)", "", a.N);
    const bool h2 = a.bf16 == 3;
    if ((a.a_plane != (h2 ? VN_PLANES_TILED_H2 : VN_PLANES_TILED) && (a.a_plane <= 0 || (a.a_plane & 7))) ||
        (!a.w_tiled && (a.w_plane <= 0 || (a.w_plane & 7))))
        return vn_fail(ctx, VN_ERR_INVALID, "gemm_x3: plane strides must be positive multiples of 8 elements (or the tiled layout of the format)%s", "");
    if (((uintptr_t)a.A | (uintptr_t)a.W) & 15) return vn_fail(ctx, VN_ERR_INVALID, "gemm_x3: operands must be 16-byte aligned%s", "");
    // folded RMSNorm: producers write 128-column group sums (N a multiple of 128, planes in the operands' tiled format), consumers read K / 128 of them
    if ((a.ssq_out || a.X16) && (epilogue != VN_EPI_RESIDUAL || !a.ssq_out || !a.X16 || (a.N & 127) || a.ldc != a.N ||
                                 a.x16_plane != (h2 ? VN_PLANES_TILED_H2 : VN_PLANES_TILED)))
        return vn_fail(ctx, VN_ERR_INVALID, "gemm_x3: folded-norm outputs need the RESIDUAL epilogue, N %% 128 == 0, ldc == N, tiled planes + ssq%s", "");
    if (a.x16_only && (h2 || !a.X16)) return vn_fail(ctx, VN_ERR_INVALID, "gemm_x3: x16_only needs exact (bf16x3) planes%s", "");
    if (a.ssq_in && ((a.K & 127) || a.K > 2048)) return vn_fail(ctx, VN_ERR_INVALID, "gemm_x3: folded-norm input needs K %% 128 == 0 and K <= 2048%s", "");
    if (!(ctx->attr_mask & VN_ATTR_GEMM_X3)) {
        int rc;
        if ((rc = x3_attrs<VN_EPI_STORE>(ctx)) || (rc = x3_attrs<VN_EPI_BIAS>(ctx)) || (rc = x3_attrs<VN_EPI_RESIDUAL>(ctx)) ||
            (rc = x3_attrs<VN_EPI_GEGLU>(ctx)) || (rc = x3_attrs<VN_EPI_QKV>(ctx)) || (rc = x3_attrs<VN_EPI_QKV3>(ctx)) ||
            (rc = x3_attrs<VN_EPI_CONV>(ctx)))
            return rc;
        if ((rc = x3_attrs<VN_EPI_STORE, 1>(ctx)) || (rc = x3_attrs<VN_EPI_BIAS, 1>(ctx)) || (rc = x3_attrs<VN_EPI_RESIDUAL, 1>(ctx)) ||
            (rc = x3_attrs<VN_EPI_GEGLU, 1>(ctx)) || (rc = x3_attrs<VN_EPI_QKV, 1>(ctx)) || (rc = x3_attrs<VN_EPI_QKV3, 1>(ctx)) ||
            (rc = x3_attrs<VN_EPI_CONV, 1>(ctx)))
            return rc;
        if ((rc = x3_attrs_abl<1, 1>(ctx)) || (rc = x3_attrs_abl<1, 2>(ctx)) || (rc = x3_attrs_abl<1, 3>(ctx)) ||
            (rc = x3_attrs_abl<2, 1>(ctx)) || (rc = x3_attrs_abl<2, 2>(ctx)) || (rc = x3_attrs_abl<2, 3>(ctx)) ||
            (rc = x3_attrs_abl<1, 4>(ctx)) || (rc = x3_attrs_abl<2, 4>(ctx)) || (rc = x3_attrs_abl<3, 1>(ctx)) ||
            (rc = x3_attrs_abl<3, 2>(ctx)) || (rc = x3_attrs_abl<3, 3>(ctx)) || (rc = x3_attrs_abl<3, 4>(ctx)))
            return rc;
        if ((rc = x3_attrs_abl<1, 1, 1>(ctx)) || (rc = x3_attrs_abl<1, 2, 1>(ctx)) || (rc = x3_attrs_abl<1, 3, 1>(ctx)) ||
            (rc = x3_attrs_abl<2, 1, 1>(ctx)) || (rc = x3_attrs_abl<2, 2, 1>(ctx)) || (rc = x3_attrs_abl<2, 3, 1>(ctx)) ||
            (rc = x3_attrs_abl<3, 1, 1>(ctx)) || (rc = x3_attrs_abl<3, 2, 1>(ctx)) || (rc = x3_attrs_abl<3, 3, 1>(ctx)) ||
            (rc = x3_attrs_abl<1, 4, 1>(ctx)) || (rc = x3_attrs_abl<2, 4, 1>(ctx)) || (rc = x3_attrs_abl<3, 4, 1>(ctx)))
            return rc;
        if ((rc = x3_attrs_abl<1, X3_MODE_TN>(ctx)) || (rc = x3_attrs_abl<2, X3_MODE_TN>(ctx)) || (rc = x3_attrs_abl<3, X3_MODE_TN>(ctx))) return rc;
        ctx->attr_mask |= VN_ATTR_GEMM_X3;
    }
    if (a.tn_blocks > 0) {
        if (epilogue != VN_EPI_STORE || h2 || a.a_plane != VN_PLANES_TILED || !a.w_tiled || (a.M & 31) || (a.ldc & 3) || !a.C ||
            a.tn_blocks > a.K / 16 || a.tn_blocks <= a.K / 16 - 2)
            return vn_fail(ctx, VN_ERR_INVALID, "gemm_x3/TN: STORE epilogue on tiled bf16x3 planes, M %% 32 == 0, K = the token count rounded up to 32 (M=%s%ld, K=%ld)", "", a.M, a.K);
        return x3_launch_tn(ctx, a, s);
    }
    switch (epilogue) {
        case VN_EPI_STORE: return h2 ? x3_launch<VN_EPI_STORE, 1>(ctx, a, s) : x3_launch<VN_EPI_STORE>(ctx, a, s);
        case VN_EPI_BIAS:
            if (!a.bias) return vn_fail(ctx, VN_ERR_INVALID, "gemm_x3: bias epilogue needs bias%s", "");
            return h2 ? x3_launch<VN_EPI_BIAS, 1>(ctx, a, s) : x3_launch<VN_EPI_BIAS>(ctx, a, s);
        case VN_EPI_RESIDUAL: return h2 ? x3_launch<VN_EPI_RESIDUAL, 1>(ctx, a, s) : x3_launch<VN_EPI_RESIDUAL>(ctx, a, s);
        case VN_EPI_GEGLU:
            // the planes of the result are written in the operands' format (they are the next GEMM's A operand)
            if (a.C16 && (h2 ? (a.c_plane != VN_PLANES_TILED_H2 && a.c_plane <= 0) : (a.c_plane != VN_PLANES_TILED && a.c_plane <= 0)))
                return vn_fail(ctx, VN_ERR_INVALID, "gemm_x3/geglu: C16 needs c_plane (> 0 planar, or the tiled layout of the operands' format)%s", "");
            if (a.C16 && vn_planes_tiled(a.c_plane) && (a.ldc & 31)) return vn_fail(ctx, VN_ERR_INVALID, "gemm_x3/geglu: tiled planes need ldc %% 32 == 0%s", "");
            return h2 ? x3_launch<VN_EPI_GEGLU, 1>(ctx, a, s) : x3_launch<VN_EPI_GEGLU>(ctx, a, s);
        case VN_EPI_QKV: return h2 ? x3_launch<VN_EPI_QKV, 1>(ctx, a, s) : x3_launch<VN_EPI_QKV>(ctx, a, s);
        case VN_EPI_QKV3:
            if (!a.C16 || !a.V16 || a.c_plane <= 0 || a.v_plane <= 0 || a.T <= 0 || a.H <= 0 || a.N != 3 * a.H * VN_DHEAD)
                return vn_fail(ctx, VN_ERR_INVALID, "gemm_x3: QKV plane epilogue needs C16 / V16 / plane strides / T / H and N = 3 H 64%s", "");
            return h2 ? x3_launch<VN_EPI_QKV3, 1>(ctx, a, s) : x3_launch<VN_EPI_QKV3>(ctx, a, s);
        case VN_EPI_CONV:
            if (a.conv_taps <= 0 || a.conv_cin <= 0 || (a.conv_cin % X3_KT) || a.K != a.conv_taps * a.conv_cin || a.conv_trows <= 0 ||
                a.M % a.conv_trows || a.conv_tin <= 0 || a.conv_tout <= 0 || !a.zeros16)
                return vn_fail(ctx, VN_ERR_INVALID, "gemm_x3/conv: inconsistent geometry (taps=%s%ld, C_in=%ld)", "", a.conv_taps, a.conv_cin);
            if (a.a_plane <= 0) return vn_fail(ctx, VN_ERR_INVALID, "gemm_x3/conv: the activation planes must be planar%s", "");
            if ((!a.C && !a.Y2 && !a.C16) || ((a.Y2 || a.C16) && !a.alpha))
                return vn_fail(ctx, VN_ERR_INVALID, "gemm_x3/conv: needs an output, and alpha for the snake outputs%s", "");
            // C16 planes: planar, in the operands' format (c_plane > 0 for bf16x3, < -2 for f16x2: the producers' stride convention)
            if ((((uintptr_t)a.C | (uintptr_t)a.Y2 | (uintptr_t)a.resid | (uintptr_t)a.bias | (uintptr_t)a.alpha) & 15) || ((uintptr_t)a.C16 & 7) ||
                (a.C16 && (h2 ? (a.c_plane >= VN_PLANES_TILED_H2 || ((-a.c_plane) & 3)) : (a.c_plane <= 0 || (a.c_plane & 3)))))
                return vn_fail(ctx, VN_ERR_INVALID, "gemm_x3/conv: outputs / bias / alpha must be 16-byte aligned%s", "");
            return h2 ? x3_launch<VN_EPI_CONV, 1>(ctx, a, s) : x3_launch<VN_EPI_CONV>(ctx, a, s);
    }
    return vn_fail(ctx, VN_ERR_INVALID, "gemm_x3: unknown epilogue %s%ld", "", epilogue);
}

// ---- planar planes -> tiled planes (weights at load time): one thread per 16-byte run of a piece row
__global__ __launch_bounds__(256) void vn_tile_planes_kernel(const uint16_t* __restrict__ planes, long plane, uint16_t* __restrict__ tiled,
                                                             long rows, int K) {
    const long n16 = 3L * rows * (K >> 3);                  // 16-byte runs of the output
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n16; i += (long)gridDim.x * 256L) {
        const long piece = i >> 6;                          // 64 runs per 1 KiB piece
        const int in_piece = (int)(i & 63), r = in_piece >> 2, slot = in_piece & 3;
        const int pt = (int)(piece % 3);
        const long bk = piece / 3, kt = bk % (K >> 5), rb = bk / (K >> 5);
        *(u32x4*)(tiled + i * 8) = *(const u32x4*)(planes + (size_t)pt * plane + (size_t)(rb * 16 + r) * K + kt * 32 + slot * 8);
    }
}
int vn_launch_tile_planes(vn_ctx* ctx, const uint16_t* planes, long plane, uint16_t* tiled, long rows, int K, hipStream_t s) {
    if (rows <= 0 || K <= 0 || (rows & 15) || (K & 31) || (plane & 7))
        return vn_fail(ctx, VN_ERR_INVALID, "tile_planes: rows %% 16, K %% 32, plane %% 8 must be 0 (rows=%s%ld, K=%ld)", "", rows, K);
    const long n16 = 3L * rows * (K >> 3);
    const int blocks = (int)((n16 + 255) / 256 < 8192 ? (n16 + 255) / 256 : 8192);
    hipLaunchKernelGGL(vn_tile_planes_kernel, dim3(blocks), dim3(256), 0, s, planes, plane, tiled, rows, K);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

// ---- plane builder (weights at load time, tests): dst[q][i] = q-th split term of src[i]
__global__ __launch_bounds__(256) void vn_split3_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, long n4,
                                                        long plane) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256L) {
        const f32x4 v = ((const f32x4*)src)[i];
        vn_store_bf16x4(dst + 4 * i, plane, v);
    }
}

extern "C" int vn_split3_f32(vn_ctx* ctx, const float* src, void* dst16, int64_t n, int64_t plane_stride, void* stream) {
    if (!ctx || !src || !dst16 || n <= 0 || (n & 3) || plane_stride < n || (plane_stride & 7)) return VN_ERR_INVALID;
    const long n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(vn_split3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (uint16_t*)dst16, n4, (long)plane_stride);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

// single-op entry (tests / tuning): A3 [3][M][K] and W3 [3][N][K] bf16 planes -> fp32 C
extern "C" int vn_gemm_bf16x3(vn_ctx* ctx, const void* A3, int64_t a_plane, const void* W3, int64_t w_plane, const float* bias,
                              float* C, int M, int N, int K, int epilogue, void* stream) {
    if (!ctx || !A3 || !W3 || !C) return VN_ERR_INVALID;
    if (epilogue < VN_EPI_STORE || epilogue > VN_EPI_GEGLU) return vn_fail(ctx, VN_ERR_INVALID, "bad epilogue%s", "");
    vn_gemm_args a{};
    a.A = (const float*)A3; a.W = (const float*)W3; a.bias = bias; a.C = C; a.M = M; a.N = N; a.K = K;
    a.ldc = epilogue == VN_EPI_GEGLU ? N / 2 : N;
    a.bf16 = 2; a.a_plane = a_plane; a.w_plane = w_plane;
    a.w_tiled = w_plane == VN_PLANES_TILED;               // -1 for either stride: that operand is given in the tiled layout
    return vn_launch_gemm_x3(ctx, a, epilogue, (hipStream_t)stream);
}

// single-op entry of the TN operand mode (tests / tuning): At3 / Wt3 = TILED bf16x3 planes of the token-major matrices [tokens][M] / [tokens][N]
// (rows past `tokens` inside the last 16-row block zero), C [M][N] = At^T Wt
extern "C" int vn_gemm_bf16x3_tn(vn_ctx* ctx, const void* At3, const void* Wt3, float* C, int M, int N, int tokens, void* stream) {
    if (!ctx || !At3 || !Wt3 || !C || tokens <= 0) return VN_ERR_INVALID;
    vn_gemm_args a{};
    a.A = (const float*)At3; a.W = (const float*)Wt3; a.C = C; a.M = M; a.N = N; a.K = (tokens + 31) & ~31; a.ldc = N;
    a.bf16 = 2; a.a_plane = VN_PLANES_TILED; a.w_plane = VN_PLANES_TILED; a.w_tiled = 1;
    a.tn_blocks = (tokens + 15) >> 4;
    return vn_launch_gemm_x3(ctx, a, VN_EPI_STORE, (hipStream_t)stream);
}

// ---- f16x2 (vn_common.h vn_split2h): plane builders and the single-op entry -------------------------------------------------------
// fp32 [rows][K] -> f16x2 planes, planar (dst[q][rows][K], `plane` elements apart) or tiled ([rows / 16][K / 32][2][16][32])
__global__ __launch_bounds__(256) void vn_split2h_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, long rows, int K,
                                                         long plane, unsigned* sat) {
    const long n4 = rows * (K >> 2);
    bool bad = false;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256L) {
        const long row = i / (K >> 2);
        const int col = (int)(i - row * (K >> 2)) * 4;
        vn_store_planes4(dst, plane, row, col, K, ((const f32x4*)src)[i], bad);
    }
    vn_sat_report(sat, VN_SAT_WEIGHT, bad);
}
// fp32 [rows][K], every column k multiplied by scale[k] first (one fp32 rounding) -> TILED split planes of either format (plane =
// VN_PLANES_TILED: three bf16 planes, VN_PLANES_TILED_H2: two fp16 planes): the consumer weights of a model with folded RMSNorms,
// W' = W (.) w_norm (engine.hip).  rows % 16 == 0, K % 32 == 0.
__global__ __launch_bounds__(256) void vn_fold_planes_kernel(const float* __restrict__ src, const float* __restrict__ scale,
                                                             uint16_t* __restrict__ dst, long rows, int K, long plane, unsigned* sat) {
    const long n4 = rows * (K >> 2);
    bool bad = false;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256L) {
        const long row = i / (K >> 2);
        const int col = (int)(i - row * (K >> 2)) * 4;
        f32x4 v = ((const f32x4*)src)[i];
        const f32x4 sc = *(const f32x4*)(scale + col);
        v[0] *= sc[0]; v[1] *= sc[1]; v[2] *= sc[2]; v[3] *= sc[3];
        vn_store_planes4(dst, plane, row, col, K, v, bad);
    }
    vn_sat_report(sat, VN_SAT_WEIGHT, bad);
}
int vn_launch_fold_planes(vn_ctx* ctx, const float* src, const float* scale, uint16_t* dst, long rows, int K, long plane, hipStream_t s) {
    if (rows <= 0 || K <= 0 || (rows & 15) || (K & 31) || !vn_planes_tiled(plane) || !scale)
        return vn_fail(ctx, VN_ERR_INVALID, "fold_planes: tiled planes only (rows %% 16, K %% 32) (rows=%s%ld, K=%ld)", "", rows, K);
    const long n4 = rows * (K >> 2);
    const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(vn_fold_planes_kernel, dim3(blocks), dim3(256), 0, s, src, scale, dst, rows, K, plane, ctx->sat);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}
int vn_launch_split2h(vn_ctx* ctx, const float* src, uint16_t* dst, long rows, int K, long plane, hipStream_t s) {
    if (rows <= 0 || K <= 0 || (K & 3) || !vn_planes_h2(plane) || (plane == VN_PLANES_TILED_H2 && ((rows & 15) || (K & 31))) ||
        (plane != VN_PLANES_TILED_H2 && ((-plane) & 7)))
        return vn_fail(ctx, VN_ERR_INVALID, "split2h: K %% 4 == 0 (tiled: rows %% 16, K %% 32; planar: stride %% 8) (rows=%s%ld, K=%ld)", "", rows, K);
    const long n4 = rows * (K >> 2);
    const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(vn_split2h_kernel, dim3(blocks), dim3(256), 0, s, src, dst, rows, K, plane, ctx->sat);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}
// C-ABI: tiled != 0 -> the tiled layout (rows % 16 == 0, K % 32 == 0), else two planar planes plane_stride elements apart
extern "C" int vn_split2_f16(vn_ctx* ctx, const float* src, void* dst16, int64_t rows, int K, int64_t plane_stride, int tiled, void* stream) {
    if (!ctx || !src || !dst16) return VN_ERR_INVALID;
    if (!tiled && (plane_stride < rows * K || (plane_stride & 7))) return vn_fail(ctx, VN_ERR_INVALID, "split2_f16: plane stride %s%ld too small or not a multiple of 8", "", (long)plane_stride);
    return vn_launch_split2h(ctx, src, (uint16_t*)dst16, (long)rows, K, tiled ? VN_PLANES_TILED_H2 : -(long)plane_stride, (hipStream_t)stream);
}
// single-op entry (tests / tuning): A2 [2][M][K] and W2 [2][N][K] fp16 planes (a stride of -1: that operand in the tiled layout) -> fp32 C
extern "C" int vn_gemm_f16x2(vn_ctx* ctx, const void* A2, int64_t a_plane, const void* W2, int64_t w_plane, const float* bias,
                             float* C, int M, int N, int K, int epilogue, void* stream) {
    if (!ctx || !A2 || !W2 || !C) return VN_ERR_INVALID;
    if (epilogue < VN_EPI_STORE || epilogue > VN_EPI_GEGLU) return vn_fail(ctx, VN_ERR_INVALID, "bad epilogue%s", "");
    vn_gemm_args a{};
    a.A = (const float*)A2; a.W = (const float*)W2; a.bias = bias; a.C = C; a.M = M; a.N = N; a.K = K;
    a.ldc = epilogue == VN_EPI_GEGLU ? N / 2 : N;
    a.bf16 = 3; a.a_plane = a_plane == -1 ? VN_PLANES_TILED_H2 : a_plane; a.w_plane = w_plane;
    a.w_tiled = w_plane == -1;
    return vn_launch_gemm_x3(ctx, a, epilogue, (hipStream_t)stream);
}

// ---- CONVT: a convolution whose OUTPUT CHANNELS are the tile's rows (vn_common.h VN_EPI_CONVT).  Offered for C_out = 192 (CFG 3, one
// 192-row tile) and C_out = 96 (CFG 4, one 96-row tile with the k-steps split between the wave groups), bf16x3 operands.  a.A = the tiled
// weight planes [C_out][taps C_in], a.W = the planar activation planes (w_plane apart), M = C_out, N = B T_rows, K = taps C_in.
static bool x3_convt_shape(int h2, int C_out, int C_in, int taps) {
    return !h2 && (C_out == 192 || C_out == 96) && C_in % X3_KT == 0 && taps > 0;
}
static int x3_launch_convt(vn_ctx* ctx, const vn_gemm_args& a, hipStream_t s) {
    if (!(ctx->attr_mask & VN_ATTR_GEMM_X3_CONVT)) {
        int rc;
        if ((rc = x3_attr(ctx, vn_gemm_x3_kernel<VN_EPI_CONVT, 3, 0, 0>, x3_lds_bytes<3, 3>())) ||
            (rc = x3_attr(ctx, vn_gemm_x3_kernel<VN_EPI_CONVT, 4, 0, 0>, x3_lds_bytes<4, 3>())))
            return rc;
        ctx->attr_mask |= VN_ATTR_GEMM_X3_CONVT;
    }
    const double fl = 2.0 * a.M * (double)a.N * a.K;
    const double rows_in = (double)(a.N / a.conv_trows) * a.conv_tin;
    const double pbytes = 6.0 * (rows_in * a.conv_cin + (double)a.M * a.K) +
                          (double)a.M * a.N * ((a.C ? 4.0 : 0.0) + (a.Y2 ? 4.0 : 0.0) + (a.C16 ? 6.0 : 0.0) + (a.resid ? 4.0 : 0.0));
    const int pi = vn_prof_pre(ctx, vn_conv_class(fl, pbytes, VN_PROF_CONV_X3, 2500.0 / 6.0), fl, s, pbytes);
    const int rc = a.M == 192 ? x3_go<VN_EPI_CONVT, 3, 0, 0>(ctx, a, 1, s) : x3_go<VN_EPI_CONVT, 4, 0, 0>(ctx, a, 1, s);
    vn_prof_post(ctx, pi, s);
    return rc;
}

// ---- the DAC convolutions on the bf16x3 pipe (codec rows a18 / a19; PARITY UNPINNED like conv1d_f32.hip) -------------------------
// y[b][t_out][co] = act(bias[co] + sum_{j < taps} sum_ci w[co][j][ci] x[b][t' in_stride + j dil - pad][ci] (+ resid)), t_out = t'
// out_stride + out_off — conv1d_f32.hip's operator with the products on the bf16 matrix cores at fp32 grade: x16 = three split planes
// of the channels-last input [3][B T_in][C_in] (x_plane elements apart; written by the producing layer's epilogue), w_tiled = the
// TILED planes of w [C_out][taps C_in] (vn_split3_f32 + vn_tile_planes_bf16x3 at load).  Outputs as in vn_conv1d_f32, plus
// y2_16 = snake(y) as split planes (y2_plane apart) for a consumer on this pipe.
static int conv1d_planes(vn_ctx* ctx, int h2, const void* x16, int64_t x_plane, const void* w_tiled, const float* bias, const float* resid,
                         const float* alpha, float* y, float* y2, void* y2_16, int64_t y2_plane, int B, int T_in, int T_rows,
                         int T_out, int C_in, int C_out, int taps, int in_stride, int dil, int pad, int out_stride, int out_off,
                         int act, void* stream) {
    if (!ctx || !x16 || !w_tiled || (!y && !y2 && !y2_16)) return VN_ERR_INVALID;
    if (B <= 0 || T_rows <= 0 || C_out <= 0 || taps <= 0 || C_in <= 0) return vn_fail(ctx, VN_ERR_INVALID, "conv1d_bf16x3: empty problem%s", "");
    if ((long)B * T_rows > 0x7fffffffL) return vn_fail(ctx, VN_ERR_INVALID, "conv1d_bf16x3: too many rows%s", "");
    if (!ctx->zero_page) {
        VN_HIP_CHECK(ctx, vn_dev_malloc((void**)&ctx->zero_page, 1024));
        VN_HIP_CHECK(ctx, hipMemset(ctx->zero_page, 0, 1024));
    }
    if (x3_convt_shape(h2, C_out, C_in, taps) && ctx->tune.x3_convt) {
        // output channels on the tile's row axis: exactly one 192- / 96-row tile instead of 128-wide column tiles that are 3/4 full
        if ((((uintptr_t)y | (uintptr_t)y2 | (uintptr_t)resid | (uintptr_t)bias | (uintptr_t)alpha) & 15) || ((uintptr_t)y2_16 & 7) ||
            (y2_16 && (y2_plane <= 0 || (y2_plane & 3))) || x_plane <= 0 || (x_plane & 7) || (((uintptr_t)x16 | (uintptr_t)w_tiled) & 15) ||
            ((y2 || y2_16) && !alpha) || T_in <= 0 || T_out <= 0)
            return vn_fail(ctx, VN_ERR_INVALID, "conv1d_bf16x3: outputs / bias / alpha must be 16-byte aligned, planes planar%s", "");
        vn_gemm_args t{};
        t.A = (const float*)w_tiled; t.W = (const float*)x16; t.bias = bias; t.C = y; t.C16 = (uint16_t*)y2_16; t.c_plane = y2_plane;
        t.bf16 = 2; t.a_plane = VN_PLANES_TILED; t.w_plane = x_plane; t.w_tiled = 0;
        t.M = C_out; t.N = B * T_rows; t.K = taps * C_in; t.ldc = C_out;
        t.conv_taps = taps; t.conv_cin = C_in; t.conv_tin = T_in; t.conv_trows = T_rows; t.conv_in_stride = in_stride; t.conv_dil = dil;
        t.conv_pad = pad; t.conv_tout = T_out; t.conv_out_stride = out_stride; t.conv_out_off = out_off; t.conv_act = act;
        t.zeros16 = (const uint16_t*)ctx->zero_page; t.resid = resid; t.alpha = alpha; t.Y2 = y2;
        return x3_launch_convt(ctx, t, (hipStream_t)stream);
    }
    vn_gemm_args a{};
    a.A = (const float*)x16; a.W = (const float*)w_tiled; a.bias = bias; a.C = y; a.C16 = (uint16_t*)y2_16;
    a.c_plane = h2 ? -y2_plane : y2_plane;               // planar planes; a negative stride names the f16x2 format (vn_common.h)
    a.bf16 = h2 ? 3 : 2; a.a_plane = x_plane; a.w_tiled = 1;
    a.M = B * T_rows; a.N = C_out; a.K = taps * C_in; a.ldc = C_out;
    a.conv_taps = taps; a.conv_cin = C_in; a.conv_tin = T_in; a.conv_trows = T_rows; a.conv_in_stride = in_stride; a.conv_dil = dil;
    a.conv_pad = pad; a.conv_tout = T_out; a.conv_out_stride = out_stride; a.conv_out_off = out_off; a.conv_act = act;
    a.zeros16 = (const uint16_t*)ctx->zero_page; a.resid = resid; a.alpha = alpha; a.Y2 = y2;
    return vn_launch_gemm_x3(ctx, a, VN_EPI_CONV, (hipStream_t)stream);
}
extern "C" int vn_conv1d_bf16x3(vn_ctx* ctx, const void* x16, int64_t x_plane, const void* w_tiled, const float* bias, const float* resid,
                                const float* alpha, float* y, float* y2, void* y2_16, int64_t y2_plane, int B, int T_in, int T_rows,
                                int T_out, int C_in, int C_out, int taps, int in_stride, int dil, int pad, int out_stride, int out_off,
                                int act, void* stream) {
    return conv1d_planes(ctx, 0, x16, x_plane, w_tiled, bias, resid, alpha, y, y2, y2_16, y2_plane, B, T_in, T_rows, T_out, C_in, C_out, taps,
                         in_stride, dil, pad, out_stride, out_off, act, stream);
}
// the same operator on f16x2 operands: x16 = TWO planar fp16 planes [2][B T_in][C_in] (vn_split2_f16), w_tiled = the tiled f16x2 planes of
// w (vn_split2_f16 with tiled = 1), y2_16 = snake(y) as two planar fp16 planes
extern "C" int vn_conv1d_f16x2(vn_ctx* ctx, const void* x16, int64_t x_plane, const void* w_tiled, const float* bias, const float* resid,
                               const float* alpha, float* y, float* y2, void* y2_16, int64_t y2_plane, int B, int T_in, int T_rows,
                               int T_out, int C_in, int C_out, int taps, int in_stride, int dil, int pad, int out_stride, int out_off,
                               int act, void* stream) {
    return conv1d_planes(ctx, 1, x16, x_plane, w_tiled, bias, resid, alpha, y, y2, y2_16, y2_plane, B, T_in, T_rows, T_out, C_in, C_out, taps,
                         in_stride, dil, pad, out_stride, out_off, act, stream);
}

// planar split planes [3][rows][K] (plane_stride elements apart) -> the tiled layout the bf16x3 GEMM / convolution read weights in
extern "C" int vn_tile_planes_bf16x3(vn_ctx* ctx, const void* planes, int64_t plane_stride, void* tiled, int64_t rows, int K, void* stream) {
    if (!ctx || !planes || !tiled) return VN_ERR_INVALID;
    return vn_launch_tile_planes(ctx, (const uint16_t*)planes, (long)plane_stride, (uint16_t*)tiled, (long)rows, K, (hipStream_t)stream);
}
