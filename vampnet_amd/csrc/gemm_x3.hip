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
// Kernel: one 512-thread block (8 waves as 4 x 2, wave tile 32 x 64) per 128 x 128 output tile, one block per CU (two
// waves per SIMD cover each other's LDS latency and barriers).
//   * k-tile = 32 bf16 = 64 bytes per row.  A stage holds the six plane tiles (3 x 128 rows of A, 3 x 128 of W) = 48 KiB,
//     double-buffered = 96 KiB of LDS, filled by LDS-DMA (global_load_lds_dwordx4: one wave instruction = 16 rows x 64 B).
//   * Loading each plane tile ONCE and using it in two or three of the six products is what separates this kernel
//     from running a K' = 6K bf16 GEMM over concatenated planes: 6 plane-tile loads per 6 MFMA groups instead of 12.
//     Per k-tile and CU: 48 KiB of DMA and 144 KiB of fragment reads against 1536 matrix-pipe cycles per SIMD
//     (32 B/clk and 94 B/clk; the LDS moves 256 B/clk for ds_read_b128).
//   * LDS image is lane-linear (DMA constraint), so the bank swizzle sits on the SOURCE address: 16-byte slot s of row r
//     is stored at slot s ^ ((r >> 2) & 3).  A ds_read_b128 is serviced in 16-lane groups whose rows are
//     {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (+32): with 64-byte rows the four rows that share r % 4 (hence a
//     256-byte bank-row quarter) have four different (r >> 2) & 3, so every group touches 16 distinct slots.
//   * Fragment of v_mfma_f32_32x32x16_bf16: lane l holds row l & 31, k = 8 (l >> 5) .. + 7 of the 16-wide step, i.e.
//     slot 2 s + (l >> 5) of the row for sub-step s in {0, 1}.
//   * Epilogues as in gemm_f32.hip (store / bias / residual / GEGLU gate / QKV head-major scatter); the GEGLU result is
//     written as three planes again (it is only ever the A operand of the next GEMM).
//   * rows >= M and columns >= N are clamped on load and masked on store.
//   * schedule (PIPE 3, default): fragments of the next k-step are read into a second register set behind the first two
//     MFMAs of the current one, ONE barrier per k-tile sits between its two k-steps, the six DMA pieces of tile kt + 2 are
//     issued one per MFMA pair after it.  PIPE 1 issues them back to back; PIPE 0 / 2 and the ABL ablations are tuning
//     variants of the store epilogue (profiles/r01_gemm_x3_ablations.txt: the kernel is bound by the DMA volume).
// Data-parallel tile walk, XCD-aware (same remap and 8-row grouping as gemm_f32.hip).  Shapes whose 128 x 128 tiles fill
// 256 CUs badly (the N = 1280 projections: 360 tiles at B = 8, 110 / 50 for c2f / one sequence) are split along K in two
// passes: gridDim.y splits store raw images to a workspace, vn_splitk_reduce_kernel adds them in fixed order and applies
// the residual.  Deterministic in both forms.  On the GPU the result is within 7.5e-6 of float64 at 4600 x 3840 x 1280
// (the fp32-input MFMA kernel: 9e-6; both accumulate K = 1280 sequentially in fp32).
#include <stdlib.h>
#include "vn_common.h"

#define X3_BN 128
#define X3_KT 32                                  // bf16 per k-tile
#define X3_BPLANE (128 * 16)                      // floats of one W plane tile: 128 rows x 64 B

// Geometry by tile height BM = 128 MI.  A stage = three A plane tiles (BM rows x 64 B) then three W plane tiles.
template <int MI>
struct x3_geo {
    static constexpr int BM = 128 * MI;
    static constexpr int APLANE = BM * 16;                          // floats
    static constexpr int STAGE = 3 * (APLANE + X3_BPLANE);          // floats: 48 KiB (MI 1) / 72 KiB (MI 2)
    static constexpr int NQ = 3 * (BM + 128) / 16;                  // 1 KiB DMA wave-instructions per stage
    static constexpr int NPW = NQ / 8;                              // per wave: 6 / 9
};

__device__ __forceinline__ int x3_xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ void x3_tile_coords(int t, int tiles_m, int tiles_n, int& tm, int& tn) {
    constexpr int GROUP_M = 8;                    // 8 x 8 patches of tiles share A / W panels in one XCD's L2
    const int per_group = GROUP_M * tiles_n;
    const int grp = t / per_group;
    const int first_m = grp * GROUP_M;
    const int gsz = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
    const int in_grp = t - grp * per_group;
    tm = first_m + in_grp % gsz;
    tn = in_grp / gsz;
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

// PIPE 3: lock-step schedule of round 1 (128 x 128 only): register-prefetched fragments, one barrier per k-tile, DMA pieces
//         spread over the MFMA pairs, two LDS buffers.
// PIPE 4: "ping-pong".  The eight waves form two groups (waves 0-3 / 4-7: one wave of each group on every SIMD) that run
//         the same step sequence ONE PHASE APART: while a group issues the 12 MI MFMAs of a k-step (s_setprio 1), the other
//         reads its next fragments and issues LDS-DMA; two s_barriers per k-step keep the alternation.  Counted vmcnt,
//         raw s_barrier: DMA stays in flight across barriers.  MI = 1: three LDS buffers (tile kt + 2 is issued in the two
//         load phases of tile kt, waited for a tile later); MI = 2 (256 x 128): two buffers of 72 KiB (tile kt + 1 is
//         issued in the first load phase and between the MFMAs of the first compute phase of tile kt).
// ABL (tuning only, results invalid): bit 0 = no DMA inside the k-loop, bit 1 = no fragment reads inside the k-loop
template <int EPI, int PIPE, int MI, int ABL = 0>
__global__ __launch_bounds__(512, 2) void vn_gemm_x3_kernel(vn_gemm_args p, int tiles_m, int tiles_n) {
    using G = x3_geo<MI>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int tm, tn;
    x3_tile_coords(x3_xcd_remap(blockIdx.x, tiles_m * tiles_n), tiles_m, tiles_n, tm, tn);
    const int m0 = tm * G::BM, n0 = tn * X3_BN;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const uint16_t* A16 = (const uint16_t*)p.A;
    const uint16_t* W16 = (const uint16_t*)p.W;
    // split-K (gridDim.y > 1, store epilogue only): split sp multiplies k-tiles [kb, ke) into its own image of C
    const int nk_all = p.K / X3_KT;
    const int kb = (int)((long)nk_all * blockIdx.y / gridDim.y), ke = (int)((long)nk_all * (blockIdx.y + 1) / gridDim.y);
    if (gridDim.y > 1) p.C += (size_t)blockIdx.y * p.M * p.N;

    // per-lane DMA sources: instruction q = wave * NPW + j fills 16 rows of one plane tile (A: q < 24 MI, plane q / (8 MI))
    const uint16_t* src[G::NPW];
    const int drow = lane >> 2, dslot = (lane & 3) ^ ((lane >> 4) & 3);     // (row >> 2) & 3 == (lane >> 4) & 3
#pragma unroll
    for (int j = 0; j < G::NPW; ++j) {
        const int q = wave * G::NPW + j;
        if (q < 24 * MI) {
            const int pt = q / (8 * MI), row = (q % (8 * MI)) * 16 + drow;
            int g = m0 + row;
            g = g < p.M ? g : p.M - 1;
            src[j] = A16 + (size_t)pt * p.a_plane + (size_t)g * p.K + dslot * 8 + kb * X3_KT;
        } else {
            const int qb = q - 24 * MI;
            const int pt = qb >> 3, row = (qb & 7) * 16 + drow;
            int g = n0 + row;
            g = g < p.N ? g : p.N - 1;
            src[j] = W16 + (size_t)pt * p.w_plane + (size_t)g * p.K + dslot * 8 + kb * X3_KT;
        }
    }
    auto stage_piece = [&](int buf, int k0, int j) {
        float* base = lds + buf * G::STAGE + wave * (G::NPW * 256);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + k0),
                                         (__attribute__((address_space(3))) void*)(base + j * 256), 16, 0, 0);
    };
    auto stage = [&](int buf, int k0) {
#pragma unroll
        for (int j = 0; j < G::NPW; ++j) stage_piece(buf, k0, j);
    };

    // fragment offsets (floats) inside a plane tile: row * 16 + ((2 s + h) ^ ((row >> 2) & 3)) * 4
    const int l31 = lane & 31, h = lane >> 5, sw = (lane >> 2) & 3;
    const int aRow = (wm * 32 * MI + l31) * 16;
    const int bRow = (wn * 64 + l31) * 16;

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    struct Frags { bf16x8 a[3][MI], b[3][2]; };
    auto load_frags = [&](Frags& f, int buf, int s) {
        const float* sA = lds + buf * G::STAGE;
        const float* sB = sA + 3 * G::APLANE;
        const int off = ((2 * s + h) ^ sw) * 4;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
                f.a[q][i] = __builtin_bit_cast(bf16x8, *(const f32x4*)(sA + q * G::APLANE + aRow + i * 32 * 16 + off));
#pragma unroll
            for (int j = 0; j < 2; ++j)
                f.b[q][j] = __builtin_bit_cast(bf16x8, *(const f32x4*)(sB + q * X3_BPLANE + bRow + j * 32 * 16 + off));
        }
    };
    // the six plane products of one 16-wide k-step, smallest terms first (t = 0: A0 W2, then A2 W0, A1 W1, A0 W1, A1 W0,
    // A0 W0); consecutive MFMAs go to different accumulators
    auto mac_prod = [&](const Frags& f, int t) {
        const int qa = t == 1 ? 2 : (t == 2 || t == 4) ? 1 : 0, qb = t == 0 ? 2 : (t == 2 || t == 3) ? 1 : 0;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[qa][i], f.b[qb][j], acc[i][j], 0, 0, 0);
    };
    auto mac = [&](const Frags& f) {
#pragma unroll
        for (int t = 0; t < 6; ++t) mac_prod(f, t);
    };

    const int nk = ke - kb;
    if constexpr (PIPE == 3) {
        static_assert(PIPE != 3 || MI == 1, "the lock-step schedule is built for the 128 x 128 tile");
        Frags f0, f1;
        stage(0, 0);
        if (nk > 1) stage(1, X3_KT);
        X3_VMCNT(G::NPW);
        if (nk == 1) X3_VMCNT(0);
        __syncthreads();
        load_frags(f0, 0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            const bool more = kt + 2 < nk;
            // The reads of the NEXT group are issued behind the first product of the current one: the s_waitcnt the
            // compiler puts in front of a group is lgkmcnt(0) (it cannot count across the loop edge), which must only
            // cover fragments requested a whole group ago, not the ones just issued.
            __builtin_amdgcn_s_setprio(1);
            mac_prod(f0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_frags(f1, cur, 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 1; t < 6; ++t) mac_prod(f0, t);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            X3_VMCNT(0);                  // this wave's share of tile kt + 1 has landed (hipcc does not always emit it for a glds)
            __syncthreads();              // + lgkmcnt(0): f1 is in registers; after the barrier tile kt + 1 is complete
            mac_prod(f1, 0);
            if (more) stage_piece(cur, (kt + 2) * X3_KT, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < nk) load_frags(f0, cur ^ 1, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 1; t < 6; ++t) {
                mac_prod(f1, t);
                if (more) stage_piece(cur, (kt + 2) * X3_KT, t);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        static_assert(PIPE == 3 || PIPE == 4, "unknown schedule");
        const int grp = wave >> 2;                        // waves w and w + 4 share a SIMD: one of each group per SIMD
        Frags f;
        auto compute = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            mac(f);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (MI == 1) {
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
                    for (int j = 0; j < 3; ++j) stage_piece(b2, (kt + 2) * X3_KT, j);
                }
                X3_LGKM0();
                X3_BARRIER();
                compute();
                X3_BARRIER();
                // ---- k-step 1
                if constexpr (!(ABL & 2)) load_frags(f, b, 1);
                if (more) {
#pragma unroll
                    for (int j = 3; j < 6; ++j) stage_piece(b2, (kt + 2) * X3_KT, j);
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
            // two buffers of 72 KiB; tile kt lives in buffer kt & 1.  Tile kt + 1 goes into the buffer tile kt - 1 used (last
            // read in I_4kt-1, retired before that interval's barrier) and is first read in I_4kt+4: each wave issues five
            // of its nine pieces in the first load phase of tile kt and four between the products of its first compute phase
            // (group 0: I_4kt, I_4kt+1; group 1: I_4kt+1, I_4kt+2) and waits for them at the last point that still has a
            // barrier between it and the first read: group 0 at the end of its second compute phase (I_4kt+3), group 1 at
            // the end of its second load phase (I_4kt+3).
            stage(0, 0);
            X3_VMCNT(0);
            X3_BARRIER();
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
                    for (int j = 0; j < 5; ++j) stage_piece(b ^ 1, k1, j);
                }
                X3_LGKM0();
                X3_BARRIER();
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    mac_prod(f, t);
                    if (t < 4 && more) stage_piece(b ^ 1, k1, 5 + t);
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
    }

    // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int colw = n0 + wn * 64 + l31;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 * MI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row >= p.M) continue;
            if constexpr (EPI == VN_EPI_GEGLU) {
                // wave tile = 64 packed columns = 32 value (j = 0) + 32 gate (j = 1), interleaved at pack time
                const int ocol = (n0 + wn * 64) / 2 + l31;
                if (2 * ocol >= p.N) continue;
                const float o = acc[i][0][r] * vn_gelu_tanh(acc[i][1][r]);
                if (p.C16) {
                    uint16_t t0, t1, t2;
                    vn_split3(o, t0, t1, t2);
                    uint16_t* d = p.C16 + (size_t)row * p.ldc + ocol;
                    d[0] = t0; d[p.c_plane] = t1; d[2 * p.c_plane] = t2;
                } else {
                    p.C[(size_t)row * p.ldc + ocol] = o;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
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
                    }
                }
            }
        }
    }
}

#define X3_WS_FLOATS (32L << 20)          // 128 MiB of split-K partial images, allocated once (graph-safe: never re-allocated)

static int x3_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// schedule: VN_X3_PIPE = 3 lock-step 128 x 128 (round 1), 4 ping-pong 128 x 128, 5 ping-pong 256 x 128
static int g_x3_pipe = -1, g_x3_split = -2, g_x3_abl = -1;      // vn_debug_x3_config overrides (tests / tuning)
static int x3_pipe() {
    static const int pipe = x3_env("VN_X3_PIPE", 4);
    return g_x3_pipe >= 3 ? g_x3_pipe : pipe;
}
extern "C" int vn_debug_x3_config(int pipe, int splitk, int abl) {
    if (pipe != -1 && (pipe < 3 || pipe > 5)) return VN_ERR_INVALID;
    g_x3_pipe = pipe; g_x3_split = splitk < 0 ? -2 : splitk; g_x3_abl = abl < 0 ? -1 : (abl & 3);
    return VN_OK;
}
static int x3_bm() { return x3_pipe() == 5 ? 256 : 128; }

template <int EPI, int PIPE, int MI, int ABL = 0>
static void x3_go(const vn_gemm_args& a, int tiles_m, int tiles_n, int nsplit, hipStream_t s) {
    constexpr size_t lds_bytes = (size_t)x3_geo<MI>::STAGE * 4 * (PIPE == 4 && MI == 1 ? 3 : 2);
    hipLaunchKernelGGL((vn_gemm_x3_kernel<EPI, PIPE, MI, ABL>), dim3(tiles_m * tiles_n, nsplit), dim3(512), lds_bytes, s, a, tiles_m,
                       tiles_n);
}
template <int EPI>
static void x3_go_pipe(const vn_gemm_args& a, int nsplit, hipStream_t s) {
    const int tiles_n = vn_cdiv(a.N, X3_BN);
    switch (x3_pipe()) {
        case 3: x3_go<EPI, 3, 1>(a, vn_cdiv(a.M, 128), tiles_n, nsplit, s); break;
        case 5: x3_go<EPI, 4, 2>(a, vn_cdiv(a.M, 256), tiles_n, nsplit, s); break;
        default: x3_go<EPI, 4, 1>(a, vn_cdiv(a.M, 128), tiles_n, nsplit, s); break;
    }
}

// split count for the store / residual epilogues: a launch costs ceil(tiles * ns / 256) rounds of K / ns, plus the reduce
// pass over (ns + 1 or 2) images of C.  Constants from profiles/r01_gemm_x3_vs_f32.txt (1.45 us per k-tile and round,
// ~3.5 TB/s for the reduce).
static int x3_pick_split(const vn_gemm_args& a, bool residual) {
    static const int forced_env = x3_env("VN_X3_SPLITK", -1);      // 0 / 1: off, 2 / 4: forced
    const int forced = g_x3_split != -2 ? g_x3_split : forced_env;
    const int bm = x3_bm();
    const int tiles = vn_cdiv(a.M, bm) * vn_cdiv(a.N, X3_BN), nk = a.K / X3_KT;
    if (forced == 0 || forced == 1 || (a.N & 3) || (a.ldc & 3)) return 1;
    int best = 1;
    double best_cost = 1e300;
    for (int ns = 1; ns <= 4; ns *= 2) {
        if (ns > 1 && (nk / ns < 8 || (double)ns * a.M * a.N > (double)X3_WS_FLOATS)) continue;
        if (forced > 1 && ns != forced && ns != 1) continue;
        double cost = ceil(tiles * ns / 256.0) * (nk / (double)ns) * 1.45 * (bm / 128);
        if (ns > 1) cost += (ns + (residual ? 2 : 1)) * 4.0 * a.M * (double)a.N / 3.5e6;
        if (forced > 1 && ns == forced) cost = 0;
        if (cost < best_cost) { best_cost = cost; best = ns; }
    }
    return best;
}

template <int EPI>
static int x3_launch(vn_ctx* ctx, const vn_gemm_args& a, hipStream_t s) {
    const double n_out = (EPI == VN_EPI_GEGLU) ? a.N / 2 : a.N;
    const double bytes = 6.0 * ((double)a.M * a.K + (double)a.N * a.K) +
                         ((EPI == VN_EPI_GEGLU && a.C16) ? 6.0 : 4.0) * (double)a.M * n_out * (EPI == VN_EPI_RESIDUAL ? 2 : 1);
    const int pi = vn_prof_pre(ctx, 0, 2.0 * a.M * (double)a.N * a.K, s, bytes);       // algorithmic (fp32-equivalent) flops
    if constexpr (EPI == VN_EPI_STORE) {
        static const int abl_env = x3_env("VN_X3_ABL", 0) & 3;             // ablations (tuning only; results invalid)
        const int abl = g_x3_abl >= 0 ? g_x3_abl : abl_env;
        if (abl) {
            const int tiles_n = vn_cdiv(a.N, X3_BN);
            if (x3_pipe() == 5) {
                if (abl == 1) x3_go<VN_EPI_STORE, 4, 2, 1>(a, vn_cdiv(a.M, 256), tiles_n, 1, s);
                else if (abl == 2) x3_go<VN_EPI_STORE, 4, 2, 2>(a, vn_cdiv(a.M, 256), tiles_n, 1, s);
                else x3_go<VN_EPI_STORE, 4, 2, 3>(a, vn_cdiv(a.M, 256), tiles_n, 1, s);
            } else {
                if (abl == 1) x3_go<VN_EPI_STORE, 4, 1, 1>(a, vn_cdiv(a.M, 128), tiles_n, 1, s);
                else if (abl == 2) x3_go<VN_EPI_STORE, 4, 1, 2>(a, vn_cdiv(a.M, 128), tiles_n, 1, s);
                else x3_go<VN_EPI_STORE, 4, 1, 3>(a, vn_cdiv(a.M, 128), tiles_n, 1, s);
            }
            vn_prof_post(ctx, pi, s);
            VN_LAUNCH_CHECK(ctx);
            return VN_OK;
        }
    }
    if constexpr (EPI == VN_EPI_STORE || EPI == VN_EPI_RESIDUAL) {
        const int ns = x3_pick_split(a, EPI == VN_EPI_RESIDUAL);
        if (ns > 1) {
            if (!ctx->x3_ws) VN_HIP_CHECK(ctx, hipMalloc((void**)&ctx->x3_ws, (size_t)X3_WS_FLOATS * sizeof(float)));
            vn_gemm_args q = a;                       // every split stores its raw image [M][N] into the workspace
            q.C = ctx->x3_ws;
            q.ldc = a.N;
            x3_go_pipe<VN_EPI_STORE>(q, ns, s);
            VN_LAUNCH_CHECK(ctx);
            const int rc = vn_launch_splitk_reduce(ctx, ctx->x3_ws, ns, a.C, a.M, a.N, a.ldc, EPI == VN_EPI_RESIDUAL, s);
            vn_prof_post(ctx, pi, s);
            return rc;
        }
    }
    x3_go_pipe<EPI>(a, 1, s);
    vn_prof_post(ctx, pi, s);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

template <typename K>
static int x3_attr(vn_ctx* ctx, K kernel, size_t bytes) {
    VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return VN_OK;
}
template <int EPI>
static int x3_attrs(vn_ctx* ctx) {
    int rc;
    if ((rc = x3_attr(ctx, vn_gemm_x3_kernel<EPI, 3, 1>, (size_t)x3_geo<1>::STAGE * 8))) return rc;
    if ((rc = x3_attr(ctx, vn_gemm_x3_kernel<EPI, 4, 1>, (size_t)x3_geo<1>::STAGE * 12))) return rc;
    return x3_attr(ctx, vn_gemm_x3_kernel<EPI, 4, 2>, (size_t)x3_geo<2>::STAGE * 8);
}

int vn_launch_gemm_x3(vn_ctx* ctx, const vn_gemm_args& a, int epilogue, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return vn_fail(ctx, VN_ERR_INVALID, "gemm_x3: empty problem%s", "");
    if (a.K % X3_KT) return vn_fail(ctx, VN_ERR_INVALID, "gemm_x3: K=%s%ld must be a multiple of 32", "", a.K);
    if (a.N % 64) return vn_fail(ctx, VN_ERR_INVALID, "gemm_x3: N=%s%ld must be a multiple of 64", "", a.N);
    if (a.a_plane <= 0 || a.w_plane <= 0 || (a.a_plane & 7) || (a.w_plane & 7))
        return vn_fail(ctx, VN_ERR_INVALID, "gemm_x3: plane strides must be positive multiples of 8 elements%s", "");
    if (((uintptr_t)a.A | (uintptr_t)a.W) & 15) return vn_fail(ctx, VN_ERR_INVALID, "gemm_x3: operands must be 16-byte aligned%s", "");
    if (!(ctx->attr_mask & VN_ATTR_GEMM_X3)) {
        int rc;
        if ((rc = x3_attrs<VN_EPI_STORE>(ctx)) || (rc = x3_attrs<VN_EPI_BIAS>(ctx)) || (rc = x3_attrs<VN_EPI_RESIDUAL>(ctx)) ||
            (rc = x3_attrs<VN_EPI_GEGLU>(ctx)) || (rc = x3_attrs<VN_EPI_QKV>(ctx)))
            return rc;
        if ((rc = x3_attr(ctx, vn_gemm_x3_kernel<VN_EPI_STORE, 4, 1, 1>, (size_t)x3_geo<1>::STAGE * 12)) ||
            (rc = x3_attr(ctx, vn_gemm_x3_kernel<VN_EPI_STORE, 4, 1, 2>, (size_t)x3_geo<1>::STAGE * 12)) ||
            (rc = x3_attr(ctx, vn_gemm_x3_kernel<VN_EPI_STORE, 4, 1, 3>, (size_t)x3_geo<1>::STAGE * 12)) ||
            (rc = x3_attr(ctx, vn_gemm_x3_kernel<VN_EPI_STORE, 4, 2, 1>, (size_t)x3_geo<2>::STAGE * 8)) ||
            (rc = x3_attr(ctx, vn_gemm_x3_kernel<VN_EPI_STORE, 4, 2, 2>, (size_t)x3_geo<2>::STAGE * 8)) ||
            (rc = x3_attr(ctx, vn_gemm_x3_kernel<VN_EPI_STORE, 4, 2, 3>, (size_t)x3_geo<2>::STAGE * 8)))
            return rc;
        ctx->attr_mask |= VN_ATTR_GEMM_X3;
    }
    switch (epilogue) {
        case VN_EPI_STORE: return x3_launch<VN_EPI_STORE>(ctx, a, s);
        case VN_EPI_BIAS:
            if (!a.bias) return vn_fail(ctx, VN_ERR_INVALID, "gemm_x3: bias epilogue needs bias%s", "");
            return x3_launch<VN_EPI_BIAS>(ctx, a, s);
        case VN_EPI_RESIDUAL: return x3_launch<VN_EPI_RESIDUAL>(ctx, a, s);
        case VN_EPI_GEGLU:
            if (a.C16 && a.c_plane <= 0) return vn_fail(ctx, VN_ERR_INVALID, "gemm_x3/geglu: C16 needs c_plane%s", "");
            return x3_launch<VN_EPI_GEGLU>(ctx, a, s);
        case VN_EPI_QKV: return x3_launch<VN_EPI_QKV>(ctx, a, s);
    }
    return vn_fail(ctx, VN_ERR_INVALID, "gemm_x3: unknown epilogue %s%ld", "", epilogue);
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
    return vn_launch_gemm_x3(ctx, a, epilogue, (hipStream_t)stream);
}
