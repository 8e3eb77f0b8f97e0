// Exact-fp32 GEMM for gfx950:  C[M][N] (op)= A[M][K] * W[N][K]^T   (torch F.linear semantics)
//
// Replaces the reference's nn.Linear / lora.Linear / 1x1 WNConv1d calls on the hot path
// (vampnet/modules/transformer.py:81-84 FFN, :229-231 QKV, :255 fc, :632 classifier).
//
// Design (MI355X_MICROARCH.md / cdna_hip_programming.md §3 "FP32-input MFMA"):
//   * v_mfma_f32_32x32x2_f32: exact f32 (bitwise an fmaf chain), 64 cycles/instr/SIMD = the f32
//     vector peak (157.3 TF chip).  One wave per SIMD with >= 1 independent 32x32 accumulator already
//     issues back to back (dependent latency == issue interval == 64 cycles).
//   * block = 256 threads = 2x2 waves; block tile BM x BN in {128,64}^2, wave tile BM/2 x BN/2 made of
//     32x32 MFMA tiles; BK = 32 floats (one 128-byte line per row per k-tile).
//   * global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB = 8 rows x 128 B per wave instruction),
//     double-buffered: tile t+1 streams in while tile t is multiplied; one barrier per k-tile.
//   * LDS image is lane-linear (DMA constraint), so the bank-conflict swizzle is applied on the SOURCE
//     address: the 16-B slot s of row r is stored at slot s ^ ((r >> 1) & 7) (guide §5.4 rule 21), and
//     fragments are read with the same XOR by ds_read_b128.  With 128-byte rows the bank half is r & 1, so the
//     16 rows of a ds_read_b128 lane group ({0-3,12-15,20-27} / {4-11,16-19,28-31}) land on 16 distinct 16-byte
//     slots: conflict-free (the first version XORed r & 7 and paid a 2-way conflict on every fragment read).
//   * k-order inside a tile: lane half h = lane>>5 reads float4 at k = 8*s + 4*h .. +3 (s = 0..3) and
//     feeds element e to MFMA #e, i.e. MFMA (s,e) contracts k-pair {8s+e, 8s+4+e}.  Any bijection of k
//     is a valid fp32 summation order (the reference's own order is MKL's, not specified).
//   * epilogues fused: bias, residual add, GEGLU gate (value/gate columns interleaved at pack time so
//     they sit in the same lane of adjacent MFMA tiles), QKV head-major scatter.
//   * rows >= M are clamped on load (duplicate last row) and masked on store, so M needs no padding.
//   * blockIdx -> tile mapping is XCD-aware (block b runs on XCD b % 8): each XCD owns a contiguous
//     range of the tile walk, and the walk goes in 8-row groups so co-resident tiles form 8x8 patches
//     that share A/W panels through the XCD's 4 MiB L2.
//
// Two schedulers over the same tile body:
//   data-parallel  one block per output tile.  A CU retires about one 128x128-equivalent tile per unit
//                  time whether it holds 1 or 2 blocks (they share the matrix pipes), so a launch costs
//                  ceil(tiles / 256) units: M = 4600 x N = 3840 (1080 tiles) wastes 16 % in the last round.
//   stream-K       2 persistent blocks per CU, each given an EQUAL share of all (tile, k-tile) iterations;
//                  a tile split between blocks b < b+1 < ... is finished by the block that ran its FIRST
//                  k-range (the last thing that block does), adding the later ranges' partial sums, which
//                  their owners computed first and published to a workspace slab — so nobody waits in
//                  steady state and the summation order is fixed (deterministic, run-to-run bitwise).
//                  Hand-off follows the guide's §6 G16 recipe R1: write-through (sc1) 16-byte slab stores ->
//                  per-wave vmcnt(0) -> __syncthreads -> one-lane relaxed agent flag store; consumer: relaxed poll ->
//                  one agent-scope acquire -> __syncthreads -> plain loads; the consumer re-zeroes the flag.
#include <stdlib.h>
#include "vn_common.h"

#define BK 32
#ifndef VN_GEMM_PRIO
#define VN_GEMM_PRIO 1          // s_setprio during a wave's MFMA phase: the co-resident wave of the other block issues its DMA /
                                // LDS traffic in the shadow (same-box A/B: 4096^3 136.8 -> 139.0 TF, model shapes +0..1 %)
#endif
#ifndef VN_GEMM_SPREAD
#define VN_GEMM_SPREAD 0        // 1: issue the next tile DMA one part per sub-step; 0: all glds up front (default: same-box A/B on MI355X: spread +3..6 % on some data-parallel B=8 shapes, -5..20 % on small-M shapes, -1.4 % end to end)
#endif

template <int BM, int BN>
struct GemmCfg {
    static constexpr int MI = BM / 64;            // 32x32 tiles per wave along M
    static constexpr int NI = BN / 64;            // along N
    static constexpr int A_FLOATS = BM * BK;
    static constexpr int B_FLOATS = BN * BK;
    static constexpr int STAGE_FLOATS = A_FLOATS + B_FLOATS;
    static constexpr int LDS_BYTES = 2 * STAGE_FLOATS * 4;
    static constexpr int A_INSTR = BM / 32;       // DMA wave-instructions per wave for A (BM/8 total / 4 waves)
    static constexpr int B_INSTR = BN / 32;
    static constexpr int SLAB_FLOATS = BM * BN;   // stream-K partial-sum slab per block
};

// XCD-aware bijective remap of a linear block id (guide §5: "XCD swizzle must be bijective")
__device__ __forceinline__ int vn_xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// position t of the tile walk -> (tm, tn).  order 1: row-groups of 8 tiles (tm fastest inside a group, then tn):
// 64 consecutive positions form an 8 x 8 patch sharing 8 A and 8 W row-panels; order 0: column-major.
__device__ __forceinline__ void vn_tile_coords(int t, int tiles_m, int tiles_n, int order, int& tm, int& tn) {
    if (order == 1) {
        constexpr int GROUP_M = 8;
        const int per_group = GROUP_M * tiles_n;
        const int grp = t / per_group;
        const int first_m = grp * GROUP_M;
        const int gsz = (tiles_m - first_m) < GROUP_M ? (tiles_m - first_m) : GROUP_M;
        const int in_grp = t - grp * per_group;
        tm = first_m + in_grp % gsz;
        tn = in_grp / gsz;
    } else {
        tn = t / tiles_m;
        tm = t - tn * tiles_m;
    }
}

// floats per operand row / k-tiles per row.  bf16 fast mode: A and W hold bf16 (the pointers are typed float* only to
// share the staging code), a k-tile is still 128 bytes per row = 64 bf16.
template <bool BF>
__device__ __forceinline__ int vn_row_floats(const vn_gemm_args& p) { return BF ? p.K / 2 : p.K; }
template <bool BF>
__device__ __forceinline__ int vn_ktiles(const vn_gemm_args& p) { return vn_row_floats<BF>(p) / BK; }

// acc += A[m0.., kb*BK .. ke*BK) * W[n0.., same k)^T   for one block tile; all 256 threads participate.
// BF = true: the 16-byte fragment a lane reads is 8 bf16 = the operand of ONE v_mfma_f32_32x32x16_bf16 (lanes 0-31
// carry k 0..7, lanes 32-63 k 8..15 of the 16-wide step), fp32 accumulate; same LDS image, DMA and swizzle.
template <int BM, int BN, bool BF = false>
__device__ __forceinline__ void vn_gemm_mac(const vn_gemm_args& p, float* lds, int m0, int n0, int kb, int ke,
                                            f32x16 (&acc)[GemmCfg<BM, BN>::MI][GemmCfg<BM, BN>::NI]) {
    using Cfg = GemmCfg<BM, BN>;
    constexpr int MI = Cfg::MI, NI = Cfg::NI;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ldk = vn_row_floats<BF>(p);

    // per-lane DMA source pointers (pidx = 16-byte slot index inside the tile image)
    const float* srcA[Cfg::A_INSTR];
    const float* srcB[Cfg::B_INSTR];
#pragma unroll
    for (int q = 0; q < Cfg::A_INSTR; ++q) {
        const int pidx = (wave * Cfg::A_INSTR + q) * 64 + lane;
        const int row = pidx >> 3, slot = (pidx & 7) ^ ((row >> 1) & 7);
        int gm = m0 + row;
        gm = gm < p.M ? gm : p.M - 1;
        srcA[q] = p.A + (size_t)gm * ldk + slot * 4;
    }
#pragma unroll
    for (int q = 0; q < Cfg::B_INSTR; ++q) {
        const int pidx = (wave * Cfg::B_INSTR + q) * 64 + lane;
        const int row = pidx >> 3, slot = (pidx & 7) ^ ((row >> 1) & 7);
        int gn = n0 + row;
        gn = gn < p.N ? gn : p.N - 1;
        srcB[q] = p.W + (size_t)gn * ldk + slot * 4;
    }
    // DMA of one stage, in 4 parts (instructions q with q % 4 == part).  The parts of tile t+1 are issued one per
    // sub-step of tile t instead of back to back: a burst of 8 glds per wave x 8 waves right after the barrier fills the
    // CU's vector-memory issue queue and stalls the issuing waves (and their MFMAs) behind it — ablation on MI355X:
    // no DMA 140 TF, burst 130 TF, spread 137 TF at 4096^3.
    auto stage_part = [&](int buf, int k0, int part) {
        float* dA = lds + buf * Cfg::STAGE_FLOATS;
        float* dB = dA + Cfg::A_FLOATS;
#pragma unroll
        for (int q = 0; q < Cfg::A_INSTR; ++q)
            if ((q & 3) == part)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(srcA[q] + k0),
                    (__attribute__((address_space(3))) void*)(dA + (wave * Cfg::A_INSTR + q) * 256), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < Cfg::B_INSTR; ++q)
            if ((q & 3) == part)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(srcB[q] + k0),
                    (__attribute__((address_space(3))) void*)(dB + (wave * Cfg::B_INSTR + q) * 256), 16, 0, 0);
    };
    auto stage = [&](int buf, int k0) {
#pragma unroll
        for (int part = 0; part < 4; ++part) stage_part(buf, k0, part);
    };

    // fragment read offsets (floats) inside a stage: row*32 + ((2s + h) ^ ((row>>1)&7))*4 ; (row>>1)&7 == (lane>>1)&7
    const int l31 = lane & 31, h = lane >> 5, sw = (lane >> 1) & 7;
    const int aRow = (wm * (BM / 2) + l31) * BK;
    const int bRow = (wn * (BN / 2) + l31) * BK;

    stage(0, kb * BK);
    __syncthreads();   // glds in flight -> hipcc emits vmcnt(0) before the barrier (guide §5)
    for (int kt = kb; kt < ke; ++kt) {
        const int cur = (kt - kb) & 1;
        const bool more = kt + 1 < ke;
        const float* sA = lds + cur * Cfg::STAGE_FLOATS;
        const float* sB = sA + Cfg::A_FLOATS;
#if !VN_GEMM_SPREAD
        if (more) stage(cur ^ 1, (kt + 1) * BK);
#endif
#if VN_GEMM_PRIO
        __builtin_amdgcn_s_setprio(VN_GEMM_PRIO);  // favour the wave that is in its MFMA phase
#endif
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#if VN_GEMM_SPREAD
            if (more) stage_part(cur ^ 1, (kt + 1) * BK, s);
            __builtin_amdgcn_sched_barrier(0);      // keep this sub-step's DMA issue in front of its MFMAs
#endif
            const int off = ((2 * s + h) ^ sw) * 4;
            f32x4 a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = *(const f32x4*)(sA + aRow + i * 32 * BK + off);
#pragma unroll
            for (int j = 0; j < NI; ++j) b[j] = *(const f32x4*)(sB + bRow + j * 32 * BK + off);
            if constexpr (BF) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]),
                                                                            __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
            }
        }
#if VN_GEMM_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        __syncthreads();   // tile kt consumed by all waves; tile kt+1 landed (vmcnt(0) + barrier)
    }
}

// C/D map of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
template <int BM, int BN, int EPI>
__device__ __forceinline__ void vn_gemm_epilogue(const vn_gemm_args& p, int m0, int n0,
                                                 f32x16 (&acc)[GemmCfg<BM, BN>::MI][GemmCfg<BM, BN>::NI]) {
    constexpr int MI = GemmCfg<BM, BN>::MI, NI = GemmCfg<BM, BN>::NI;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    const int colw = n0 + wn * (BN / 2) + l31;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row >= p.M) continue;
            if constexpr (EPI == VN_EPI_GEGLU) {
                static_assert(NI == 2 || EPI != VN_EPI_GEGLU, "GEGLU needs value+gate tiles in one wave");
                // wave tile = 64 packed columns = 32 value (tile j=0) + 32 gate (tile j=1)
                const int ocol = (n0 + wn * (BN / 2)) / 2 + l31;
                const float v = acc[i][0][r], g = acc[i][NI - 1][r];
                if (p.C16) p.C16[(size_t)row * p.ldc + ocol] = vn_f32_to_bf16(v * vn_gelu_tanh(g));   // A operand of w_2
                else p.C[(size_t)row * p.ldc + ocol] = v * vn_gelu_tanh(g);
            } else {
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int col = colw + j * 32;
                    if (col >= p.N) continue;
                    float v = acc[i][j][r];
                    if constexpr (EPI == VN_EPI_STORE) {
                        p.C[(size_t)row * p.ldc + col] = v;
                    } else if constexpr (EPI == VN_EPI_BIAS) {
                        p.C[(size_t)row * p.ldc + col] = v + p.bias[col];
                    } else if constexpr (EPI == VN_EPI_RESIDUAL) {
                        float* c = p.C + (size_t)row * p.ldc + col;
                        *c = *c + v;
                    } else if constexpr (EPI == VN_EPI_QKV) {
                        // col in [0, 3D): which = col / D; head = (col % D) / 64; d = col % 64
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

// De-phase the two blocks that share a CU: blocks are dispatched round-robin over the 8 XCDs and, inside an XCD, over
// its 32 CUs, so dispatch index (blockIdx/8) and (blockIdx/8 + 32) are co-resident and would otherwise run in lock
// step and hit their per-k-tile barriers together (matrix pipes idle).  The second-slot block sleeps first.
__device__ __forceinline__ void vn_stagger(int order) {
    const int units = (order >> 8) & 0xff;
    if (units && (((blockIdx.x >> 3) >> 5) & 1)) {
        for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(16);      // 16 * 64 = 1024 cycles per unit
    }
}

template <int MI, int NI>
__device__ __forceinline__ void vn_acc_zero(f32x16 (&acc)[MI][NI]) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
}

// ---------------------------------------------------------------------------------------------
// data-parallel scheduler: one block per tile
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int EPI, bool BF = false>
__global__ __launch_bounds__(256, 2) void vn_gemm_f32_kernel(vn_gemm_args p, int tiles_m, int tiles_n, int order) {
    using Cfg = GemmCfg<BM, BN>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int tm, tn;
    vn_tile_coords(vn_xcd_remap(blockIdx.x, tiles_m * tiles_n), tiles_m, tiles_n, order, tm, tn);
    f32x16 acc[Cfg::MI][Cfg::NI];
    vn_acc_zero(acc);
    vn_gemm_mac<BM, BN, BF>(p, lds, tm * BM, tn * BN, 0, vn_ktiles<BF>(p), acc);
    vn_gemm_epilogue<BM, BN, EPI>(p, tm * BM, tn * BN, acc);
}

// ---------------------------------------------------------------------------------------------
// stream-K scheduler: G persistent blocks, equal shares of the tiles*nk iteration space
// ---------------------------------------------------------------------------------------------
#define SK_SPIN_LIMIT (1 << 24)

template <int BM, int BN, int EPI, bool BF = false>
__global__ __launch_bounds__(256, 2) void vn_gemm_f32_sk_kernel(vn_gemm_args p, int tiles_m, int tiles_n, int order,
                                                               float* __restrict__ slabs, unsigned* flags,
                                                               unsigned* errword) {
    using Cfg = GemmCfg<BM, BN>;
    constexpr int MI = Cfg::MI, NI = Cfg::NI;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    vn_stagger(order);
    order &= 1;
    const int G = gridDim.x;
    const int nk = vn_ktiles<BF>(p);
    const int ntiles = tiles_m * tiles_n;
    const int b = vn_xcd_remap(blockIdx.x, G);          // neighbours in the walk sit on one XCD
    const int tid = threadIdx.x;
    typedef __attribute__((address_space(1))) unsigned gu32;

    // Hybrid schedule ("data-parallel + two-tile stream-K"): whole rounds of G tiles run data-parallel inside the
    // persistent blocks — in round r block b takes walk position r*G + b, so the 64 blocks co-resident on an XCD work on
    // one 8x8 patch and share A/W panels in L2 — and only the last 0.5..1.5 tiles per block are split along K.  (A pure
    // stream-K split of a 2.8-tile range spreads the co-resident blocks over distant tiles: rocprof showed 2x the L2
    // miss traffic, 0.86 GB per W1 launch.)
    int dp_rounds = ntiles / G;
    if (dp_rounds > 0 && (ntiles - dp_rounds * G) * 2 < G) --dp_rounds;
    for (int r = 0; r < dp_rounds; ++r) {
        int tm, tn;
        vn_tile_coords(r * G + b, tiles_m, tiles_n, order, tm, tn);
        f32x16 acc[MI][NI];
        vn_acc_zero(acc);
        vn_gemm_mac<BM, BN, BF>(p, lds, tm * BM, tn * BN, 0, nk, acc);
        vn_gemm_epilogue<BM, BN, EPI>(p, tm * BM, tn * BN, acc);
        __syncthreads();
    }
    const int tile0 = dp_rounds * G;
    const long total = (long)(ntiles - tile0) * nk;
    long it = total * b / G;
    const long end = total * (b + 1) / G;

    while (it < end) {
        const int tile = (int)(it / nk);                 // index inside the stream-K part
        const int kb = (int)(it - (long)tile * nk);
        const long rem = end - it;
        const int ke = (long)(nk - kb) < rem ? nk : kb + (int)rem;
        int tm, tn;
        vn_tile_coords(tile0 + tile, tiles_m, tiles_n, order, tm, tn);
        const int m0 = tm * BM, n0 = tn * BN;
        f32x16 acc[MI][NI];
        vn_acc_zero(acc);
        vn_gemm_mac<BM, BN, BF>(p, lds, m0, n0, kb, ke, acc);

        if (kb != 0) {
            // a later k-range of a tile that an earlier block owns: publish the partial sums (only ever the FIRST
            // segment of a block, so one slab per block suffices)
            // write-through (sc1) 16-byte stores: the data goes to memory past this XCD's L2, so no release fence
            // (no L2 write-back of everybody's dirty C tiles) is needed before the flag (guide G16 recipe R1)
            __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                slabs + (size_t)b * Cfg::SLAB_FLOATS, 0, Cfg::SLAB_FLOATS * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc,
                                                               (((i * NI + j) * 4 + q) * 256 + tid) * 16, 0, 16 /*sc1*/);
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // EVERY storing wave drains its stores
            __syncthreads();
            if (tid == 0) __hip_atomic_store((gu32*)(flags + b), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (ke < nk) {
                // owner of the tile's first k-range: add the later ranges (blocks b+1, b+2, ... in k order)
                const long tile_end = (long)(tile + 1) * nk;
                for (int pb = b + 1; pb < G; ++pb) {
                    if (tid == 0) {
                        unsigned spins = 0;
                        while (__hip_atomic_load((gu32*)(flags + pb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                            __builtin_amdgcn_s_sleep(8);
                            if (++spins > SK_SPIN_LIMIT) {   // never hang the GPU: record and fall through
                                __hip_atomic_store((gu32*)errword, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                break;
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    }
                    __syncthreads();
                    const f32x4* slab = (const f32x4*)(slabs + (size_t)pb * Cfg::SLAB_FLOATS);
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const f32x4 v = slab[((i * NI + j) * 4 + q) * 256 + tid];
                                acc[i][j][4 * q] += v[0];
                                acc[i][j][4 * q + 1] += v[1];
                                acc[i][j][4 * q + 2] += v[2];
                                acc[i][j][4 * q + 3] += v[3];
                            }
                    __syncthreads();
                    if (tid == 0) __hip_atomic_store((gu32*)(flags + pb), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (total * (pb + 1) / G >= tile_end) break;      // block pb reached the end of this tile
                }
            }
            vn_gemm_epilogue<BM, BN, EPI>(p, m0, n0, acc);
        }
        it += ke - kb;
    }
}

// ---------------------------------------------------------------------------------------------
// two-pass split-K for SMALL-M shapes (one sequence: M = 575 rows -> 180 tiles of 64 x 64 cannot fill 256 CUs, and each
// block walks all K / 32 k-tiles serially at one wave per SIMD).  Pass 1: grid (tiles, S), split s accumulates k-tiles
// [s*nk/S, (s+1)*nk/S) and stores its raw partial tile to workspace[s][M][N]; pass 2 sums the S partials in fixed order and
// applies the epilogue (residual add / plain store).  No in-kernel hand-off: the per-launch fix-up of the stream-K path
// (slab write-through, flag, acquire) costs more than these kernels last.  Deterministic.
// ---------------------------------------------------------------------------------------------
template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void vn_gemm_f32_splitk_kernel(vn_gemm_args p, int tiles_m, int tiles_n, int order,
                                                                    int nsplit, float* __restrict__ partial) {
    using Cfg = GemmCfg<BM, BN>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int tm, tn;
    vn_tile_coords(vn_xcd_remap(blockIdx.x, tiles_m * tiles_n), tiles_m, tiles_n, order, tm, tn);
    const int nk = vn_ktiles<false>(p), sp = blockIdx.y;
    const int kb = (int)((long)nk * sp / nsplit), ke = (int)((long)nk * (sp + 1) / nsplit);
    f32x16 acc[Cfg::MI][Cfg::NI];
    vn_acc_zero(acc);
    if (ke > kb) vn_gemm_mac<BM, BN, false>(p, lds, tm * BM, tn * BN, kb, ke, acc);
    vn_gemm_args q = p;                                   // raw store of this split's partial tile
    q.C = partial + (size_t)sp * p.M * p.N;
    q.ldc = p.N;
    vn_gemm_epilogue<BM, BN, VN_EPI_STORE>(q, tm * BM, tn * BN, acc);
}

template <bool RESID>
__global__ __launch_bounds__(256) void vn_splitk_reduce_kernel(const float* __restrict__ partial, int nsplit, float* __restrict__ C,
                                                               int M, int N4, int ldc4) {
    const long total = (long)M * N4, plane = total;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
        f32x4 a = ((const f32x4*)partial)[i];
        for (int sp = 1; sp < nsplit; ++sp) {
            const f32x4 b = ((const f32x4*)partial)[i + sp * plane];
            a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
        }
        const long row = i / N4, c4 = i - row * N4;
        f32x4* dst = (f32x4*)C + row * ldc4 + c4;
        if (RESID) {
            const f32x4 r = *dst;
            a[0] += r[0]; a[1] += r[1]; a[2] += r[2]; a[3] += r[3];
        }
        *dst = a;
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// tuning hooks (scripts/gemm_sweep.py): force a tile / scheduler / tile order
// (state in vn_ctx::tune: f32_order, f32_stagger, f32_bm / f32_bn, f32_sched -1 auto / 0 data-parallel / 1 stream-K, f32_splitk
// -1 auto / 0 off / >= 2 forced split count for the small-M shapes; defaults from VN_GEMM_ORDER / _TILE / _SCHED / _SPLITK when the
// context is created)
extern "C" int vn_debug_gemm_config(vn_ctx* ctx, int bm, int bn, int order) {
    if (!ctx) return VN_ERR_INVALID;
    vn_tune& t = ctx->tune;
    t.f32_bm = bm; t.f32_bn = bn;
    // bits: [0] walk, [2:1] sched + 1, [15:8] stagger + 1
    if (order >= 0) { t.f32_order = order & 1; t.f32_sched = ((order >> 1) & 3) - 1; if (order >> 8) t.f32_stagger = ((order >> 8) & 0xff) - 1; }
    ++t.epoch;
    return VN_OK;
}

// stream-K workspace (per process; GEMMs of one ctx run on one stream at a time, see vampnet_hip.h)
#define SK_MAX_BLOCKS 512
// per-context workspace: ctx->sk_flags = [SK_MAX_BLOCKS] flags + [1] error word
static int sk_workspace(vn_ctx* ctx) {
    if (ctx->sk_slabs) return VN_OK;
    VN_HIP_CHECK(ctx, vn_dev_malloc((void**)&ctx->sk_slabs, (size_t)SK_MAX_BLOCKS * 128 * 128 * sizeof(float)));
    VN_HIP_CHECK(ctx, vn_dev_malloc((void**)&ctx->sk_flags, (SK_MAX_BLOCKS + 64) * sizeof(unsigned)));
    VN_HIP_CHECK(ctx, hipMemset(ctx->sk_flags, 0, (SK_MAX_BLOCKS + 64) * sizeof(unsigned)));
    VN_HIP_CHECK(ctx, hipDeviceSynchronize());      // the fill runs on the null stream; the first launch may be on a stream that does not wait for it
    return VN_OK;
}

// stream-K spin-limit indicator (a finisher gave up waiting for a partial-sum slab): 0 = healthy
extern "C" int vn_health_check(vn_ctx* ctx, void* stream) {
    if (!ctx) return VN_ERR_INVALID;
    if (!ctx->sk_flags) return VN_OK;
    unsigned w = 0;
    VN_HIP_CHECK(ctx, hipMemcpyAsync(&w, ctx->sk_flags + SK_MAX_BLOCKS, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream));
    VN_HIP_CHECK(ctx, hipStreamSynchronize((hipStream_t)stream));
    if (w) return vn_fail(ctx, VN_ERR_HIP, "stream-K GEMM: a tile owner timed out waiting for a partial-sum slab%s", "");
    return VN_OK;
}

template <int BM, int BN, int EPI, bool BF = false>
static int launch_cfg(vn_ctx* ctx, const vn_gemm_args& a, bool streamk, hipStream_t s) {
    using Cfg = GemmCfg<BM, BN>;
    const int tiles_m = vn_cdiv(a.M, BM), tiles_n = vn_cdiv(a.N, BN);
    if (streamk) {
        int rc = sk_workspace(ctx);
        if (rc) return rc;
    }
    const double n_out = (EPI == VN_EPI_GEGLU) ? a.N / 2 : a.N;
    const double bytes = (BF ? 2.0 : 4.0) * ((double)a.M * a.K + (double)a.N * a.K) +
                         4.0 * (double)a.M * n_out * (EPI == VN_EPI_RESIDUAL ? 2 : 1);
    const int pi = vn_prof_pre(ctx, BF ? 3 : 0, 2.0 * a.M * (double)a.N * a.K, s, bytes);
    if (streamk) {
        const long total = (long)tiles_m * tiles_n * ((BF ? a.K / 2 : a.K) / BK);
        int G = SK_MAX_BLOCKS;
        if (total < G) G = (int)total;
        hipLaunchKernelGGL((vn_gemm_f32_sk_kernel<BM, BN, EPI, BF>), dim3(G), dim3(256), Cfg::LDS_BYTES, s, a, tiles_m,
                           tiles_n, ctx->tune.f32_order | (ctx->tune.f32_stagger << 8), ctx->sk_slabs, ctx->sk_flags, ctx->sk_flags + SK_MAX_BLOCKS);
    } else {
        hipLaunchKernelGGL((vn_gemm_f32_kernel<BM, BN, EPI, BF>), dim3(tiles_m * tiles_n), dim3(256), Cfg::LDS_BYTES, s, a,
                           tiles_m, tiles_n, ctx->tune.f32_order);
    }
    vn_prof_post(ctx, pi, s);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

// Cost model, in units of (tile area / efficiency), calibrated on MI355X (profiles/history/r01_gemm_sweep.txt):
//   a CU retires one block-tile per (bm*bn/eff) whether it hosts 1 or 2 blocks (shared matrix pipes), so
//   data-parallel costs ceil(blocks/256) rounds; stream-K costs the exact share + a fix-up overhead of about
//   0.2 x (128x128 tile at K=1280) = ~16-20 us (slab write + read, pipeline refill per segment).
static double dp_cost(int M, int N, int K, int bm, int bn, double eff) {
    const double blocks = (double)vn_cdiv(M, bm) * vn_cdiv(N, bn);
    return ceil(blocks / 256.0) * (double)bm * bn / eff;
}
static double sk_cost(int M, int N, int K, int bm, int bn, double eff) {
    const double blocks = (double)vn_cdiv(M, bm) * vn_cdiv(N, bn);
    return blocks / 256.0 * (double)bm * bn / eff + 0.2 * 16384.0 * 1280.0 / (double)K;
}

// small-M residual / plain-store GEMMs: two-pass split-K (see vn_gemm_f32_splitk_kernel)
template <int EPI>
static int launch_splitk(vn_ctx* ctx, const vn_gemm_args& a, int nsplit, hipStream_t s) {
    using Cfg = GemmCfg<64, 64>;
    int rc = sk_workspace(ctx);
    if (rc) return rc;
    const int tiles_m = vn_cdiv(a.M, 64), tiles_n = vn_cdiv(a.N, 64);
    const double bytes = 4.0 * ((double)a.M * a.K + (double)a.N * a.K) + 4.0 * (double)a.M * a.N * (EPI == VN_EPI_RESIDUAL ? 2 : 1);
    const int pi = vn_prof_pre(ctx, 0, 2.0 * a.M * (double)a.N * a.K, s, bytes);
    hipLaunchKernelGGL((vn_gemm_f32_splitk_kernel<64, 64>), dim3(tiles_m * tiles_n, nsplit), dim3(256), Cfg::LDS_BYTES, s, a, tiles_m,
                       tiles_n, ctx->tune.f32_order, nsplit, ctx->sk_slabs);
    const long total4 = (long)a.M * (a.N / 4);
    const int blocks = (int)((total4 + 255) / 256 < 2048 ? (total4 + 255) / 256 : 2048);
    hipLaunchKernelGGL((vn_splitk_reduce_kernel<EPI == VN_EPI_RESIDUAL>), dim3(blocks), dim3(256), 0, s, ctx->sk_slabs, nsplit, a.C,
                       a.M, a.N / 4, a.ldc / 4);
    vn_prof_post(ctx, pi, s);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

int vn_launch_splitk_reduce(vn_ctx* ctx, const float* partial, int nsplit, float* C, int M, int N, int ldc, bool residual,
                            hipStream_t s) {
    if ((N & 3) || (ldc & 3)) return vn_fail(ctx, VN_ERR_INVALID, "split-K reduce: N and ldc must be multiples of 4%s", "");
    const long total4 = (long)M * (N / 4);
    const int blocks = (int)((total4 + 255) / 256 < 2048 ? (total4 + 255) / 256 : 2048);
    if (residual) hipLaunchKernelGGL((vn_splitk_reduce_kernel<true>), dim3(blocks), dim3(256), 0, s, partial, nsplit, C, M, N / 4, ldc / 4);
    else hipLaunchKernelGGL((vn_splitk_reduce_kernel<false>), dim3(blocks), dim3(256), 0, s, partial, nsplit, C, M, N / 4, ldc / 4);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

template <int EPI>
static int launch_epi(vn_ctx* ctx, const vn_gemm_args& a, hipStream_t s) {
    if constexpr (EPI == VN_EPI_RESIDUAL || EPI == VN_EPI_STORE) {
        // one sequence (M <= 640): 64 x 64 tiles fill at most 70 % of the CUs at one wave per SIMD -> split K so that every CU
        // hosts two blocks; needs fp32 operands, a 4-aligned row stride and room in the 32 MiB slab workspace
        if (!a.bf16 && !ctx->tune.f32_bm && ctx->tune.f32_sched != 1 && ctx->tune.f32_splitk != 0 && a.M <= 1280 && a.K >= 1024 && a.N % 4 == 0 && a.ldc % 4 == 0) {
            const int tiles = vn_cdiv(a.M, 64) * vn_cdiv(a.N, 64);
            int ns = ctx->tune.f32_splitk >= 2 ? ctx->tune.f32_splitk : (tiles >= 512 ? 0 : (tiles >= 256 ? 2 : (tiles >= 128 ? 4 : 8)));
            while (ns > 1 && (a.K / BK) / ns < 4) ns >>= 1;
            if (ns >= 2 && (size_t)ns * a.M * a.N <= (size_t)SK_MAX_BLOCKS * 128 * 128) return launch_splitk<EPI>(ctx, a, ns, s);
        }
    }
    int bm = ctx->tune.f32_bm, bn = ctx->tune.f32_bn;
    bool sk = ctx->tune.f32_sched == 1;
    if (!bm) {
        const bool geglu = (EPI == VN_EPI_GEGLU);
        struct { int bm, bn; double eff; } cand[] = {{128, 128, 1.00}, {64, 128, 0.94}, {128, 64, 0.93}, {64, 64, 0.89}};
        double best = 1e300;
        for (auto& c : cand) {
            if (geglu && c.bn != 128) continue;
            if (ctx->tune.f32_sched != 1) {
                const double cost = dp_cost(a.M, a.N, a.K, c.bm, c.bn, c.eff);
                if (cost < best) { best = cost; bm = c.bm; bn = c.bn; sk = false; }
            }
            if (ctx->tune.f32_sched != 0) {
                const double cost = sk_cost(a.M, a.N, a.K, c.bm, c.bn, c.eff);
                if (cost < best) { best = cost; bm = c.bm; bn = c.bn; sk = true; }
            }
        }
    }
    if (a.bf16) {      // fast mode: 128x128 data-parallel (kernels last 25-80 us: the ~20 us stream-K fix-up does not pay;
                       // measured DP 590-770 TF vs SK 400-670 TF on the model's shapes)
        return launch_cfg<128, 128, EPI, true>(ctx, a, ctx->tune.f32_sched == 1, s);
    }
    if (bm == 128 && bn == 128) return launch_cfg<128, 128, EPI>(ctx, a, sk, s);
    if (bm == 64 && bn == 128) return launch_cfg<64, 128, EPI>(ctx, a, sk, s);
    if constexpr (EPI != VN_EPI_GEGLU) {
        if (bm == 128 && bn == 64) return launch_cfg<128, 64, EPI>(ctx, a, sk, s);
        if (bm == 64 && bn == 64) return launch_cfg<64, 64, EPI>(ctx, a, sk, s);
    }
    return vn_fail(ctx, VN_ERR_INVALID, "gemm: unsupported tile %s%ldx%ld", "", bm, bn);
}

int vn_launch_gemm_f32(vn_ctx* ctx, const vn_gemm_args& a, int epilogue, hipStream_t s) {
    if (a.bf16 >= 2) return vn_launch_gemm_x3(ctx, a, epilogue, s);            // 2: bf16x3 planes, 3: f16x2 planes
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return vn_fail(ctx, VN_ERR_INVALID, "gemm: empty problem%s", "");
    if (a.K % (a.bf16 ? 2 * BK : BK) != 0) return vn_fail(ctx, VN_ERR_INVALID, "gemm: K=%s%ld must be a multiple of 32 (64 for bf16)", "", a.K);
    if (a.N % 64 != 0) return vn_fail(ctx, VN_ERR_INVALID, "gemm: N=%s%ld must be a multiple of 64", "", a.N);
    switch (epilogue) {
        case VN_EPI_STORE: return launch_epi<VN_EPI_STORE>(ctx, a, s);
        case VN_EPI_BIAS: return launch_epi<VN_EPI_BIAS>(ctx, a, s);
        case VN_EPI_RESIDUAL: return launch_epi<VN_EPI_RESIDUAL>(ctx, a, s);
        case VN_EPI_GEGLU:
            if (a.N % 128 != 0) return vn_fail(ctx, VN_ERR_INVALID, "gemm/geglu: N=%s%ld must be a multiple of 128", "", a.N);
            return launch_epi<VN_EPI_GEGLU>(ctx, a, s);
        case VN_EPI_QKV: return launch_epi<VN_EPI_QKV>(ctx, a, s);
    }
    return vn_fail(ctx, VN_ERR_INVALID, "gemm: unknown epilogue %s%ld", "", epilogue);
}
