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
// Softmax probabilities are split into their three planes in registers (exact for any fp32 value).
//
// f16x2 precision (round 3; NP = 2 in the templates below): the same kernels on fp16 TWO-plane operands, h0 = fp16(x), h1 = fp16(x - h0)
// WITHOUT the 2^11 of the GEMM format, so that the three kept products (a0 b1, a1 b0, a0 b0) go into the ONE accumulator the six
// bf16 products use (a second S / O accumulator would cost the third wave per SIMD): half the MFMAs, 2/3 of the tile bytes, 28 fewer
// VGPRs.  An unscaled remainder falls into fp16's subnormal range for |x| < 0.125 — exact to 2^-25 absolute there instead of 2^-22
// relative — which moves a score by < 3e-7; the softmax weights (<= e^6) and V carry a factor 16 each (16 P < 6.5e3), divided out
// with l at the end.  Measured against float64 (tests/test_gpu_kernels.py): 1.2e-6 max, the bf16x3 operands 1.1e-6, the fp32-input
// kernel 2.6e-6.  B = 8: 58.8 us vs 86.5 us (profiles/history/r03_attention_f16x2_vs_bf16x3.txt).
//
// Per wave: 32 query columns.
//   S^T[key][q] = K[key][:] . Q[q][:]      A = K tile rows (LDS), B = Q (registers, loaded once)
//   O^T[d][q]   = V^T[d][key] . P^T[key][q]  A = V^T tile rows (LDS), B = P (the registers the softmax produced)
// MFMA row rho of S^T is mapped to key swap_bits_2_3(rho) (the A operand simply reads that K row): a lane (query j = lane & 31,
// half hh = lane >> 5) then holds keys 16 (r >> 3) + 8 hh + (r & 7), r = 0..15 — for each 16-key step of the second product
// EIGHT CONSECUTIVE keys, i.e. exactly its k-slots of the B operand, and the V^T operand is one 16-byte LDS read.
// K/V^T tiles (32 keys: 12 + 12 KiB for the three planes) come in by LDS-DMA (global_load_lds_dwordx4).  LDS images are
// lane-linear, so the bank swizzles sit on the DMA source address:
//   K rows are 128 B: slot s of row r lives at s ^ ((r >> 1) & 7);  V^T rows are 64 B: s ^ ((r >> 2) & 3)  (both give every
//   16-lane ds_read_b128 group 16 distinct 16-byte slots of the 256-byte bank row).
//
// Two work decompositions (vn_launch_attention_x3 picks by how many blocks exist):
//   * SHARED tiles (vn_attention_x3_kernel): block = 4 waves = 128 queries of one (b, h); the waves share every K / V^T tile
//     (double-buffered, one barrier per tile), three blocks per CU.  The many-sequence shape (B >= 4 at T = 575).
//   * KEY-SPLIT (vn_attention_x3_split_kernel<KS>): block = KS waves that all own the SAME 32 queries and walk DISJOINT key
//     tiles (wave w: tiles w, w + KS, ...), each staging its own tiles into a wave-private LDS region — no barrier in the main
//     loop — and the partial (m, l, O) triples are merged through LDS at the end in a fixed order (flash-decoding inside a
//     block).  One or two sequences: 18 x 20 q-blocks x KS waves fill the chip where 5 x 20 blocks of 128 queries leave 60 % of
//     the CUs idle, and a wave's serial chain is 19 / KS tiles long instead of 19 (round 2 fell back to the fp32-input MFMA
//     kernel here: 39.6 us at B = 1).
//
// Softmax arithmetic (round 3; the PMC of round 2 showed 7.6 VALU instructions per MFMA against ~4 that ride along for free,
// profiles/history/r02_final2_pmc_attention_x3.txt):
//   * the S^T accumulator is INITIALISED with the bias row (the MFMAs add the products onto it) instead of adding it afterwards;
//   * the running max is only raised when some score of the tile exceeds it by more than AX_THR (guide T13): P = exp(s - m_ref)
//     may then be as large as e^AX_THR, which costs nothing here — P is split EXACTLY into three bf16 planes whatever its size
//     and l / O accumulate in fp32 — and removes the 32 multiplies of O, the exp of the scale and their dependency from all
//     but the first tiles; when the reference does move, O AND l are rescaled before any P of the tile is formed;
//   * exp(t) = 2^hi * (1 + d), hi = fl(t log2 e), d = t - hi ln 2 by two fmas (Cody-Waite on the rounded product): the same
//     fp32-grade accuracy as vn_exp_neg with one instruction less per score.
// Algorithmic FLOPs: 4*T*T*64 per (b,h); executed on the bf16 pipe: 6x that.
// History of measured-and-dropped variants (software-pipelined loop, start staggers, 96-query blocks, six-wave blocks, s_setprio):
// DESIGN.md section 3 and profiles/history/r02_attention_x3_*.txt.
#include "attention_x3_dev.h"

// ---- shared-tile kernel: block = 4 waves x 32 queries of one (b, h), K / V^T tiles double-buffered for the whole block --------
// TRACE (tuning): wave 0 of EVERY block writes trace[blockIdx][8] = {wait, barrier, dma issue, tile math (s_memtime deltas summed
// over the tiles; full role only), entry time, exit time (low 32 bits of s_memtime), XCC id, HW_ID} — the launch's timeline.
// TRAIN (bf16x3 only; the training step's forward, train.hip): probability dropout from the counter-based stream `drop` (row = global
// (b, h, query), column = key — the stream the backward kernels of attention_train_x3.hip recompute) and the log-sum-exp of every
// score row written to lse[b][h][t] (transformer.py:234-254, :250).
template <int NW, bool TRACE = false, int NP = 3, bool TRAIN = false>
__global__ __launch_bounds__(NW * 64, 3) void vn_attention_x3_kernel(const uint16_t* __restrict__ q16, const uint16_t* __restrict__ k16,
                                                                    long plane_qk, const uint16_t* __restrict__ vt16, long plane_vt,
                                                                    const float* __restrict__ bias_full, float* __restrict__ out,
                                                                    uint16_t* __restrict__ out16, long plane16, int B, int H, int T,
                                                                    int stagger, unsigned* __restrict__ trace, float* __restrict__ lse,
                                                                    vn_drop drop, unsigned k_bytes, unsigned v_bytes) {
    static_assert(!TRAIN || NP == 3, "the training forward runs on bf16x3 operands");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int AXS = AX_STAGE_FLOATS_NP(NP);          // floats of one stage: NP K plane tiles, then NP V^T plane tiles
    float* bt = smem + 2 * AXS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const ax_lane L = ax_lane_init(lane);
    const bool tail_prio = (stagger >> 16) & 1;             // set by the launcher when the grid exceeds the resident slots
    stagger &= 0xffff;
    // Roles.  A head's queries are cut into 128-query blocks; a remainder of 1..64 queries (T = 575: 63) gets a TAIL block that
    // uses its four waves as 2 query sub-blocks x 2 KEY HALVES (half the tiles per wave, partial results merged through LDS) and
    // therefore lasts about half as long as a full block.  Tail blocks take the HIGHEST block indices, so they are dispatched
    // last: at B = 8 the 800 blocks no longer need a second round of 19-tile blocks on the 768 slots (three per CU) — the 32 blocks
    // that start late are short ones and, at wave priority 3 (see the tail role), end before the CUs that hold three full blocks
    // do (round 2 / 3 probes: the tail round was ~30 % of the launch).  Both index ranges are walked XCD-aware so that a head's
    // blocks share one L2.
    const int nqbf = T / (NW * 32), rq = T - nqbf * (NW * 32);
    const bool split_tail = rq > 0 && rq <= 64;
    const int nqb = nqbf + ((rq > 0 && !split_tail) ? 1 : 0);           // full-role blocks per (b, h)
    const int n_full = nqb * H * B;
    const bool tail_role = (int)blockIdx.x >= n_full;
    const int lid = tail_role ? ax_walk(blockIdx.x - n_full, gridDim.x - n_full) : ax_walk(blockIdx.x, n_full);
    const int qb = tail_role ? nqbf : lid % nqb;
    const int hbi = tail_role ? lid : lid / nqb;
    const int h = hbi % H, b = hbi / H;
    // key tiles of this item in global token rows: tile g holds tokens 32 g .. 32 g + 31, key index t = m - b T
    const int m_lo = b * T;
    const int g_lo = m_lo / AX_KT, NT = (m_lo + T - 1) / AX_KT - g_lo + 1;
    const int MT = (B * T + AX_KT - 1) / AX_KT;
    const size_t head = (size_t)b * H + h;
    const uint16_t* Qp = q16 + head * (size_t)T * VN_DHEAD;
    const int q0 = qb * (NW * 32) + (tail_role ? (wave & 1) : wave) * 32;
    const bool active = q0 < T;                         // waves past the end only help with the DMA and the barriers
    const int qrow = q0 + L.l31;
    const int qrow_c = qrow < T ? qrow : T - 1;

    unsigned long long tr_t = 0, tr_start = 0;
    unsigned tr_acc[4] = {0, 0, 0, 0};
    if constexpr (TRACE) { tr_t = tr_start = __builtin_readcyclecounter(); }
    auto trace_out = [&]() {
        if constexpr (TRACE) {
            if (trace && wave == 0 && lane == 0) {
                unsigned* t = trace + (size_t)blockIdx.x * 8;
#pragma unroll
                for (int i = 0; i < 4; ++i) t[i] = tr_acc[i];
                t[4] = (unsigned)tr_start;
                t[5] = (unsigned)__builtin_readcyclecounter();
                t[6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);           // HW_REG_XCC_ID
                t[7] = __builtin_amdgcn_s_getreg((31 << 11) | 4);            // HW_REG_HW_ID
            }
        }
    };

    const int nb = 2 * T - 1;
    for (int i = tid; i < nb; i += NW * 64) bt[i] = bias_full[(size_t)h * nb + i];

    f32x4 qf[NP][4];
    ax_load_q<NP>(qf, Qp, plane_qk, qrow_c, L.hh);
    uint32_t rk = 0;
    if constexpr (TRAIN) rk = vn_drop_rowkey(drop, (long)head * T + qrow_c);

    // LDS-DMA: 24 wave-instructions of 1 KiB per stage (K: 3 planes x 4, each 8 rows x 128 B; V^T: 3 planes x 4, each 16 rows x 64 B);
    // wave w issues piece w of every plane tile, as buffer loads: descriptor (SGPRs) + ONE per-lane byte offset per operand that
    // is the same for every plane and tile (VGPR) + a scalar offset for plane / tile — a tile costs six scalar adds, no vector
    // address arithmetic and two address VGPRs in all (the flat-address form kept six 64-bit per-lane pointers live, which the
    // register allocator spilled once the softmax changed shape).  K rows are NOT clamped to
    // the item: the rows of a first / last tile that belong to the neighbouring items (or to the q planes in front of / the
    // 32-row padding behind the k planes) are read as they are and masked to -inf before the softmax.
    static_assert(NW == 4, "one piece of every plane tile per wave");
    const int krow = 8 * wave + (lane >> 3), vrow = 16 * wave + (lane >> 2);
    const unsigned kvoff = (unsigned)(krow * VN_DHEAD + ((lane & 7) ^ ((krow >> 1) & 7)) * 8) * 2u;      // bytes inside a K tile
    const unsigned vvoff = (unsigned)(vrow * AX_KT + ((lane & 3) ^ ((vrow >> 2) & 3)) * 8) * 2u;         // bytes inside a V^T tile
    const ax_src src = ax_src_init(k16, vt16, plane_qk, plane_vt, b, h, H, T, m_lo, g_lo, MT, k_bytes, v_bytes);
    auto stage = [&](int buf, int kt) {                       // tile kt -> stage buf
        if (kt >= NT) return;
        float* base = smem + buf * AXS + wave * 256;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            ax_dma(src.krs, base + (4 * p) * 256, kvoff, src.k0 + p * src.kplane + kt * (AX_KT * VN_DHEAD * 2));
            ax_dma(src.vrs, base + (4 * NP + 4 * p) * 256, vvoff, src.v0 + p * src.vplane + kt * (VN_DHEAD * AX_KT * 2));
        }
    };

    float m_run = -INFINITY, l_run = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;

    if (tail_role) {
        // waves (qs, kh): query sub-block qs = wave & 1, key half kh = wave >> 1.  The two stages hold, SINGLE-buffered, the tile of
        // key half 0 (tile i) and of key half 1 (tile NH + i) of the same iteration; all four waves stage both (piece `wave` of every
        // plane tile, as in the full role).  Two barriers per iteration: tiles landed / tiles read.
        // The DMA of the next iteration is issued as soon as BOTH waves of a key half have read the operand it overwrites (K after the
        // S^T products, V^T after PV), so it flies under the softmax / the next S^T instead of being waited for (a first version
        // issued it after the whole tile: the exposed L2 latency under load made a 10-tile tail block last as long as a 19-tile full
        // block).  Every wave always issues 6 K and 6 V^T pieces per iteration (tiles past the end re-fetch the last tile), so the
        // counted vmcnt below is exact.
        // When some blocks have to wait for a slot, tail blocks run at wave priority 3.  The SQ issues oldest-first, so a tail block that starts late (the 32 of the 800 blocks
        // at B = 8 that find no free slot) was starved by the two older full blocks of its CU: 88k cycles instead of the 52k it
        // takes alone, ending 15-40k cycles after every other CU was idle.  With priority the tail blocks of the first wave end at
        // ~75k, the late ones start at ~70k and end at ~130k — before the CUs that hold three full blocks (~160k) — and the launch
        // is 7 % shorter (profiles/history/r03_attention_x3_timeline.txt).  When every block is resident from the start the priority only
        // slows the full blocks that bound the launch (B = 3: 47.0 vs 41.9 us), hence the launcher's condition.
        if (tail_prio) __builtin_amdgcn_s_setprio(3);
        const int kh = wave >> 1, NH = (NT + 1) >> 1;
        auto stage_op = [&](int kt0, int kt1, bool v_op) {
            kt0 = kt0 < NT ? kt0 : NT - 1;
            kt1 = kt1 < NT ? kt1 : NT - 1;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                float* d0 = smem + wave * 256 + (v_op ? 4 * NP + 4 * p : 4 * p) * 256;
                if (v_op) {
                    ax_dma(src.vrs, d0, vvoff, src.v0 + p * src.vplane + kt0 * (VN_DHEAD * AX_KT * 2));
                    ax_dma(src.vrs, d0 + AXS, vvoff, src.v0 + p * src.vplane + kt1 * (VN_DHEAD * AX_KT * 2));
                } else {
                    ax_dma(src.krs, d0, kvoff, src.k0 + p * src.kplane + kt0 * (AX_KT * VN_DHEAD * 2));
                    ax_dma(src.krs, d0 + AXS, kvoff, src.k0 + p * src.kplane + kt1 * (AX_KT * VN_DHEAD * 2));
                }
            }
        };
        stage_op(0, NH, false);
        stage_op(0, NH, true);
        const float* Ks = smem + kh * AXS;
        const float* Vs = Ks + NP * AX_PLANE_FLOATS;
        for (int i = 0; i < NH; ++i) {
            const int kt = kh ? NH + i : i;
            const bool valid = active && kt < NT, more = i + 1 < NH;
            const int key0 = (g_lo + kt) * AX_KT - m_lo;
            const bool full = key0 >= 0 && key0 + AX_KT <= T;
            f32x16 sacc;
            f32x4 pf[NP][2];
            // this wave's K pieces of the iteration landed (V^T may still fly: 2 NP pieces); lgkmcnt: first time round, its part of the
            // bias table
            if constexpr (NP == 3) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            AX_RAW_BARRIER();
            if (valid) {
                if (full) ax_bias_init<true>(sacc, bt, key0, L.hh, qrow_c, T);
                else ax_bias_init<false>(sacc, bt, key0, L.hh, qrow_c, T);
                ax_qk<NP>(sacc, Ks, qf, L);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            AX_RAW_BARRIER();                                        // both K stages have been read by everybody
            if (more) stage_op(i + 1, NH + i + 1, false);
            if (valid) {
                if constexpr (TRAIN) ax_softmax_train(sacc, pf, m_run, l_run, o, key0, L.hh, T, full, drop, rk);
                else ax_softmax<NP>(sacc, pf, m_run, l_run, o, key0, L.hh, T, full);
            }
            if (more) {                                                      // V^T of this iteration landed (the next K flies)
                if constexpr (NP == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            AX_RAW_BARRIER();
            if (valid) ax_pv<NP>(o, Vs, pf, L);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            AX_RAW_BARRIER();                                        // both V^T stages have been read
            if (more) stage_op(i + 1, NH + i + 1, true);
        }
        // merge the two key halves of each query sub-block (fixed order: half 0, then half 1), as the key-split kernel does.  Image per
        // wave in the (free) stages: 8 groups of 64 lanes x 16 B, then m, l per lane
        float* mine = smem + wave * 2304;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(f32x4*)(mine + ((dt * 4 + g) * 64 + lane) * 4) = f32x4{o[dt][4 * g], o[dt][4 * g + 1], o[dt][4 * g + 2], o[dt][4 * g + 3]};
        mine[2048 + lane] = m_run;
        mine[2112 + lane] = l_run;
        __syncthreads();
        const float* im0 = smem + (wave & 1) * 2304;            // key half 0 of this query sub-block
        const float* im1 = smem + ((wave & 1) + 2) * 2304;      // key half 1
        const float m0 = im0[2048 + lane], m1 = im1[2048 + lane];
        const float m_all = fmaxf(m0, m1);
        const float s0 = vn_exp_neg(m0 - m_all), s1 = vn_exp_neg(m1 - m_all);      // a half without tiles holds m = -inf, l = 0
        const float l_all = im0[2112 + lane] * s0 + im1[2112 + lane] * s1;
        const float l_tot = (l_all + __shfl_xor(l_all, 32)) * (NP == 2 ? 16.0f : 1.0f);       // f16: l' = 16 l, O' = 256 O
        if (active && qrow < T) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int G = kh * 4 + i;                       // this wave finishes four of the eight column groups
                const f32x4 a = *(const f32x4*)(im0 + (G * 64 + lane) * 4) * s0 + *(const f32x4*)(im1 + (G * 64 + lane) * 4) * s1;
                ax_store4(a, l_tot, out, out16, plane16, (long)b * T + qrow, h, H, L.hh, G >> 2, G & 3);
            }
            if constexpr (TRAIN) { if (lane < 32 && kh == 0) lse[head * T + qrow] = m_all + logf(l_tot); }
        }
        trace_out();
        return;
    }

    if (stagger > 0) {                                      // de-phase the blocks that share this CU (tuning)
        const int slot = __builtin_amdgcn_s_getreg((3 << 11) | 4) % 3;         // HW_REG_HW_ID bits [3:0]: wave slot in the SIMD
        for (int i = 0; i < slot * stagger; ++i) __builtin_amdgcn_s_sleep(1);
    }
    const bool tracing = TRACE && trace && wave == 0;
    auto tick = [&](int slot) {
        if constexpr (TRACE) {
            if (tracing) {
                const unsigned long long now = __builtin_readcyclecounter();
                tr_acc[slot] += (unsigned)(now - tr_t);
                tr_t = now;
            }
        }
    };
    if constexpr (TRACE) { tr_t = __builtin_readcyclecounter(); }
    stage(0, 0);
    for (int kt = 0; kt < NT; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pieces of tile kt have landed
        tick(0);
        __syncthreads();                                        // all pieces landed; everybody is done with tile kt - 1
        tick(1);
        stage((kt + 1) & 1, kt + 1);
        tick(2);
        if (!active) continue;
        const float* St = smem + (kt & 1) * AXS;
        ax_tile<NP, TRAIN>(St, St + NP * AX_PLANE_FLOATS, bt, qf, L, m_run, l_run, o, (g_lo + kt) * AX_KT - m_lo, qrow_c, T, drop, rk);
        tick(3);
    }
    // ---- finish: the two lanes of a query add their row sums; normalise; store
    const float l_tot = (l_run + __shfl_xor(l_run, 32)) * (NP == 2 ? 16.0f : 1.0f);
    if (active && qrow < T) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 a = {o[dt][4 * g], o[dt][4 * g + 1], o[dt][4 * g + 2], o[dt][4 * g + 3]};
                ax_store4(a, l_tot, out, out16, plane16, (long)b * T + qrow, h, H, L.hh, dt, g);
            }
        if constexpr (TRAIN) { if (lane < 32) lse[head * T + qrow] = m_run + logf(l_tot); }
    }
    trace_out();
}

// ---- key-split kernel: block = KS waves on the SAME 32 queries, wave w walks key tiles w, w + KS, ... ---------------------------
// LDS: KS wave-private stages (K 12 KiB + V^T 12 KiB, single-buffered each: the K DMA of the wave's next tile is issued right after
// the S^T products have read the current one and flies under softmax + PV, the V^T DMA after PV and flies under the next S^T +
// softmax; vmcnt counts this wave's own pieces in issue order, so no barrier is needed), then the bias table.  After the loop every
// wave leaves (m, l, O) in its stage, and after ONE barrier wave w merges 8 / KS of the eight 4-column groups in the fixed order
// w' = 0 .. KS - 1 (deterministic), normalises and stores them.
template <int KS, int NP = 3>
__global__ __launch_bounds__(KS * 64, 2) void vn_attention_x3_split_kernel(const uint16_t* __restrict__ q16, const uint16_t* __restrict__ k16,
                                                                          long plane_qk, const uint16_t* __restrict__ vt16, long plane_vt,
                                                                          const float* __restrict__ bias_full, float* __restrict__ out,
                                                                          uint16_t* __restrict__ out16, long plane16, int B, int H, int T,
                                                                          unsigned k_bytes, unsigned v_bytes) {
    static_assert(KS == 1 || KS == 2 || KS == 4, "the merge hands 8 / KS column groups to every wave");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int AXS = AX_STAGE_FLOATS_NP(NP);
    float* bt = smem + KS * AXS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const ax_lane L = ax_lane_init(lane);
    const int nqb = (T + 31) / 32;
    const int lid = ax_walk(blockIdx.x, gridDim.x);
    const int qb = lid % nqb, h = (lid / nqb) % H, b = lid / (nqb * H);
    const int m_lo = b * T;
    const int g_lo = m_lo / AX_KT, NT = (m_lo + T - 1) / AX_KT - g_lo + 1;
    const int MT = (B * T + AX_KT - 1) / AX_KT;
    const size_t head = (size_t)b * H + h;
    const uint16_t* Qp = q16 + head * (size_t)T * VN_DHEAD;
    const int qrow = qb * 32 + L.l31;
    const int qrow_c = qrow < T ? qrow : T - 1;

    const int nb = 2 * T - 1;
    for (int i = tid; i < nb; i += KS * 64) bt[i] = bias_full[(size_t)h * nb + i];

    f32x4 qf[NP][4];
    ax_load_q<NP>(qf, Qp, plane_qk, qrow_c, L.hh);

    // this wave stages whole tiles: pieces w' = 0..3 of every plane.  K piece w' = rows 8 w' .. + 7: the swizzle key ((row >> 1) & 7)
    // = (4 w' + (lane >> 4)) & 7 differs between even and odd w', so two per-lane offsets; V^T piece w' = rows 16 w' .. + 15:
    // ((row >> 2) & 3) = (lane >> 4) & 3 for every w', one offset.  Piece w' adds 1 KiB to both source and destination.
    const int kr0 = lane >> 3, vr0 = lane >> 2;
    const unsigned kvoff_e = (unsigned)(kr0 * VN_DHEAD + ((lane & 7) ^ ((kr0 >> 1) & 7)) * 8) * 2u;
    const unsigned kvoff_o = (unsigned)(kr0 * VN_DHEAD + ((lane & 7) ^ (((kr0 >> 1) + 4) & 7)) * 8) * 2u;
    const unsigned vvoff = (unsigned)(vr0 * AX_KT + ((lane & 3) ^ ((vr0 >> 2) & 3)) * 8) * 2u;
    const ax_src src = ax_src_init(k16, vt16, plane_qk, plane_vt, b, h, H, T, m_lo, g_lo, MT, k_bytes, v_bytes);
    float* mine = smem + wave * AXS;
    auto stage_k = [&](int kt) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const unsigned so = src.k0 + p * src.kplane + kt * (AX_KT * VN_DHEAD * 2);
            float* dst = mine + (4 * p) * 256;               // the instruction offset moves BOTH the source and the LDS address
            ax_dma<0>(src.krs, dst, kvoff_e, so);
            ax_dma<1024>(src.krs, dst, kvoff_o, so);
            ax_dma<2048>(src.krs, dst, kvoff_e, so);
            ax_dma<3072>(src.krs, dst, kvoff_o, so);
        }
    };
    auto stage_v = [&](int kt) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const unsigned so = src.v0 + p * src.vplane + kt * (VN_DHEAD * AX_KT * 2);
            float* dst = mine + (4 * NP + 4 * p) * 256;
            ax_dma<0>(src.vrs, dst, vvoff, so);
            ax_dma<1024>(src.vrs, dst, vvoff, so);
            ax_dma<2048>(src.vrs, dst, vvoff, so);
            ax_dma<3072>(src.vrs, dst, vvoff, so);
        }
    };

    float m_run = -INFINITY, l_run = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;

    if (wave < NT) { stage_k(wave); stage_v(wave); }
    __syncthreads();                                            // the bias table is complete
    const float* Ks = mine;
    const float* Vs = mine + NP * AX_PLANE_FLOATS;
    for (int kt = wave; kt < NT; kt += KS) {
        const int key0 = (g_lo + kt) * AX_KT - m_lo;
        const bool full = key0 >= 0 && key0 + AX_KT <= T;
        const bool more = kt + KS < NT;
        f32x16 sacc;
        f32x4 pf[NP][2];
        if (full) ax_bias_init<true>(sacc, bt, key0, L.hh, qrow_c, T);
        else ax_bias_init<false>(sacc, bt, key0, L.hh, qrow_c, T);
        // K of this tile landed (its 4 NP V^T pieces may still fly)
        if constexpr (NP == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        ax_qk<NP, true>(sacc, Ks, qf, L);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // every K fragment has been read: the K stage is free
        __builtin_amdgcn_sched_barrier(0);
        if (more) stage_k(kt + KS);
        ax_softmax<NP>(sacc, pf, m_run, l_run, o, key0, L.hh, T, full);
        if (more) {                                             // V^T of this tile landed (the next K flies)
            if constexpr (NP == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        ax_pv<NP, true>(o, Vs, pf, L);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (more) stage_v(kt + KS);
    }

    // ---- merge the KS partial results.  Image per wave (its own stage, free now): 8 groups of 64 lanes x 16 B, then m, l per lane
    if constexpr (KS > 1) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(f32x4*)(mine + ((dt * 4 + g) * 64 + lane) * 4) = f32x4{o[dt][4 * g], o[dt][4 * g + 1], o[dt][4 * g + 2], o[dt][4 * g + 3]};
        mine[2048 + lane] = m_run;
        mine[2112 + lane] = l_run;
        __syncthreads();
        float m_all = -INFINITY;
#pragma unroll
        for (int w = 0; w < KS; ++w) m_all = fmaxf(m_all, smem[w * AXS + 2048 + lane]);
        float sc[KS], l_all = 0.f;
#pragma unroll
        for (int w = 0; w < KS; ++w) {                          // a wave without tiles holds m = -inf, l = 0: scale 0
            sc[w] = vn_exp_neg(smem[w * AXS + 2048 + lane] - m_all);
            l_all += smem[w * AXS + 2112 + lane] * sc[w];
        }
        const float l_tot = (l_all + __shfl_xor(l_all, 32)) * (NP == 2 ? 16.0f : 1.0f);       // f16: l' = 16 l, O' = 256 O
        if (qrow < T) {
#pragma unroll
            for (int i = 0; i < 8 / KS; ++i) {
                const int G = wave * (8 / KS) + i;              // uniform
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < KS; ++w) a += *(const f32x4*)(smem + w * AXS + (G * 64 + lane) * 4) * sc[w];
                ax_store4(a, l_tot, out, out16, plane16, (long)b * T + qrow, h, H, L.hh, G >> 2, G & 3);
            }
        }
    } else {
        const float l_tot = (l_run + __shfl_xor(l_run, 32)) * (NP == 2 ? 16.0f : 1.0f);
        if (qrow < T) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 a = {o[dt][4 * g], o[dt][4 * g + 1], o[dt][4 * g + 2], o[dt][4 * g + 3]};
                    ax_store4(a, l_tot, out, out16, plane16, (long)b * T + qrow, h, H, L.hh, dt, g);
                }
        }
    }
}

// ---- pair-split kernel (round 5; one or two sequences): block = 2 KH waves = 2 query sub-blocks of 32 x KH key parts ------------------
// The key-split kernel above gives every wave a private stage (24 KiB), so a CU holds six waves at most and a wave's chain of
// dependent tiles is 19 / 2 long at its best occupancy: 28 us at B = 1 for 4.5 us of matrix work, the chip 70 % empty (0.7 waves per
// SIMD).  Here the two query sub-blocks of a 64-query block SHARE the tile of their key part (the shared kernel's tail role, with KH
// parts instead of two halves): KH = 4 -> eight waves = two per SIMD on one CU, 4 x 24 KiB of single-buffered stages, a chain of
// ceil(19 / 4) = 5 tiles per wave.  All waves run the same four-barrier iteration (K landed / K read / V^T landed / V^T read), the DMA of
// the next iteration's K is issued right after the S^T products and flies under the softmax, the next V^T after PV.  Waves w and w ^ 1
// share key part w >> 1; wave w stages piece (w & 3) of every plane tile of two stages (waves 0-3: parts 0 / 1, waves 4-7: parts 2 / 3),
// so every wave issues 2 NP K and 2 NP V^T pieces per iteration and the counted vmcnt is exact.  The KH partial (m, l, O) triples of a
// query sub-block are merged through LDS in the fixed order 0 .. KH - 1 (deterministic), each wave finishing 8 / KH column groups.
template <int KH, int NP = 3>
__global__ __launch_bounds__(2 * KH * 64) void vn_attention_x3_pair_kernel(const uint16_t* __restrict__ q16, const uint16_t* __restrict__ k16,
                                                                          long plane_qk, const uint16_t* __restrict__ vt16, long plane_vt,
                                                                          const float* __restrict__ bias_full, float* __restrict__ out,
                                                                          uint16_t* __restrict__ out16, long plane16, int B, int H, int T,
                                                                          unsigned k_bytes, unsigned v_bytes) {
    static_assert(KH == 2 || KH == 4, "two stages per staging wave group, 8 / KH column groups per wave in the merge");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // a key part's LDS: TWO K buffers (tile i in buffer i & 1) and one V^T buffer — the K tile of iteration i + 2 is issued as soon as
    // QK(i) has read its buffer and has two iterations to land; V^T(i + 1) follows PV(i) and flies under QK + softmax of the next one
    constexpr int KB = NP * AX_PLANE_FLOATS, AXS = 3 * KB, NW = 2 * KH;
    float* bt = smem + KH * AXS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const ax_lane L = ax_lane_init(lane);
    const int nqb = (T + 63) / 64;
    const int lid = ax_walk(blockIdx.x, gridDim.x);
    const int qb = lid % nqb, hbi = lid / nqb;
    const int h = hbi % H, b = hbi / H;
    const int m_lo = b * T;
    const int g_lo = m_lo / AX_KT, NT = (m_lo + T - 1) / AX_KT - g_lo + 1;
    const int MT = (B * T + AX_KT - 1) / AX_KT;
    const size_t head = (size_t)b * H + h;
    const uint16_t* Qp = q16 + head * (size_t)T * VN_DHEAD;
    const int qs = wave & 1, kh = wave >> 1;
    const int q0 = qb * 64 + qs * 32;
    const bool active = q0 < T;                         // the second sub-block of a head's last block may be empty: it only stages
    const int qrow = q0 + L.l31;
    const int qrow_c = qrow < T ? qrow : T - 1;

    const int nb = 2 * T - 1;
    for (int i = tid; i < nb; i += NW * 64) bt[i] = bias_full[(size_t)h * nb + i];

    f32x4 qf[NP][4];
    ax_load_q<NP>(qf, Qp, plane_qk, qrow_c, L.hh);

    const int pw = wave & 3, sA = 2 * (wave >> 2);      // piece of every plane tile / first of the two stages this wave fills
    const int krow = 8 * pw + (lane >> 3), vrow = 16 * pw + (lane >> 2);
    const unsigned kvoff = (unsigned)(krow * VN_DHEAD + ((lane & 7) ^ ((krow >> 1) & 7)) * 8) * 2u;
    const unsigned vvoff = (unsigned)(vrow * AX_KT + ((lane & 3) ^ ((vrow >> 2) & 3)) * 8) * 2u;
    const ax_src src = ax_src_init(k16, vt16, plane_qk, plane_vt, b, h, H, T, m_lo, g_lo, MT, k_bytes, v_bytes);
    const int NH = (NT + KH - 1) / KH;
    auto stage_op = [&](int i, bool v_op) {             // iteration i: tiles sA NH + i and (sA + 1) NH + i (clamped: re-fetch the last tile)
        int kt0 = sA * NH + i, kt1 = (sA + 1) * NH + i;
        kt0 = kt0 < NT ? kt0 : NT - 1;
        kt1 = kt1 < NT ? kt1 : NT - 1;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            float* d0 = smem + sA * AXS + pw * 256 + (v_op ? 2 * KB : (i & 1) * KB) + 4 * p * 256;
            if (v_op) {
                ax_dma(src.vrs, d0, vvoff, src.v0 + p * src.vplane + kt0 * (VN_DHEAD * AX_KT * 2));
                ax_dma(src.vrs, d0 + AXS, vvoff, src.v0 + p * src.vplane + kt1 * (VN_DHEAD * AX_KT * 2));
            } else {
                ax_dma(src.krs, d0, kvoff, src.k0 + p * src.kplane + kt0 * (AX_KT * VN_DHEAD * 2));
                ax_dma(src.krs, d0 + AXS, kvoff, src.k0 + p * src.kplane + kt1 * (AX_KT * VN_DHEAD * 2));
            }
        }
    };

    float m_run = -INFINITY, l_run = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;

    stage_op(0, false);
    stage_op(0, true);
    if (NH > 1) stage_op(1, false);
    const float* Vs = smem + kh * AXS + 2 * KB;
    // The two waves of a SIMD are w and w + 4.  With all eight waves in lock-step their matrix phases (QK, PV) coincide and so do their
    // VALU phases (softmax): the pipe idles through every softmax.  Waves 4-7 (key parts 2 / 3: their own stages, their own DMA) therefore
    // run ONE barrier slot behind waves 0-3 — while one wave of a SIMD is in its softmax the other issues MFMAs — by passing one extra
    // barrier here (waves 0-3 pass one more after the loop); the waves that share a stage are in the same group.
    const bool late = KH == 4 && wave >= 4;
    if (late) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's part of the bias table is written
        AX_RAW_BARRIER();
    }
    for (int i = 0; i < NH; ++i) {
        const float* Ks = smem + kh * AXS + (i & 1) * KB;
        const int kt = kh * NH + i;
        const bool valid = active && kt < NT, more = i + 1 < NH;
        const int key0 = (g_lo + kt) * AX_KT - m_lo;
        const bool full = key0 >= 0 && key0 + AX_KT <= T;
        f32x16 sacc;
        f32x4 pf[NP][2];
        // K(i) landed.  Issue order of a wave's batches (2 NP pieces each): K(0), V^T(0), K(1) [prologue]; K(2), V^T(1) [iteration 0]; K(3),
        // V^T(2) [iteration 1]; ...  Iteration 0: V^T(0) and K(1) are younger than K(0).  Iteration 1: K(1) is YOUNGER than V^T(0), so the
        // wait for V^T(0) did not cover it — K(2) (if issued) and V^T(1) are behind it.  From iteration 2 on K(i) is older than V^T(i - 1),
        // which iteration i - 1 waited for.  lgkmcnt: first time round, this wave's part of the bias table
        if (i == 0) {
            if (NH > 1) { if constexpr (NP == 3) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); }
            else { if constexpr (NP == 3) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); }
        } else if (i == 1) {
            if (NH > 2) { if constexpr (NP == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
            else { if constexpr (NP == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        }
        AX_RAW_BARRIER();
        if (valid) {
            if (full) ax_bias_init<true>(sacc, bt, key0, L.hh, qrow_c, T);
            else ax_bias_init<false>(sacc, bt, key0, L.hh, qrow_c, T);
            ax_qk<NP, true>(sacc, Ks, qf, L);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        AX_RAW_BARRIER();                                        // this K buffer has been read by both of its waves
        const bool more2 = i + 2 < NH;
        if (more2) stage_op(i + 2, false);
        if (valid) ax_softmax<NP>(sacc, pf, m_run, l_run, o, key0, L.hh, T, full);
        // V^T(i) landed.  Younger than it in the queue: K(i + 1) if it was issued BEFORE V^T(i)... it was not: the issue order is
        // K(i + 1) [iteration i - 1], V^T(i) [end of iteration i - 1], K(i + 2) [just now] — only K(i + 2) may still fly.  (Iteration 0:
        // K(0), V^T(0), K(1), K(2): K(1) and K(2) are younger.)
        if (i == 0) {
            const int young = (NH > 1 ? 1 : 0) + (more2 ? 1 : 0);
            if (young == 2) { if constexpr (NP == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
            else if (young == 1) { if constexpr (NP == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (more2) {
            if constexpr (NP == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        AX_RAW_BARRIER();
        if (valid) ax_pv<NP, true>(o, Vs, pf, L);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        AX_RAW_BARRIER();                                        // every V^T stage has been read
        if (more) stage_op(i + 1, true);
    }
    // the early group passes the late group's last barrier with it: nobody writes a merge image (they overlay the stages) before every
    // wave of the block has read its last V^T fragments
    if (KH == 4 && !late) AX_RAW_BARRIER();
    // merge the KH key parts of each query sub-block (fixed order).  Image per wave in the (free) stages: 8 groups of 64 lanes x 16 B, m, l
    float* mine = smem + wave * 2304;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *(f32x4*)(mine + ((dt * 4 + g) * 64 + lane) * 4) = f32x4{o[dt][4 * g], o[dt][4 * g + 1], o[dt][4 * g + 2], o[dt][4 * g + 3]};
    mine[2048 + lane] = m_run;
    mine[2112 + lane] = l_run;
    __syncthreads();
    float m_all = -INFINITY;
#pragma unroll
    for (int k = 0; k < KH; ++k) m_all = fmaxf(m_all, smem[(qs + 2 * k) * 2304 + 2048 + lane]);
    float sc[KH], l_all = 0.f;
#pragma unroll
    for (int k = 0; k < KH; ++k) {                               // a part without tiles holds m = -inf, l = 0: scale 0
        sc[k] = vn_exp_neg(smem[(qs + 2 * k) * 2304 + 2048 + lane] - m_all);
        l_all += smem[(qs + 2 * k) * 2304 + 2112 + lane] * sc[k];
    }
    const float l_tot = (l_all + __shfl_xor(l_all, 32)) * (NP == 2 ? 16.0f : 1.0f);           // f16: l' = 16 l, O' = 256 O
    if (active && qrow < T) {
#pragma unroll
        for (int i = 0; i < 8 / KH; ++i) {
            const int G = kh * (8 / KH) + i;                     // uniform: this wave finishes 8 / KH of the eight column groups
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < KH; ++k) a += *(const f32x4*)(smem + (qs + 2 * k) * 2304 + (G * 64 + lane) * 4) * sc[k];
            ax_store4(a, l_tot, out, out16, plane16, (long)b * T + qrow, h, H, L.hh, G >> 2, G & 3);
        }
    }
}

// LDS bytes of the two decompositions; the launcher (and the engine's choice of attention kernel) need them to fit the CU
size_t vn_attention_x3_lds_bytes(int T, int key_split, int np) {
    // key_split: 0 shared tiles (two stages), 1 / 2 / 4 key-split waves (a private stage each), 8 = the pair-split kernel (four shared stages)
    const size_t stages = key_split > 0 ? (size_t)key_split : 2;
    // pair-split: four key parts of two K buffers + one V^T buffer = six stages' worth
    const size_t bytes = ((key_split == 8 ? 6 : stages) * AX_STAGE_FLOATS_NP(np) + 2 * (size_t)T - 1 + 3) * sizeof(float);
    // shared tiles: the tail blocks merge their key halves through four 9 KiB images laid over the stages (and, for two-plane
    // stages and a short bias table, past them); pair-split: eight such images
    const size_t merge = key_split == 8 ? 8 * 2304 * sizeof(float) : key_split > 0 ? 0 : 4 * 2304 * sizeof(float);
    return bytes > merge ? bytes : merge;
}

// which decomposition: 0 = shared tiles (128-query blocks + key-split tail blocks), KS > 0 = key-split with KS waves per 32-query block
int vn_attention_x3_plan(const vn_ctx* ctx, int B, int H, int T, int cus) {
    if (ctx->tune.ax_split >= 0) return ctx->tune.ax_split;
    // the key-split shape (two key halves per 32-query block: 52.6 KiB, three blocks per CU) halves the serial chain of tiles a
    // wave walks and wins while ALL its blocks are resident at once; from its second round on the shared-tile kernel (a quarter of
    // the blocks, K / V^T staged once per 128 queries, key-split tail blocks) is ahead (T = 575: B = 2 33.5 vs 35.4 us, B = 3
    // 57.4 vs 40.9 us; T = 173: B = 4 15.2 vs 15.9, B = 8 25.5 vs 20.4 — profiles/history/r03_attention_x3_tail_role.txt)
    // one sequence (or the four c2f chunks of one): the pair-split kernel (64-query blocks of eight waves, four key parts: a chain of
    // ceil(NT / 4) tiles per wave at two waves per SIMD) while ALL its blocks are resident at once, one per CU — measured
    // (profiles/r05_attention_pair_split.txt): T = 575, B = 1: 23.3 vs 28.1 us for the key-split pair; T = 173, B = 4: 13.3 vs 15.0; from
    // its second round on it loses (B = 2, T = 575: 44.1 vs 34.0).  VN_ATTN_X3_PAIR=0: off
    if (ctx->tune.ax_pair && (long)B * H * ((T + 63) / 64) <= (long)cus && vn_attention_x3_lds_bytes(T, 8, 3) <= 160 * 1024) return 8;
    if ((long)B * H * ((T + 31) / 32) <= 3L * cus) return 2;
    return 0;
}

// np = 3: q16 / k16 / vt16 are bf16x3 planes; np = 2: fp16 two-plane operands (second plane unscaled, V^T times 16), the attention
// format of the f16x2 precision
int vn_launch_attention_x3(vn_ctx* ctx, const uint16_t* q16, const uint16_t* k16, long plane_qk, const uint16_t* vt16, long plane_vt,
                           const float* relbias_full, float* out, uint16_t* out16, long plane16, int B, int H, int T, int cus, int np,
                           hipStream_t s) {
    if (B <= 0 || T <= 0) return VN_OK;
    if (np != 2 && np != 3) return vn_fail(ctx, VN_ERR_INVALID, "attention_x3: %s%ld planes per operand (2 or 3)", "", np);
    const int ks = vn_attention_x3_plan(ctx, B, H, T, cus);
    size_t lds = vn_attention_x3_lds_bytes(T, ks, np);
    if (lds > 160 * 1024) return vn_fail(ctx, VN_ERR_INVALID, "attention_x3: T=%s%ld too long for the LDS bias table", "", T);
    if (ks == 0 && ctx->tune.ax_lds > (int)lds && ctx->tune.ax_lds <= 160 * 1024) lds = ctx->tune.ax_lds;
    if (!(ctx->attr_mask & VN_ATTR_ATTN_X3)) {
        VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)(vn_attention_x3_kernel<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)(vn_attention_x3_kernel<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)(vn_attention_x3_split_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)(vn_attention_x3_split_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)(vn_attention_x3_split_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)(vn_attention_x3_kernel<4, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)(vn_attention_x3_kernel<4, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)(vn_attention_x3_split_kernel<1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)(vn_attention_x3_split_kernel<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)(vn_attention_x3_split_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)(vn_attention_x3_pair_kernel<4, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)(vn_attention_x3_pair_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ctx->attr_mask |= VN_ATTR_ATTN_X3;
    }
    if (3 * plane_qk * 2 + AX_K_PAD * 2 >= (1L << 31) || 3 * plane_vt * 2 >= (1L << 31))
        return vn_fail(ctx, VN_ERR_UNSUPPORTED, "attention_x3: %s%ld elements per plane exceed the 32-bit buffer descriptors", "", plane_qk);
    const unsigned kb = ax_k_extent(q16, k16, plane_qk), vb = (unsigned)(3 * plane_vt * 2);      // true extents (attention_x3_dev.h)
    const int pi = vn_prof_pre(ctx, 1, 4.0 * T * (double)T * VN_DHEAD * H * B, s, 16.0 * T * VN_DHEAD * (double)H * B);
    if (ks == 0) {
        // four waves (128 queries) per block, three blocks per CU (LDS: 2 x 24 KiB stages + the bias table)
        const int nqbf = T / 128, rq = T - 128 * nqbf;              // full 128-query blocks (+ one for a remainder > 64) + key-split tail blocks
        const dim3 grid((nqbf + (rq > 0 ? 1 : 0)) * H * B);
        const long per_cu = (long)(160 * 1024 / lds) < 3 ? (long)(160 * 1024 / lds) : 3;      // registers allow three waves per SIMD
        const long slots = per_cu * cus;
        const int knobs = (ctx->tune.ax_stagger & 0xffff) | ((long)grid.x > slots && !(ctx->tune.ax_stagger >> 16) ? 0x10000 : 0);
#define AX_SHARED_GO(TR, NP) hipLaunchKernelGGL((vn_attention_x3_kernel<4, TR, NP>), grid, dim3(256), lds, s, q16, k16, plane_qk, vt16, plane_vt, \
                                                relbias_full, out, out16, plane16, B, H, T, knobs, TR ? ctx->tune.ax_trace : (unsigned*)nullptr, (float*)nullptr, vn_drop{}, kb, vb)
        if (ctx->tune.ax_trace) { if (np == 3) AX_SHARED_GO(true, 3); else AX_SHARED_GO(true, 2); }
        else { if (np == 3) AX_SHARED_GO(false, 3); else AX_SHARED_GO(false, 2); }
#undef AX_SHARED_GO
    } else if (ks == 8) {
        const dim3 grid(vn_cdiv(T, 64) * H * B);
        if (np == 3) hipLaunchKernelGGL((vn_attention_x3_pair_kernel<4, 3>), grid, dim3(512), lds, s, q16, k16, plane_qk, vt16, plane_vt, relbias_full, out, out16, plane16, B, H, T, kb, vb);
        else hipLaunchKernelGGL((vn_attention_x3_pair_kernel<4, 2>), grid, dim3(512), lds, s, q16, k16, plane_qk, vt16, plane_vt, relbias_full, out, out16, plane16, B, H, T, kb, vb);
    } else {
        const dim3 grid(vn_cdiv(T, 32) * H * B);
#define AX_SPLIT_GO(KS, NP) hipLaunchKernelGGL((vn_attention_x3_split_kernel<KS, NP>), grid, dim3(KS * 64), lds, s, q16, k16, plane_qk, vt16, \
                                               plane_vt, relbias_full, out, out16, plane16, B, H, T, kb, vb)
        if (np == 3) {
            if (ks == 1) AX_SPLIT_GO(1, 3);
            else if (ks == 2) AX_SPLIT_GO(2, 3);
            else AX_SPLIT_GO(4, 3);
        } else {
            if (ks == 1) AX_SPLIT_GO(1, 2);
            else if (ks == 2) AX_SPLIT_GO(2, 2);
            else AX_SPLIT_GO(4, 2);
        }
#undef AX_SPLIT_GO
    }
    vn_prof_post(ctx, pi, s);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

// the training step's forward (train.hip): always the shared-tile decomposition; dropout stream d, lse[b][h][t] out
int vn_launch_attention_x3_train_fwd(vn_ctx* ctx, const uint16_t* q16, const uint16_t* k16, long plane_qk, const uint16_t* vt16, long plane_vt,
                                     const float* relbias_full, float* out, float* lse, int B, int H, int T, int cus, const vn_drop& d,
                                     hipStream_t s, uint16_t* out16, long plane16) {
    if (B <= 0 || T <= 0) return VN_OK;
    const size_t lds = vn_attention_x3_lds_bytes(T, 0, 3);
    if (lds > 160 * 1024) return vn_fail(ctx, VN_ERR_INVALID, "attention_x3 (training): T=%s%ld too long for the LDS bias table", "", T);
    if (!(ctx->attr_mask & VN_ATTR_ATTN_X3_TRAIN)) {
        VN_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)(vn_attention_x3_kernel<4, false, 3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ctx->attr_mask |= VN_ATTR_ATTN_X3_TRAIN;
    }
    if (3 * plane_qk * 2 + AX_K_PAD * 2 >= (1L << 31) || 3 * plane_vt * 2 >= (1L << 31))
        return vn_fail(ctx, VN_ERR_UNSUPPORTED, "attention_x3 (training): %s%ld elements per plane exceed the 32-bit buffer descriptors", "", plane_qk);
    const unsigned kb = ax_k_extent(q16, k16, plane_qk), vb = (unsigned)(3 * plane_vt * 2);
    const int pi = vn_prof_pre(ctx, 1, 4.0 * T * (double)T * VN_DHEAD * H * B, s, 16.0 * T * VN_DHEAD * (double)H * B);
    const int nqbf = T / 128, rq = T - 128 * nqbf;
    const dim3 grid((nqbf + (rq > 0 ? 1 : 0)) * H * B);
    const long per_cu = (long)(160 * 1024 / lds) < 3 ? (long)(160 * 1024 / lds) : 3;
    const int knobs = (long)grid.x > per_cu * cus ? 0x10000 : 0;
    hipLaunchKernelGGL((vn_attention_x3_kernel<4, false, 3, true>), grid, dim3(256), lds, s, q16, k16, plane_qk, vt16, plane_vt, relbias_full, out,
                       out16, plane16, B, H, T, knobs, (unsigned*)nullptr, lse, d, kb, vb);
    vn_prof_post(ctx, pi, s);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}

// tuning hook of one context (scripts/attn_probe.py; include/vampnet_hip_debug.h)
extern "C" int vn_debug_attention_x3_config(vn_ctx* ctx, int split, int lds_bytes, int stagger, void* trace_dev) {
    if (!ctx) return VN_ERR_INVALID;
    if (split != -1 && split != 0 && split != 1 && split != 2 && split != 4 && split != 8)
        return vn_fail(ctx, VN_ERR_INVALID, "attention_x3: key split %s%ld is not -1 / 0 / 1 / 2 / 4 / 8 (8 = the pair-split kernel)", "", split);
    ctx->tune.ax_split = split;
    ctx->tune.ax_lds = lds_bytes;
    if (stagger >= 0) ctx->tune.ax_stagger = stagger;
    ctx->tune.ax_trace = (unsigned*)trace_dev;
    ++ctx->tune.epoch;
    return VN_OK;
}
