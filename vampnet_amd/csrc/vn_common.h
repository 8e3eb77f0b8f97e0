// Internal helpers shared by the gfx950 kernels of libvampnet_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include "../../include/vampnet_hip.h"
#include "../../include/vampnet_hip_debug.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef float f32x8 __attribute__((ext_vector_type(8)));

// fp32 -> bf16, round to nearest even (what torch's .to(bfloat16) does): the native conversion, v_cvt_pk_bf16_f32 on gfx950
// (a third of the integer-arithmetic form's instructions); NaN not expected on this path
__device__ __forceinline__ uint16_t vn_f32_to_bf16(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }

__device__ __forceinline__ float vn_bf16_to_f32(uint16_t b) { return __builtin_bit_cast(float, (unsigned)b << 16); }

// fp32 -> three bf16 terms with p0 + p1 + p2 == x EXACTLY (8 + 8 + 8 significand bits; both remainders are exact fp32
// subtractions).  The "bf16x3" GEMM mode multiplies such triples on the bf16 matrix cores (gemm_x3.hip).
__device__ __forceinline__ void vn_split3(float x, uint16_t& p0, uint16_t& p1, uint16_t& p2) {
    p0 = vn_f32_to_bf16(x);
    const float r1 = x - vn_bf16_to_f32(p0);
    p1 = vn_f32_to_bf16(r1);
    p2 = vn_f32_to_bf16(r1 - vn_bf16_to_f32(p1));
}
// eight values at once (the B operand of one MFMA k-step): planes as packed bf16x8.  The bf16 -> fp32 widening of a plane is
// taken from the PACKED conversion result (low half << 16, high half & 0xffff0000: one integer op per value) — written as
// __builtin_convertvector(p0, f32x8) hipcc re-converts every value on its own (a single-source v_cvt_pk_bf16_f32 + a shift per value
// and level: +2 VALU per value; the softmax of attention_x3.hip is made of this).  Same planes bit for bit.
__device__ __forceinline__ f32x8 vn_bf16x8_widen(const bf16x8& p) {
    const u32x4 u = __builtin_bit_cast(u32x4, p);
    f32x8 f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __builtin_bit_cast(float, u[i] << 16);
        f[2 * i + 1] = __builtin_bit_cast(float, u[i] & 0xffff0000u);
    }
    return f;
}
__device__ __forceinline__ void vn_split3_x8(const f32x8& x, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
    p0 = __builtin_convertvector(x, bf16x8);
    const f32x8 r1 = x - vn_bf16x8_widen(p0);
    p1 = __builtin_convertvector(r1, bf16x8);
    p2 = __builtin_convertvector(r1 - vn_bf16x8_widen(p1), bf16x8);
}
// four consecutive values -> one 8-byte store per plane; plane == 0: single bf16 plane (fast mode)
__device__ __forceinline__ void vn_store_bf16x4(uint16_t* dst, long plane, const f32x4& o) {
    if (plane == 0) {
        uint2 pk;
        pk.x = vn_f32_to_bf16(o[0]) | ((unsigned)vn_f32_to_bf16(o[1]) << 16);
        pk.y = vn_f32_to_bf16(o[2]) | ((unsigned)vn_f32_to_bf16(o[3]) << 16);
        *(uint2*)dst = pk;
        return;
    }
    uint16_t t[3][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) vn_split3(o[e], t[0][e], t[1][e], t[2][e]);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        uint2 pk;
        pk.x = t[q][0] | ((unsigned)t[q][1] << 16);
        pk.y = t[q][2] | ((unsigned)t[q][3] << 16);
        *(uint2*)(dst + q * plane) = pk;
    }
}

// fp32 -> TWO fp16 terms ("f16x2", gemm_x3.hip's second operand format): h0 = fp16(x), h1 = fp16((x - h0) * 2^11).  fp16 carries 11
// significand bits, the remainder of a round-to-nearest conversion is at most 2^-11 |x| and exact in fp32, and scaling it by 2^11
// puts it in the SAME binade range as h0 — so x = h0 + 2^-11 h1 to within 2^-22 |x| for every |x| in fp16's normal range
// (6.1e-5 .. 65504; below that the absolute error is 2^-25: fp16 subnormals are produced by v_cvt_f16_f32 and kept by the f16 MFMA
// on gfx950 — scripts/ubench/f16_denorm_probe.hip).  A product of such pairs needs THREE matrix-core products instead of the six
// of the bf16 triples: a0 b0 into one accumulator, a0 b1 + a1 b0 into a second one that enters the result times 2^-11 (a1 b1 is
// 2^-22 of the leading term and dropped).  Operand error 2^-22 is four times the representation error of fp32 itself and random in
// sign: on the model's shapes the result is 7e-8 rms from the float64 product, an fp32-accumulating GEMM of the unsplit operands
// 3.4e-7 (tests/test_gpu_f16x2.py).  Values beyond +-65504 SATURATE (v_med3) instead of turning into inf.
#define VN_H2_SCALE 2048.0f
#define VN_H2_INV_SCALE (1.0f / 2048.0f)
// SATURATION LEDGER of the fp16 plane writers.  Every value that is turned into fp16 planes passes vn_split2h / vn_split2u, which clamp;
// each of them also ORs "this value did not fit (|x| >= 65504, or NaN)" into a per-thread flag, and the kernel reports the flag once per
// thread into the context's sticky device words (vn_ctx::sat; read and cleared by vn_saturation_flags, include/vampnet_hip.h):
//   word 0  GEMM-operand planes (normalised rows, attention output, GEGLU output, codec activations)
//   word 1  attention operands (q / 8, k, 16 v — so |v| >= 4094 is reported)
//   word 2  weight planes (vn_split2_f16 / vn_model_set_f16x2)
// The host side (vampnet_amd/engine.py) reads the words after every generate() in the f16x2 precision and re-runs a call that
// saturated on bf16x3, so a clamped value never reaches a caller silently.  Cost: one v_cmp + s_or per value.
enum { VN_SAT_OPERAND = 0, VN_SAT_ATTN = 1, VN_SAT_WEIGHT = 2, VN_SAT_WORDS = 4 };
__device__ __forceinline__ void vn_sat_note(bool& bad, float x) { bad |= !(__builtin_fabsf(x) < 65504.0f); }
__device__ __forceinline__ void vn_sat_report(unsigned* sat, int word, bool bad) {
    if (bad && sat) sat[word] = 1u;          // every writer stores the same value: no atomic needed
}
__device__ __forceinline__ void vn_split2h(float x, uint16_t& h0, uint16_t& h1, bool& bad) {
    vn_sat_note(bad, x);
    x = __builtin_fminf(__builtin_fmaxf(x, -65504.0f), 65504.0f);
    const _Float16 a = (_Float16)x;
    const _Float16 b = (_Float16)((x - (float)a) * VN_H2_SCALE);
    h0 = __builtin_bit_cast(uint16_t, a);
    h1 = __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float vn_h2_value(uint16_t h0, uint16_t h1) {
    return (float)__builtin_bit_cast(_Float16, h0) + (float)__builtin_bit_cast(_Float16, h1) * VN_H2_INV_SCALE;
}
// The attention operands of the f16x2 precision (attention_x3.hip; written by the QKV3 epilogue of gemm_x3.hip) are two fp16 planes
// WITHOUT the 2^11 on the second one, h1 = fp16(x - h0): all three kept products then go into ONE accumulator (a second S / O
// accumulator would cost the kernel its third wave per SIMD).  The remainder of a value below 0.125 falls into fp16's subnormal
// range, i.e. the split is exact to 2^-25 absolute instead of 2^-22 relative — harmless for q / 8 and k (a score moves by < 3e-7)
// and for softmax weights and values once those carry a factor 16 (P <= e^6: 16 P < 6.5e3; the factors leave with the final division).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void vn_split2u(float x, uint16_t& h0, uint16_t& h1, bool& bad) {
    vn_sat_note(bad, x);
    x = __builtin_fminf(__builtin_fmaxf(x, -65504.0f), 65504.0f);
    const _Float16 a = (_Float16)x;
    const _Float16 b = (_Float16)(x - (float)a);
    h0 = __builtin_bit_cast(uint16_t, a);
    h1 = __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ void vn_split2u_x8(const f32x8& x, f16x8& h0, f16x8& h1) {        // |x| < 65504 (softmax weights)
    h0 = __builtin_convertvector(x, f16x8);
    h1 = __builtin_convertvector(x - __builtin_convertvector(h0, f32x8), f16x8);
}
// four consecutive values -> one 8-byte store per f16x2 plane
__device__ __forceinline__ void vn_store_h2x4(uint16_t* dst, long plane, const f32x4& o, bool& bad) {
    uint16_t t[2][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) vn_split2h(o[e], t[0][e], t[1][e], bad);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        uint2 pk;
        pk.x = t[q][0] | ((unsigned)t[q][1] << 16);
        pk.y = t[q][2] | ((unsigned)t[q][3] << 16);
        *(uint2*)(dst + q * plane) = pk;
    }
}

// Split planes of a [rows][K] GEMM operand come in two layouts: PLANAR, three [rows][K] images `plane` elements apart, and TILED
// (plane stride VN_PLANES_TILED): [row / 16][k / 32][plane][row % 16][k % 32] — the 16-row x 32-k block of each plane is one
// contiguous 1 KiB piece = exactly what one LDS-DMA instruction of gemm_x3.hip fetches (eight whole cache lines instead of sixteen
// half lines from sixteen rows).  The model path uses TILED for weights and activations; PLANAR stays for the single-op test entries
// and the training GEMMs' on-the-fly splits.
// The plane-stride argument of every producer (RMSNorm, attention output, GEGLU epilogue) also names the FORMAT, so that the
// model path switches formats without a second argument anywhere:
//   stride > 0                planar bf16x3, three planes `stride` elements apart
//   VN_PLANES_TILED (-1)      tiled bf16x3
//   VN_PLANES_TILED_H2 (-2)   tiled f16x2: [row / 16][k / 32][2 planes][16][32]
//   stride < -2               planar f16x2, two planes -stride elements apart (single-op tests)
#define VN_PLANES_TILED (-1L)
#define VN_PLANES_TILED_H2 (-2L)
__host__ __device__ __forceinline__ bool vn_planes_h2(long plane) { return plane <= VN_PLANES_TILED_H2; }
__host__ __device__ __forceinline__ bool vn_planes_tiled(long plane) { return plane == VN_PLANES_TILED || plane == VN_PLANES_TILED_H2; }
__host__ __device__ __forceinline__ size_t vn_tiled_off_np(long row, int k, int K, int NP) {       // plane 0; plane q: + 512 q
    return (((size_t)(row >> 4) * (K >> 5) + (k >> 5)) * NP) * 512 + (size_t)((row & 15) * 32 + (k & 31));
}
__host__ __device__ __forceinline__ size_t vn_tiled_off(long row, int k, int K) { return vn_tiled_off_np(row, k, K, 3); }
// four consecutive columns col .. col + 3 (col % 4 == 0) of row `row` of a [rows][ld] matrix -> its planes
// (one call of vn_store_bf16x4 with a selected address / stride: with a call per layout hipcc 7.2 tail-merged the three store
// sequences of the D = 256 RMSNorm kernel and stored a stale register as the last dword of the planar form)
// `bad`: the caller's saturation flag (touched by the f16x2 formats only; see the saturation ledger above)
__device__ __forceinline__ void vn_store_planes4(uint16_t* base, long plane, long row, int col, int ld, const f32x4& o, bool& bad) {
    const bool tiled = vn_planes_tiled(plane);
    if (vn_planes_h2(plane)) {
        const size_t off = tiled ? vn_tiled_off_np(row, col, ld, 2) : (size_t)row * ld + col;
        vn_store_h2x4(base + off, tiled ? 512L : -plane, o, bad);
        return;
    }
    const size_t off = tiled ? vn_tiled_off(row, col, ld) : (size_t)row * ld + col;
    vn_store_bf16x4(base + off, tiled ? 512L : plane, o);
}

#define VN_WAVE 64
#define VN_DHEAD 64

#ifdef __HIPCC__
// the fp32 values of eight / four consecutive columns of a row from its TILED bf16x3 planes: p0 + p1 + p2, exact (vn_split3)
__device__ __forceinline__ f32x8 vn_load_planes8_bf16x3_tiled(const uint16_t* base, long row, int col, int ld) {
    const uint16_t* d = base + vn_tiled_off(row, col, ld);
    const f32x8 a = vn_bf16x8_widen(__builtin_bit_cast(bf16x8, *(const u32x4*)d));
    const f32x8 b = vn_bf16x8_widen(__builtin_bit_cast(bf16x8, *(const u32x4*)(d + 512)));
    const f32x8 c = vn_bf16x8_widen(__builtin_bit_cast(bf16x8, *(const u32x4*)(d + 1024)));
    return (a + b) + c;
}
__device__ __forceinline__ f32x4 vn_load_planes4_bf16x3_tiled(const uint16_t* base, long row, int col, int ld) {
    const uint16_t* d = base + vn_tiled_off(row, col, ld);
    f32x4 v;
    const uint2 a = *(const uint2*)d, b = *(const uint2*)(d + 512), c = *(const uint2*)(d + 1024);
    const unsigned aw[2] = {a.x, a.y}, bw[2] = {b.x, b.y}, cw[2] = {c.x, c.y};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        v[2 * i] = (__builtin_bit_cast(float, aw[i] << 16) + __builtin_bit_cast(float, bw[i] << 16)) + __builtin_bit_cast(float, cw[i] << 16);
        v[2 * i + 1] = (__builtin_bit_cast(float, aw[i] & 0xffff0000u) + __builtin_bit_cast(float, bw[i] & 0xffff0000u)) +
                       __builtin_bit_cast(float, cw[i] & 0xffff0000u);
    }
    return v;
}
// eight consecutive columns col .. col + 7 (col % 8 == 0) of row `row` -> one 16-byte store per plane, TILED layouts only (the eight
// columns sit in one 32-column block of the piece)
__device__ __forceinline__ void vn_store_planes8_tiled(uint16_t* base, long plane, long row, int col, int ld, const f32x8& o, bool& bad) {
    if (vn_planes_h2(plane)) {
        uint16_t t[2][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) vn_split2h(o[e], t[0][e], t[1][e], bad);
        uint16_t* d = base + vn_tiled_off_np(row, col, ld, 2);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            u32x4 pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = t[q][2 * e] | ((unsigned)t[q][2 * e + 1] << 16);
            *(u32x4*)(d + 512 * q) = pk;
        }
        return;
    }
    bf16x8 p0, p1, p2;
    vn_split3_x8(o, p0, p1, p2);
    uint16_t* d = base + vn_tiled_off(row, col, ld);
    *(u32x4*)d = __builtin_bit_cast(u32x4, p0);
    *(u32x4*)(d + 512) = __builtin_bit_cast(u32x4, p1);
    *(u32x4*)(d + 1024) = __builtin_bit_cast(u32x4, p2);
}
#endif

#ifdef __HIPCC__
// Cross-lane sums by DPP (one VALU instruction per step; a __shfl_xor is an LDS-crossbar round trip of ~100 cycles, and five of
// them per row piece made the folded-norm epilogue cost what the norm kernel it replaced did).  Fixed order: deterministic.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float vn_dpp(float x) {     // lanes outside ROW_MASK receive 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xF, false));
}
// sum over each group of 16 consecutive lanes, left in every lane of the group: xor 1, xor 2 (quad_perm), row_half_mirror, row_mirror
__device__ __forceinline__ float vn_sum16(float x) {
    x += vn_dpp<0xB1>(x);
    x += vn_dpp<0x4E>(x);
    x += vn_dpp<0x141>(x);
    x += vn_dpp<0x140>(x);
    return x;
}
// sum over each half-wave of 32 lanes, VALID IN LANES 16-31 / 48-63 only (row_bcast:15 hands the first row's sum to the second)
__device__ __forceinline__ float vn_sum32_hi(float x) {
    x = vn_sum16(x);
    return x + vn_dpp<0x142, 0xA>(x);
}
// Folded RMSNorm (vn_gemm_args::ssq_out): sums of squares as one fma chain per thread
__device__ __forceinline__ float vn_ssq4(const f32x4& v, float s = 0.0f) {
    s = fmaf(v[0], v[0], s);
    s = fmaf(v[1], v[1], s);
    s = fmaf(v[2], v[2], s);
    return fmaf(v[3], v[3], s);
}
#endif

#ifdef __HIPCC__
__device__ __forceinline__ uint4 vn_philox4x32_10(uint4 c, uint2 k) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
        const uint32_t hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += W0;
        k.y += W1;
    }
    return c;
}

__device__ __forceinline__ float vn_gelu_tanh(float x) {
    // activations.py:16-26: 0.5*x*(1+tanh(sqrt(2/pi)*(x+0.044715*x^3)))
    const float c = 0.7978845608028654f;
    float x3 = x * x * x;
    return 0.5f * x * (1.0f + tanhf(c * (x + 0.044715f * x3)));
}
#endif

#ifdef __HIPCC__
// Snake1d of the DAC stacks (vampnet/modules/layers.py:12-18 via lac): v + sin(alpha v)^2 / (alpha + 1e-9), inv = 1 / (alpha + 1e-9).
// sin^2 through an explicit reduction instead of ocml's sinf (half the instructions; the epilogues of the codec's convolutions are
// VALU-bound on it): k = rint(x 2/pi), r = x - k pi/2 by a three-constant Cody-Waite reduction (exact products for |k| < 2^13),
// s = sin(r) on [-pi/4, pi/4] by the degree-7 odd polynomial (2^-25 relative), sin^2(x) = s^2 for even k, 1 - s^2 (= cos^2 r) for odd k.
// |x| > 8192 (never seen on this path) takes sinf.
__device__ __forceinline__ float vn_sin_sq(float x) {
    if (fabsf(x) > 8192.0f) { const float s = sinf(x); return s * s; }
    const float kf = rintf(x * 0.636619772367581343f);
    float r = fmaf(kf, -1.5703125f, x);
    r = fmaf(kf, -4.837512969970703125e-4f, r);
    r = fmaf(kf, -7.549789954891882e-8f, r);
    const float r2 = r * r;
    float pl = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    pl = fmaf(r2, pl, -1.6666654611e-1f);
    const float sn = fmaf(r * r2, pl, r);
    const float s2 = sn * sn;
    return ((int)kf & 1) ? 1.0f - s2 : s2;
}
__device__ __forceinline__ float vn_snake(float v, float alpha, float inv) { return fmaf(inv, vn_sin_sq(alpha * v), v); }
#endif

#ifdef __HIPCC__
// exp(x) for x <= 0 with fp32-level accuracy at a third of ocml expf's instruction count: x*log2(e) is split into a
// rounded product and its exact fma remainder (plus the constant's low part), v_exp_f32 evaluates 2^hi (1 ulp) and the
// remainder is applied to first order (|lo| < 2^-22, so the dropped term is < 2^-45 relative).
template <bool FINITE = false>
__device__ __forceinline__ float vn_exp_neg(float x) {
    if constexpr (!FINITE) x = fmaxf(x, -104.0f);              // -inf (masked key / first tile) -> exp2(-150) = 0, no NaN
    // FINITE: the caller guarantees x > -inf (then hi, lo are finite whatever x is and exp2 flushes to 0 by itself)
    const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-08f;
    const float hi = x * L2E_HI;
    const float lo = fmaf(x, L2E_HI, -hi) + x * L2E_LO;
    const float e = __builtin_amdgcn_exp2f(hi);               // v_exp_f32; exp2(-inf) = 0, flushes below 2^-126
    return fmaf(e, lo * 0.693147182464599609375f, e);
}
#endif

// ---- training-mode dropout (train_kernels.hip, attention_train.hip) ---------------------------
// Counter-based keep-mask that does not depend on launch geometry, layout or sharding: element (row, col) of a
// dropout site is kept iff a 16-bit slice of  mix32(rowkey(row) ^ (col >> 1))  is >= thresh16 = round(p * 65536)
// (two columns per hash: 32-bit integer multiplies are quarter rate on CDNA, Philox would cost more than the
// attention MFMAs it sits between).  `row` is a GLOBAL row index (batch-sharded ranks draw disjoint streams);
// `key` already mixes (seed, optimiser step, layer, site) on the host.  Scale is the nominal 1/(1-p), as torch's.
struct vn_drop {
    uint32_t key;        // host: vn_drop_key(seed, step, site)
    uint32_t thresh16;   // 0 = dropout off (p == 0): kernels skip the hash
    float scale;         // 1 / (1 - p)
    long row0;           // global index of row 0 of this launch
};
static inline uint32_t vn_mix32_host(uint32_t x) {
    x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15; x *= 0x735a2d97u; x ^= x >> 15;
    return x;
}
static inline uint32_t vn_drop_key(uint64_t seed, uint32_t step, uint32_t site) {
    return vn_mix32_host((uint32_t)seed ^ vn_mix32_host((uint32_t)(seed >> 32) ^ vn_mix32_host(step * 0x9E3779B9u + site)));
}
#ifdef __HIPCC__
__device__ __forceinline__ uint32_t vn_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15; x *= 0x735a2d97u; x ^= x >> 15;
    return x;
}
__device__ __forceinline__ uint32_t vn_drop_rowkey(const vn_drop& d, long row) {
    const long g = d.row0 + row;
    return vn_mix32((uint32_t)g ^ vn_mix32(d.key ^ (uint32_t)(g >> 32)));
}
// keep-multiplier (0 or scale) of column `col` given the row key
__device__ __forceinline__ uint32_t vn_drop_bits(uint32_t rowkey, int col) { return vn_mix32(rowkey ^ (uint32_t)(col >> 1)); }
__device__ __forceinline__ float vn_drop_mul(const vn_drop& d, uint32_t bits, int col) {
    const uint32_t v = (col & 1) ? (bits >> 16) : (bits & 0xFFFFu);
    return v >= d.thresh16 ? d.scale : 0.0f;
}
#endif

struct vn_prof {
    bool on = false;
    int cap = 0, n = 0;
    unsigned stride = 1, seen = 0; // vn_profile_set_stride: bracket ~1 of every `stride` launches (hash-selected)
    hipEvent_t* ev = nullptr;      // 2 events per launch
    int* cls = nullptr;            // class per launch (VN_PROF_*)
    double* flops = nullptr;
    double* bytes = nullptr;       // algorithmic operand bytes of the launch
};

// Per-context tuning / test state.  Defaults come from the environment ONCE, when the context is created (vn_tune_init, engine.hip:
// valid-result knobs only); the vn_debug_* entries of include/vampnet_hip_debug.h change them for ONE context and bump `epoch`, so
// forward graphs captured under an older setting are re-captured.  Nothing here is process-global: two contexts may run different
// settings concurrently on their own streams (tests/test_gpu_kernels.py::test_two_contexts_tune_independently).
struct vn_tune {
    // gemm_f32.hip: tile walk order, stream-K start stagger, forced tile, scheduler (-1 auto, 0 data-parallel, 1 stream-K), split-K
    int f32_order, f32_stagger, f32_bm, f32_bn, f32_sched, f32_splitk;
    // gemm_x3.hip: forced tile height (0 = by shape), forced k-split (-2 = cost model), ablation bits (0 = none; results INVALID,
    // settable only through vn_debug_x3_config), fused reduce + norm, LDS-staged epilogues, tile-walk group height (0 = 8)
    int x3_bm, x3_split, x3_abl, x3_fuse_norm, x3_staged, x3_group_m;
    int x3_tile96;           // gemm_x3.hip: the planner may pick the 96-row k-split tile (VN_X3_TILE96, default 1; bf16x3 operands)
    int x3_convt;            // gemm_x3.hip: 96- / 192-channel convolutions with the channels on the tile's row axis (VN_X3_CONVT, default 1)
    // attention_x3.hip: decomposition (-1 by shape, 0 shared tiles, 1 / 2 / 4 key-split waves), dynamic-LDS override of the shared
    // kernel (occupancy probe), start stagger, phase-trace buffer (device)
    int ax_split, ax_lds, ax_stagger;
    int ax_pair;             // attention_x3.hip: the by-shape plan may pick the pair-split kernel for one or two sequences (VN_ATTN_X3_PAIR, default 1)
    unsigned* ax_trace;
    // engine.hip: bf16x3 models take the split-plane attention path (-1 by shape / LDS fit, 0 never, 1 always); operand plane layouts
    int attn_x3, a_tiled, w_tiled;
    int fold_norm;           // engine.hip: split-plane models fold the RMSNorms into their consumer GEMMs (VN_FOLD_NORM, default 1)
    int fold_x16_only;       // ... and bf16x3 models keep the residual stream in its (exact) planes only (VN_FOLD_X16ONLY, default 1)
    unsigned epoch;
};
void vn_tune_init(vn_tune* t);

struct vn_ctx {
    int device;
    char err[512];
    vn_prof prof;
    vn_tune tune;
    int cus;                 // compute units of the device (vn_ctx_create)
    // lazily allocated per-context device scratch (freed by vn_ctx_destroy): stream-K partial-sum slabs + hand-off flags
    // (gemm_f32.hip; launches of one context must be stream-ordered with each other, see include/vampnet_hip.h) and the
    // zero page the conv kernel DMAs its padding from (conv1d_f32.hip)
    float* sk_slabs;
    unsigned* sk_flags;
    float* zero_page;
    float* x3_ws;            // split-K partial tiles of the bf16x3 GEMM (gemm_x3.hip), fixed size, allocated on first use
    unsigned* sat;           // VN_SAT_WORDS sticky saturation words of the fp16 plane writers (allocated with the context)
    // kernels whose dynamic-LDS limit was raised on THIS context's device (hipFuncSetAttribute is per device, and one
    // process may hold contexts on several)
    unsigned attr_mask;
};
enum { VN_ATTR_ATTN = 1u, VN_ATTR_ATTN_TRAIN = 2u, VN_ATTR_REMASK = 4u, VN_ATTR_MT_JUMP = 8u, VN_ATTR_GEMM_X3 = 16u, VN_ATTR_ATTN_X3 = 32u, VN_ATTR_ATTN_X3_TRAIN = 64u,
       VN_ATTR_ATTN_X3_BWD = 128u, VN_ATTR_GEMM_X3_CONVT = 256u };

// launch classes of vn_profile_end (four doubles each: launches, ms, algorithmic flops, algorithmic bytes).  The codec's convolutions
// are booked by the roofline that bounds them: arithmetic intensity (flops / operand bytes) at or above the ridge of the pipe the layer
// runs on (split-plane pipe: 2500 / 6 TF over 8 TB/s = 52 flop / B; fp32-input MFMA: 157.3 TF over 8 TB/s = 19.7) -> that pipe's class,
// below it -> the HBM class, whichever kernel runs it
enum { VN_PROF_GEMM = 0, VN_PROF_ATTN = 1, VN_PROF_CONV_X3 = 2, VN_PROF_GEMM_BF16 = 3, VN_PROF_CONV_F32 = 4, VN_PROF_CONV_HBM = 5, VN_PROF_CLASSES = 6 };
static inline int vn_conv_class(double flops, double bytes, int pipe_class, double pipe_peak_tf) {
    const double ridge = pipe_peak_tf * 1e12 / 8.0e12;
    return bytes > 0.0 && flops / bytes < ridge ? VN_PROF_CONV_HBM : pipe_class;
}

// bracket a launch with events when profiling is on (no-ops otherwise)
static inline int vn_prof_pre(vn_ctx* ctx, int cls, double flops, hipStream_t s, double bytes = 0.0) {
    vn_prof& p = ctx->prof;
    if (!p.on || p.n >= p.cap) return -1;
    if (p.stride > 1) {            // unbiased sub-sampling: a multiplicative hash of the launch counter decides
        const unsigned k = p.seen++ * 2654435761u;
        if ((k >> 16) % p.stride) return -1;
    }
    const int i = p.n++;
    p.cls[i] = cls;
    p.flops[i] = flops;
    p.bytes[i] = bytes;
    (void)hipEventRecord(p.ev[2 * i], s);
    return i;
}
static inline void vn_prof_post(vn_ctx* ctx, int i, hipStream_t s) {
    if (i >= 0) (void)hipEventRecord(ctx->prof.ev[2 * i + 1], s);
}

static inline int vn_fail(vn_ctx* ctx, int code, const char* fmt, const char* a = "", long b = 0, long c = 0) {
    if (ctx) snprintf(ctx->err, sizeof(ctx->err), fmt, a, b, c);
    return code;
}

#define VN_HIP_CHECK(ctx, expr)                                                            \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            if (ctx) snprintf((ctx)->err, sizeof((ctx)->err), "%s failed: %s (%s:%d)", #expr, \
                              hipGetErrorString(_e), __FILE__, __LINE__);                  \
            return VN_ERR_HIP;                                                             \
        }                                                                                  \
    } while (0)

#define VN_LAUNCH_CHECK(ctx) VN_HIP_CHECK(ctx, hipGetLastError())

static inline int vn_cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- launchers implemented in the .hip files ------------------------------------------------
// QKV3 (gemm_x3.hip only): the operands of attention_x3.hip as split planes from ONE QKV GEMM — q (x 1/8) and k head-major,
// V TRANSPOSED (through the epilogue's LDS image) and blocked by tiles of 32 global token rows
// CONV (gemm_x3.hip only): implicit-GEMM 1-D convolution of the DAC stacks — A rows gathered per tap, epilogue = bias (+ residual)
// (+ tanh) -> y, snake(y) -> y2 as fp32 and / or split planes (conv1d_f32.hip's epilogue on the bf16x3 pipe)
// CONVT (gemm_x3.hip only, round 6): the same convolution with the operands' roles swapped — Y^T[C_out][positions] = W [C_out][taps C_in] X^T:
// the OUTPUT CHANNELS are the tile's rows (M = C_out = exactly one 96- or 192-row tile: nothing multiplies padding, where 128-wide column
// tiles waste a quarter of a 192-channel layer), the positions its columns; the tap gather runs on the W side, the epilogue transposes
enum { VN_EPI_STORE = 0, VN_EPI_BIAS = 1, VN_EPI_RESIDUAL = 2, VN_EPI_GEGLU = 3, VN_EPI_QKV = 4, VN_EPI_QKV3 = 5, VN_EPI_CONV = 6, VN_EPI_CONVT = 7 };

struct vn_gemm_args {
    const float* A;      // [M][K] row-major, lda = K
    const float* W;      // [N][K] row-major
    const float* bias;   // [N] or null
    float* C;            // epilogue dependent
    uint16_t* C16;       // GEGLU epilogue: write bf16 instead of C (fast mode / bf16x3 planes)
    int bf16;            // 1: A and W hold bf16 (K counts elements), fp32 accumulate; 2: bf16x3 split planes, 3: f16x2 split planes (gemm_x3.hip)
    long a_plane, w_plane, c_plane;   // bf16x3 / f16x2: elements between the planes of A / W / C16 (> 0 planar, VN_PLANES_TILED[_H2])
    int M, N, K;
    int ldc;             // row stride of C (floats)
    // QKV scatter: C = qkv base [3][B][H][T][64]; row m = b*T + t
    int T, H;
    long qkv_plane;      // B*H*T*64
    uint16_t* V16;       // QKV3 epilogue: V^T planes [3][H][ceil(M / 32)][64][32], v_plane elements apart (C16 = q then k planes)
    long v_plane;
    unsigned* sat;       // gemm_x3.hip: the context's saturation words (set by the launcher; f16x2 plane epilogues report into them)
    int staged;          // gemm_x3.hip: epilogue through LDS with 16-byte global accesses (set by the launcher when alignment allows)
    // gemm_x3.hip: W given as TILED planes (vn_launch_tile_planes): the 16-row x 32-k block of each plane is one contiguous 1 KiB piece,
    // [row / 16][k / 32][plane][row % 16][k % 32] — an LDS-DMA instruction then fetches eight whole cache lines instead of sixteen
    // half lines (w_plane is ignored)
    int w_tiled;
    int group_m;         // gemm_x3.hip tuning: rows of tiles per group of the XCD-aware tile walk (0 = 8)
    // RESIDUAL epilogue only, optional: the RMSNorm that follows this GEMM in the layer (y = RMSNorm(C) with weight norm_w).  A launch
    // that is split along K runs it inside its reduce pass (vn_splitk_reduce_rmsnorm_kernel) and sets *norm_done = 1; otherwise the
    // caller launches the norm itself.  norm_y16 / norm_plane as vn_launch_rmsnorm's y16 / plane16.
    const float* norm_w;
    float* norm_y;
    uint16_t* norm_y16;
    long norm_plane;
    float norm_eps;
    int* norm_done;
    // RMSNorm FOLDED into its consumer (engine.hip, models with m->folded): y W^T = r (.) (x (W (.) w)^T), r = rsqrt(mean(x^2) + eps) per row,
    // the norm weight w multiplied into the consumer's weight columns when its planes are built.  No norm kernel runs:
    //   PRODUCERS — the RESIDUAL epilogue (and the reduce pass of a split launch) also write X16 = the split planes of the NEW residual
    //   rows (x16_plane: VN_PLANES_TILED / VN_PLANES_TILED_H2; row length N) and ssq_out[t][row] (t < N / 128, leading dimension M) =
    //   the sum of squares of the row's 128 columns of column tile t (a row's total is the sum over t, added in index order by the
    //   consumer; group-major so that the consumer's lanes = rows read contiguously);
    //   CONSUMERS — the QKV3 / QKV / GEGLU / BIAS epilogues multiply every accumulator row by r = 1 / sqrt(sum_t ssq_in[t][row] / K +
    //   fold_eps) before anything else (K = the consumer's K = the residual width; K / 128 groups).
    uint16_t* X16;
    long x16_plane;
    int x16_only;        // producers, bf16x3 planes only (their sum IS the fp32 value): the residual stream lives in X16 alone — the old
                         // row is read back from its planes (p0 + p1 + p2, exact) and C is neither read nor written
    float* ssq_out;
    const float* ssq_in;
    float fold_eps;
    // CONV epilogue / operand (gemm_x3.hip): A = channels-last activation planes [3][B * T_in][C_in] (PLANAR, a_plane apart);
    // GEMM row m = (b, t') of M = B * T_rows reads input row t' * in_stride + j * dil - pad for tap j (k = j * C_in + c; rows outside
    // [0, T_in) come from the zero page); W = [C_out][taps * C_in] tiled planes; N = C_out, K = taps * C_in.
    // Output row t_out = t' * out_stride + out_off (rows outside [0, T_out) are dropped: the phases of a transposed convolution);
    // v = acc + bias (+ resid) (tanh if act) -> C (fp32, optional); snake(v, alpha) -> Y2 (fp32, optional) and / or C16 planes
    // (planar, c_plane apart, optional).  All [B][T_out][C_out].
    int conv_taps, conv_cin, conv_tin, conv_trows, conv_in_stride, conv_dil, conv_pad;
    int conv_tout, conv_out_stride, conv_out_off, conv_act;
    const uint16_t* zeros16;
    // gemm_x3.hip, STORE epilogue, bf16x3 planes: tn_blocks > 0 = the TN operand mode (X3_MODE_TN) — C[M][N] = A^T W, A [tokens][M] and W
    // [tokens][N] token-major TILED planes, K = the token count rounded up to 32, tn_blocks = ceil(tokens / 16) 16-token blocks exist (rows
    // past the last token inside the last block must be ZERO in at least one operand and finite in the other); M % 32 == 0
    int tn_blocks;
    const float* resid;
    const float* alpha;
    float* Y2;
};
int vn_launch_gemm_f32(vn_ctx* ctx, const vn_gemm_args& a, int epilogue, hipStream_t s);
int vn_launch_gemm_x3(vn_ctx* ctx, const vn_gemm_args& a, int epilogue, hipStream_t s);      // a.bf16 == 2
// planar split planes [3][rows][K] (plane elements apart) -> tiled planes [rows / 16][K / 32][3][16][32]; rows % 16 == 0, K % 32 == 0
int vn_launch_tile_planes(vn_ctx* ctx, const uint16_t* planes, long plane, uint16_t* tiled, long rows, int K, hipStream_t s);
// fp32 [rows][K] with column k scaled by scale[k] -> tiled split planes (plane = VN_PLANES_TILED or VN_PLANES_TILED_H2)
int vn_launch_fold_planes(vn_ctx* ctx, const float* src, const float* scale, uint16_t* dst, long rows, int K, long plane, hipStream_t s);
// fp32 [rows][K] -> f16x2 planes: plane = VN_PLANES_TILED_H2 (rows % 16 == 0, K % 32 == 0) or -(planar stride)
int vn_launch_split2h(vn_ctx* ctx, const float* src, uint16_t* dst, long rows, int K, long plane, hipStream_t s);
// C[M][N] (row stride ldc) (+)= sum over the nsplit partial images partial[s][M][N], in fixed order (gemm_f32.hip)
int vn_launch_splitk_reduce(vn_ctx* ctx, const float* partial, int nsplit, float* C, int M, int N, int ldc, bool residual,
                            hipStream_t s);

// folded-norm producers (elementwise.hip): x[rows][D] += sum of the nsplit images (nsplit = 0: x as it is, nothing written back), then
// x16 = the split planes of the rows (plane16: tiled bf16x3 / f16x2) and ssq[t][row] (t < D / 128) = sums of squares per 128-column group
// from_planes != 0 (tiled bf16x3 planes only): the old rows are read from x16 itself (exact) and x is not touched
int vn_launch_rowprep(vn_ctx* ctx, const float* partial, int nsplit, float* x, uint16_t* x16, long plane16, float* ssq, int rows, int D,
                      hipStream_t s, int from_planes = 0);
// x[rows][D] += sum of the nsplit images partial[s][rows][D] (fixed order), then y = RMSNorm(x) from the same registers (elementwise.hip)
int vn_launch_splitk_reduce_rmsnorm(vn_ctx* ctx, const float* partial, int nsplit, float* x, const float* w, float* y, uint16_t* y16,
                                    long plane16, int rows, int D, float eps, hipStream_t s);

// y16 / out16: bf16 image of the output for the next GEMM; plane16 == 0 one plane, > 0 three split planes that far apart
// both: write y (fp32) AND y16 (a y16 alone replaces y: the inference path's normalised rows are only ever a GEMM operand)
int vn_launch_rmsnorm(vn_ctx* ctx, const float* x, const float* w, float* y, int rows, int D, float eps, hipStream_t s,
                      uint16_t* y16 = nullptr, long plane16 = 0, bool both = false);
int vn_launch_embed(vn_ctx* ctx, const int32_t* codes, const float* tables, const float* wt, const float* b,
                    float* x, int B, int C, int T, int V1, int latent, int D, hipStream_t s);
int vn_launch_attention(vn_ctx* ctx, const float* q, const float* k, const float* v, const float* relbias_full,
                        float* out, int B, int H, int T, hipStream_t s, uint16_t* out16 = nullptr, long plane16 = 0);
// bf16x3 attention (attention_x3.hip): q16 / k16 planes [3][B][H][T][64] (plane_qk elements apart, q pre-scaled by 1/8),
// vt16 planes [3][H][ceil(B T / 32)][64][32] over global token rows (plane_vt apart); out fp32 [B][T][H*64] or out16 split planes (plane16 apart)
int vn_launch_attention_x3(vn_ctx* ctx, const uint16_t* q16, const uint16_t* k16, long plane_qk, const uint16_t* vt16, long plane_vt,
                           const float* relbias_full, float* out, uint16_t* out16, long plane16, int B, int H, int T, int cus,
                           int np, hipStream_t s);       // np = 3: bf16x3 planes; 2: fp16 two-plane operands (f16x2 precision)
// decomposition it will use (0 = shared 128-query tiles, KS = key-split waves per 32-query block) and the dynamic LDS that needs
int vn_attention_x3_plan(const vn_ctx* ctx, int B, int H, int T, int cus);
size_t vn_attention_x3_lds_bytes(int T, int key_split, int np);
// The q / k planes of a call: [planes][q: n | k: n | rest of the plane stride] + AX_K_PAD elements behind the last plane — every
// allocation of them is made with ax_qk_elems() (engine.hip, train.hip), and the kernels' descriptors are derived from the same two
// numbers, so the pad a last key tile over-reads into is owned by the buffer by construction.
#define AX_K_PAD (32 * VN_DHEAD)
static inline size_t ax_qk_elems(int planes, long plane_qk) { return (size_t)planes * (size_t)plane_qk + AX_K_PAD; }
// bytes from k16 (= q16 + n) to the end of that allocation; planes = 3 (the f16x2 format's two planes live in a three-plane buffer)
static inline unsigned ax_k_extent(const uint16_t* q16, const uint16_t* k16, long plane_qk) {
    return (unsigned)((ax_qk_elems(3, plane_qk) - (size_t)(k16 - q16)) * 2);
}
// every device allocation of the library (devmem.hip): plain hipMalloc / hipFree, or guard blocks when VN_GUARD_ALLOC / vn_debug_guard_mode say so
hipError_t vn_dev_malloc(void** p, size_t bytes);
void vn_dev_free(void* p);
static inline int vn_num_cus(const vn_ctx* ctx) { return ctx->cus > 0 ? ctx->cus : 256; }
// expands [num_buckets][H] into per-head tables over rel = key - query in [-(T-1), T-1]: out[h][rel + T - 1]
void vn_bucket_lut_host(int T, int num_buckets, int max_distance, int32_t* lut /* [2T-1] */);
int vn_launch_bias_expand(vn_ctx* ctx, const float* rel_bias, const int32_t* lut_dev, float* out, int H, int T,
                          hipStream_t s);

struct vn_sample_args {
    const float* logits;      // [B][N][V]   N = T*Cp
    const int32_t* z;         // [B][C][T] current tokens (MASK = V)
    const float* exp_noise;   // [B*N][V] or null
    int32_t* sampled;         // [B][N]
    float* psel;              // [B][N]  (+inf where not masked)
    int B, T, C, n_cond, V;
    float temperature;        // already resolved (>0)
    int do_sample;
    uint64_t seed;
    uint32_t step;
    long batch_offset;        // global index of item 0 (device RNG only)
    int call_batch, global_batch;   // RNG item id = (b / call_batch) * global_batch + batch_offset + b % call_batch
};
int vn_launch_sample(vn_ctx* ctx, const vn_sample_args& a, hipStream_t s);
// in-place nucleus filter on the logits of the masked rows (transformer.py:1001-1016)
int vn_launch_top_p(vn_ctx* ctx, float* logits, const int32_t* z, int B, int T, int C, int n_cond, int V, float top_p,
                    hipStream_t s);

struct vn_remask_args {
    const int32_t* sampled;   // [B][N]
    const float* psel;        // [B][N]
    const float* unif_noise;  // [B][N] or null
    int32_t* z;               // [B][C][T] in/out (only codebooks >= n_cond rewritten)
    int32_t* out_sampled;     // [B][C][T] or null: sampled tokens unflattened (+ cond codebooks from z)
    int B, T, C, n_cond, V;
    float mask_temp;          // mask_temperature * (1 - r)
    const int64_t* k_sched;   // device [B]: floor(gamma(r) * N0) per item (items of different calls may be batched)
    int last_step;
    uint64_t seed;
    uint32_t step;
    long batch_offset;
    int call_batch, global_batch;
};
int vn_launch_remask(vn_ctx* ctx, const vn_remask_args& a, hipStream_t s);

int vn_launch_i64_to_i32(vn_ctx* ctx, const int64_t* in, int32_t* out, long n, hipStream_t s);
int vn_launch_i32_to_i64(vn_ctx* ctx, const int32_t* in, int64_t* out, long n, hipStream_t s);
// z = mask ? V : tokens ; also counts masked tokens into *count (device int32, pre-zeroed)
int vn_launch_apply_mask(vn_ctx* ctx, const int64_t* tokens, const int64_t* mask, int32_t* z, int32_t* count,
                         long n, int V, hipStream_t s);
