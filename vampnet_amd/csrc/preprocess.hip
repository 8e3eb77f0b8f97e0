// Interface._preprocess on the device (vampnet/interface.py:206-217: clone -> resample -> to_mono -> normalize(-24 LUFS) ->
// ensure_max_of_audio(1.0) -> codec.preprocess): the loudness measurement, the gain, the peak limit and the right-pad to the codec's hop
// for a batch of mono signals that are already at the codec's sample rate.  Row f3 of SURVEY.md section 8 ("next").  PARITY UNPINNED:
// `audiotools` is not part of the reference tree; the arithmetic is ITU-R BS.1770-4 gated integrated loudness as vampnet_amd/codec.py
// restates it on the host (integrated_loudness: K-weighting = two biquads, 400 ms blocks every 100 ms, absolute gate -70 LUFS, relative
// gate -10 LU) — tests/test_gpu_codec.py holds the two to 1e-6.
//
// The K-weighting filters are IIR: a sample depends on every earlier one.  They are still evaluated in parallel over time, exactly: the
// cascade of the two biquads (direct form II transposed, as scipy.signal.lfilter runs them) is a linear system with a 4-vector state,
//     s[n + 1] = A s[n] + B x[n],      y[n] = C s[n] + D x[n],
// so the state at the start of chunk j is  S_j = A^CH S_(j-1) + e_(j-1)  with e_j the end state of chunk j run from a ZERO state.
//   pass 1  one thread per (item, chunk): the chunk from a zero state -> e_j                               (parallel over time)
//   scan    one thread per item: S_j for every chunk (a 4 x 4 product per chunk; A^CH comes from the host)   (100 steps for 10 s)
//   pass 2  one thread per (item, chunk): the chunk again from its true state -> sum of y^2, and max |x|   (parallel over time)
//   gate    one thread per item: block energies from four consecutive chunk sums (chunk = the 100 ms hop), the two gates, LUFS,
//           gain 10^((target - LUFS) / 20) (unchanged at or below -70 LUFS: silence stays silent), peak limit to 1.0
//   apply   y = g x, zero-padded to the hop
// All of it in fp64 like the host twin (the recurrences amplify fp32 rounding); ~1 ms for eight 10 s clips against ~10 ms PER CLIP of
// scipy on the host.
#include "vn_common.h"

struct vn_kw { double b1[3], a1[3], b2[3], a2[3]; };            // the two biquads (a[0] = 1)
struct vn_kw_pow { double m[16]; };                              // A^CH, row major

// one step of the cascade: s = (z1a, z2a, z1b, z2b) -> y
__device__ __forceinline__ double kw_step(const vn_kw& k, double (&s)[4], double x) {
    const double y1 = k.b1[0] * x + s[0];
    s[0] = k.b1[1] * x - k.a1[1] * y1 + s[1];
    s[1] = k.b1[2] * x - k.a1[2] * y1;
    const double y2 = k.b2[0] * y1 + s[2];
    s[2] = k.b2[1] * y1 - k.a2[1] * y2 + s[3];
    s[3] = k.b2[2] * y1 - k.a2[2] * y2;
    return y2;
}

// pass 1 (true_state == nullptr): end state of every chunk from a zero state -> st[b][j][4]
// pass 2: from the true start state true_state[b][j][4] -> seg[b][j] = sum of y^2 over the chunk, peak[b][j] = max |x|
__global__ __launch_bounds__(64) void vn_kweight_chunks_kernel(const float* __restrict__ x, int B, int T, int CH, int NCH, vn_kw k,
                                                              const double* __restrict__ true_state, double* __restrict__ st,
                                                              double* __restrict__ seg, float* __restrict__ peak) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= B * NCH) return;
    const int b = i / NCH, j = i - b * NCH;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if (true_state) {
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] = true_state[(size_t)i * 4 + q];
    }
    const float* xr = x + (size_t)b * T;
    const int n0 = j * CH;
    double acc = 0.0;
    float pk = 0.f;
    for (int n = 0; n < CH; ++n) {
        const int t = n0 + n;
        const float xv = t < T ? xr[t] : 0.0f;                  // a signal shorter than one block is zero-extended (as the host twin pads)
        const double y = kw_step(k, s, (double)xv);
        if (t < T) acc += y * y;                                // (the host twin zero-pads the FILTERED signal of a clip shorter than a block)
        pk = fmaxf(pk, fabsf(xv));
    }
    if (true_state) { seg[i] = acc; peak[i] = pk; }
    else {
#pragma unroll
        for (int q = 0; q < 4; ++q) st[(size_t)i * 4 + q] = s[q];
    }
}

// scan: S_0 = 0, S_(j+1) = A^CH S_j + e_j ; in place: st[b][j] (end states from zero) -> start states
__global__ void vn_kweight_scan_kernel(double* __restrict__ st, int B, int NCH, vn_kw_pow P) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double S[4] = {0.0, 0.0, 0.0, 0.0};
    for (int j = 0; j < NCH; ++j) {
        double* e = st + ((size_t)b * NCH + j) * 4;
        const double e0 = e[0], e1 = e[1], e2 = e[2], e3 = e[3];
        e[0] = S[0]; e[1] = S[1]; e[2] = S[2]; e[3] = S[3];       // the chunk's true start state
        double N[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) N[r] = P.m[4 * r] * S[0] + P.m[4 * r + 1] * S[1] + P.m[4 * r + 2] * S[2] + P.m[4 * r + 3] * S[3];
        S[0] = N[0] + e0; S[1] = N[1] + e1; S[2] = N[2] + e2; S[3] = N[3] + e3;
    }
}

// gating (codec.py integrated_loudness) -> gain[b] (linear, peak limit included), lufs[b]
__global__ void vn_loudness_gate_kernel(const double* __restrict__ seg, const float* __restrict__ peak, int B, int NCH, int nblocks, int CH,
                                        float target, float* __restrict__ gain, float* __restrict__ lufs_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double* sg = seg + (size_t)b * NCH;
    const double inv = 1.0 / (4.0 * CH);
    // absolute gate, then the relative gate 10 LU under the mean energy of the blocks above it (all sums in block order)
    double sum1 = 0.0;
    int n1 = 0;
    for (int i = 0; i < nblocks; ++i) {
        const double z = (sg[i] + sg[i + 1] + sg[i + 2] + sg[i + 3]) * inv;
        const double lk = -0.691 + 10.0 * log10(fmax(z, 1e-12));
        if (lk > -70.0) { sum1 += z; ++n1; }
    }
    double lufs = -70.0;
    if (n1 > 0) {
        const double rel = -0.691 + 10.0 * log10(fmax(sum1 / n1, 1e-12)) - 10.0;
        double sum2 = 0.0;
        int n2 = 0;
        for (int i = 0; i < nblocks; ++i) {
            const double z = (sg[i] + sg[i + 1] + sg[i + 2] + sg[i + 3]) * inv;
            const double lk = -0.691 + 10.0 * log10(fmax(z, 1e-12));
            if (lk > -70.0 && lk > rel) { sum2 += z; ++n2; }
        }
        if (n2 > 0) lufs = -0.691 + 10.0 * log10(fmax(sum2 / n2, 1e-12));
    }
    // normalize(target): the gain in float64 like the host twin's Python arithmetic, applied as an fp32 factor; silence stays silent
    const float g = lufs > -70.0 ? (float)pow(10.0, ((double)target - lufs) / 20.0) : 1.0f;
    float pk = 0.f;
    for (int j = 0; j < NCH; ++j) pk = fmaxf(pk, peak[(size_t)b * NCH + j]);
    pk *= g;                                                     // ensure_max_of_audio(1.0): the peak of the NORMALISED signal (rounding is monotone)
    gain[b] = g;
    gain[B + b] = pk > 1.0f ? fmaxf(pk, 1e-12f) : 0.0f;          // divisor of the peak limit, 0 = not limited
    if (lufs_out) lufs_out[b] = (float)lufs;
}

__global__ __launch_bounds__(256) void vn_gain_pad_kernel(const float* __restrict__ x, const float* __restrict__ gain, float* __restrict__ y,
                                                          int B, int T, int Tp) {
    const long n = (long)B * Tp;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        const int b = (int)(i / Tp), t = (int)(i - (long)b * Tp);
        float v = 0.0f;
        if (t < T) {                                             // the host twin's two roundings: x * g, then / peak where the item is limited
            v = x[(size_t)b * T + t] * gain[b];
            const float dv = gain[B + b];
            if (dv > 0.0f) v = v / dv;
        }
        y[i] = v;
    }
}

extern "C" int vn_preprocess_workspace(int B, int T, int sample_rate, int64_t* n_bytes) {
    if (B <= 0 || T <= 0 || sample_rate <= 0 || !n_bytes) return VN_ERR_INVALID;
    const int CH = sample_rate / 10, blk = (int)(0.4 * sample_rate);
    const int Tl = T < blk ? blk : T;
    const long NCH = (Tl + CH - 1) / CH + 4;
    *n_bytes = (int64_t)B * NCH * (4 * 8 + 8 + 4) + 2 * B * 4 + 256;
    return VN_OK;
}

// kw = the 12 biquad coefficients (b1[3], a1[3], b2[3], a2[3]) and pw = A^CH (16 doubles, row major; CH = sample_rate / 10) of the
// cascade for `sample_rate` — computed by the caller in float64 (vampnet_amd/codec.py: _k_weighting / kweight_state_power)
extern "C" int vn_preprocess_f32(vn_ctx* ctx, const float* x, float* y, int B, int T, int Tp, int sample_rate, float target_lufs,
                                 const double* kw12, const double* pow16, void* workspace, float* lufs_out, void* stream) {
    if (!ctx || !x || !y || !kw12 || !pow16 || !workspace) return VN_ERR_INVALID;
    if (B <= 0 || T <= 0 || Tp < T) return vn_fail(ctx, VN_ERR_INVALID, "preprocess: B=%s%ld, T=%ld must be positive and the padded length >= T", "", B, T);
    const int CH = sample_rate / 10, blk = (int)(0.4 * sample_rate);
    if (CH <= 0 || blk != 4 * CH)
        return vn_fail(ctx, VN_ERR_UNSUPPORTED, "preprocess: a 400 ms block must be four 100 ms hops at the sample rate (%s%ld Hz)", "", sample_rate);
    hipStream_t s = (hipStream_t)stream;
    const int Tl = T < blk ? blk : T;
    const int nblocks = 1 + (Tl - blk) / CH;
    const int NCH = (Tl + CH - 1) / CH + 4;                      // chunks (zero-extended past the signal); blocks read chunks i .. i + 3
    vn_kw k;
    vn_kw_pow P;
    for (int i = 0; i < 3; ++i) { k.b1[i] = kw12[i]; k.a1[i] = kw12[3 + i]; k.b2[i] = kw12[6 + i]; k.a2[i] = kw12[9 + i]; }
    for (int i = 0; i < 16; ++i) P.m[i] = pow16[i];
    char* w = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    double* st = (double*)w;
    double* seg = st + (size_t)B * NCH * 4;
    float* peak = (float*)(seg + (size_t)B * NCH);
    float* gain = peak + (size_t)B * NCH;
    const int nthreads = B * NCH;
    hipLaunchKernelGGL(vn_kweight_chunks_kernel, dim3(vn_cdiv(nthreads, 64)), dim3(64), 0, s, x, B, T, CH, NCH, k, (const double*)nullptr, st, seg, peak);
    hipLaunchKernelGGL(vn_kweight_scan_kernel, dim3(vn_cdiv(B, 64)), dim3(64), 0, s, st, B, NCH, P);
    hipLaunchKernelGGL(vn_kweight_chunks_kernel, dim3(vn_cdiv(nthreads, 64)), dim3(64), 0, s, x, B, T, CH, NCH, k, (const double*)st, st, seg, peak);
    hipLaunchKernelGGL(vn_loudness_gate_kernel, dim3(vn_cdiv(B, 64)), dim3(64), 0, s, (const double*)seg, (const float*)peak, B, NCH, nblocks, CH,
                       target_lufs, gain, lufs_out);
    const long n = (long)B * Tp;
    hipLaunchKernelGGL(vn_gain_pad_kernel, dim3((unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192)), dim3(256), 0, s, x, (const float*)gain, y, B,
                       T, Tp);
    VN_LAUNCH_CHECK(ctx);
    return VN_OK;
}
