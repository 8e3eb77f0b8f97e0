// The DAC codec for a host WITHOUT Python: vn_codec_create_from_weights builds, in C, everything vampnet_amd/codec.py builds on the host —
// the re-laid weights (channels-last convolution weights, the phases of the transposed convolutions, the tiled split planes of the
// layers that run on the matrix-core pipe), the layer loop of one direction as a program of vn_codec_op launches, and ONE activation
// arena planned from the buffers' live ranges — and returns the vn_codec that vn_dac_encode / vn_dac_decode (codec.hip) execute.
// `vn_dac_encode / vn_dac_decode`; rows a18 / a19.  PARITY UNPINNED like every codec kernel (`lac` is not part
// of the reference tree): the architecture is the published DAC design as SURVEY.md App. D reads it from the reference's call sites
// (vampnet/interface.py:203-224, vampnet/modules/transformer.py:661-684, vampnet/modules/layers.py:145).
//
// Input: a flat fp32 device blob holding the codec's tensors under their state_dict names, weight-norm already folded
// (`<conv>.weight` = g v / ||v||), in PyTorch's own layouts (Conv1d (C_out, C_in, k), ConvTranspose1d (C_in, C_out, 2 s), Snake alpha
// (1, C, 1)); vn_codec_tensor_name / vn_codec_tensor_offset say where each one goes.  The five tensors of the quantizer levels are
// stored level after level (`quantizer.quantizers.<i>.<t>` = row i of a stacked region), which is the layout the RVQ kernels read.
//
// The program is the SAME launch sequence the Python recorder produces (tests/test_gpu_codec.py holds the two to bitwise equality):
// same routing rule (which convolutions run on the split-plane pipe), same formats handed from producer to consumer, same kernels.
#include <math.h>
#include <algorithm>
#include <new>
#include <string>
#include <vector>
#include "vn_common.h"
#include "vn_train.h"        // vn_launch_split3_tiled

struct vn_codec;             // codec.hip
vn_codec* vn_codec_adopt(vn_ctx* ctx, int direction, std::vector<vn_codec_op>&& ops, std::vector<void*>&& owned);

namespace {

struct Tensor { std::string name; long off, count; };

static long al64(long n) { return (n + 63) & ~63L; }

struct Layout {
    std::vector<Tensor> t;
    long total = 0;
    void add(const std::string& name, long count) { t.push_back({name, total, count}); total += al64(count); }
    // level-stacked quantizer tensors: one region of n * count floats, names per level
    void add_levels(int n, const char* leaf, long count) {
        const long base = total;
        for (int i = 0; i < n; ++i) t.push_back({"quantizer.quantizers." + std::to_string(i) + "." + leaf, base + i * count, count});
        total += al64((long)n * count);
    }
    const Tensor* find(const char* name) const {
        for (const Tensor& x : t) if (x.name == name) return &x;
        return nullptr;
    }
};

static int cfg_ok(const vn_codec_cfg* c) {
    if (!c || c->encoder_dim <= 0 || c->decoder_dim <= 0 || c->n_rates <= 0 || c->n_rates > 8 || c->n_codebooks <= 0 || c->codebook_size <= 0 ||
        c->codebook_dim <= 0)
        return 0;
    for (int i = 0; i < c->n_rates; ++i)
        if (c->encoder_rates[i] <= 0 || c->decoder_rates[i] <= 0) return 0;
    return 1;
}
static int latent_of(const vn_codec_cfg* c) { return c->latent_dim > 0 ? c->latent_dim : c->encoder_dim << c->n_rates; }

static void res_unit_tensors(Layout& L, const std::string& q, long C) {
    L.add(q + ".block.0.alpha", C);
    L.add(q + ".block.1.weight", C * C * 7);
    L.add(q + ".block.1.bias", C);
    L.add(q + ".block.2.alpha", C);
    L.add(q + ".block.3.weight", C * C);
    L.add(q + ".block.3.bias", C);
}

static Layout make_layout(const vn_codec_cfg* c) {
    Layout L;
    const int n = c->n_rates, lat = latent_of(c);
    long C = c->encoder_dim;
    L.add("encoder.block.0.weight", C * 7);
    L.add("encoder.block.0.bias", C);
    for (int i = 0; i < n; ++i) {
        const std::string p = "encoder.block." + std::to_string(1 + i);
        for (int j = 0; j < 3; ++j) res_unit_tensors(L, p + ".block." + std::to_string(j), C);
        L.add(p + ".block.3.alpha", C);
        L.add(p + ".block.4.weight", 2 * C * C * 2 * c->encoder_rates[i]);
        L.add(p + ".block.4.bias", 2 * C);
        C *= 2;
    }
    L.add("encoder.block." + std::to_string(n + 1) + ".alpha", C);
    L.add("encoder.block." + std::to_string(n + 2) + ".weight", (long)lat * C * 3);
    L.add("encoder.block." + std::to_string(n + 2) + ".bias", lat);
    const long cd = c->codebook_dim;
    L.add_levels(c->n_codebooks, "in_proj.weight", cd * lat);
    L.add_levels(c->n_codebooks, "in_proj.bias", cd);
    L.add_levels(c->n_codebooks, "codebook.weight", (long)c->codebook_size * cd);
    L.add_levels(c->n_codebooks, "out_proj.weight", lat * cd);
    L.add_levels(c->n_codebooks, "out_proj.bias", lat);
    long D = c->decoder_dim;
    L.add("decoder.model.0.weight", D * lat * 7);
    L.add("decoder.model.0.bias", D);
    for (int i = 0; i < n; ++i) {
        const std::string p = "decoder.model." + std::to_string(1 + i);
        L.add(p + ".block.0.alpha", D);
        L.add(p + ".block.1.weight", D * (D / 2) * 2 * c->decoder_rates[i]);
        L.add(p + ".block.1.bias", D / 2);
        for (int j = 0; j < 3; ++j) res_unit_tensors(L, p + ".block." + std::to_string(2 + j), D / 2);
        D /= 2;
    }
    L.add("decoder.model." + std::to_string(n + 1) + ".alpha", D);
    L.add("decoder.model." + std::to_string(n + 2) + ".weight", D * 7);
    L.add("decoder.model." + std::to_string(n + 2) + ".bias", 1);
    return L;
}

// ---- weight re-layout kernels (create time only) ----------------------------------------------------------------------------
// Conv1d weight (C_out, C_in, k) -> [C_out][k][C_in]
__global__ void perm_conv_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin, int k) {
    const long n = (long)cout * cin * k;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        const int ci = (int)(i % cin);
        const long r = i / cin;
        const int j = (int)(r % k), co = (int)(r / k);
        out[i] = w[((long)co * cin + ci) * k + j];
    }
}
// ConvTranspose1d weight (C_in, C_out, 2 s) -> phase r: [C_out][2][C_in] = w[ci][co][r + jj s]
__global__ void perm_convT_kernel(const float* __restrict__ w, float* __restrict__ out, int cin, int cout, int s) {
    const long per = (long)cout * 2 * cin, n = per * s;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        const int r = (int)(i / per);
        const long q = i - (long)r * per;
        const int ci = (int)(q % cin), jj = (int)((q / cin) & 1), co = (int)(q / (2L * cin));
        out[i] = w[((long)ci * cout + co) * (2 * s) + r + jj * s];
    }
}

struct Conv {
    const float* w = nullptr;        // [C_out][k][C_in] (convT: phase 0; the phases are w + r * cout * 2 * cin)
    const float* b = nullptr;
    int cout = 0, cin = 0, k = 0, s = 0;
    const uint16_t* w16 = nullptr;   // tiled planes (convT: phase 0; phases w16 + r * NP * cout * 2 * cin)
};
struct Res { const float *a1, *a2; Conv c7, c1; };

#define FAKE_BASE (1ULL << 60)
struct PBuf { size_t nbytes; int first = -1, last = -1, io = 0; };
struct Act { int f32 = -1, p16 = -1; };

struct Builder {
    vn_ctx* ctx;
    const vn_codec_cfg* cfg;
    const float* blob;
    Layout lay;
    int precision;                   // 0 f32, 2 bf16x3, 3 f16x2
    hipStream_t s = nullptr;
    std::vector<void*> owned;
    std::vector<PBuf> bufs;
    std::vector<vn_codec_op> ops;
    int rc = VN_OK;

    int np() const { return precision == 3 ? 2 : 3; }
    const float* T(const std::string& name) {
        const Tensor* t = lay.find(name.c_str());
        if (!t) { rc = vn_fail(ctx, VN_ERR_INVALID, "codec: no tensor named %s", name.c_str()); return nullptr; }
        return blob + t->off;
    }
    template <typename X>
    X* dev(size_t n) {
        void* p = nullptr;
        if (vn_dev_malloc(&p, n * sizeof(X)) != hipSuccess) { rc = vn_fail(ctx, VN_ERR_OOM, "codec: hipMalloc of %s%ld bytes", "", (long)(n * sizeof(X))); return nullptr; }
        owned.push_back(p);
        return (X*)p;
    }
    // the routing rule of vampnet_amd/codec.py (_on_x3): on the split-plane pipe where that is MFMA-bound
    bool on_x3(const Conv& c, int taps) const {
        const double eff = c.cout / (128.0 * ceil(c.cout / 128.0));
        if (precision == 2 && (c.cout == 96 || c.cout == 192) && c.cin % 32 == 0) {     // CONVT (gemm_x3.hip): no padding
            static const int min_work = [] { const char* e = getenv("VN_CODEC_CONVT_MIN_WORK"); return e && *e ? atoi(e) : 512; }();
            return taps * c.cin >= min_work;
        }
        return (precision == 2 || precision == 3) && c.cout >= 128 && c.cout % 16 == 0 && c.cin % 32 == 0 && taps * c.cin * eff >= 512;
    }
    const uint16_t* tile_planes(const float* w2d, long rows, int K) {
        uint16_t* t = dev<uint16_t>((size_t)np() * rows * K);
        if (!t) return nullptr;
        int r;
        if (precision == 3) r = vn_launch_split2h(ctx, w2d, t, rows, K, VN_PLANES_TILED_H2, s);
        else r = vn_launch_split3_tiled(ctx, w2d, t, (int)rows, K, K, s);
        if (r) rc = r;
        return t;
    }
    Conv conv(const std::string& key, int cout, int cin, int k) {
        Conv c;
        c.cout = cout; c.cin = cin; c.k = k;
        const float* w = T(key + ".weight");
        c.b = T(key + ".bias");
        float* o = dev<float>((size_t)cout * cin * k);
        if (!w || !o) return c;
        hipLaunchKernelGGL(perm_conv_kernel, dim3(256), dim3(256), 0, s, w, o, cout, cin, k);
        c.w = o;
        if (on_x3(c, k)) c.w16 = tile_planes(o, cout, k * cin);
        return c;
    }
    Conv convT(const std::string& key, int cin, int cout, int st) {
        Conv c;
        c.cout = cout; c.cin = cin; c.k = 2; c.s = st;
        const float* w = T(key + ".weight");
        c.b = T(key + ".bias");
        float* o = dev<float>((size_t)st * cout * 2 * cin);
        if (!w || !o) return c;
        hipLaunchKernelGGL(perm_convT_kernel, dim3(256), dim3(256), 0, s, w, o, cin, cout, st);
        c.w = o;
        if (on_x3(c, 2)) {
            uint16_t* t = dev<uint16_t>((size_t)st * np() * cout * 2 * cin);
            if (!t) return c;
            for (int r = 0; r < st && rc == VN_OK; ++r) {
                const float* src = o + (size_t)r * cout * 2 * cin;
                uint16_t* dst = t + (size_t)r * np() * cout * 2 * cin;
                const int q = precision == 3 ? vn_launch_split2h(ctx, src, dst, cout, 2 * cin, VN_PLANES_TILED_H2, s)
                                             : vn_launch_split3_tiled(ctx, src, dst, cout, 2 * cin, 2 * cin, s);
                if (q) rc = q;
            }
            c.w16 = t;
        }
        return c;
    }
    Res res(const std::string& q, int C) {
        Res r;
        r.a1 = T(q + ".block.0.alpha");
        r.c7 = conv(q + ".block.1", C, C, 7);
        r.a2 = T(q + ".block.2.alpha");
        r.c1 = conv(q + ".block.3", C, C, 1);
        return r;
    }

    // ---- program recording ----------------------------------------------------------------------------------------------
    int new_buf(size_t nbytes, int io = 0) { bufs.push_back(PBuf{nbytes, -1, -1, io}); return (int)bufs.size() - 1; }
    int new_f32(long B, long Tn, long C, int io = 0) { return new_buf((size_t)B * Tn * C * 4, io); }
    int new_planes(long rows, long cols) { return new_buf((size_t)np() * rows * cols * 2); }
    void* ptr(int id) const {
        if (id < 0) return nullptr;
        if (bufs[id].io == 1) return (void*)VN_CODEC_PTR_IN;
        if (bufs[id].io == 2) return (void*)VN_CODEC_PTR_OUT;
        return (void*)(FAKE_BASE + (unsigned long long)id);
    }
    void emit(vn_codec_op& o) {
        const int k = (int)ops.size();
        for (int j = 0; j < 12; ++j) {
            const unsigned long long v = (unsigned long long)o.p[j];
            if (v >= FAKE_BASE && v < FAKE_BASE + bufs.size()) {
                PBuf& b = bufs[v - FAKE_BASE];
                if (b.first < 0) b.first = k;
                b.last = k;
            }
        }
        ops.push_back(o);
    }
    bool fmt_x3(const Conv& c, int taps = -1) const { return on_x3(c, taps < 0 ? c.k : taps); }

    // fp32 [B][T][C] -> Act with split planes as well (a producer outside the conv stack feeding the pipe)
    Act planes_of(int x, long R, long K) {
        Act a;
        a.f32 = x;
        a.p16 = new_planes(R, K);
        vn_codec_op o{};
        o.p[0] = ptr(x); o.p[1] = ptr(a.p16);
        if (precision == 3) { o.kind = VN_CODEC_OP_SPLIT2; o.l[0] = R; o.i[0] = (int)K; o.l[1] = R * K; o.i[1] = 0; }
        else { o.kind = VN_CODEC_OP_SPLIT3; o.l[0] = R * K; o.l[1] = R * K; }
        emit(o);
        return a;
    }

    // one convolution launch (vampnet_amd/codec.py: _conv).  s_fmt: 0 none, 1 "f32", 2 "x3", 3 "both"
    struct ConvOut { int y; Act sn; };
    ConvOut conv_op(const Act& x, const Conv& c, int B, int T_in, int T_rows, int T_out, int phase = -1, int taps = -1, int in_stride = 1,
                    int dil = 1, int pad = 0, int out_stride = 1, int out_off = 0, int resid = -1, const float* alpha = nullptr,
                    bool want_raw = true, int s_fmt = 0, int act = 0, int out_y = -1, const Act* out_sn = nullptr) {
        if (taps < 0) taps = c.k;
        const bool x3 = on_x3(c, taps);
        const int src = x3 ? x.p16 : x.f32;
        if (src < 0) { rc = vn_fail(ctx, VN_ERR_INVALID, "codec: producer / consumer format mismatch in the layer graph%s", ""); return {-1, Act{}}; }
        int y = out_y;
        Act sn = out_sn ? *out_sn : Act{};
        const bool have_out = out_y >= 0 || out_sn != nullptr;
        if (want_raw && y < 0 && !have_out) y = new_f32(B, T_out, c.cout);
        if (s_fmt && !out_sn) {
            if (s_fmt & 1) sn.f32 = new_f32(B, T_out, c.cout);
            if (s_fmt & 2) sn.p16 = new_planes((long)B * T_out, c.cout);
        }
        const long plane = (long)B * T_out * c.cout;
        vn_codec_op o{};
        const long wph = (long)c.cout * 2 * c.cin;           // floats of one phase of a transposed convolution
        if (x3) {
            o.kind = precision == 3 ? VN_CODEC_OP_CONV1D_F16X2 : VN_CODEC_OP_CONV1D_BF16X3;
            o.p[0] = ptr(src);
            o.l[0] = (long)B * T_in * c.cin;
            o.p[1] = (void*)(phase < 0 ? c.w16 : c.w16 + (size_t)phase * np() * wph);
            o.l[1] = plane;
        } else {
            o.kind = VN_CODEC_OP_CONV1D_F32;
            o.p[0] = ptr(src);
            o.p[1] = (void*)(phase < 0 ? c.w : c.w + (size_t)phase * wph);
            o.l[0] = precision == 3 ? -plane : plane;     // the fp32 kernel writes two fp16 planes when the stride is negative
        }
        o.p[2] = (void*)c.b; o.p[3] = ptr(resid); o.p[4] = (void*)alpha; o.p[5] = ptr(y); o.p[6] = ptr(sn.f32); o.p[7] = ptr(sn.p16);
        const int iv[13] = {B, T_in, T_rows, T_out, c.cin, c.cout, taps, in_stride, dil, pad, out_stride, out_off, act};
        for (int j = 0; j < 13; ++j) o.i[j] = iv[j];
        emit(o);
        return {y, sn};
    }
    // ResidualUnit: x + conv1(snake(conv7(snake(x))))
    ConvOut res_unit(int x, const Act& sx, const Res& r, int dil, const float* alpha_next, int B, int Tn, bool next_x3, bool next_both_none = false) {
        ConvOut h = conv_op(sx, r.c7, B, Tn, Tn, Tn, -1, -1, 1, dil, 3 * dil, 1, 0, -1, r.a2, false, fmt_x3(r.c1) ? 2 : 1);
        (void)next_both_none;
        return conv_op(h.sn, r.c1, B, Tn, Tn, Tn, -1, -1, 1, 1, 0, 1, 0, x, alpha_next, true, next_x3 ? 2 : 1);
    }

    // first-fit arena plan over the buffers' live ranges (vampnet_amd/codec.py: _Recorder.plan)
    size_t plan(std::vector<size_t>& off, size_t align = 256) {
        struct Live { int last; size_t o, sz; };
        struct Free { size_t o, sz; };
        std::vector<std::pair<int, int>> order;
        for (int i = 0; i < (int)bufs.size(); ++i)
            if (bufs[i].first >= 0 && bufs[i].io == 0) order.push_back({bufs[i].first, i});
        std::sort(order.begin(), order.end());
        std::vector<Live> live;
        std::vector<Free> freeb;
        size_t end = 0;
        off.assign(bufs.size(), 0);
        for (auto& fi : order) {
            const int first = fi.first, i = fi.second;
            for (size_t q = 0; q < live.size();) {
                if (live[q].last < first) { freeb.push_back({live[q].o, live[q].sz}); live.erase(live.begin() + q); }
                else ++q;
            }
            const size_t need = (bufs[i].nbytes + align - 1) / align * align;
            std::sort(freeb.begin(), freeb.end(), [](const Free& a, const Free& b) { return a.o != b.o ? a.o < b.o : a.sz < b.sz; });
            int slot = -1;
            for (size_t q = 0; q < freeb.size(); ++q)
                if (freeb[q].sz >= need) { slot = (int)q; break; }
            size_t o;
            if (slot < 0) { o = end; end += need; }
            else {
                o = freeb[slot].o;
                const size_t sz = freeb[slot].sz;
                freeb.erase(freeb.begin() + slot);
                if (sz > need) freeb.push_back({o + need, sz - need});
            }
            off[i] = o;
            live.push_back({bufs[i].last, o, need});
        }
        return end;
    }
};

}  // namespace

extern "C" int vn_codec_weights_size(const vn_codec_cfg* cfg, int64_t* n_floats) {
    if (!cfg_ok(cfg) || !n_floats) return VN_ERR_INVALID;
    *n_floats = make_layout(cfg).total;
    return VN_OK;
}
extern "C" int vn_codec_tensor_count(const vn_codec_cfg* cfg, int* n) {
    if (!cfg_ok(cfg) || !n) return VN_ERR_INVALID;
    *n = (int)make_layout(cfg).t.size();
    return VN_OK;
}
extern "C" int vn_codec_tensor_name(const vn_codec_cfg* cfg, int index, char* name, int name_len, int64_t* offset, int64_t* count) {
    if (!cfg_ok(cfg) || !name || name_len <= 0 || !offset || !count) return VN_ERR_INVALID;
    const Layout L = make_layout(cfg);
    if (index < 0 || index >= (int)L.t.size()) return VN_ERR_INVALID;
    snprintf(name, (size_t)name_len, "%s", L.t[index].name.c_str());
    *offset = L.t[index].off;
    *count = L.t[index].count;
    return VN_OK;
}
extern "C" int vn_codec_tensor_offset(const vn_codec_cfg* cfg, const char* name, int64_t* offset, int64_t* count) {
    if (!cfg_ok(cfg) || !name || !offset || !count) return VN_ERR_INVALID;
    const Layout L = make_layout(cfg);
    const Tensor* t = L.find(name);
    if (!t) return VN_ERR_INVALID;
    *offset = t->off;
    *count = t->count;
    return VN_OK;
}

extern "C" int vn_codec_create_from_weights(vn_ctx* ctx, const vn_codec_cfg* cfg, const float* blob_dev, int direction, int B, int n,
                                            int precision, vn_codec** out) {
    if (!ctx || !out) return VN_ERR_INVALID;
    *out = nullptr;
    if (!cfg_ok(cfg)) return vn_fail(ctx, VN_ERR_INVALID, "vn_codec_create_from_weights: bad configuration%s", "");
    if (!blob_dev || ((uintptr_t)blob_dev & 255)) return vn_fail(ctx, VN_ERR_INVALID, "vn_codec_create_from_weights: the weight blob must be 256-byte aligned%s", "");
    if ((direction != 0 && direction != 1) || B <= 0 || n <= 0) return vn_fail(ctx, VN_ERR_INVALID, "vn_codec_create_from_weights: direction 0 / 1, B > 0, length > 0%s", "");
    if (precision != 0 && precision != 2 && precision != 3)
        return vn_fail(ctx, VN_ERR_INVALID, "vn_codec_create_from_weights: precision %s%ld is not 0 (f32) / 2 (bf16x3) / 3 (f16x2)", "", precision);
    if (cfg->codebook_dim != 8) return vn_fail(ctx, VN_ERR_UNSUPPORTED, "vn_codec_create_from_weights: the RVQ kernels take codebook_dim 8 (got %s%ld)", "", cfg->codebook_dim);
    VN_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const int nr = cfg->n_rates, lat = latent_of(cfg), ncb = cfg->n_codebooks;
    long hop = 1;
    for (int i = 0; i < nr; ++i) hop *= cfg->encoder_rates[i];
    if (direction == 0 && n % hop) return vn_fail(ctx, VN_ERR_INVALID, "vn_codec_create_from_weights: %s%ld samples are not a multiple of the hop %ld", "", n, hop);
    Builder bd;
    bd.ctx = ctx; bd.cfg = cfg; bd.blob = blob_dev; bd.lay = make_layout(cfg); bd.precision = precision;
    auto fail = [&](int rc) {
        (void)hipDeviceSynchronize();
        for (void* p : bd.owned) (void)vn_dev_free(p);
        return rc;
    };
    const float* win = bd.T("quantizer.quantizers.0.in_proj.weight");
    const float* bin = bd.T("quantizer.quantizers.0.in_proj.bias");
    const float* cb = bd.T("quantizer.quantizers.0.codebook.weight");
    const float* wout = bd.T("quantizer.quantizers.0.out_proj.weight");
    const float* bout = bd.T("quantizer.quantizers.0.out_proj.bias");
    if (direction == 0) {
        // ---- encoder (vampnet_amd/codec.py: _encode_layers)
        const int L = n;
        std::vector<std::vector<Res>> rs(nr);
        std::vector<const float*> a_blk(nr);
        std::vector<Conv> down(nr);
        int C = cfg->encoder_dim;
        const float* stem_w = bd.T("encoder.block.0.weight");
        const float* stem_b = bd.T("encoder.block.0.bias");
        const int C0 = C;
        for (int i = 0; i < nr; ++i) {
            const std::string p = "encoder.block." + std::to_string(1 + i);
            for (int j = 0; j < 3; ++j) rs[i].push_back(bd.res(p + ".block." + std::to_string(j), C));
            a_blk[i] = bd.T(p + ".block.3.alpha");
            down[i] = bd.conv(p + ".block.4", 2 * C, C, 2 * cfg->encoder_rates[i]);
            C *= 2;
        }
        const float* a_out = bd.T("encoder.block." + std::to_string(nr + 1) + ".alpha");
        const Conv outc = bd.conv("encoder.block." + std::to_string(nr + 2), lat, C, 3);
        if (bd.rc) return fail(bd.rc);
        const int xin = bd.new_buf((size_t)B * L * 4, 1);
        const int cur0 = bd.new_f32(B, L, C0), s0 = bd.new_f32(B, L, C0);
        {
            vn_codec_op o{};
            o.kind = VN_CODEC_OP_CONV_IN;
            o.p[0] = bd.ptr(xin); o.p[1] = (void*)stem_w; o.p[2] = (void*)stem_b; o.p[3] = (void*)rs[0][0].a1; o.p[4] = bd.ptr(cur0); o.p[5] = bd.ptr(s0);
            o.i[0] = B; o.i[1] = L; o.i[2] = C0;
            bd.emit(o);
        }
        int cur = cur0;
        Act sx;
        if (bd.fmt_x3(rs[0][0].c7)) sx = bd.planes_of(s0, (long)B * L, C0);
        else sx.f32 = s0;
        int Tn = L;
        static const int dils[3] = {1, 3, 9};
        for (int bi = 0; bi < nr; ++bi) {
            for (int j = 0; j < 3; ++j) {
                const float* a_next = j < 2 ? rs[bi][j + 1].a1 : a_blk[bi];
                const bool nx = j < 2 ? bd.fmt_x3(rs[bi][j + 1].c7) : bd.fmt_x3(down[bi]);
                Builder::ConvOut r = bd.res_unit(cur, sx, rs[bi][j], dils[j], a_next, B, Tn, nx);
                cur = r.y; sx = r.sn;
            }
            const int st = cfg->encoder_rates[bi];
            const int pad = (st + 1) / 2;
            const int T_out = (Tn + 2 * pad - 2 * st) / st + 1;
            const float* a_next = bi + 1 < nr ? rs[bi + 1][0].a1 : a_out;
            const bool nx = bi + 1 < nr ? bd.fmt_x3(rs[bi + 1][0].c7) : bd.fmt_x3(outc);
            Builder::ConvOut r = bd.conv_op(sx, down[bi], B, Tn, T_out, T_out, -1, -1, st, 1, pad, 1, 0, -1, a_next, bi + 1 < nr, nx ? 2 : 1);
            cur = r.y; sx = r.sn;
            Tn = T_out;
        }
        Builder::ConvOut z = bd.conv_op(sx, outc, B, Tn, Tn, Tn, -1, -1, 1, 1, 1);
        const int codes = bd.new_buf((size_t)B * ncb * Tn * 8, 2);
        vn_codec_op o{};
        o.kind = VN_CODEC_OP_RVQ_ENCODE;
        o.p[0] = bd.ptr(z.y); o.p[1] = (void*)win; o.p[2] = (void*)bin; o.p[3] = (void*)cb; o.p[4] = (void*)wout; o.p[5] = (void*)bout; o.p[6] = bd.ptr(codes);
        o.i[0] = B; o.i[1] = Tn; o.i[2] = lat; o.i[3] = ncb; o.i[4] = cfg->codebook_size;
        bd.emit(o);
    } else {
        // ---- decoder (vampnet_amd/codec.py: _decode_layers)
        int Tn = n;
        int D = cfg->decoder_dim;
        const Conv inc = bd.conv("decoder.model.0", D, lat, 7);
        std::vector<const float*> a_blk(nr);
        std::vector<Conv> up(nr);
        std::vector<std::vector<Res>> rs(nr);
        for (int i = 0; i < nr; ++i) {
            const std::string p = "decoder.model." + std::to_string(1 + i);
            a_blk[i] = bd.T(p + ".block.0.alpha");
            up[i] = bd.convT(p + ".block.1", D, D / 2, cfg->decoder_rates[i]);
            for (int j = 0; j < 3; ++j) rs[i].push_back(bd.res(p + ".block." + std::to_string(2 + j), D / 2));
            D /= 2;
        }
        const float* a_out = bd.T("decoder.model." + std::to_string(nr + 1) + ".alpha");
        const float* head_src = bd.T("decoder.model." + std::to_string(nr + 2) + ".weight");       // (1, C, 7) -> [7][C]
        const float* head_bias = bd.T("decoder.model." + std::to_string(nr + 2) + ".bias");
        float* head_w = bd.dev<float>((size_t)7 * D);
        if (bd.rc || !head_w) return fail(bd.rc ? bd.rc : VN_ERR_OOM);
        hipLaunchKernelGGL(perm_conv_kernel, dim3(64), dim3(256), 0, bd.s, head_src, head_w, 1, D, 7);
        float head_b = 0.f;
        if (hipMemcpy(&head_b, head_bias, sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return fail(VN_ERR_HIP);
        const int codes = bd.new_buf((size_t)B * ncb * Tn * 8, 1);
        const int zq = bd.new_f32(B, Tn, lat);
        {
            vn_codec_op o{};
            o.kind = VN_CODEC_OP_RVQ_DECODE;
            o.p[0] = bd.ptr(codes); o.p[1] = (void*)cb; o.p[2] = (void*)wout; o.p[3] = (void*)bout; o.p[4] = bd.ptr(zq);
            o.i[0] = B; o.i[1] = Tn; o.i[2] = lat; o.i[3] = ncb; o.i[4] = cfg->codebook_size;
            bd.emit(o);
        }
        Act zin;
        if (bd.fmt_x3(inc)) zin = bd.planes_of(zq, (long)B * Tn, lat);
        else zin.f32 = zq;
        Builder::ConvOut r0 = bd.conv_op(zin, inc, B, Tn, Tn, Tn, -1, -1, 1, 1, 3, 1, 0, -1, a_blk[0], false, bd.fmt_x3(up[0], 2) ? 2 : 1);
        Act sx = r0.sn;
        int cur = -1;
        static const int dils[3] = {1, 3, 9};
        for (int bi = 0; bi < nr; ++bi) {
            const int st = cfg->decoder_rates[bi];
            const int pad = (st + 1) / 2;
            const int T_out = (Tn - 1) * st - 2 * pad + 2 * st;
            const bool f0 = bd.fmt_x3(rs[bi][0].c7);
            const int y = bd.new_f32(B, T_out, up[bi].cout);
            Act y2;
            if (f0) y2.p16 = bd.new_planes((long)B * T_out, up[bi].cout);
            else y2.f32 = bd.new_f32(B, T_out, up[bi].cout);
            for (int ph = 0; ph < st; ++ph)      // polyphase: output rows t = t' st + ph - pad read x[t'] and x[t' - 1]
                bd.conv_op(sx, up[bi], B, Tn, Tn + 1, T_out, ph, 2, 1, -1, 0, st, ph - pad, -1, rs[bi][0].a1, true, f0 ? 2 : 1, 0, y, &y2);
            cur = y; sx = y2; Tn = T_out;
            for (int j = 0; j < 3; ++j) {
                const float* a_next;
                bool nx;
                if (j < 2) { a_next = rs[bi][j + 1].a1; nx = bd.fmt_x3(rs[bi][j + 1].c7); }
                else if (bi + 1 < nr) { a_next = a_blk[bi + 1]; nx = bd.fmt_x3(up[bi + 1], 2); }
                else { a_next = a_out; nx = false; }            // the 1-channel head reads fp32
                Builder::ConvOut r = bd.res_unit(cur, sx, rs[bi][j], dils[j], a_next, B, Tn, nx);
                cur = r.y; sx = r.sn;
            }
        }
        const int audio = bd.new_buf((size_t)B * Tn * 4, 2);
        vn_codec_op o{};
        o.kind = VN_CODEC_OP_CONV_OUT;
        o.p[0] = bd.ptr(sx.f32); o.p[1] = (void*)head_w; o.f[0] = head_b; o.p[2] = bd.ptr(audio);
        o.i[0] = B; o.i[1] = Tn; o.i[2] = D;
        bd.emit(o);
    }
    if (bd.rc) return fail(bd.rc);
    // ---- the arena, then the real addresses
    std::vector<size_t> off;
    const size_t total = bd.plan(off);
    char* arena = bd.dev<char>(total > 0 ? total : 256);
    if (!arena) return fail(VN_ERR_OOM);
    for (vn_codec_op& o : bd.ops)
        for (int j = 0; j < 12; ++j) {
            const unsigned long long v = (unsigned long long)o.p[j];
            if (v >= FAKE_BASE && v < FAKE_BASE + bd.bufs.size()) o.p[j] = arena + off[v - FAKE_BASE];
        }
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return fail(VN_ERR_HIP);      // the re-laid weights are in place
    vn_codec* c = vn_codec_adopt(ctx, direction, std::move(bd.ops), std::move(bd.owned));
    if (!c) return fail(VN_ERR_OOM);
    *out = c;
    return VN_OK;
}
