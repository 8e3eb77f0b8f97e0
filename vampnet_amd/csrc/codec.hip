// The DAC codec as ONE C call per direction (vn_dac_encode / vn_dac_decode; rows a18 / a19, PARITY UNPINNED like
// the layer kernels it runs: `lac` is not part of the reference tree).
//
// Replaces the per-layer ctypes calls of round 3 (vampnet_amd/codec.py issued ~70 / ~150 calls per encode / decode).  The host builds,
// once per (direction, batch, length, precision), a PROGRAM — the ordered launches of the layer loop with their shapes and pointers into
// ONE arena whose offsets the host planned from the buffers' live ranges (vampnet_amd/codec.py: _Recorder) — and hands it over
// (vn_codec_create copies it).  A request is then vn_dac_encode(prog, audio, codes, stream) / vn_dac_decode(prog, codes, audio,
// stream): the executor below walks the list and calls the same launchers the single-layer entry points call.  Nothing is allocated
// and nothing synchronises; all launches go to the caller's stream in order.
//
// Reference call sites: Interface.encode -> codec.encode(...)["codes"] (vampnet/interface.py:219-224), Interface.decode ->
// VampNet.decode -> codec.quantizer.from_latents + codec.decode (vampnet/interface.py:203-204, vampnet/modules/transformer.py:661-684).
#include <new>
#include <vector>
#include "vn_common.h"

struct vn_codec {
    vn_ctx* ctx;
    int direction;                         // 0 = encode (audio -> codes), 1 = decode (codes -> audio)
    std::vector<vn_codec_op> ops;
    std::vector<void*> owned;              // device allocations the program owns (vn_codec_create_from_weights: re-laid weights + arena)
};

// a program whose weights and arena were built in C (codec_plan.hip): takes the op list and the device allocations over
vn_codec* vn_codec_adopt(vn_ctx* ctx, int direction, std::vector<vn_codec_op>&& ops, std::vector<void*>&& owned) {
    vn_codec* c = new (std::nothrow) vn_codec();
    if (!c) return nullptr;
    c->ctx = ctx;
    c->direction = direction;
    c->ops = std::move(ops);
    c->owned = std::move(owned);
    return c;
}

extern "C" int vn_codec_create(vn_ctx* ctx, const vn_codec_op* ops, int n_ops, int direction, vn_codec** out) {
    if (!ctx || !ops || n_ops <= 0 || !out || (direction != 0 && direction != 1)) return VN_ERR_INVALID;
    *out = nullptr;
    for (int k = 0; k < n_ops; ++k)
        if (ops[k].kind < 0 || ops[k].kind >= VN_CODEC_OP__COUNT) return vn_fail(ctx, VN_ERR_INVALID, "vn_codec_create: op %s%ld has an unknown kind", "", k);
    vn_codec* c = new (std::nothrow) vn_codec();
    if (!c) return VN_ERR_OOM;
    c->ctx = ctx;
    c->direction = direction;
    c->ops.assign(ops, ops + n_ops);
    *out = c;
    return VN_OK;
}

extern "C" void vn_codec_destroy(vn_codec* c) {
    if (!c) return;
    for (void* p : c->owned) (void)vn_dev_free(p);     // hipFree waits for the device: no launch of this program is in flight afterwards
    delete c;
}

// pointer slot of an op: the program's own pointers are absolute device addresses; VN_CODEC_PTR_IN / _OUT stand for the call's arguments
static inline void* bind(void* p, const void* in, void* out) {
    if (p == (void*)VN_CODEC_PTR_IN) return (void*)in;
    if (p == (void*)VN_CODEC_PTR_OUT) return out;
    return p;
}

static int codec_run(vn_codec* c, const void* in, void* out, void* stream) {
    vn_ctx* ctx = c->ctx;
    for (size_t k = 0; k < c->ops.size(); ++k) {
        const vn_codec_op& o = c->ops[k];
        void* p[12];
        for (int j = 0; j < 12; ++j) p[j] = bind(o.p[j], in, out);
        const int32_t* i = o.i;
        int rc = VN_ERR_INVALID;
        switch (o.kind) {
            case VN_CODEC_OP_CONV1D_F32:
                rc = vn_conv1d_f32(ctx, (const float*)p[0], (const float*)p[1], (const float*)p[2], (const float*)p[3], (const float*)p[4],
                                   (float*)p[5], (float*)p[6], p[7], o.l[0], i[0], i[1], i[2], i[3], i[4], i[5], i[6], i[7], i[8], i[9], i[10],
                                   i[11], i[12], stream);
                break;
            case VN_CODEC_OP_CONV1D_BF16X3:
                rc = vn_conv1d_bf16x3(ctx, p[0], o.l[0], p[1], (const float*)p[2], (const float*)p[3], (const float*)p[4], (float*)p[5],
                                      (float*)p[6], p[7], o.l[1], i[0], i[1], i[2], i[3], i[4], i[5], i[6], i[7], i[8], i[9], i[10], i[11], i[12],
                                      stream);
                break;
            case VN_CODEC_OP_CONV1D_F16X2:
                rc = vn_conv1d_f16x2(ctx, p[0], o.l[0], p[1], (const float*)p[2], (const float*)p[3], (const float*)p[4], (float*)p[5],
                                     (float*)p[6], p[7], o.l[1], i[0], i[1], i[2], i[3], i[4], i[5], i[6], i[7], i[8], i[9], i[10], i[11], i[12],
                                     stream);
                break;
            case VN_CODEC_OP_CONV_IN:
                rc = vn_dac_conv_in_f32(ctx, (const float*)p[0], (const float*)p[1], (const float*)p[2], (const float*)p[3], (float*)p[4],
                                        (float*)p[5], i[0], i[1], i[2], stream);
                break;
            case VN_CODEC_OP_CONV_OUT:
                rc = vn_dac_conv_out_f32(ctx, (const float*)p[0], (const float*)p[1], o.f[0], (float*)p[2], i[0], i[1], i[2], stream);
                break;
            case VN_CODEC_OP_RVQ_ENCODE:
                rc = vn_rvq_encode_f32(ctx, (const float*)p[0], (const float*)p[1], (const float*)p[2], (const float*)p[3], (const float*)p[4],
                                       (const float*)p[5], (int64_t*)p[6], i[0], i[1], i[2], i[3], i[4], stream);
                break;
            case VN_CODEC_OP_RVQ_DECODE:
                rc = vn_rvq_decode_f32(ctx, (const int64_t*)p[0], (const float*)p[1], (const float*)p[2], (const float*)p[3], (float*)p[4],
                                       i[0], i[1], i[2], i[3], i[4], stream);
                break;
            case VN_CODEC_OP_SPLIT3:
                rc = vn_split3_f32(ctx, (const float*)p[0], p[1], o.l[0], o.l[1], stream);
                break;
            case VN_CODEC_OP_SPLIT2:
                rc = vn_split2_f16(ctx, (const float*)p[0], p[1], o.l[0], i[0], o.l[1], i[1], stream);
                break;
        }
        if (rc != VN_OK) {
            if (!ctx->err[0] || rc == VN_ERR_INVALID) {
                char keep[256];
                snprintf(keep, sizeof(keep), "%s", ctx->err);
                snprintf(ctx->err, sizeof(ctx->err), "codec program: op %ld (kind %d) failed with status %d: %s", (long)k, o.kind, rc, keep);
            }
            return rc;
        }
    }
    return VN_OK;
}

extern "C" int vn_dac_encode(vn_codec* c, const float* audio_dev, int64_t* codes_dev, void* stream) {
    if (!c || !audio_dev || !codes_dev) return VN_ERR_INVALID;
    if (c->direction != 0) return vn_fail(c->ctx, VN_ERR_INVALID, "vn_dac_encode: this program decodes%s", "");
    return codec_run(c, audio_dev, codes_dev, stream);
}

extern "C" int vn_dac_decode(vn_codec* c, const int64_t* codes_dev, float* audio_dev, void* stream) {
    if (!c || !codes_dev || !audio_dev) return VN_ERR_INVALID;
    if (c->direction != 1) return vn_fail(c->ctx, VN_ERR_INVALID, "vn_dac_decode: this program encodes%s", "");
    return codec_run(c, codes_dev, audio_dev, stream);
}
