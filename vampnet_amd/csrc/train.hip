// Host side of the VampNet TRAINING step (scripts/exp/train.py:237-304, conf/vampnet.yml) for gfx950:
// activation stash, kernel schedule of forward-with-dropout / cross-entropy / backward / clip + AdamW, and the
// extern "C" entry points (include/vampnet_hip.h, "training" section).  Everything is enqueued on the caller's
// stream; there is no host<->device synchronisation inside a step (loss, gradient norm and the valid-target count
// stay on the device).
//
// Parameter vector ("train vector"): [ packed inference blob | classifier weight_g | classifier weight_v ].
// The blob's VN_W_CLS_W region is DERIVED (g * v / ||v||, re-folded after every update); its slot in the gradient
// vector is scratch for dW and is zeroed before the norm.  Gradients, Adam moments and parameters share the layout,
// so the optimiser is element-wise and a data-parallel job all-reduces one flat buffer (or per-stage slices of it).
//
// Every linear y = x W^T needs dX = dY W and dW = dY^T X.  The one GEMM kernel (gemm_f32.hip) computes A B^T with both
// operands contiguous along the contraction axis, so
//   dX: A = dY [M][N], B = W^T [K][N]     -> W^T copies of all GEMM weights, refreshed once per update (vn_train_sync)
//   dW: A = dY^T [N][Mp], B = X^T [K][Mp] -> 64x64 LDS transposes of the two activations, Mp = M rounded up to 32 (zero
//                                            filled), ~7 % of a step's time — the fp32 kernel and VN_TRAIN_TN=0.
//   dW on the split-plane pipe (round 6, default): NO transposes.  gemm_x3.hip's TN operand mode contracts over the token axis of the
//   token-major tiled planes themselves — dY's planes are the ones its dX GEMM reads anyway, X's planes are the ones the forward GEMM
//   read, now kept in the layer's stash instead of a shared scratch buffer (vn_layer_stash::*_16).
#include <new>
#include <stdlib.h>
#include <vector>
#include "vn_common.h"
#include "vn_model.h"
#include "vn_train.h"

enum { SITE_ATTN = 0, SITE_RES1 = 1, SITE_FFN = 2, SITE_RES2 = 3 };

struct vn_layer_stash {
    float *x_in, *y1, *qkv, *lse, *a, *x_mid, *y3, *u, *g;
    uint16_t *qk16, *vt16;         // attention on the split-plane pipe (vn_train::ax3): the q / k planes and the blocked V^T planes instead of qkv
    // vn_train::tn: the tiled planes of the A operands of the layer's four forward GEMMs (y1 -> QKV, a -> Wo, y3 -> W1, g -> W2), kept for
    // the dW GEMMs of the backward pass; rows past the token count inside the last 16-row block stay ZERO (vn_train::tn_rows)
    uint16_t *y1_16, *a_16, *y3_16, *g_16;
};

struct vn_train {
    vn_model* m;
    float* params;                 // train vector (caller owned); params == m->blob
    long wsize, n_total, off_g, off_v;
    int NV;                        // classifier rows = Cp * vocab
    long Mp_max;
    // transposed GEMM weights
    float* wT;                     // [L][qkvT | woT | w1T | w2T] then clsT
    long wT_layer, wT_cls;
    std::vector<vn_layer_stash> st;
    float *x_last, *y_f;           // X[L], final-norm output
    // scratch
    float *tmp, *dxa, *dxb, *dh, *dg, *du, *dy, *dqkv, *da, *At, *Bt, *partial, *row_loss, *delta, *scal;
    float* dbias_partial;          // [L][max_batch*H*ceil(max_T/64)][64] per-block bucket sums of the bias-table gradient
    long dbias_slab;               // floats per layer slab
    double* npartial;
    int32_t *t32, *n_valid;
    int B, T;                      // shape of the stashed forward (0 = none)
    // LoRA fine-tuning (vn_train_enable_lora): the blob holds W_eff = W + s B A; w_base keeps the frozen W of the five
    // LoRA'd linears [L][qkv 3D^2 | wo D^2 | w1 4D^2 | w2 2D^2]; lora = caller-owned [L][5][At | B] vector
    bool lora;
    float lora_scale;
    float *lora_params, *w_base, *h8, *dh8;
    long n_lora;
    // The GEMMs of the step on the split-plane pipe (gemm_x3.hip; VN_TRAIN_X3, default on): tiled bf16x3 planes of the GEMM weights
    // and of their transposes (rebuilt by vn_train_sync after every update, same offsets x 3 as params / wT), and scratch planes for
    // the activation operand of a forward / dX GEMM (a16) and the two transposed operands of a dW GEMM (at16, bt16)
    bool x3;
    uint16_t *w16, *wT16, *a16, *at16, *bt16;
    // dW without transposes (gemm_x3.hip X3_MODE_TN; VN_TRAIN_TN, default on with x3): yf16 = the planes of the final norm's output (the
    // classifier's A operand), the layers' in vn_layer_stash.  tn_rows = the token count whose pad rows are known to be zero in all of them
    // (a forward with another token count clears the buffers first)
    bool tn;
    uint16_t* yf16;
    int tn_rows;
    // The layers' dW GEMMs on a SIDE stream (VN_TRAIN_OVERLAP, default on with tn): a weight gradient feeds nothing in the backward pass, so
    // its GEMM need not sit in the chain dY -> dX -> norm backward -> ... of the caller's stream.  It runs from a second context (its own
    // split-K workspace) on a low-priority stream, ordered by events: ready[i] (caller's stream: the planes of dY are in plane buffer i),
    // done[i] (side stream: its GEMM has read buffer i — the caller's stream waits for it before that buffer is written again), join (end
    // of every vn_train_backward call: the gradients are complete when the call's work on the caller's stream is).  The chain's small
    // kernels and the half-empty last rounds of its GEMMs then share the chip with the weight-gradient GEMMs.  Same kernels, same
    // operands: bitwise the same gradients.
    enum { NB = 4 };
    bool overlap;
    vn_ctx* ctx2;
    hipStream_t side;
    hipEvent_t ev_ready[NB], ev_done[NB], ev_join;
    // ... and the re-split of the weights after an update (vn_train_sync) runs there too, layer by layer: the NEXT forward waits for
    // layer l's planes (ev_w[l]; ev_w[L] = the classifier's) just before its first GEMM of that layer, so the splitters of the deeper
    // layers run beside the forward pass of the first ones.  Only the internal planes lag — the parameters themselves are final when
    // vn_train_update returns on the caller's stream.
    std::vector<hipEvent_t> ev_w;
    hipEvent_t ev_upd;
    bool w_pending;
    bool buf_busy[NB];             // a side GEMM of the current backward call reads this buffer
    bool side_used;
    int buf_next;
    uint16_t* a16r[NB];            // rotating plane buffers for the dY operands (a16r[0] == a16)
    // ... and the attention of the step as well (attention_x3.hip TRAIN forward, attention_train_x3.hip backward; VN_TRAIN_ATTN_X3, default
    // on with x3 when D % 128 == 0 and max_T fits the backward's LDS): the QKV GEMM writes the attention operands' planes (its
    // inference epilogue) into the layer's stash — no fp32 q / k / v exist; ax_ws = the backward's transposed / row-major plane images
    bool ax3;
    uint16_t* ax_ws;
    long qk_plane, vt_plane;
    int near_T, near_r;            // near_r (attention_train_x3.hip) of the T it was derived for
};

enum { LORA_Q = 0, LORA_V = 1, LORA_FC = 2, LORA_W1 = 3, LORA_W2 = 4 };
#define LORA_R 8
static void lora_shape(const vn_dims* d, int which, long* K, long* N) {
    const long D = d->d_model;
    *K = which == LORA_W2 ? 2 * D : D;
    *N = which == LORA_W1 ? 4 * D : D;
}
static long lora_layer_floats(const vn_dims* d) { return 14L * LORA_R * d->d_model; }
// offset of At (ab = 0, [K][8]) or B (ab = 1, [N][8]) of linear `which` in layer `layer`
static long lora_offset(const vn_dims* d, int layer, int which, int ab) {
    long off = lora_layer_floats(d) * layer;
    for (int w = 0; w < which; ++w) {
        long K, N;
        lora_shape(d, w, &K, &N);
        off += LORA_R * (K + N);
    }
    if (ab) {
        long K, N;
        lora_shape(d, which, &K, &N);
        off += LORA_R * K;
    }
    return off;
}

extern "C" int vn_lora_param_size(const vn_dims* dims, int64_t* n_floats) {
    if (!dims || !n_floats) return VN_ERR_INVALID;
    *n_floats = lora_layer_floats(dims) * dims->n_layers;
    return VN_OK;
}

extern "C" int vn_lora_param_offset(const vn_dims* dims, int layer, int which, int ab, int64_t* offset, int64_t* count) {
    if (!dims || !offset || !count || layer < 0 || layer >= dims->n_layers || which < 0 || which > 4 || ab < 0 || ab > 1)
        return VN_ERR_INVALID;
    long K, N;
    lora_shape(dims, which, &K, &N);
    *offset = lora_offset(dims, layer, which, ab);
    *count = LORA_R * (ab ? N : K);
    return VN_OK;
}

static long al64(long n) { return (n + 63) & ~63L; }

extern "C" int vn_train_param_size(const vn_dims* dims, int64_t* n_floats) {
    int64_t w = 0;
    int rc = vn_weights_size(dims, &w);
    if (rc) return rc;
    const long NV = (long)(dims->n_codebooks - dims->n_cond) * dims->vocab;
    *n_floats = w + al64(NV) + al64(NV * dims->d_model);
    return VN_OK;
}

extern "C" int vn_train_param_offset(const vn_dims* dims, int which, int64_t* offset, int64_t* count) {
    int64_t w = 0;
    int rc = vn_weights_size(dims, &w);
    if (rc) return rc;
    if (!offset || !count) return VN_ERR_INVALID;
    const long NV = (long)(dims->n_codebooks - dims->n_cond) * dims->vocab;
    if (which == 0) { *offset = w; *count = NV; return VN_OK; }
    if (which == 1) { *offset = w + al64(NV); *count = NV * dims->d_model; return VN_OK; }
    return VN_ERR_INVALID;
}

template <typename T>
static int talloc(vn_ctx* ctx, T** p, size_t n) {
    void* q = nullptr;
    if (vn_dev_malloc(&q, n * sizeof(T)) != hipSuccess) {
        vn_fail(ctx, VN_ERR_OOM, "hipMalloc of %s%ld bytes failed (training workspace)", "", (long)(n * sizeof(T)));
        return VN_ERR_OOM;
    }
    *p = (T*)q;
    return VN_OK;
}

extern "C" void vn_train_destroy(vn_train* t) {
    if (!t) return;
    (void)vn_dev_free(t->yf16);
    for (int i = 1; i < vn_train::NB; ++i) (void)vn_dev_free(t->a16r[i]);
    if (t->side) {
        (void)hipStreamSynchronize(t->side);
        for (int i = 0; i < vn_train::NB; ++i) { (void)hipEventDestroy(t->ev_ready[i]); (void)hipEventDestroy(t->ev_done[i]); }
        for (hipEvent_t e : t->ev_w) (void)hipEventDestroy(e);
        (void)hipEventDestroy(t->ev_upd);
        (void)hipEventDestroy(t->ev_join);
        (void)hipStreamDestroy(t->side);
    }
    if (t->ctx2) vn_ctx_destroy(t->ctx2);
    for (auto& s : t->st) {
        (void)vn_dev_free(s.qk16); (void)vn_dev_free(s.vt16);
        (void)vn_dev_free(s.y1_16); (void)vn_dev_free(s.a_16); (void)vn_dev_free(s.y3_16); (void)vn_dev_free(s.g_16);
        float* a[] = {s.x_in, s.y1, s.qkv, s.lse, s.a, s.x_mid, s.y3, s.u, s.g};
        for (float* p : a) (void)vn_dev_free(p);
    }
    float* b[] = {t->wT, t->x_last, t->y_f, t->tmp, t->dxa, t->dxb, t->dh, t->dg, t->du, t->dy, t->dqkv, t->da, t->At, t->Bt,
                  t->partial, t->row_loss, t->delta, t->scal, t->dbias_partial};
    for (float* p : b) (void)vn_dev_free(p);
    (void)vn_dev_free(t->npartial);
    (void)vn_dev_free(t->t32);
    (void)vn_dev_free(t->n_valid);
    (void)vn_dev_free(t->w_base);
    (void)vn_dev_free(t->h8);
    (void)vn_dev_free(t->dh8);
    uint16_t* c[] = {t->w16, t->wT16, t->a16, t->at16, t->bt16, t->ax_ws};
    for (uint16_t* p : c) (void)vn_dev_free(p);
    delete t;
}

static const float* P(const vn_train* t, int id, int layer = 0) { return t->params + vn_tensor_offset(&t->m->d, id, layer); }
static float* G(const vn_train* t, float* grads, int id, int layer = 0) { return grads + vn_tensor_offset(&t->m->d, id, layer); }

extern "C" int vn_train_create(vn_model* m, float* params, vn_train** out) {
    if (!m || !params || !out) return VN_ERR_INVALID;
    *out = nullptr;
    vn_ctx* ctx = m->ctx;
    if (params != m->blob) return vn_fail(ctx, VN_ERR_INVALID, "vn_train_create: params must be the model's weight blob (the train vector's prefix)%s", "");
    if (m->blob16) return vn_fail(ctx, VN_ERR_UNSUPPORTED, "vn_train_create: the model is in bf16 fast mode; training is fp32%s", "");
    vn_train* t = new (std::nothrow) vn_train();
    if (!t) return VN_ERR_OOM;
    t->m = m; t->params = params;
    t->lora = false; t->lora_params = t->w_base = t->h8 = t->dh8 = nullptr; t->n_lora = 0; t->lora_scale = 0.f;
    t->dbias_partial = nullptr;
    t->w16 = t->wT16 = t->a16 = t->at16 = t->bt16 = nullptr;
    { const char* e = getenv("VN_TRAIN_X3"); t->x3 = !(e && e[0] == '0'); }
    { const char* e = getenv("VN_TRAIN_TN"); t->tn = t->x3 && !(e && e[0] == '0'); }
    t->yf16 = nullptr; t->tn_rows = 0;
    t->overlap = false; t->ctx2 = nullptr; t->side = nullptr; t->buf_next = 0; t->side_used = false; t->w_pending = false;
    for (int i = 0; i < vn_train::NB; ++i) { t->a16r[i] = nullptr; t->buf_busy[i] = false; }
    const vn_dims& d = m->d;
    const long D = m->D, L = m->L, rows = m->max_rows;
    t->ax_ws = nullptr; t->near_T = 0; t->near_r = 0;
    {
        const char* e = getenv("VN_TRAIN_ATTN_X3");
        t->ax3 = t->x3 && !(e && e[0] == '0') && !(D & 127) && d.num_buckets <= 64 &&
                 vn_attention_x3_bwd_dq_lds(d.max_T, d.max_T - 1) <= 160 * 1024 && vn_attention_x3_lds_bytes(d.max_T, 0, 3) <= 160 * 1024;
        if (t->ax3) {       // the DMA descriptors of the backward address a plane set with 32-bit byte offsets
            vn_ax_bwd_ws w;
            vn_attention_x3_bwd_ws_layout(d.max_batch, m->H, d.max_T, &w);
            if (3 * (2 * rows * D) * 2 >= (1L << 31) || 3 * w.plane_r * 2 >= (1L << 31) || 3 * w.plane_t * 2 >= (1L << 31)) t->ax3 = false;
        }
    }
    t->qk_plane = 2 * rows * D;
    t->vt_plane = (long)m->H * ((rows + 31) / 32) * (VN_DHEAD * 32);
    t->NV = m->Cp * d.vocab;
    int64_t w = 0, n = 0;
    vn_weights_size(&d, &w);
    vn_train_param_size(&d, &n);
    t->wsize = w; t->n_total = n; t->off_g = w; t->off_v = w + al64(t->NV);
    t->Mp_max = (rows + 31) & ~31L;
    t->wT_layer = 3 * D * D + D * D + 4 * D * D + 2 * D * D;
    t->wT_cls = t->wT_layer * L;
    int rc = VN_OK;
    auto A = [&](float** p, size_t cnt) { if (rc == VN_OK) rc = talloc(ctx, p, cnt); };
    A(&t->wT, (size_t)t->wT_layer * L + (size_t)t->NV * D);
    t->st.resize(L);
    for (auto& s : t->st) {
        memset(&s, 0, sizeof(s));
        A(&s.x_in, rows * D); A(&s.y1, rows * D); A(&s.lse, (size_t)d.max_batch * m->H * d.max_T);
        if (!t->ax3) A(&s.qkv, 3 * rows * D);
        else {
            // masked rows / keys of a last tile are read as they are (and multiplied by P = 0): they must be finite, so start from zeros.
            // Exact sizes: the k planes own a 32-row pad (a last key tile reads up to 31 rows past T), the V^T planes are tile-exact;
            // the kernels' buffer descriptors carry these extents (an access past them would return 0, not fault) and the whole step
            // runs under guard pages in tests/test_gpu_guard.py
            const size_t nqk = ax_qk_elems(3, t->qk_plane), nvt = (size_t)3 * t->vt_plane;
            if (rc == VN_OK) rc = talloc(ctx, &s.qk16, nqk);
            if (rc == VN_OK) rc = talloc(ctx, &s.vt16, nvt);
            if (rc == VN_OK && (hipMemset(s.qk16, 0, nqk * 2) != hipSuccess || hipMemset(s.vt16, 0, nvt * 2) != hipSuccess)) rc = VN_ERR_HIP;
        }
        A(&s.a, rows * D); A(&s.x_mid, rows * D); A(&s.y3, rows * D); A(&s.u, 4 * rows * D); A(&s.g, 2 * rows * D);
    }
    const long wide = 4 * D > t->NV ? 4 * D : t->NV;
    A(&t->x_last, rows * D); A(&t->y_f, rows * D); A(&t->tmp, rows * D); A(&t->dxa, rows * D); A(&t->dxb, rows * D);
    A(&t->dh, rows * D); A(&t->dg, 2 * rows * D); A(&t->du, 4 * rows * D); A(&t->dy, rows * D); A(&t->dqkv, 3 * rows * D);
    A(&t->da, rows * D); A(&t->At, (size_t)wide * t->Mp_max); A(&t->Bt, (size_t)2 * D * t->Mp_max);
    size_t part = (size_t)vn_rmsnorm_bwd_blocks((int)rows) * D;
    const size_t cs = (size_t)vn_cdiv((int)rows, 64) * (size_t)(t->NV > D ? t->NV : D);
    if (cs > part) part = cs;
    const size_t eb = (size_t)vn_embed_bwd_partial_floats(d.max_batch, d.max_T, d.n_codebooks, d.latent_dim, (int)D);
    if (eb > part) part = eb;
    const size_t lp = (size_t)vn_lora_up_partial_floats((int)rows, (int)(4 * D));
    if (lp > part) part = lp;
    A(&t->partial, part);
    A(&t->row_loss, (size_t)rows * m->Cp + (size_t)d.num_buckets * m->H);   // + room for the LoRA-mode bias-gradient sink
    A(&t->delta, (size_t)d.max_batch * m->H * d.max_T); A(&t->scal, 16);
    t->dbias_slab = (long)d.max_batch * m->H * vn_cdiv(d.max_T, 64) * 64;
    A(&t->dbias_partial, (size_t)t->dbias_slab * L);
    if (t->x3) {
        const size_t rows16 = ((size_t)rows + 15) & ~(size_t)15;
        if (rc == VN_OK) rc = talloc(ctx, &t->w16, (size_t)3 * w);
        if (rc == VN_OK) rc = talloc(ctx, &t->wT16, (size_t)3 * ((size_t)t->wT_layer * L + (size_t)t->NV * D));
        if (rc == VN_OK) rc = talloc(ctx, &t->a16, (size_t)3 * rows16 * (size_t)wide);
        // (a TN GEMM contracts over the pad rows of a last 16-token block too: zero in the stash planes, and FINITE here — start from zeros)
        if (rc == VN_OK && hipMemset(t->a16, 0, (size_t)3 * rows16 * (size_t)wide * 2) != hipSuccess) rc = VN_ERR_HIP;
        if (rc == VN_OK) rc = talloc(ctx, &t->at16, (size_t)3 * wide * t->Mp_max);
        if (rc == VN_OK) rc = talloc(ctx, &t->bt16, (size_t)3 * 2 * D * t->Mp_max);
        t->a16r[0] = t->a16;
        bool want = false;
        { const char* e = getenv("VN_TRAIN_OVERLAP"); want = t->tn && rc == VN_OK && !(e && e[0] == '0'); }
        if (want) {
            for (int i = 1; i < vn_train::NB && rc == VN_OK; ++i) {
                rc = talloc(ctx, &t->a16r[i], (size_t)3 * rows16 * (size_t)wide);
                if (rc == VN_OK && hipMemset(t->a16r[i], 0, (size_t)3 * rows16 * (size_t)wide * 2) != hipSuccess) rc = VN_ERR_HIP;
            }
            // the side stream's context: its own split-K workspace and zero page, allocated here (nothing allocates inside a step)
            if (rc == VN_OK) rc = vn_ctx_create(ctx->device, &t->ctx2);
            if (rc == VN_OK) {
                t->ctx2->tune = ctx->tune;
                if (vn_dev_malloc((void**)&t->ctx2->x3_ws, (size_t)(32L << 20) * sizeof(float)) != hipSuccess) rc = VN_ERR_OOM;
            }
            if (rc == VN_OK) {
                int lo = 0, hi = 0;
                (void)hipDeviceGetStreamPriorityRange(&lo, &hi);                   // lo = the LEAST urgent
                // (VN_TRAIN_SIDE_PRIO=high|normal for A/B runs: the chain of the caller's stream is the critical path, so the default is low)
                const char* pe = getenv("VN_TRAIN_SIDE_PRIO");
                const int prio = pe && pe[0] == 'h' ? hi : pe && pe[0] == 'n' ? 0 : lo;
                bool ok = hipStreamCreateWithPriority(&t->side, hipStreamNonBlocking, prio) == hipSuccess;
                for (int i = 0; i < vn_train::NB && ok; ++i)
                    ok = hipEventCreateWithFlags(&t->ev_ready[i], hipEventDisableTiming) == hipSuccess &&
                         hipEventCreateWithFlags(&t->ev_done[i], hipEventDisableTiming) == hipSuccess;
                ok = ok && hipEventCreateWithFlags(&t->ev_join, hipEventDisableTiming) == hipSuccess &&
                     hipEventCreateWithFlags(&t->ev_upd, hipEventDisableTiming) == hipSuccess;
                for (int l = 0; l <= (int)L && ok; ++l) {
                    hipEvent_t e;
                    ok = hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
                    if (ok) t->ev_w.push_back(e);
                }
                if (!ok) rc = VN_ERR_HIP;
            }
            t->overlap = rc == VN_OK;
        }
        if (t->tn) {
            const size_t nd = (size_t)3 * rows16 * D;
            auto Z = [&](uint16_t** p, size_t cnt) {
                if (rc == VN_OK) rc = talloc(ctx, p, cnt);
                if (rc == VN_OK && hipMemset(*p, 0, cnt * 2) != hipSuccess) rc = VN_ERR_HIP;
            };
            for (auto& st : t->st) { Z(&st.y1_16, nd); Z(&st.a_16, nd); Z(&st.y3_16, nd); Z(&st.g_16, 2 * nd); }
            Z(&t->yf16, nd);
        }
    }
    if (t->ax3) {
        vn_ax_bwd_ws w;
        vn_attention_x3_bwd_ws_layout(d.max_batch, m->H, d.max_T, &w);
        if (rc == VN_OK) rc = talloc(ctx, &t->ax_ws, (size_t)w.total);
        if (rc == VN_OK && hipMemset(t->ax_ws, 0, (size_t)w.total * 2) != hipSuccess) rc = VN_ERR_HIP;
    }
    if (rc == VN_OK) rc = talloc(ctx, &t->npartial, 1024);
    if (rc == VN_OK) rc = talloc(ctx, &t->t32, (size_t)rows * m->Cp);
    if (rc == VN_OK) rc = talloc(ctx, &t->n_valid, 4);
    // (the zero fills above are asynchronous on the null stream; the trainer's own streams do not wait for that stream)
    if (rc == VN_OK && hipDeviceSynchronize() != hipSuccess) rc = VN_ERR_HIP;
    if (rc != VN_OK) { vn_train_destroy(t); return rc; }
    *out = t;
    return VN_OK;
}

// tuning / measurement hook: run the layers' weight-gradient GEMMs on the side stream (1) or in the caller's stream (0); -1 = as created
// (VN_TRAIN_OVERLAP).  Only between steps; returns the state in effect.
extern "C" int vn_debug_train_overlap(vn_train* t, int on) {
    if (!t) return VN_ERR_INVALID;
    const bool built = t->side != nullptr;
    if (on < 0) { const char* e = getenv("VN_TRAIN_OVERLAP"); t->overlap = built && !(e && e[0] == '0'); }
    else t->overlap = built && on != 0;
    return t->overlap ? 1 : 0;
}

// re-derive everything that is a function of the parameters: folded classifier weight, W^T copies, bias table
extern "C" int vn_train_sync(vn_train* t, void* stream) {
    if (!t) return VN_ERR_INVALID;
    vn_model* m = t->m;
    vn_ctx* ctx = m->ctx;
    hipStream_t s = (hipStream_t)stream;
    const int D = m->D;
    int rc;
    float* clsW = t->params + vn_tensor_offset(&m->d, VN_W_CLS_W, 0);
    if ((rc = vn_launch_weight_norm_fold(ctx, t->params + t->off_g, t->params + t->off_v, clsW, t->NV, D, s))) return rc;
    // every dX GEMM on the split-plane pipe (D and NV multiples of 64): the fp32 W^T copies are never read — their planes come straight
    // from W through the transposing splitter; otherwise (odd widths, VN_TRAIN_X3=0) the fp32 copies feed the fp32 kernel
    const bool all_x3 = t->x3 && !(D & 63) && !(t->NV & 63);
    if (all_x3) {
        // planes of w [rows][K] (dst) and of w^T [K][rows] (dstT) in ONE pass over w (the transposing splitter writes both from its tile)
        // vn_train::overlap: the splitters run on the side stream behind everything the caller's stream has done so far (the update, the
        // classifier fold above); the next forward waits layer by layer (ev_w)
        const bool aside = t->overlap && t->side;
        vn_ctx* const cx = aside ? t->ctx2 : ctx;
        if (aside) {
            VN_HIP_CHECK(ctx, hipEventRecord(t->ev_upd, s));
            VN_HIP_CHECK(ctx, hipStreamWaitEvent(t->side, t->ev_upd, 0));
            s = t->side;
        }
        auto both = [&](const float* w, uint16_t* dst, uint16_t* dstT, int rows, int K) {
            return vn_launch_transpose_split3_tiled(cx, w, dstT, rows, K, K, rows, s, dst);
        };
        for (int l = 0; l < m->L; ++l) {
            uint16_t* b16 = t->wT16 + 3 * (t->wT_layer * l);
            const float *wq = P(t, VN_W_QKV, l), *wo = P(t, VN_W_WO, l), *w1 = P(t, VN_W_W1, l), *w2 = P(t, VN_W_W2, l);
            if ((rc = both(wq, t->w16 + 3 * vn_tensor_offset(&m->d, VN_W_QKV, l), b16, 3 * D, D)) ||
                (rc = both(wo, t->w16 + 3 * vn_tensor_offset(&m->d, VN_W_WO, l), b16 + 3 * (3L * D * D), D, D)) ||
                (rc = both(w1, t->w16 + 3 * vn_tensor_offset(&m->d, VN_W_W1, l), b16 + 3 * (4L * D * D), 4 * D, D)) ||
                (rc = both(w2, t->w16 + 3 * vn_tensor_offset(&m->d, VN_W_W2, l), b16 + 3 * (8L * D * D), D, 2 * D)))
                return aside ? vn_fail(ctx, rc, "weight planes on the side stream: %s", cx->err) : rc;
            if (aside) VN_HIP_CHECK(ctx, hipEventRecord(t->ev_w[l], s));
        }
        if ((rc = both(clsW, t->w16 + 3 * vn_tensor_offset(&m->d, VN_W_CLS_W, 0), t->wT16 + 3 * t->wT_cls, t->NV, D)))
            return aside ? vn_fail(ctx, rc, "weight planes on the side stream: %s", cx->err) : rc;
        if (aside) { VN_HIP_CHECK(ctx, hipEventRecord(t->ev_w[m->L], s)); t->w_pending = true; }
        m->bias_T = -1;
        return VN_OK;
    }
    for (int l = 0; l < m->L; ++l) {
        float* base = t->wT + t->wT_layer * l;
        if ((rc = vn_launch_transpose(ctx, P(t, VN_W_QKV, l), base, 3 * D, D, D, 3 * D, s))) return rc;
        if ((rc = vn_launch_transpose(ctx, P(t, VN_W_WO, l), base + 3L * D * D, D, D, D, D, s))) return rc;
        if ((rc = vn_launch_transpose(ctx, P(t, VN_W_W1, l), base + 4L * D * D, 4 * D, D, D, 4 * D, s))) return rc;
        if ((rc = vn_launch_transpose(ctx, P(t, VN_W_W2, l), base + 8L * D * D, D, 2 * D, 2 * D, D, s))) return rc;
    }
    if ((rc = vn_launch_transpose(ctx, clsW, t->wT + t->wT_cls, t->NV, D, D, t->NV, s))) return rc;
    if (t->x3) {
        // the split planes of every GEMM weight and of its transpose, in the tiled layout (gemm_x3.hip): 10 bytes of traffic per
        // weight and copy, ~2 ms per update of the coarse model — what lets a forward / dX GEMM of the step split only its activation
        auto planes = [&](const float* w, uint16_t* dst, int rows, int K) { return vn_launch_split3_tiled(ctx, w, dst, rows, K, K, s); };
        for (int l = 0; l < m->L; ++l) {
            const float* base = t->wT + t->wT_layer * l;
            uint16_t* b16 = t->wT16 + 3 * (t->wT_layer * l);
            if ((rc = planes(P(t, VN_W_QKV, l), t->w16 + 3 * vn_tensor_offset(&m->d, VN_W_QKV, l), 3 * D, D)) ||
                (rc = planes(P(t, VN_W_WO, l), t->w16 + 3 * vn_tensor_offset(&m->d, VN_W_WO, l), D, D)) ||
                (rc = planes(P(t, VN_W_W1, l), t->w16 + 3 * vn_tensor_offset(&m->d, VN_W_W1, l), 4 * D, D)) ||
                (rc = planes(P(t, VN_W_W2, l), t->w16 + 3 * vn_tensor_offset(&m->d, VN_W_W2, l), D, 2 * D)) ||
                (rc = planes(base, b16, D, 3 * D)) || (rc = planes(base + 3L * D * D, b16 + 3 * (3L * D * D), D, D)) ||
                (rc = planes(base + 4L * D * D, b16 + 3 * (4L * D * D), D, 4 * D)) ||
                (rc = planes(base + 8L * D * D, b16 + 3 * (8L * D * D), 2 * D, D)))
                return rc;
        }
        if ((rc = planes(clsW, t->w16 + 3 * vn_tensor_offset(&m->d, VN_W_CLS_W, 0), t->NV, D)) ||
            (rc = planes(t->wT + t->wT_cls, t->wT16 + 3 * t->wT_cls, D, t->NV)))
            return rc;
    }
    m->bias_T = -1;                 // the relative-position table is a parameter too
    return VN_OK;
}

static vn_drop make_drop(const vn_train_params* p, int layer, int site, long row0) {
    vn_drop d;
    d.key = vn_drop_key(p->seed, (uint32_t)p->step, (uint32_t)(layer * 4 + site));
    const double th = (double)p->dropout * 65536.0;
    d.thresh16 = p->dropout > 0.f ? (uint32_t)(th + 0.5) : 0u;
    d.scale = 1.0f / (1.0f - p->dropout);
    d.row0 = row0;
    return d;
}

// The GEMMs of the training step — forward, dX and dW — on the split-plane pipe (gemm_x3.hip, bf16x3: six bf16-MFMA products of
// exact three-way operand splits, fp32 accumulation: fp32-grade, and bf16 keeps fp32's exponent range, which the tiny dlogits / dY
// magnitudes of the backward pass need; an fp16-based split would not).  Default since round 5 (VN_TRAIN_X3=0: the fp32-input MFMA
// kernel).  Round 2's opt-in form split BOTH fp32 operands of every GEMM on the fly into planar planes (two extra HBM passes per GEMM)
// and came out slower than the fp32 kernel; now
//   * the weights and their transposes are split ONCE per update, into the tiled layout (vn_train_sync),
//   * a forward / dX GEMM splits only its activation operand (one pass, tiled: vn_launch_split3_tiled),
//   * a dW GEMM gets both operands from the transposes it needed anyway, which now write tiled planes instead of fp32
//     (vn_launch_transpose_split3_tiled: 6 instead of 4 bytes written per element, no separate split pass).
// Shapes the kernel does not take (N % 64, K % 32) stay on the fp32 kernel.
static const uint16_t* weight_planes(const vn_train* t, const float* W) {
    if (W >= t->params && W < t->params + t->wsize) return t->w16 + 3 * (W - t->params);
    const long wT_total = t->wT_layer * t->m->L + (long)t->NV * t->m->D;
    if (W >= t->wT && W < t->wT + wT_total) return t->wT16 + 3 * (W - t->wT);
    return nullptr;
}
// a: an fp32 GEMM (A [M][K] row-major, W one of the step's weight tensors or transposes)
// can this GEMM shape run on the split-plane pipe ?  (its producer may then write the planes of A into t->a16 itself: a_ready)
static bool x3_shape(const vn_train* t, int N, int K) { return t->x3 && !(N & 63) && !(K & 31); }
// a16: where the planes of A live / go (default: the shared scratch t->a16; the forward pass of vn_train::tn names the layer's stash)
static int gemm_args(vn_train* t, vn_gemm_args a, int epi, hipStream_t s, bool a_ready = false, uint16_t* a16 = nullptr) {
    vn_ctx* ctx = t->m->ctx;
    const uint16_t* w16 = t->x3 ? weight_planes(t, a.W) : nullptr;
    if (!w16 || !x3_shape(t, a.N, a.K)) return vn_launch_gemm_f32(ctx, a, epi, s);
    if (!a16) a16 = t->a16;
    int rc;
    if (!a_ready && (rc = vn_launch_split3_tiled(ctx, a.A, a16, a.M, a.K, a.K, s))) return rc;
    a.A = (const float*)a16; a.W = (const float*)w16; a.bf16 = 2; a.a_plane = VN_PLANES_TILED; a.w_plane = VN_PLANES_TILED; a.w_tiled = 1;
    return vn_launch_gemm_x3(ctx, a, epi, s);
}

static int gemm(vn_train* t, const float* A, const float* W, const float* bias, float* C, int M, int N, int K, int epi,
                hipStream_t s, bool a_ready = false, uint16_t* a16 = nullptr) {
    vn_gemm_args a{};
    a.A = A; a.W = W; a.bias = bias; a.C = C; a.M = M; a.N = N; a.K = K; a.ldc = N;
    return gemm_args(t, a, epi, s, a_ready, a16);
}

static int params_ok(vn_ctx* ctx, const vn_train_params* p) {
    if (!p) return vn_fail(ctx, VN_ERR_INVALID, "train params is NULL%s", "");
    if (!(p->dropout >= 0.f && p->dropout < 1.f)) return vn_fail(ctx, VN_ERR_INVALID, "dropout must be in [0, 1)%s", "");
    if (!(p->label_smoothing >= 0.f && p->label_smoothing < 1.f)) return vn_fail(ctx, VN_ERR_INVALID, "label_smoothing must be in [0, 1)%s", "");
    if (p->step < 1) return vn_fail(ctx, VN_ERR_INVALID, "optimiser step index must be >= 1%s", "");
    if (p->world_size < 1) return vn_fail(ctx, VN_ERR_INVALID, "world_size must be >= 1%s", "");
    return VN_OK;
}


// ---- LoRA ------------------------------------------------------------------------------------------
static float* blob_weight(vn_train* t, int which, int l, long* row0_floats) {
    const vn_dims* d = &t->m->d;
    const long D = d->d_model;
    *row0_floats = 0;
    switch (which) {
        case LORA_Q: return t->params + vn_tensor_offset(d, VN_W_QKV, l);
        case LORA_V: *row0_floats = 2 * D * D; return t->params + vn_tensor_offset(d, VN_W_QKV, l);
        case LORA_FC: return t->params + vn_tensor_offset(d, VN_W_WO, l);
        case LORA_W1: return t->params + vn_tensor_offset(d, VN_W_W1, l);
        default: return t->params + vn_tensor_offset(d, VN_W_W2, l);
    }
}
static const float* base_weight(const vn_train* t, int which, int l) {
    const long D = t->m->D;
    const float* b = t->w_base + 10L * D * D * l;
    switch (which) {
        case LORA_Q: return b;
        case LORA_V: return b + 2 * D * D;
        case LORA_FC: return b + 3 * D * D;
        case LORA_W1: return b + 4 * D * D;
        default: return b + 8 * D * D;
    }
}

// blob <- W_base + s * B * A for every LoRA'd linear
static int lora_merge_all(vn_train* t, hipStream_t s) {
    vn_ctx* ctx = t->m->ctx;
    const vn_dims* d = &t->m->d;
    for (int l = 0; l < t->m->L; ++l)
        for (int w = 0; w < 5; ++w) {
            long K, N, r0;
            lora_shape(d, w, &K, &N);
            float* dst = blob_weight(t, w, l, &r0);
            int rc = vn_launch_lora_merge(ctx, base_weight(t, w, l), t->lora_params + lora_offset(d, l, w, 1),
                                          t->lora_params + lora_offset(d, l, w, 0), dst + r0, (int)N, (int)K, t->lora_scale, s);
            if (rc) return rc;
        }
    return VN_OK;
}

extern "C" int vn_train_enable_lora(vn_train* t, float* lora_params, float scaling, void* stream) {
    if (!t || !lora_params) return VN_ERR_INVALID;
    vn_model* m = t->m;
    vn_ctx* ctx = m->ctx;
    hipStream_t s = (hipStream_t)stream;
    if (t->lora) return vn_fail(ctx, VN_ERR_INVALID, "vn_train_enable_lora: already enabled%s", "");
    const long D = m->D;
    int rc;
    if ((rc = talloc(ctx, &t->w_base, (size_t)10 * D * D * m->L))) return rc;
    if ((rc = talloc(ctx, &t->h8, (size_t)m->max_rows * LORA_R))) return rc;
    if ((rc = talloc(ctx, &t->dh8, (size_t)m->max_rows * LORA_R))) return rc;
    for (int l = 0; l < m->L; ++l) {        // snapshot the frozen weights (the blob must hold the UN-merged W here)
        float* b = t->w_base + 10L * D * D * l;
        VN_HIP_CHECK(ctx, hipMemcpyAsync(b, P(t, VN_W_QKV, l), 3 * D * D * sizeof(float), hipMemcpyDeviceToDevice, s));
        VN_HIP_CHECK(ctx, hipMemcpyAsync(b + 3 * D * D, P(t, VN_W_WO, l), D * D * sizeof(float), hipMemcpyDeviceToDevice, s));
        VN_HIP_CHECK(ctx, hipMemcpyAsync(b + 4 * D * D, P(t, VN_W_W1, l), 4 * D * D * sizeof(float), hipMemcpyDeviceToDevice, s));
        VN_HIP_CHECK(ctx, hipMemcpyAsync(b + 8 * D * D, P(t, VN_W_W2, l), 2 * D * D * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    t->lora = true; t->lora_params = lora_params; t->lora_scale = scaling;
    int64_t n = 0;
    vn_lora_param_size(&m->d, &n);
    t->n_lora = n;
    if ((rc = lora_merge_all(t, s))) return rc;
    return vn_train_sync(t, stream);
}

// re-derive W_eff and its copies after the adapters were changed from outside (checkpoint load)
extern "C" int vn_train_lora_merge(vn_train* t, void* stream) {
    if (!t) return VN_ERR_INVALID;
    if (!t->lora) return vn_fail(t->m->ctx, VN_ERR_INVALID, "vn_train_lora_merge: LoRA mode is not enabled%s", "");
    int rc = lora_merge_all(t, (hipStream_t)stream);
    return rc ? rc : vn_train_sync(t, stream);
}

// inference-side adapter swap (include/vampnet_hip.h): blob <- base + s B A on the five LoRA'd linears of every layer
extern "C" int vn_model_apply_lora(vn_model* m, const float* base_blob, const float* lora, float scaling, void* stream) {
    if (!m || !base_blob || !lora) return VN_ERR_INVALID;
    vn_ctx* ctx = m->ctx;
    if (base_blob == m->blob) return vn_fail(ctx, VN_ERR_INVALID, "vn_model_apply_lora: the base blob may not alias the model's blob%s", "");
    if (m->D % 4) return vn_fail(ctx, VN_ERR_UNSUPPORTED, "vn_model_apply_lora: d_model %% 4%s", "");
    const vn_dims* d = &m->d;
    const long D = m->D;
    float* blob = const_cast<float*>(m->blob);           // caller-owned and writable: the model's weights are swapped in place
    for (int l = 0; l < m->L; ++l)
        for (int w = 0; w < 5; ++w) {
            long K, N;
            lora_shape(d, w, &K, &N);
            const int id = w <= LORA_V ? VN_W_QKV : w == LORA_FC ? VN_W_WO : w == LORA_W1 ? VN_W_W1 : VN_W_W2;
            const long off = vn_tensor_offset(d, id, l) + (w == LORA_V ? 2 * D * D : 0);
            int rc = vn_launch_lora_merge(ctx, base_blob + off, lora + lora_offset(d, l, w, 1), lora + lora_offset(d, l, w, 0), blob + off,
                                          (int)N, (int)K, scaling, (hipStream_t)stream);
            if (rc) return rc;
        }
    return VN_OK;
}

// gradients of one LoRA'd linear y = x W^T + s (x At) B^T given X [M][K] (row stride ldx) and dY [M][N] (row stride ldy)
static int lora_grads(vn_train* t, const float* X, int ldx, const float* dY, int ldy, int l, int which, float* grads, int M,
                      hipStream_t s) {
    vn_ctx* ctx = t->m->ctx;
    const vn_dims* d = &t->m->d;
    long K, N;
    lora_shape(d, which, &K, &N);
    const float* At = t->lora_params + lora_offset(d, l, which, 0);
    const float* Bm = t->lora_params + lora_offset(d, l, which, 1);
    float* gAt = grads + lora_offset(d, l, which, 0);
    float* gB = grads + lora_offset(d, l, which, 1);
    int rc;
    if ((rc = vn_launch_lora_down(ctx, X, ldx, At, t->h8, M, (int)K, 1.0f, s))) return rc;                  // h = x At
    if ((rc = vn_launch_lora_up(ctx, dY, ldy, t->h8, gB, t->partial, M, (int)N, t->lora_scale, s))) return rc;  // dB = s dY^T h
    if ((rc = vn_launch_lora_down(ctx, dY, ldy, Bm, t->dh8, M, (int)N, t->lora_scale, s))) return rc;         // dh = s dY B
    rc = vn_launch_lora_up(ctx, X, ldx, t->dh8, gAt, t->partial, M, (int)K, 1.0f, s);                      // dAt = x^T dh
    return rc;
}

// ---- forward in train() mode + loss + dlogits --------------------------------------------------
static int forward_train(vn_train* t, int B, int T, const vn_train_params* p, hipStream_t s) {
    vn_model* m = t->m;
    vn_ctx* ctx = m->ctx;
    const int D = m->D, H = m->H, M = B * T, L = m->L;
    const long plane = (long)B * H * T * VN_DHEAD;
    const long r_tok = (long)p->batch_offset * T, r_att = (long)p->batch_offset * H * T;
    int rc;
    if ((rc = vn_model_ensure_bias(m, T, s))) return rc;
    if (t->ax3 && t->near_T != T) {            // which key - query offsets the backward's per-wave bias-gradient tables must cover
        std::vector<int32_t> lut(2 * T - 1);
        vn_bucket_lut_host(T, m->d.num_buckets, m->d.max_distance, lut.data());
        t->near_r = vn_attention_x3_near_r(lut.data(), T);
        t->near_T = T;
    }
    if (t->tn && t->tn_rows != M) {
        // the producers below write rows < M only; the dW GEMMs contract over whole 16-token blocks, so what an earlier forward with
        // another token count left in the last block's pad rows must go (shape changes only)
        const size_t nd = (size_t)3 * (((size_t)m->max_rows + 15) & ~(size_t)15) * D * 2;
        for (auto& st : t->st) {
            VN_HIP_CHECK(ctx, hipMemsetAsync(st.y1_16, 0, nd, s)); VN_HIP_CHECK(ctx, hipMemsetAsync(st.a_16, 0, nd, s));
            VN_HIP_CHECK(ctx, hipMemsetAsync(st.y3_16, 0, nd, s)); VN_HIP_CHECK(ctx, hipMemsetAsync(st.g_16, 0, 2 * nd, s));
        }
        VN_HIP_CHECK(ctx, hipMemsetAsync(t->yf16, 0, nd, s));
        t->tn_rows = M;
    }
    if ((rc = vn_launch_embed(ctx, m->z, P(t, VN_W_EMB_TABLES), P(t, VN_W_EMB_WT), P(t, VN_W_EMB_B), t->st[0].x_in, B,
                              m->d.n_codebooks, T, m->d.vocab + 1, m->d.latent_dim, D, s)))
        return rc;
    for (int l = 0; l < L; ++l) {
        vn_layer_stash& S = t->st[l];
        if (t->w_pending) VN_HIP_CHECK(ctx, hipStreamWaitEvent(s, t->ev_w[l], 0));       // this layer's weight planes (vn_train_sync on the side stream)
        float* x_out = l + 1 < L ? t->st[l + 1].x_in : t->x_last;
        // where the planes of the four A operands go: the layer's stash (kept for the dW GEMMs) or the shared scratch
        uint16_t* const y1_16 = t->tn ? S.y1_16 : t->a16, * const a_16 = t->tn ? S.a_16 : t->a16, * const y3_16 = t->tn ? S.y3_16 : t->a16,
                * const g_16 = t->tn ? S.g_16 : t->a16;
        // producers write the tiled planes of the next GEMM's A operand from the registers that hold the values (t->a16) where the kernel
        // can (RMSNorm: D in {256, 1280}; GEGLU) — no split pass for those operands
        const bool n16 = (D == 256 || D == 1280) && x3_shape(t, 3 * D, D);
        if ((rc = vn_launch_rmsnorm(ctx, S.x_in, P(t, VN_W_NORM1, l), S.y1, M, D, m->d.eps, s, n16 ? y1_16 : nullptr, VN_PLANES_TILED, true))) return rc;
        vn_gemm_args a{};
        a.A = S.y1; a.W = P(t, VN_W_QKV, l); a.C = S.qkv; a.M = M; a.N = 3 * D; a.K = D; a.ldc = 3 * D;
        a.T = T; a.H = H; a.qkv_plane = plane;
        bool a16o = false;
        if (t->ax3) {
            a.C = nullptr; a.C16 = S.qk16; a.c_plane = t->qk_plane; a.V16 = S.vt16; a.v_plane = t->vt_plane;
            if ((rc = gemm_args(t, a, VN_EPI_QKV3, s, n16, y1_16))) return rc;
            // (the kernel can also write the planes of its output for the Wo GEMM — a16o = x3_shape(t, D, D), t->a16 below — but its 8-byte
            // scattered plane stores cost 13 us per layer against the 10.6 us of the split pass they replace: left off)
            rc = vn_launch_attention_x3_train_fwd(ctx, S.qk16, S.qk16 + plane, t->qk_plane, S.vt16, t->vt_plane, m->bias_full, S.a, S.lse, B, H,
                                                  T, vn_num_cus(ctx), make_drop(p, l, SITE_ATTN, r_att), s, a16o ? a_16 : nullptr, VN_PLANES_TILED);
        } else {
            if ((rc = gemm_args(t, a, VN_EPI_QKV, s, n16, y1_16))) return rc;
            rc = vn_launch_attention_train_fwd(ctx, S.qkv, S.qkv + plane, S.qkv + 2 * plane, m->bias_full, S.a, S.lse, B, H, T,
                                               make_drop(p, l, SITE_ATTN, r_att), s);
        }
        if (rc) return rc;
        if ((rc = gemm(t, S.a, P(t, VN_W_WO, l), nullptr, t->tmp, M, D, D, VN_EPI_STORE, s, a16o, a_16))) return rc;
        if ((rc = vn_launch_resid_dropout(ctx, S.x_in, t->tmp, S.x_mid, M, D, make_drop(p, l, SITE_RES1, r_tok), s))) return rc;
        if ((rc = vn_launch_rmsnorm(ctx, S.x_mid, P(t, VN_W_NORM3, l), S.y3, M, D, m->d.eps, s, n16 ? y3_16 : nullptr, VN_PLANES_TILED, true))) return rc;
        if ((rc = gemm(t, S.y3, P(t, VN_W_W1, l), nullptr, S.u, M, 4 * D, D, VN_EPI_STORE, s, n16, y3_16))) return rc;
        const bool g16 = x3_shape(t, D, 2 * D);
        if ((rc = vn_launch_geglu_train(ctx, S.u, nullptr, S.g, M, 2 * D, make_drop(p, l, SITE_FFN, r_tok), false, s, g16 ? g_16 : nullptr))) return rc;
        if ((rc = gemm(t, S.g, P(t, VN_W_W2, l), nullptr, t->tmp, M, D, 2 * D, VN_EPI_STORE, s, g16, g_16))) return rc;
        if ((rc = vn_launch_resid_dropout(ctx, S.x_mid, t->tmp, x_out, M, D, make_drop(p, l, SITE_RES2, r_tok), s))) return rc;
    }
    const bool f16 = (D == 256 || D == 1280) && x3_shape(t, t->NV, D);
    if (t->w_pending) { VN_HIP_CHECK(ctx, hipStreamWaitEvent(s, t->ev_w[L], 0)); t->w_pending = false; }
    uint16_t* const yf16 = t->tn ? t->yf16 : t->a16;
    if ((rc = vn_launch_rmsnorm(ctx, t->x_last, P(t, VN_W_FINAL_NORM), t->y_f, M, D, m->d.eps, s, f16 ? yf16 : nullptr, VN_PLANES_TILED, true))) return rc;
    return gemm(t, t->y_f, P(t, VN_W_CLS_W), P(t, VN_W_CLS_B), m->logits, M, t->NV, D, VN_EPI_BIAS, s, f16, yf16);
}

// dW[N][K] = dY^T X   (dY [M][N], X [M][K]) through the two transposes
// dy_planes (out, optional): the transposer of dY has ALSO left dY's own tiled planes in t->a16 — the A operand of the dX GEMM the caller
// runs next on the same dY (one read of dY instead of two; nothing else may write t->a16 in between)
// X16 (vn_train::tn): the tiled planes of X the forward GEMM read (the layer's stash); dy_ready: dY's tiled planes are in t->a16 already.
// Then nothing is transposed: both operands go to the kernel token-major (gemm_x3.hip X3_MODE_TN), dY's planes — written here by one
// split pass unless they exist — serve the dX GEMM that follows as before.
// the plane buffer the NEXT dY operand goes into (vn_train::overlap: one of NB rotating buffers, waited free on the caller's stream if a
// side GEMM of this backward call still reads it; otherwise always t->a16)
static int plane_buf(vn_train* t, hipStream_t s, uint16_t** buf, int* slot) {
    if (!t->overlap) { *buf = t->a16; *slot = 0; return VN_OK; }
    const int i = t->buf_next;
    t->buf_next = (i + 1) % vn_train::NB;
    if (t->buf_busy[i]) {
        VN_HIP_CHECK(t->m->ctx, hipStreamWaitEvent(s, t->ev_done[i], 0));
        t->buf_busy[i] = false;
    }
    *buf = t->a16r[i]; *slot = i;
    return VN_OK;
}
// end of a backward call: everything the side stream was given is complete when the caller's stream gets here
static int side_join(vn_train* t, hipStream_t s) {
    if (!t->overlap || !t->side_used) return VN_OK;
    vn_ctx* ctx = t->m->ctx;
    VN_HIP_CHECK(ctx, hipEventRecord(t->ev_join, t->side));
    VN_HIP_CHECK(ctx, hipStreamWaitEvent(s, t->ev_join, 0));
    for (int i = 0; i < vn_train::NB; ++i) t->buf_busy[i] = false;
    t->side_used = false;
    return VN_OK;
}

// dy16 / slot: the plane buffer of plane_buf() this dY uses (holds dY's planes already if dy_ready); side: this GEMM may run on the side
// stream (its result is read by nothing before the end of the backward call)
static int grad_weight(vn_train* t, const float* dY, const float* X, float* dW, int M, int N, int K, hipStream_t s, bool* dy_planes = nullptr,
                       const uint16_t* X16 = nullptr, bool dy_ready = false, uint16_t* dy16 = nullptr, int slot = 0, bool side = false) {
    vn_ctx* ctx = t->m->ctx;
    const int Mp = (M + 31) & ~31;
    int rc;
    if (dy_planes) *dy_planes = false;
    if (!dy16) dy16 = t->a16;
    if (t->tn && X16 && x3_shape(t, N, K) && !(K & 63)) {      // x3_shape(N, K): the forward GEMM of this weight ran on the pipe and filled X16
        if (!dy_ready && (rc = vn_launch_split3_tiled(ctx, dY, dy16, M, N, N, s))) return rc;
        if (dy_planes) *dy_planes = true;
        vn_gemm_args a{};
        a.A = (const float*)dy16; a.W = (const float*)X16; a.C = dW; a.M = N; a.N = K; a.K = Mp; a.ldc = K;
        a.bf16 = 2; a.a_plane = VN_PLANES_TILED; a.w_plane = VN_PLANES_TILED; a.w_tiled = 1;
        a.tn_blocks = (M + 15) >> 4;
        if (side && t->overlap && dy16 == t->a16r[slot]) {
            VN_HIP_CHECK(ctx, hipEventRecord(t->ev_ready[slot], s));
            VN_HIP_CHECK(ctx, hipStreamWaitEvent(t->side, t->ev_ready[slot], 0));
            rc = vn_launch_gemm_x3(t->ctx2, a, VN_EPI_STORE, t->side);
            if (rc) return vn_fail(ctx, rc, "weight-gradient GEMM on the side stream: %s", t->ctx2->err);
            VN_HIP_CHECK(ctx, hipEventRecord(t->ev_done[slot], t->side));
            t->buf_busy[slot] = true;
            t->side_used = true;
            return VN_OK;
        }
        return vn_launch_gemm_x3(ctx, a, VN_EPI_STORE, s);
    }
    if (t->x3 && !(N & 15) && !(K & 63)) {
        // both operands as tiled planes straight out of the transposes: A = dY^T [N][Mp], W = X^T [K][Mp], contraction over the tokens
        const bool both = dy_planes && !(N & 31);
        if ((rc = vn_launch_transpose_split3_tiled(ctx, dY, t->at16, M, N, N, Mp, s, both ? dy16 : nullptr))) return rc;
        if (both) *dy_planes = true;
        if ((rc = vn_launch_transpose_split3_tiled(ctx, X, t->bt16, M, K, K, Mp, s))) return rc;
        vn_gemm_args a{};
        a.A = (const float*)t->at16; a.W = (const float*)t->bt16; a.C = dW; a.M = N; a.N = K; a.K = Mp; a.ldc = K;
        a.bf16 = 2; a.a_plane = VN_PLANES_TILED; a.w_plane = VN_PLANES_TILED; a.w_tiled = 1;
        return vn_launch_gemm_x3(ctx, a, VN_EPI_STORE, s);
    }
    if ((rc = vn_launch_transpose(ctx, dY, t->At, M, N, N, Mp, s))) return rc;
    if ((rc = vn_launch_transpose(ctx, X, t->Bt, M, K, K, Mp, s))) return rc;
    vn_gemm_args a{};
    a.A = t->At; a.W = t->Bt; a.C = dW; a.M = N; a.N = K; a.K = Mp; a.ldc = K;
    return vn_launch_gemm_f32(ctx, a, VN_EPI_STORE, s);
}

// Backward over the stages hi >= ... >= lo of the stashed forward: stage L = classifier + final norm, stages L-1 .. 0 =
// transformer layers, stage -1 = codebook embedding.  Between stages the running gradient lives in t->dxa, so a caller
// may interleave its own work (e.g. the all-reduce of the gradient slices that are already final) between calls.
static int backward_body(vn_train* t, const vn_train_params* p, float* grads, int hi, int lo, hipStream_t s);
static int backward_range(vn_train* t, const vn_train_params* p, float* grads, int hi, int lo, hipStream_t s) {
    const int rc = backward_body(t, p, grads, hi, lo, s);
    const int rj = side_join(t, s);             // (also after an error: nothing of this call is left running on the side stream)
    return rc ? rc : rj;
}
static int backward_body(vn_train* t, const vn_train_params* p, float* grads, int hi, int lo, hipStream_t s) {
    vn_model* m = t->m;
    vn_ctx* ctx = m->ctx;
    const int B = t->B, T = t->T;
    const int D = m->D, H = m->H, M = B * T, L = m->L, NV = t->NV;
    const long plane = (long)B * H * T * VN_DHEAD;
    const long r_tok = (long)p->batch_offset * T, r_att = (long)p->batch_offset * H * T;
    int rc = VN_OK;
    float* dlog = m->logits;
    // LoRA mode (mark_only_lora_as_trainable): only dX flows through the frozen tensors; their own gradients are not
    // computed (norm-weight / bias-table by-products land in scratch).
    const bool lora = t->lora;
    float* junk = t->tmp;                       // >= D floats: by-product norm-weight gradients in LoRA mode
    float* dx = t->dxa;
    float* dx2 = t->dxb;
    if (hi >= L) {
    // ---- classifier (WNConv1d 1x1, transformer.py:596-604) + final norm
    bool dl16 = false;
    if (!lora && (rc = grad_weight(t, dlog, t->y_f, G(t, grads, VN_W_CLS_W), M, NV, D, s, &dl16, t->yf16))) return rc;
    if ((rc = gemm(t, dlog, t->wT + t->wT_cls, nullptr, t->dy, M, D, NV, VN_EPI_STORE, s, dl16 && x3_shape(t, D, NV)))) return rc;
    if (!lora) {
        float* dWc = G(t, grads, VN_W_CLS_W);
        if ((rc = vn_launch_colsum(ctx, dlog, M, NV, t->partial, G(t, grads, VN_W_CLS_B), s))) return rc;
        if ((rc = vn_launch_weight_norm_bwd(ctx, t->params + t->off_g, t->params + t->off_v, dWc, grads + t->off_g,
                                            grads + t->off_v, NV, D, s)))
            return rc;
        VN_HIP_CHECK(ctx, hipMemsetAsync(dWc, 0, (size_t)NV * D * sizeof(float), s));   // derived tensor: not a parameter
    }
    if ((rc = vn_launch_rmsnorm_bwd(ctx, t->x_last, P(t, VN_W_FINAL_NORM), t->dy, nullptr, dx,
                                    lora ? junk : G(t, grads, VN_W_FINAL_NORM), t->partial, M, D, m->d.eps, s)))
        return rc;
    }
    for (int l = (hi < L - 1 ? hi : L - 1); l >= (lo > 0 ? lo : 0); --l) {
        vn_layer_stash& S = t->st[l];
        const float* wTl = t->wT + t->wT_layer * l;
        // ---- feed-forward branch (transformer.py:72-85, :360-367)
        const vn_drop d2 = make_drop(p, l, SITE_RES2, r_tok);
        bool dy16 = false;                             // grad_weight left dY's planes in the plane buffer for the dX GEMM that follows it
        uint16_t* pb;                                  // the plane buffer of the current dY (plane_buf: t->a16, or one of the rotating ones)
        int ps;
        const float* dh = dx;
        // (vn_train::tn: the mask kernel writes dh's tiled planes as well — both GEMMs below read them, no split pass)
        const bool dh16 = t->tn && d2.thresh16 && x3_shape(t, 2 * D, D);
        if ((rc = plane_buf(t, s, &pb, &ps))) return rc;
        if (d2.thresh16) { if ((rc = vn_launch_dropout_bwd(ctx, dx, t->dh, M, D, d2, s, dh16 ? pb : nullptr))) return rc; dh = t->dh; }
        if (lora) { rc = lora_grads(t, S.g, 2 * D, dh, D, l, LORA_W2, grads, M, s); dy16 = dh16; }
        else rc = grad_weight(t, dh, S.g, G(t, grads, VN_W_W2, l), M, D, 2 * D, s, &dy16, S.g_16, dh16, pb, ps, true);
        if (rc) return rc;
        if ((rc = gemm(t, dh, wTl + 8L * D * D, nullptr, t->dg, M, 2 * D, D, VN_EPI_STORE, s, dy16 && x3_shape(t, 2 * D, D), pb))) return rc;
        const bool du16 = x3_shape(t, D, 4 * D);       // du's planes: the A operand of the dW GEMM (token-major form) and of the dX GEMM below; nothing in between writes the buffer
        if ((rc = plane_buf(t, s, &pb, &ps))) return rc;
        if ((rc = vn_launch_geglu_train(ctx, S.u, t->dg, t->du, M, 2 * D, make_drop(p, l, SITE_FFN, r_tok), true, s, du16 ? pb : nullptr))) return rc;
        bool du_p = false;
        if (lora) rc = lora_grads(t, S.y3, D, t->du, 4 * D, l, LORA_W1, grads, M, s);
        else rc = grad_weight(t, t->du, S.y3, G(t, grads, VN_W_W1, l), M, 4 * D, D, s, &du_p, S.y3_16, du16, pb, ps, true);
        if (rc) return rc;
        if ((rc = gemm(t, t->du, wTl + 4L * D * D, nullptr, t->dy, M, D, 4 * D, VN_EPI_STORE, s, du16 || (du_p && x3_shape(t, D, 4 * D)), pb))) return rc;
        if ((rc = vn_launch_rmsnorm_bwd(ctx, S.x_mid, P(t, VN_W_NORM3, l), t->dy, dx, dx2, lora ? junk : G(t, grads, VN_W_NORM3, l),
                                        t->partial, M, D, m->d.eps, s)))
            return rc;
        // ---- attention branch (transformer.py:211-257, :336-347)
        const vn_drop d1 = make_drop(p, l, SITE_RES1, r_tok);
        const float* dh2 = dx2;
        const bool dh216 = t->tn && d1.thresh16 && x3_shape(t, D, D);
        if ((rc = plane_buf(t, s, &pb, &ps))) return rc;
        if (d1.thresh16) { if ((rc = vn_launch_dropout_bwd(ctx, dx2, t->dh, M, D, d1, s, dh216 ? pb : nullptr))) return rc; dh2 = t->dh; }
        dy16 = false;
        if (lora) { rc = lora_grads(t, S.a, D, dh2, D, l, LORA_FC, grads, M, s); dy16 = dh216; }
        else rc = grad_weight(t, dh2, S.a, G(t, grads, VN_W_WO, l), M, D, D, s, &dy16, S.a_16, dh216, pb, ps, true);
        if (rc) return rc;
        if ((rc = gemm(t, dh2, wTl + 3L * D * D, nullptr, t->da, M, D, D, VN_EPI_STORE, s, dy16 && x3_shape(t, D, D), pb))) return rc;
        if (t->ax3)
            rc = vn_launch_attention_x3_bwd(ctx, S.qk16, t->qk_plane, S.vt16, t->vt_plane, t->ax_ws, m->bias_full, m->lut, t->near_r, S.a, t->da,
                                            S.lse, t->delta, t->dqkv, lora ? nullptr : t->dbias_partial + t->dbias_slab * l, B, H, T,
                                            m->d.num_buckets, make_drop(p, l, SITE_ATTN, r_att), s);
        else
            rc = vn_launch_attention_bwd(ctx, S.qkv, S.qkv + plane, S.qkv + 2 * plane, m->bias_full, m->lut, S.a, t->da, S.lse,
                                         t->delta, t->dqkv, lora ? nullptr : t->dbias_partial + t->dbias_slab * l, B, H, T,
                                         m->d.num_buckets, make_drop(p, l, SITE_ATTN, r_att), s);
        if (rc) return rc;
        dy16 = false;
        if (lora) {          // w_qs and w_vs carry adapters, w_ks is a plain nn.Linear (transformer.py:109-111)
            if ((rc = lora_grads(t, S.y1, D, t->dqkv, 3 * D, l, LORA_Q, grads, M, s))) return rc;
            rc = lora_grads(t, S.y1, D, t->dqkv + 2 * D, 3 * D, l, LORA_V, grads, M, s);
        } else {
            if ((rc = plane_buf(t, s, &pb, &ps))) return rc;
            rc = grad_weight(t, t->dqkv, S.y1, G(t, grads, VN_W_QKV, l), M, 3 * D, D, s, &dy16, S.y1_16, false, pb, ps, true);
        }
        if (rc) return rc;
        if ((rc = gemm(t, t->dqkv, wTl, nullptr, t->dy, M, D, 3 * D, VN_EPI_STORE, s, dy16 && x3_shape(t, D, 3 * D), lora ? nullptr : pb))) return rc;
        if ((rc = vn_launch_rmsnorm_bwd(ctx, S.x_in, P(t, VN_W_NORM1, l), t->dy, dx2, dx, lora ? junk : G(t, grads, VN_W_NORM1, l),
                                        t->partial, M, D, m->d.eps, s)))
            return rc;
    }
    if (!lora && lo <= 0 && hi >= 0) {          // layer 0 is done: every layer's slab of the shared bias-table gradient is final
        if ((rc = vn_launch_dbias_reduce(ctx, t->dbias_partial, G(t, grads, VN_W_REL_BIAS), L, t->dbias_slab, B, H,
                                         vn_cdiv(T, t->ax3 ? 128 : 64), m->d.num_buckets, false, s)))
            return rc;
    }
    if (lora || lo >= 0) return VN_OK;          // LoRA: embedding parameters are frozen
    // ---- codebook embedding (layers.py:134-163)
    return vn_launch_embed_bwd(ctx, dx, m->z, P(t, VN_W_EMB_TABLES), P(t, VN_W_EMB_WT), G(t, grads, VN_W_EMB_TABLES),
                               G(t, grads, VN_W_EMB_WT), G(t, grads, VN_W_EMB_B), t->partial, t->du, B, m->d.n_codebooks, T,
                               m->d.vocab + 1, m->d.latent_dim, D, s);
}

extern "C" int vn_train_forward_loss(vn_train* t, const int64_t* z_masked, const int64_t* target, int B, int T,
                                     const vn_train_params* p, float* grads, float* loss_dev, void* stream) {
    if (!t || !z_masked || !target || !grads || !loss_dev) return VN_ERR_INVALID;
    vn_model* m = t->m;
    vn_ctx* ctx = m->ctx;
    hipStream_t s = (hipStream_t)stream;
    int rc = params_ok(ctx, p);
    if (rc) return rc;
    if (B <= 0 || T <= 0 || B > m->d.max_batch || T > m->d.max_T)
        return vn_fail(ctx, VN_ERR_INVALID, "train step: B=%s%ld, T=%ld outside the model workspace", "", B, T);
    t->B = t->T = 0;
    VN_HIP_CHECK(ctx, hipMemsetAsync(grads, 0, (size_t)(t->lora ? t->n_lora : t->n_total) * sizeof(float), s));
    if ((rc = vn_launch_i64_to_i32(ctx, z_masked, m->z, (long)B * m->d.n_codebooks * T, s))) return rc;
    if ((rc = forward_train(t, B, T, p, s))) return rc;
    if ((rc = vn_launch_cross_entropy(ctx, m->logits, target, t->t32, (long)B * T * m->Cp, m->d.vocab, p->label_smoothing,
                                      t->n_valid, t->row_loss, loss_dev, s)))
        return rc;
    t->B = B; t->T = T;
    return VN_OK;
}

extern "C" int vn_train_backward(vn_train* t, const vn_train_params* p, float* grads, int stage_hi, int stage_lo, void* stream) {
    if (!t || !grads) return VN_ERR_INVALID;
    vn_ctx* ctx = t->m->ctx;
    int rc = params_ok(ctx, p);
    if (rc) return rc;
    if (t->B <= 0) return vn_fail(ctx, VN_ERR_INVALID, "vn_train_backward: no stashed forward (call vn_train_forward_loss first)%s", "");
    if (stage_hi > t->m->L || stage_lo < -1 || stage_lo > stage_hi)
        return vn_fail(ctx, VN_ERR_INVALID, "vn_train_backward: bad stage range [%s%ld, %ld]", "", stage_lo, stage_hi);
    return backward_range(t, p, grads, stage_hi, stage_lo, (hipStream_t)stream);
}

extern "C" int vn_train_forward_backward(vn_train* t, const int64_t* z_masked, const int64_t* target, int B, int T,
                                         const vn_train_params* p, float* grads, float* loss_dev, void* stream) {
    int rc = vn_train_forward_loss(t, z_masked, target, B, T, p, grads, loss_dev, stream);
    if (rc) return rc;
    return backward_range(t, p, grads, t->m->L, -1, (hipStream_t)stream);
}

// logits of the train()-mode forward only (parity tests): [B][T][Cp][vocab]
extern "C" int vn_train_forward(vn_train* t, const int64_t* z_masked, int B, int T, const vn_train_params* p, float* logits,
                                void* stream) {
    if (!t || !z_masked || !logits) return VN_ERR_INVALID;
    vn_model* m = t->m;
    vn_ctx* ctx = m->ctx;
    hipStream_t s = (hipStream_t)stream;
    int rc = params_ok(ctx, p);
    if (rc) return rc;
    if (B <= 0 || T <= 0 || B > m->d.max_batch || T > m->d.max_T)
        return vn_fail(ctx, VN_ERR_INVALID, "train forward: B=%s%ld, T=%ld outside the model workspace", "", B, T);
    t->B = t->T = 0;                            // the stash is overwritten: a pending staged backward must not use it
    if ((rc = vn_launch_i64_to_i32(ctx, z_masked, m->z, (long)B * m->d.n_codebooks * T, s))) return rc;
    if ((rc = forward_train(t, B, T, p, s))) return rc;
    VN_HIP_CHECK(ctx, hipMemcpyAsync(logits, m->logits, (size_t)B * T * t->NV * sizeof(float), hipMemcpyDeviceToDevice, s));
    return VN_OK;
}

// val_loop (train.py:327-377): eval()-mode forward (dropout off) + per-row loss / rank of the true token
extern "C" int vn_train_eval(vn_train* t, const int64_t* z_masked, const int64_t* target, int B, int T, float label_smoothing,
                             float* row_loss, int32_t* rank, void* stream) {
    if (!t || !z_masked || !target || !row_loss || !rank) return VN_ERR_INVALID;
    vn_model* m = t->m;
    vn_ctx* ctx = m->ctx;
    hipStream_t s = (hipStream_t)stream;
    if (B <= 0 || T <= 0 || B > m->d.max_batch || T > m->d.max_T)
        return vn_fail(ctx, VN_ERR_INVALID, "eval: B=%s%ld, T=%ld outside the model workspace", "", B, T);
    vn_train_params p{};
    p.step = 1; p.world_size = 1; p.dropout = 0.f;
    int rc;
    t->B = t->T = 0;                            // the stash is overwritten: a pending staged backward must not use it
    if ((rc = vn_launch_i64_to_i32(ctx, z_masked, m->z, (long)B * m->d.n_codebooks * T, s))) return rc;
    if ((rc = forward_train(t, B, T, &p, s))) return rc;
    return vn_launch_eval_rows(ctx, m->logits, target, (long)B * T * m->Cp, m->d.vocab, label_smoothing, row_loss, rank, s);
}

extern "C" int vn_train_update(vn_train* t, const float* grads, float* mom, float* var, const vn_train_params* p,
                               float* grad_norm_dev, void* stream) {
    if (!t || !grads || !mom || !var || !grad_norm_dev) return VN_ERR_INVALID;
    vn_model* m = t->m;
    vn_ctx* ctx = m->ctx;
    hipStream_t s = (hipStream_t)stream;
    int rc = params_ok(ctx, p);
    if (rc) return rc;
    vn_adamw_args a;
    a.lr = p->lr; a.beta1 = p->beta1; a.beta2 = p->beta2; a.eps = p->eps; a.weight_decay = p->weight_decay;
    a.bc1 = (float)(1.0 - pow((double)p->beta1, (double)p->step));
    a.bc2 = (float)(1.0 - pow((double)p->beta2, (double)p->step));
    a.gscale = 1.0f / (float)p->world_size;
    a.clip = p->grad_clip;
    if (t->lora) {               // the whole LoRA vector is trainable; then W_eff = W + s B A and the derived copies
        if ((rc = vn_launch_grad_norm(ctx, grads, t->n_lora, a.gscale, t->npartial, grad_norm_dev, s))) return rc;
        if ((rc = vn_launch_adamw(ctx, t->lora_params, grads, mom, var, t->n_lora, a, grad_norm_dev, s))) return rc;
        if ((rc = lora_merge_all(t, s))) return rc;
        return vn_train_sync(t, stream);
    }
    if ((rc = vn_launch_grad_norm(ctx, grads, t->n_total, a.gscale, t->npartial, grad_norm_dev, s))) return rc;
    auto range = [&](long lo, long hi) {
        return vn_launch_adamw(ctx, t->params + lo, grads + lo, mom + lo, var + lo, hi - lo, a, grad_norm_dev, s);
    };
    const vn_dims& d = m->d;
    const long V1 = d.vocab + 1, ld = d.latent_dim;
    const long tab = vn_tensor_offset(&d, VN_W_EMB_TABLES, 0);
    for (int c = 0; c < d.n_codebooks; ++c) {      // embedding.special.MASK rows; the codec codebooks are not parameters
        const long lo = tab + ((long)c * V1 + d.vocab) * ld;
        if ((rc = range(lo, lo + ld))) return rc;
    }
    if ((rc = range(vn_tensor_offset(&d, VN_W_EMB_WT, 0), vn_tensor_offset(&d, VN_W_CLS_W, 0)))) return rc;
    if ((rc = range(vn_tensor_offset(&d, VN_W_CLS_B, 0), t->wsize))) return rc;
    if ((rc = range(t->off_g, t->n_total))) return rc;
    return vn_train_sync(t, stream);
}

// ---- ZeRO-1 (train.py:588-590: ZeroRedundancyOptimizer(model.parameters(), AdamW) when world_size > 1) ------------------
// Each rank owns elements [lo, hi) of the flat train vector and keeps Adam moments for that slice only.  Per step:
//   reduce-scatter(sum) of the gradient vector -> the rank's slice ; vn_train_grad_sumsq on the slice ; all-reduce(sum) of the
//   doubles -> global norm ; vn_train_update_shard ; all-gather of the parameter slices ; vn_train_sync.
extern "C" int vn_train_grad_sumsq(vn_train* t, const float* grads, int64_t n, double* sumsq_dev, void* stream) {
    if (!t || !grads || !sumsq_dev || n <= 0) return VN_ERR_INVALID;
    return vn_launch_grad_sumsq(t->m->ctx, grads, (long)n, t->npartial, sumsq_dev, (hipStream_t)stream);
}

// AdamW on the trainable elements inside [lo, hi): grads_shard / mom_shard / var_shard are indexed from lo (element i of the train
// vector <-> shard[i - lo]) and grads_shard holds the SUM over ranks; *grad_norm_dev = || sum / world_size ||_2 over the WHOLE
// vector (computed by the caller from the ranks' vn_train_grad_sumsq).  Parameters are updated in place in the trainer's full
// vector; the caller all-gathers the slices and then calls vn_train_sync.
extern "C" int vn_train_update_shard(vn_train* t, const float* grads_shard, float* mom_shard, float* var_shard,
                                     const vn_train_params* p, int64_t lo, int64_t hi, const float* grad_norm_dev, void* stream) {
    if (!t || !grads_shard || !mom_shard || !var_shard || !grad_norm_dev) return VN_ERR_INVALID;
    vn_model* m = t->m;
    vn_ctx* ctx = m->ctx;
    hipStream_t s = (hipStream_t)stream;
    int rc = params_ok(ctx, p);
    if (rc) return rc;
    if (t->lora) return vn_fail(ctx, VN_ERR_UNSUPPORTED, "update_shard: LoRA-only training keeps a replicated optimiser (2.9 M parameters)%s", "");
    if (lo < 0 || hi > t->n_total || lo > hi) return vn_fail(ctx, VN_ERR_INVALID, "update_shard: bad range [%s%ld, %ld)", "", (long)lo, (long)hi);
    vn_adamw_args a;
    a.lr = p->lr; a.beta1 = p->beta1; a.beta2 = p->beta2; a.eps = p->eps; a.weight_decay = p->weight_decay;
    a.bc1 = (float)(1.0 - pow((double)p->beta1, (double)p->step));
    a.bc2 = (float)(1.0 - pow((double)p->beta2, (double)p->step));
    a.gscale = 1.0f / (float)p->world_size;
    a.clip = p->grad_clip;
    auto range = [&](long r0, long r1) {
        r0 = r0 > lo ? r0 : (long)lo;
        r1 = r1 < hi ? r1 : (long)hi;
        if (r1 <= r0) return (int)VN_OK;
        return vn_launch_adamw(ctx, t->params + r0, grads_shard + (r0 - lo), mom_shard + (r0 - lo), var_shard + (r0 - lo), r1 - r0, a,
                               grad_norm_dev, s);
    };
    const vn_dims& d = m->d;
    const long V1 = d.vocab + 1, ld = d.latent_dim;
    const long tab = vn_tensor_offset(&d, VN_W_EMB_TABLES, 0);
    for (int c = 0; c < d.n_codebooks; ++c) {      // embedding.special.MASK rows; the codec codebooks are not parameters
        const long r0 = tab + ((long)c * V1 + d.vocab) * ld;
        if ((rc = range(r0, r0 + ld))) return rc;
    }
    if ((rc = range(vn_tensor_offset(&d, VN_W_EMB_WT, 0), vn_tensor_offset(&d, VN_W_CLS_W, 0)))) return rc;
    if ((rc = range(vn_tensor_offset(&d, VN_W_CLS_B, 0), t->wsize))) return rc;
    return range(t->off_g, t->n_total);
}

extern "C" int vn_dropout_keep_mask(vn_ctx* ctx, uint64_t seed, int64_t step, int layer, int site, float p, int64_t row0,
                                    int64_t rows, int cols, uint8_t* out, void* stream) {
    if (!ctx || !out || site < 0 || site > 3) return VN_ERR_INVALID;
    vn_train_params tp{};
    tp.seed = seed; tp.step = step; tp.dropout = p;
    return vn_launch_dropout_mask(ctx, out, rows, cols, make_drop(&tp, layer, site, row0), (hipStream_t)stream);
}

// single-kernel entry point (tests / tuning): dst [C][ldd] <- src [R][C]^T, columns R..ldd-1 zero-filled
extern "C" int vn_transpose_f32(vn_ctx* ctx, const float* src, float* dst, int R, int C, int ldd, void* stream) {
    if (!ctx || !src || !dst) return VN_ERR_INVALID;
    return vn_launch_transpose(ctx, src, dst, R, C, C, ldd, (hipStream_t)stream);
}

// single-kernel entry points for the training attention (tests / tuning).  The expanded bias table and the LUT are
// rebuilt per call (the training step keeps them in the model workspace).
//   q,k,v [B][H][T][64]; out [B][T][H*64]; lse [B][H][T]
//   backward: dout [B][T][H*64] -> dqkv [B*T][3*H*64] (dq | dk | dv), dbias [num_buckets][H] (ACCUMULATED into)
extern "C" int vn_attention_train_f32(vn_ctx* ctx, const float* q, const float* k, const float* v, const float* rel_bias,
                                      float* out, float* lse, const float* dout, float* dqkv, float* dbias, int B, int H, int T,
                                      int num_buckets, int max_distance, float dropout, uint64_t seed, void* stream) {
    if (!ctx || !q || !k || !v || !rel_bias || !out || !lse || T <= 0 || H <= 0) return VN_ERR_INVALID;
    if (dout && (!dqkv || !dbias)) return VN_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    float *full = nullptr, *delta = nullptr;
    int32_t* lut_d = nullptr;
    const int n = 2 * T - 1;
    int rc = VN_OK;
    if (vn_dev_malloc((void**)&full, (size_t)H * n * sizeof(float)) != hipSuccess ||
        vn_dev_malloc((void**)&lut_d, (size_t)n * sizeof(int32_t)) != hipSuccess ||
        vn_dev_malloc((void**)&delta, (size_t)B * H * T * sizeof(float)) != hipSuccess)
        rc = vn_fail(ctx, VN_ERR_OOM, "vn_attention_train_f32: scratch allocation failed%s", "");
    std::vector<int32_t> lut(n);
    vn_bucket_lut_host(T, num_buckets, max_distance, lut.data());
    if (rc == VN_OK && hipMemcpy(lut_d, lut.data(), n * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) rc = VN_ERR_HIP;
    if (rc == VN_OK) rc = vn_launch_bias_expand(ctx, rel_bias, lut_d, full, H, T, s);
    vn_train_params tp{};
    tp.seed = seed; tp.step = 1; tp.dropout = dropout;
    const vn_drop d = make_drop(&tp, 0, SITE_ATTN, 0);
    if (rc == VN_OK) rc = vn_launch_attention_train_fwd(ctx, q, k, v, full, out, lse, B, H, T, d, s);
    float* part = nullptr;
    const long slab = (long)B * H * vn_cdiv(T, 64) * 64;
    if (rc == VN_OK && dout && vn_dev_malloc((void**)&part, (size_t)slab * sizeof(float)) != hipSuccess) rc = VN_ERR_OOM;
    if (rc == VN_OK && dout)
        rc = vn_launch_attention_bwd(ctx, q, k, v, full, lut_d, out, dout, lse, delta, dqkv, part, B, H, T, num_buckets, d, s);
    if (rc == VN_OK && dout) rc = vn_launch_dbias_reduce(ctx, part, dbias, 1, slab, B, H, vn_cdiv(T, 64), num_buckets, true, s);
    (void)hipStreamSynchronize(s);
    (void)vn_dev_free(full);
    (void)vn_dev_free(lut_d);
    (void)vn_dev_free(delta);
    (void)vn_dev_free(part);
    return rc;
}

// host-only (no GPU needed; tests/test_host_logic.py): the relative-position bucket LUT of length 2 T - 1 (index key - query + T - 1) as the
// kernels use it, and the half-width of the per-wave bias-gradient tables the split-plane backward derives from it
extern "C" int vn_attention_bwd_table_span(int T, int num_buckets, int max_distance, int32_t* lut_out, int* near_r) {
    if (T <= 0 || num_buckets <= 0 || max_distance <= 0 || !near_r) return VN_ERR_INVALID;
    std::vector<int32_t> lut(2 * T - 1);
    vn_bucket_lut_host(T, num_buckets, max_distance, lut.data());
    if (lut_out) memcpy(lut_out, lut.data(), lut.size() * sizeof(int32_t));
    *near_r = vn_attention_x3_near_r(lut.data(), T);
    return VN_OK;
}

// the same on the split-plane pipe (attention_x3.hip TRAIN forward, attention_train_x3.hip backward): fp32 q, k, v are split /
// transposed here exactly as the QKV GEMM's plane epilogue does (engine.hip vn_attn_x3_prep_kernel)
extern "C" int vn_attention_train_bf16x3(vn_ctx* ctx, const float* q, const float* k, const float* v, const float* rel_bias,
                                         float* out, float* lse, const float* dout, float* dqkv, float* dbias, int B, int H, int T,
                                         int num_buckets, int max_distance, float dropout, uint64_t seed, void* stream) {
    if (!ctx || !q || !k || !v || !rel_bias || !out || !lse || T <= 0 || H <= 0 || B <= 0) return VN_ERR_INVALID;
    if (dout && (!dqkv || !dbias)) return VN_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const long heads = (long)B * H, n = heads * T * VN_DHEAD;
    const long plane_qk = 2 * n, plane_vt = (long)H * (((long)B * T + 31) / 32) * (VN_DHEAD * 32);
    const int nb = 2 * T - 1;
    float *full = nullptr, *delta = nullptr, *part = nullptr;
    int32_t* lut_d = nullptr;
    uint16_t *qk16 = nullptr, *vt16 = nullptr, *ws = nullptr;
    vn_ax_bwd_ws w;
    vn_attention_x3_bwd_ws_layout(B, H, T, &w);
    const long slab = heads * vn_cdiv(T, 128) * 64;
    int rc = VN_OK;
    if (vn_dev_malloc((void**)&full, (size_t)H * nb * sizeof(float)) != hipSuccess || vn_dev_malloc((void**)&lut_d, (size_t)nb * sizeof(int32_t)) != hipSuccess ||
        vn_dev_malloc((void**)&delta, (size_t)heads * T * sizeof(float)) != hipSuccess || vn_dev_malloc((void**)&part, (size_t)slab * sizeof(float)) != hipSuccess ||
        vn_dev_malloc((void**)&qk16, ax_qk_elems(3, plane_qk) * 2) != hipSuccess ||
        vn_dev_malloc((void**)&vt16, (size_t)3 * plane_vt * 2) != hipSuccess || vn_dev_malloc((void**)&ws, (size_t)w.total * 2) != hipSuccess)
        rc = vn_fail(ctx, VN_ERR_OOM, "vn_attention_train_bf16x3: scratch allocation failed%s", "");
    std::vector<int32_t> lut(nb);
    vn_bucket_lut_host(T, num_buckets, max_distance, lut.data());
    const int near_r = vn_attention_x3_near_r(lut.data(), T);
    if (rc == VN_OK && hipMemcpy(lut_d, lut.data(), nb * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) rc = VN_ERR_HIP;
    if (rc == VN_OK && (hipMemsetAsync(vt16, 0, (size_t)3 * plane_vt * 2, s) != hipSuccess ||
                        hipMemsetAsync(qk16, 0, ax_qk_elems(3, plane_qk) * 2, s) != hipSuccess ||
                        hipMemsetAsync(ws, 0, (size_t)w.total * 2, s) != hipSuccess))
        rc = VN_ERR_HIP;
    if (rc == VN_OK) rc = vn_launch_bias_expand(ctx, rel_bias, lut_d, full, H, T, s);
    vn_train_params tp{};
    tp.seed = seed; tp.step = 1; tp.dropout = dropout;
    const vn_drop d = make_drop(&tp, 0, SITE_ATTN, 0);
    if (rc == VN_OK) rc = vn_launch_attn_x3_prep(ctx, q, k, v, qk16, plane_qk, vt16, plane_vt, heads, H, T, s);
    if (rc == VN_OK) rc = vn_launch_attention_x3_train_fwd(ctx, qk16, qk16 + n, plane_qk, vt16, plane_vt, full, out, lse, B, H, T, vn_num_cus(ctx), d, s);
    if (rc == VN_OK && dout)
        rc = vn_launch_attention_x3_bwd(ctx, qk16, plane_qk, vt16, plane_vt, ws, full, lut_d, near_r, out, dout, lse, delta, dqkv, part, B, H, T,
                                        num_buckets, d, s);
    if (rc == VN_OK && dout) rc = vn_launch_dbias_reduce(ctx, part, dbias, 1, slab, B, H, vn_cdiv(T, 128), num_buckets, true, s);
    (void)hipStreamSynchronize(s);
    (void)vn_dev_free(full); (void)vn_dev_free(lut_d); (void)vn_dev_free(delta); (void)vn_dev_free(part); (void)vn_dev_free(qk16); (void)vn_dev_free(vt16); (void)vn_dev_free(ws);
    return rc;
}
