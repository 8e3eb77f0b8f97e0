/*
 * vampnet_hip.h — C ABI of libvampnet_hip.so, the MI355X (gfx950) engine for the VampNet
 * `Interface.vamp()` hot path.
 *
 * The reference (hugofloresgarcia/vampnet) is pure Python/PyTorch and has NO FFI seam; the
 * path sits behind the Python class `Interface` (vampnet/interface.py:54-575).  This header
 * is the seam a maintainer would bind instead of the torch ops (see INTEGRATION.md for the
 * ctypes stub).  Each entry point cites the reference code it replaces.
 *
 * Conventions
 *   - every pointer marked "dev" is a DEVICE pointer owned by the caller (e.g. tensor.data_ptr()
 *     of a torch-ROCm tensor); the library never frees or retains it past the call, except the
 *     weight blob of vn_model_create which must outlive the model.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); all
 *     work is enqueued asynchronously on it; nothing synchronises the device.
 *   - every function returns VN_OK (0) or a negative vn_status; vn_last_error(ctx) gives text.
 *     No C++ exception crosses the ABI.  No allocation happens after *_create (workspace is sized
 *     at vn_model_create from dims.max_batch/max_T), so calls are hipGraph-capturable.
 *   - a vn_ctx and the models created from it may be used by ONE host thread at a time.
 *   - all arithmetic is fp32 (exact-f32 MFMA v_mfma_f32_32x32x2_f32 / 16x16x4_f32) — the parity
 *     mode of the CPU reference.
 *   - tokens are int64 at the boundary like the reference's LongTensors; MASK == dims.vocab.
 */
#ifndef VAMPNET_HIP_H
#define VAMPNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vn_ctx vn_ctx;
typedef struct vn_model vn_model;

typedef enum {
    VN_OK = 0,
    VN_ERR_INVALID = -1,     /* bad argument / unsupported shape */
    VN_ERR_HIP = -2,         /* a HIP runtime call failed */
    VN_ERR_OOM = -3,
    VN_ERR_UNSUPPORTED = -4
} vn_status;

/* Model hyper-parameters = VampNet.__init__ kwargs (vampnet/modules/transformer.py:535-552). */
typedef struct {
    int32_t n_layers;      /* 20 coarse / 16 c2f                    */
    int32_t n_heads;       /* 20; d_model / n_heads must be 64      */
    int32_t d_model;       /* 1280                                  */
    int32_t n_codebooks;   /* 4 coarse / 14 c2f                     */
    int32_t n_cond;        /* n_conditioning_codebooks: 0 / 4       */
    int32_t vocab;         /* 1024 (MASK token id == vocab)         */
    int32_t latent_dim;    /* 8                                     */
    int32_t num_buckets;   /* 32  (transformer.py:96)               */
    int32_t max_distance;  /* 128 (transformer.py:97)               */
    float   eps;           /* 1e-6 RMSNorm (transformer.py:38)      */
    int32_t max_batch;     /* workspace sizing                      */
    int32_t max_T;         /* workspace sizing (575 / 173)          */
} vn_dims;

/* Packed fp32 weight blob.  The library defines the layout; the host packer asks for offsets.
 * tensor ids (layer = -1 for non-layer tensors):                                              */
enum {
    VN_W_EMB_TABLES = 0,  /* [C][vocab+1][latent]   codec codebook rows + MASK row (layers.py:134-150) */
    VN_W_EMB_WT     = 1,  /* [C*latent][D]          embedding.out_proj.weight TRANSPOSED (layers.py:132) */
    VN_W_EMB_B      = 2,  /* [D]                                                                  */
    VN_W_REL_BIAS   = 3,  /* [num_buckets][H]       layers.0.self_attn.relative_attention_bias    */
    VN_W_FINAL_NORM = 4,  /* [D]                    transformer.norm.weight                       */
    VN_W_CLS_W      = 5,  /* [Cp*vocab][D]  weight-norm folded, rows re-ordered to (c, p)  (transformer.py:596-604,634) */
    VN_W_CLS_B      = 6,  /* [Cp*vocab]     same row order                                        */
    VN_W_NORM1      = 7,  /* per layer [D]          norm_1.weight                                 */
    VN_W_QKV        = 8,  /* per layer [3D][D]      rows: w_qs | w_ks | w_vs  (transformer.py:109-111) */
    VN_W_WO         = 9,  /* per layer [D][D]       self_attn.fc.weight                           */
    VN_W_NORM3      = 10, /* per layer [D]          norm_3.weight                                 */
    VN_W_W1         = 11, /* per layer [4D][D]      feed_forward.w_1.weight, rows interleaved in 32-blocks:
                             packed row 64*g + i      (i < 32) = original row 32*g + i        (value half p1)
                             packed row 64*g + 32 + i (i < 32) = original row 2D + 32*g + i   (gate half p2)
                             so one wave tile holds matching value/gate columns (activations.py:33-35) */
    VN_W_W2         = 12, /* per layer [D][2D]      feed_forward.w_2.weight                       */
    VN_W__COUNT     = 13
};

/* Sampling parameters = VampNet.generate kwargs (transformer.py:687-710). */
typedef struct {
    int32_t steps;               /* _sampling_steps                                             */
    float   temperature;         /* <= 0 behaves as 1.0 (transformer.py:1019-1023)              */
    float   mask_temperature;    /* 10.5                                                        */
    double  sample_cutoff;       /* sample iff (i/steps) <= sample_cutoff, compared in double like
                                    the reference's Python floats (transformer.py:852)          */
    float   top_p;               /* nucleus filtering when 0 < top_p < 1 (transformer.py:1001-1016);
                                    otherwise disabled                                          */
    int64_t n0_override;         /* < 0: N0 = masked count over THIS batch (transformer.py:766);
                                    >= 0: the global batch's N0 when this call sees a shard     */
    uint64_t seed;               /* device-RNG seed (used only where a noise pointer is NULL)   */
    int64_t batch_offset;        /* index of this call's first item inside the GLOBAL batch: the
                                    device-RNG stream is indexed by global item, so a batch sharded
                                    over GPUs draws the same noise as the unsharded batch         */
    int32_t call_batch;          /* items per reference generate() call when several calls are batched
                                    into this launch (item i belongs to call i / call_batch); 0 = B */
    int32_t global_batch;        /* global batch size of one call; 0 = call_batch.  Device-RNG item id =
                                    (i / call_batch) * global_batch + batch_offset + i % call_batch  */
    void* const* step_events;    /* NULL, or `steps` hipEvent_t handles: before the sampling kernels of step i read the
                                    caller-supplied noise, the stream waits for step_events[i] (noise for later steps may
                                    still be in production on another stream while earlier steps run)                 */
} vn_sample_params;

/* ---- context ------------------------------------------------------------------------------
 * One context per device (and per thread that drives it).  A context owns a small device scratch (stream-K partial-sum
 * slabs, hand-off flags): all work enqueued through ONE context must be stream-ordered (one stream, or streams chained
 * by events); use separate contexts for concurrent streams.  The calling thread's current HIP device must be the
 * context's device for every call (vn_ctx_create / vn_model_create select it; a host that switches devices must switch
 * back, as torch.cuda.device does).  The reference itself is single-threaded.                           */
int  vn_ctx_create(int device, vn_ctx** out);
void vn_ctx_destroy(vn_ctx* ctx);
const char* vn_last_error(const vn_ctx* ctx);
/* library build info, e.g. "vampnet_hip 0.1 gfx950 f32-mfma" */
const char* vn_version(void);

/* ---- kernel timing (bench.py's roofline leg) ------------------------------------------------
 * While enabled, every launch of the MFMA kernels (vn_gemm_f32[_sk]_kernel, vn_attention_kernel,
 * vn_conv1d_f32_kernel) made through this ctx is bracketed by hipEvents on the launch stream.
 * vn_profile_end synchronises the recorded events and returns, per class c in {0: GEMMs of the precision in use, 1: attention,
 * 2: codec convolutions bound by the split-plane matrix pipe, 3: gemm bf16 (fast mode), 4: codec convolutions bound by the fp32-input
 * MFMA, 5: codec convolutions bound by their operand BYTES (arithmetic intensity below the ridge of the pipe they run on: the
 * audio-rate and k = 1 layers)} — 6 classes x 4 doubles:
 *   stats[4c+0] = launches, stats[4c+1] = total kernel time in ms, stats[4c+2] = algorithmic FLOPs
 *   (2*M*N*K per GEMM / conv launch; 4*T*T*64 per (b,h) for attention), stats[4c+3] = algorithmic operand bytes
 *   (every fp32 operand read once + every result written once).                                  */
int vn_profile_begin(vn_ctx* ctx, int max_launches);
/* Bracket only ~1 of every `stride` MFMA launches (selected by a hash of the launch counter, so every shape of a
 * periodic schedule is sampled): two hipEventRecords cost ~7 us of stream time per bracketed launch (2.3 % of a B = 8
 * vamp() step, 11 % at B = 1 with stride 1).  The returned statistics then cover the sampled launches only.       */
int vn_profile_set_stride(vn_ctx* ctx, int stride);
int vn_profile_end(vn_ctx* ctx, double* stats24);

/* ---- weights ------------------------------------------------------------------------------ */
/* total number of floats in the packed blob */
int vn_weights_size(const vn_dims* dims, int64_t* n_floats);
/* offset (in floats) and element count of one tensor inside the blob */
int vn_weights_offset(const vn_dims* dims, int tensor_id, int layer, int64_t* offset, int64_t* count);

/* ---- model -------------------------------------------------------------------------------- */
/* replaces VampNet.__init__ + load_state_dict (transformer.py:535-615; interface.py:27-50).
 * `blob_dev`: packed weights (layout above) in device memory, must outlive the model.        */
int  vn_model_create(vn_ctx* ctx, const vn_dims* dims, const float* blob_dev, vn_model** out);
void vn_model_destroy(vn_model* model);

/* Optional bf16 FAST MODE (not bit-exact; the reference itself runs bf16 autocast on a GPU: interface.py:364,428).
 * `blob_bf16_dev`: the packed weight blob converted element-wise to bf16 (same element offsets), device memory that
 * must outlive the model; NULL switches back to exact fp32.  The GEMM A/W operands become bf16
 * (v_mfma_f32_32x32x16_bf16, fp32 accumulate); attention, norms, residual stream, softmax and sampling stay fp32. */
int vn_model_set_bf16(vn_model* model, const void* blob_bf16_dev);

/* Optional "bf16x3" mode: fp32-GRADE GEMMs on the bf16 matrix cores.  Every GEMM operand is held as three bf16 planes
 * whose sum is the fp32 value exactly (vn_split3_f32), and A W^T is evaluated as the six plane products down to 2^-16 of
 * the leading one with fp32 accumulation: error <= that of an fp32-accumulating fp32 GEMM, at 6/16 of the
 * exact-fp32 MFMA's matrix time.  `blob_planes_dev`: vn_split3_f32 of the packed blob, planes `plane_stride` elements
 * apart (>= vn_weights_size, multiple of 8), 16-byte aligned device memory that must outlive the model; NULL switches
 * back to exact fp32.  Attention, norms, residual stream, softmax and sampling are the fp32 path unchanged.            */
int vn_model_set_bf16x3(vn_model* model, const void* blob_planes_dev, int64_t plane_stride);
/* Optional, OPT-IN "f16x2" mode (NOT the default: its operands are NARROWER than fp32 — 22 significand bits, fp16's exponent range —
 * so it is a fast mode with measured fp32-level results on well-scaled models, not an fp32-equivalent): GEMMs as THREE fp16
 * matrix-core products.  Every GEMM operand is two fp16 planes, h0 = fp16(x) and h1 = fp16((x - h0) * 2^11) (vn_split2_f16 below):
 * x = h0 + 2^-11 h1 to within 2^-22 |x| (four times fp32's own representation error) — at half of bf16x3's matrix time.
 * on != 0: the engine builds the weight planes from the fp32 blob it already holds; on == 0: back to exact fp32.
 * Attention runs on fp16 TWO-PLANE operands too (attention_x3.hip, NP = 2): q / 8 and k as h0 = fp16(x), h1 = fp16(x - h0) (second
 * plane unscaled: values below 0.125 keep 2^-25 absolute instead of 2^-22 relative), V^T times 16 in the same form.
 * RANGE: a GEMM operand with |x| >= 65504, or an attention value with |v| >= 4094 (16 v >= 65504), is CLAMPED — and recorded on the
 * context's saturation ledger (vn_saturation_flags below); a caller must read the ledger after the work and repeat it in another
 * precision if a word is set (vampnet_amd/engine.py does: the call is re-run on bf16x3).                                          */
int vn_model_set_f16x2(vn_model* model, int on);
/* The saturation ledger of the fp16 plane writers (f16x2 mode, vn_split2_f16, vn_conv1d_f16x2, vn_attention_f16x2): sticky device
 * words, set when a value that was turned into fp16 planes did not fit (|x| >= 65504 or NaN) and was clamped.
 * flags4[0] GEMM-operand planes (normalised rows, attention output, GEGLU output, codec activations), flags4[1] attention operands
 * (q / 8, k, 16 v), flags4[2] weight planes, flags4[3] reserved.  Synchronises `stream`; clear != 0 resets the words.  A context that
 * never runs an fp16-plane kernel always reads zeros.                                                                             */
int vn_saturation_flags(vn_ctx* ctx, uint32_t* flags4, int clear, void* stream);
/* dst16[q * plane_stride + i] = q-th split term of src[i], q = 0..2 (n % 4 == 0, src 16-byte aligned) */
int vn_split3_f32(vn_ctx* ctx, const float* src, void* dst16, int64_t n, int64_t plane_stride, void* stream);

/* replaces embedding.from_codes + VampNet.forward (layers.py:134-163; transformer.py:617-639).
 * codes  dev int64 [B][C][T] (MASK = vocab allowed in any codebook)
 * logits dev f32   [B][T][Cp][vocab]  (== reference logits[b, p, t*Cp + c] transposed so each
 *                                         (t, c) owns `vocab` contiguous values)              */
int vn_forward(vn_model* model, const int64_t* codes, int B, int T, float* logits, void* stream);

/* replaces VampNet.generate (transformer.py:686-946; spec SURVEY.md App. A), return_signal=False,
 * ctrls=None, cfg_guidance=None.
 * start_tokens dev int64 [B][C][T]; mask dev int64 [B][C][T] in {0,1};
 * num_to_mask_sched host int64 [steps][B] or NULL: floor(gamma((i+1)/steps) * N0) per step AND per item
 *     (transformer.py:903) as the caller computed it (bit-exact torch fp32).  Per item because the caller may
 *     batch items that belong to different reference generate() calls (e.g. the 4 coarse-to-fine chunks of
 *     interface.py:360-374), each with its own batch-wide N0.  NULL = computed here from one N0;
 * exp_noise  dev f32 [steps][B*T*Cp][vocab] Exp(1) draws (multinomial replay) or NULL
 *     = device Philox stream;  unif_noise dev f32 [steps][B][T*Cp] U(1e-20,1) or NULL likewise;
 * out_tokens dev int64 [B][C][T].                                                              */
int vn_generate(vn_model* model, const int64_t* start_tokens, const int64_t* mask, int B, int T,
                const vn_sample_params* params, const int64_t* num_to_mask_sched,
                const float* exp_noise, const float* unif_noise, int64_t* out_tokens, void* stream);

/* One sampling step on caller-provided logits (teacher-forced parity tests):
 * sample_from_logits + the where()/inf bookkeeping + mask_by_random_topk + re-mask
 * (transformer.py:852-927, 952-1074).
 * z_masked dev int64 [B][C][T] in/out; logits dev f32 [B][T][Cp][vocab] (filtered IN PLACE when top_p is
 * active, like the reference's `logits[indices_to_remove] = -inf`, transformer.py:1016);
 * exp_noise dev f32 [B*T*Cp][vocab] or NULL; unif_noise dev f32 [B][T*Cp] or NULL;
 * sampled_out dev int64 [B][C][T] (tokens before re-masking; conditioning codebooks copied).   */
int vn_sample_step(vn_model* model, int64_t* z_masked, float* logits, int B, int T,
                   int step, const vn_sample_params* params, const int64_t* num_to_mask_sched /* host [B] */,
                   const float* exp_noise, const float* unif_noise, int64_t* sampled_out, void* stream);

/* ---- single kernels (unit tests / profiling; same kernels the model uses) ------------------ */
/* RMSNorm (transformer.py:55-58): y[r][:] = w * (x[r][:] * rsqrt(mean(x^2) + eps))             */
int vn_rmsnorm_f32(vn_ctx* ctx, const float* x, const float* w, float* y, int rows, int D, float eps, void* stream);

/* C[M][N] (op)= A[M][K] * W[N][K]^T  (torch F.linear).  epilogue:
 *   0 store; 1 store + bias[N]; 2 C += (residual, transformer.py:347,367);
 *   3 GEGLU: W rows interleaved as VN_W_W1, C is [M][N/2] (activations.py:16-35)                */
int vn_gemm_f32(vn_ctx* ctx, const float* A, const float* W, const float* bias, float* C,
                int M, int N, int K, int epilogue, void* stream);

/* bf16-operand GEMM of the fast mode (A16 [M][K], W16 [N][K] bf16, K % 64 == 0; fp32 accumulate/output;
 * epilogue 0 store, 1 bias, 2 residual)                                                                        */
int vn_gemm_bf16(vn_ctx* ctx, const void* A16, const void* W16, const float* bias, float* C, int M, int N, int K,
                 int epilogue, void* stream);

/* bf16x3 GEMM as a single op (tests / tuning): A3 [3][M][K] and W3 [3][N][K] split planes (plane strides in elements),
 * fp32 C; epilogues 0..3 as vn_gemm_f32; K % 32 == 0, N % 64 == 0.  A plane stride of -1 says that operand is given in the
 * TILED layout the model path uses for weights and activations — [row / 16][k / 32][plane][row % 16][k % 32] bf16, rows padded to
 * a multiple of 16: the 16-row x 32-k block of a plane is one contiguous 1 KiB piece, i.e. one LDS-DMA instruction of whole
 * cache lines (vn_model_set_bf16x3 re-lays the weight planes it is given this way; the engine's norm / attention / GEGLU
 * producers write it directly).                                                                                         */
int vn_gemm_bf16x3(vn_ctx* ctx, const void* A3, int64_t a_plane, const void* W3, int64_t w_plane, const float* bias,
                   float* C, int M, int N, int K, int epilogue, void* stream);
/* The same kernel in its TN operand mode: C [M][N] = At^T Wt with BOTH operands token-major, At [tokens][M] and Wt [tokens][N], given
 * as their TILED planes ([token / 16][column / 32][plane][16][32]: the layout every activation operand already has).  The weight
 * gradients of training (dW = dY^T X, train.hip) use it: no transposing pass, the fragment reads transpose in LDS (ds_read_b64_tr_b16).
 * M % 32 == 0, N % 64 == 0; rows past `tokens` inside the last 16-row block must be zero.                                  */
int vn_gemm_bf16x3_tn(vn_ctx* ctx, const void* At3, const void* Wt3, float* C, int M, int N, int tokens, void* stream);

/* "f16x2": the second fp32-grade operand format of the same kernel (gemm_x3.hip).  An operand is TWO fp16 planes, h0 = fp16(x)
 * and h1 = fp16((x - h0) * 2^11): x = h0 + 2^-11 h1 to within 2^-22 |x| (four times fp32's own representation error, random in
 * sign) over fp16's whole normal range, and A W^T takes THREE matrix-core products (a0 b0; a0 b1 + a1 b0 into a second accumulator
 * that joins times 2^-11) instead of bf16x3's six — half the matrix time at an error that stays below the rounding noise of an
 * fp32-accumulating fp32 GEMM (7e-8 vs 3.4e-7 rms on the model's shapes).  |x| > 65504 saturates.
 * vn_split2_f16: fp32 [rows][K] -> planes; tiled != 0: [rows / 16][K / 32][2][16][32] (rows % 16 == 0, K % 32 == 0), else two
 * planar planes plane_stride elements apart.  vn_gemm_f16x2: as vn_gemm_bf16x3 (plane stride -1 = that operand tiled).        */
int vn_split2_f16(vn_ctx* ctx, const float* src, void* dst16, int64_t rows, int K, int64_t plane_stride, int tiled, void* stream);
int vn_gemm_f16x2(vn_ctx* ctx, const void* A2, int64_t a_plane, const void* W2, int64_t w_plane, const float* bias,
                  float* C, int M, int N, int K, int epilogue, void* stream);

/* The self-attention core as a SINGLE OP (fp32 q, k, v in, scratch allocated and freed inside, synchronous) is a test entry, not
 * part of the boundary: vn_attention_f32 / _bf16 / _bf16x3 / _f16x2 and the training pair live in vampnet_hip_debug.h.  The path
 * itself runs attention inside vn_forward / vn_generate / vn_train_* on the workspace of vn_model_create / vn_train_create.  */

/* ---- DAC codec layers (Interface.encode / Interface.decode; PARITY UNPINNED: the codec source
 * `lac` is not part of the reference tree) -------------------------------------------------------------------
 * All activations are channels-last [B][T][C] fp32.
 *
 * vn_conv1d_f32: y[b][t_out][co] = act( bias[co] + sum_{j<taps} sum_ci w[co][j][ci] * x[b][t_in][ci] (+ resid) )
 *   for t' in [0, T_rows): t_in = t'*in_stride + j*dil - pad (zero outside [0, T_in)),
 *   t_out = t'*out_stride + out_off (rows outside [0, T_out) are dropped).
 *   w dev [C_out][taps][C_in] (C_in % 32 == 0); y raw result or NULL; y2 = snake(result, alpha) or NULL
 *   (Snake1d of the NEXT layer fused: vampnet/modules/layers.py:12-18); act 1 = tanh; y2_16 = the same snake output as three
 *   exact split bf16 planes [3][B*T_out][C_out] (y2_plane elements apart) for a consumer on the bf16x3 pipe, or NULL.
 *   Covers WNConv1d (any k/dilation/stride) and, one call per phase, WNConvTranspose1d(k=2s, stride s).       */
int vn_conv1d_f32(vn_ctx* ctx, const float* x, const float* w, const float* bias, const float* resid,
                  const float* alpha, float* y, float* y2, void* y2_16, int64_t y2_plane, int B, int T_in, int T_rows, int T_out,
                  int C_in, int C_out, int taps, int in_stride, int dil, int pad, int out_stride, int out_off, int act,
                  void* stream);
/* The same operator with the products on the bf16 matrix cores at fp32 grade (gemm_x3.hip's implicit-GEMM mode: six bf16-MFMA
 * products of exact 3-way operand splits, fp32 accumulate): x16 = split planes of the input [3][B*T_in][C_in] (x_plane elements
 * apart; C_in % 32 == 0), w_tiled = the TILED planes of w viewed as [C_out][taps*C_in] (vn_split3_f32, then vn_tile_planes_bf16x3;
 * C_out % 16 == 0).  Outputs as vn_conv1d_f32 (at least one of y / y2 / y2_16).                                              */
int vn_conv1d_bf16x3(vn_ctx* ctx, const void* x16, int64_t x_plane, const void* w_tiled, const float* bias, const float* resid,
                     const float* alpha, float* y, float* y2, void* y2_16, int64_t y2_plane, int B, int T_in, int T_rows, int T_out,
                     int C_in, int C_out, int taps, int in_stride, int dil, int pad, int out_stride, int out_off, int act,
                     void* stream);
/* ... and on f16x2 operands (three fp16-MFMA products, vn_gemm_f16x2's format): x16 = TWO planar fp16 planes [2][B*T_in][C_in]
 * (vn_split2_f16, tiled = 0), w_tiled = vn_split2_f16(w as [C_out][taps*C_in], tiled = 1); y2_16 = snake(y) as two planar fp16 planes
 * y2_plane apart.  (vn_conv1d_f32 writes that format when its y2_plane is NEGATIVE: two planes -y2_plane apart.)                  */
int vn_conv1d_f16x2(vn_ctx* ctx, const void* x16, int64_t x_plane, const void* w_tiled, const float* bias, const float* resid,
                    const float* alpha, float* y, float* y2, void* y2_16, int64_t y2_plane, int B, int T_in, int T_rows, int T_out,
                    int C_in, int C_out, int taps, int in_stride, int dil, int pad, int out_stride, int out_off, int act,
                    void* stream);
/* planar split planes [3][rows][K] (plane_stride elements apart; rows % 16 == 0, K % 32 == 0) -> the tiled layout
 * [rows/16][K/32][3][16][32] in which the bf16x3 GEMM / convolution read their weights (`tiled`: 3*rows*K bf16)              */
int vn_tile_planes_bf16x3(vn_ctx* ctx, const void* planes, int64_t plane_stride, void* tiled, int64_t rows, int K, void* stream);
/* encoder stem WNConv1d(1 -> C, k=7, pad 3): x dev [B][T], w dev [C][7]; y / y2 as above                      */
int vn_dac_conv_in_f32(vn_ctx* ctx, const float* x, const float* w, const float* bias, const float* alpha,
                       float* y, float* y2, int B, int T, int C, void* stream);
/* decoder head WNConv1d(C -> 1, k=7, pad 3) + tanh on the snake-activated input xs [B][T][C]; w dev [7][C]   */
int vn_dac_conv_out_f32(vn_ctx* ctx, const float* xs, const float* w, float bias, float* y, int B, int T, int C,
                        void* stream);
/* residual VQ encode (codec.encode(...)["codes"], interface.py:223): z dev [B*T][L] -> codes dev int64 [B][n][T];
 * win [n][8][L], bin [n][8], cb [n][Kc][8], wout [n][L][8], bout [n][L]                                       */
int vn_rvq_encode_f32(vn_ctx* ctx, const float* z, const float* win, const float* bin, const float* cb,
                      const float* wout, const float* bout, int64_t* codes, int B, int T, int L, int n_levels,
                      int codebook_size, void* stream);
/* codes -> z_q = sum_i out_proj_i(codebook_i[code_i])  (quantizer.from_latents(from_codes(z)), transformer.py:671-672) */
int vn_rvq_decode_f32(vn_ctx* ctx, const int64_t* codes, const float* cb, const float* wout, const float* bout,
                      float* zq, int B, int T, int L, int n_levels, int codebook_size, void* stream);

/* ---- the codec as ONE call per direction (Interface.encode: vampnet/interface.py:219-224 -> codec.encode(...)["codes"];
 * Interface.decode: vampnet/interface.py:203-204 -> VampNet.decode, vampnet/modules/transformer.py:661-684).  PARITY UNPINNED like the
 * layer entry points above.  The host records the layer loop of one direction for one (batch, length, precision) as a PROGRAM: the
 * ordered launches (each = one of the single-layer entry points above with its arguments) over device pointers the host owns — the
 * weights and ONE activation arena planned from the buffers' live ranges (vampnet_amd/codec.py).  vn_codec_create copies the list;
 * vn_dac_encode / vn_dac_decode walk it on `stream`: no allocation, no synchronisation, ~70 / ~150 launches per call instead of as
 * many host round trips.  Argument slots of an op, in the order of the entry point it names: pointers -> p[], int -> i[], int64 ->
 * l[], float -> f[]; the pointer values VN_CODEC_PTR_IN / VN_CODEC_PTR_OUT stand for the call's input / output.               */
typedef struct vn_codec vn_codec;
enum { VN_CODEC_OP_CONV1D_F32 = 0, VN_CODEC_OP_CONV1D_BF16X3 = 1, VN_CODEC_OP_CONV1D_F16X2 = 2, VN_CODEC_OP_CONV_IN = 3,
       VN_CODEC_OP_CONV_OUT = 4, VN_CODEC_OP_RVQ_ENCODE = 5, VN_CODEC_OP_RVQ_DECODE = 6, VN_CODEC_OP_SPLIT3 = 7, VN_CODEC_OP_SPLIT2 = 8,
       VN_CODEC_OP__COUNT = 9 };
#define VN_CODEC_PTR_IN  1          /* pointer slot = the audio (encode) / the codes (decode) of the call     */
#define VN_CODEC_PTR_OUT 2          /* pointer slot = the codes (encode) / the audio (decode) of the call     */
typedef struct {
    int32_t kind;                   /* VN_CODEC_OP_*                                                          */
    int32_t i[16];
    int64_t l[2];
    float   f[2];
    void*   p[12];
} vn_codec_op;
int  vn_codec_create(vn_ctx* ctx, const vn_codec_op* ops, int n_ops, int direction /* 0 encode, 1 decode */, vn_codec** out);
void vn_codec_destroy(vn_codec* codec);
/* The same program built WITHOUT a Python host (csrc/codec_plan.hip): the library re-lays the weights (channels-last convolution
 * weights, the phases of the transposed convolutions, tiled split planes of the layers that run on the matrix-core pipe), records the
 * layer loop of `direction` for batch B and length n (encode: samples per item, a multiple of the hop; decode: tokens per item) and plans
 * the arena; the returned vn_codec owns all of it.  `blob_dev`: the codec's tensors as one flat fp32 device blob (256-byte aligned), each
 * under its state_dict name with weight-norm already folded (`<conv>.weight` = g v / ||v||), in PyTorch's layouts — Conv1d (C_out, C_in,
 * k), ConvTranspose1d (C_in, C_out, 2 s), alpha (1, C, 1); vn_codec_tensor_name enumerates (name, offset, count), vn_codec_tensor_offset
 * looks one up.  Names (the lac / DAC module tree as vampnet's call sites use it): encoder.block.0, encoder.block.<1+i>
 * .block.<j>.block.{0.alpha, 1, 2.alpha, 3}, encoder.block.<1+i>.block.{3.alpha, 4}, encoder.block.<n+1>.alpha, encoder.block.<n+2>,
 * quantizer.quantizers.<l>.{in_proj, codebook, out_proj}, decoder.model.0, decoder.model.<1+i>.block.{0.alpha, 1, <2+j>...},
 * decoder.model.<n+1>.alpha, decoder.model.<n+2>.  precision: 0 = fp32-input MFMA everywhere, 2 = bf16x3, 3 = f16x2 (the routing rule
 * of the Python host: the MFMA-bound convolutions on the split-plane pipe).  Synchronises the device (a setup call).              */
typedef struct vn_codec_cfg {
    int32_t encoder_dim;            /* 64                                                                     */
    int32_t n_rates;                /* number of encoder / decoder blocks (<= 8)                               */
    int32_t encoder_rates[8];       /* 2, 4, 8, 12 ...                                                         */
    int32_t decoder_dim;            /* 1536                                                                   */
    int32_t decoder_rates[8];       /* 12, 8, 4, 2 ...                                                         */
    int32_t n_codebooks, codebook_size, codebook_dim;
    int32_t latent_dim;             /* 0: encoder_dim * 2^n_rates                                              */
} vn_codec_cfg;
int  vn_codec_weights_size(const vn_codec_cfg* cfg, int64_t* n_floats);
int  vn_codec_tensor_count(const vn_codec_cfg* cfg, int* n_tensors);
int  vn_codec_tensor_name(const vn_codec_cfg* cfg, int index, char* name, int name_len, int64_t* offset, int64_t* count);
int  vn_codec_tensor_offset(const vn_codec_cfg* cfg, const char* name, int64_t* offset, int64_t* count);
int  vn_codec_create_from_weights(vn_ctx* ctx, const vn_codec_cfg* cfg, const float* blob_dev, int direction, int B, int n,
                                  int precision, vn_codec** out);
/* audio dev f32 [B][L] (mono, L a multiple of the hop: codec.preprocess pads) -> codes dev int64 [B][n_codebooks][L / hop]   */
int  vn_dac_encode(vn_codec* codec, const float* audio_dev, int64_t* codes_dev, void* stream);
/* codes dev int64 [B][n_codebooks][T] -> audio dev f32 [B][T * hop]                                                           */
int  vn_dac_decode(vn_codec* codec, const int64_t* codes_dev, float* audio_dev, void* stream);

/* Interface._preprocess on the device (vampnet/interface.py:206-217: normalize(-24 LUFS) -> ensure_max_of_audio(1.0) -> codec.preprocess's
 * right-pad), for B mono signals already at the codec's sample rate: x dev f32 [B][T] -> y dev f32 [B][Tp] (Tp >= T, zero padded).
 * ITU-R BS.1770-4 gated integrated loudness in float64 (K-weighting biquads evaluated in parallel over 100 ms chunks through their
 * state-space form; csrc/preprocess.hip), gain to `target_lufs` (items at or below -70 LUFS keep their level), then the peak limit.
 * kw12 = the two K-weighting biquads for `sample_rate` (b1[3], a1[3], b2[3], a2[3], a[0] = 1), pow16 = the 4 x 4 state-transition matrix
 * of their cascade raised to the chunk length sample_rate / 10 (row major) — HOST pointers to float64 values the caller computes
 * (vampnet_amd/codec.py).  workspace: device scratch of vn_preprocess_workspace bytes.  lufs_out: dev f32 [B] or NULL.  Resampling and
 * the mono mix stay with the caller.  PARITY UNPINNED (audiotools is not part of the reference tree).                              */
int vn_preprocess_workspace(int B, int T, int sample_rate, int64_t* n_bytes);
int vn_preprocess_f32(vn_ctx* ctx, const float* x, float* y, int B, int T, int Tp, int sample_rate, float target_lufs,
                      const double* kw12, const double* pow16, void* workspace, float* lufs_out, void* stream);

/* Synchronises `stream` and reports whether any stream-K GEMM of this process ever hit its bounded-spin give-up
 * (results would be wrong): VN_OK or VN_ERR_HIP.  The GEMM never hangs the GPU; this is how a caller finds out.   */
int vn_health_check(vn_ctx* ctx, void* stream);

/* ---- training step ----------------------------------------------
 * Replaces scripts/exp/train.py:237-304 (`train_loop`) after the codec/masking front end: VampNet.forward in train()
 * mode (dropout at transformer.py:82, :250, :347, :367), CrossEntropyLoss(label_smoothing) over the masked targets
 * (train.py:267-278), backward, clip_grad_norm_ (train.py:296-298), torch.optim.AdamW (train.py:299) with the learning
 * rate of vampnet/scheduler.py:38-46 supplied by the host.  fp32 (conf/vampnet.yml:15 amp: false).
 *
 * Train vector = [ packed inference blob (vn_weights_size floats) | classifier weight_g | classifier weight_v ]
 * (vn_train_param_size floats; vn_train_param_offset(which = 0: g, 1: v), rows in the packed (c, p) order of VN_W_CLS_W).
 * Parameters, gradients and the two Adam moments are four caller-owned device buffers of that layout, so a data-parallel
 * job all-reduces the gradient buffer between vn_train_forward_backward and vn_train_update (world_size = number of
 * summed ranks).  Trainable: everything VampNet owns (all Linear/Conv weights incl. the shared relative-position table,
 * norms, embedding.special.MASK); NOT the codec codebook rows inside VN_W_EMB_TABLES.  (With the real loralib the five
 * LoRA'd linears would be frozen and only lora_A/B trained; loralib is absent from the reference tree, see DESIGN.md.) */
typedef struct vn_train vn_train;
typedef struct {
    float lr, beta1, beta2, eps, weight_decay;  /* AdamW (torch defaults 1e-3, 0.9, 0.999, 1e-8, 1e-2)              */
    float grad_clip;                            /* clip_grad_norm_ max norm; <= 0 disables (conf/vampnet.yml: 5.0)  */
    float label_smoothing;                      /* conf/vampnet.yml:17  0.1                                         */
    float dropout;                              /* conf/vampnet.yml:33  0.1; 0 = off                                */
    uint64_t seed;                              /* dropout stream (counter based: independent of batch sharding)    */
    int64_t step;                               /* optimiser step t >= 1: Adam bias correction + dropout stream     */
    int64_t batch_offset;                       /* global index of this rank's first batch item (dropout stream)    */
    int32_t world_size;                         /* gradient buffer holds the SUM over this many ranks (mean = / ws) */
} vn_train_params;

int  vn_train_param_size(const vn_dims* dims, int64_t* n_floats);
int  vn_train_param_offset(const vn_dims* dims, int which, int64_t* offset, int64_t* count);
/* `params`: the train vector in device memory; the model must have been created on its prefix (params == blob_dev).
 * Allocates the activation stash for the model's max_batch x max_T; no allocation happens inside a step.           */
int  vn_train_create(vn_model* model, float* params, vn_train** out);
void vn_train_destroy(vn_train* tr);
/* Re-derives what depends on the parameters (folded classifier weight, transposed GEMM weights, bias table).  Call once
 * after filling `params` and after any external change to it; vn_train_update calls it itself.                      */
int  vn_train_sync(vn_train* tr, void* stream);
/* z_masked dev int64 [B][C][T] (MASK = vocab where masked), target dev int64 [B][T*Cp] in the reference's
 * codebook_flatten order ("b c t -> b (t c)", util.py:35-40) with -100 = ignore (train.py:68).  Overwrites `grads`
 * (train vector layout) with d(loss)/d(param) and *loss_dev with the mean loss over the valid targets of THIS call.   */
int  vn_train_forward_backward(vn_train* tr, const int64_t* z_masked, const int64_t* target, int B, int T,
                               const vn_train_params* p, float* grads, float* loss_dev, void* stream);
/* The same step in pieces, for overlapping the data-parallel gradient exchange with the backward pass:
 * vn_train_forward_loss = zero `grads`, forward, loss, d(loss)/d(logits); vn_train_backward runs the backward stages
 * stage_hi >= ... >= stage_lo (n_layers = classifier + final norm, n_layers-1 .. 0 = transformer layers, -1 = embedding).
 * After stage s has run, every gradient produced by stages >= s is final EXCEPT the shared relative-position table
 * (VN_W_REL_BIAS, final after stage 0).  vn_train_forward_backward == forward_loss + backward(n_layers, -1).            */
int  vn_train_forward_loss(vn_train* tr, const int64_t* z_masked, const int64_t* target, int B, int T,
                           const vn_train_params* p, float* grads, float* loss_dev, void* stream);
int  vn_train_backward(vn_train* tr, const vn_train_params* p, float* grads, int stage_hi, int stage_lo, void* stream);
/* train()-mode forward only; logits dev f32 [B][T][Cp][vocab] (parity tests / validation with dropout = 0)          */
int  vn_train_forward(vn_train* tr, const int64_t* z_masked, int B, int T, const vn_train_params* p, float* logits,
                      void* stream);
/* val_loop (train.py:327-377): eval()-mode forward, then for every (b, t, c) row the label-smoothed CE against
 * target (dev int64 [B][T*Cp], every entry a valid token) -> row_loss dev f32 [B*T*Cp], and the number of classes whose
 * logit is strictly greater than the target's -> rank dev int32 [B*T*Cp] (accuracy top-k = rank < k, train.py:155-183).    */
int  vn_train_eval(vn_train* tr, const int64_t* z_masked, const int64_t* target, int B, int T, float label_smoothing,
                   float* row_loss, int32_t* rank, void* stream);
/* *grad_norm_dev = || grads / world_size ||_2 ; clip ; AdamW on every trainable element ; vn_train_sync.             */
/* ZeRO-1 (train.py:588-590, ZeroRedundancyOptimizer): the optimiser state is sharded over the ranks — rank r keeps Adam moments
 * for elements [lo, hi) of the train vector only.  vn_train_grad_sumsq: *sumsq_dev = sum of squares (double) of n gradient
 * elements (the ranks add theirs, the root x 1/world_size is the global norm).  vn_train_update_shard: clip by *grad_norm_dev and
 * AdamW on the trainable elements inside [lo, hi); grads_shard (SUM over ranks) / mom_shard / var_shard are indexed from lo.
 * The caller all-gathers the parameter slices and then calls vn_train_sync.                                              */
int  vn_train_grad_sumsq(vn_train* tr, const float* grads, int64_t n, double* sumsq_dev, void* stream);
int  vn_train_update_shard(vn_train* tr, const float* grads_shard, float* mom_shard, float* var_shard, const vn_train_params* p,
                           int64_t lo, int64_t hi, const float* grad_norm_dev, void* stream);
int  vn_train_update(vn_train* tr, const float* grads, float* adam_m, float* adam_v, const vn_train_params* p,
                     float* grad_norm_dev, void* stream);
/* LoRA-only fine-tuning: train.py:696 `lora.mark_only_lora_as_trainable(model)` on loralib `Linear(r=8, lora_alpha=1)`
 * (transformer.py:67-68, :109-114; loralib semantics per SURVEY.md App. C — the package is absent from the reference
 * tree, parity unpinned): y = x W^T + s (x A^T) B^T with s = alpha / r; every other parameter is frozen.
 * LoRA vector = per layer, for which in {0: w_qs, 1: w_vs, 2: fc, 3: w_1, 4: w_2}: A TRANSPOSED [K][8] then B [N][8]
 * (w_1's B rows in the packed order of VN_W_W1); vn_lora_param_offset(layer, which, ab = 0: At, 1: B).
 * vn_train_enable_lora: the model blob must hold the UN-merged base weights; they are snapshotted, the blob becomes
 * W + s B A, and from then on `grads` / `adam_m` / `adam_v` of vn_train_forward_backward / vn_train_update are buffers of
 * vn_lora_param_size floats and the update modifies `lora_params` only (then re-merges).                              */
int  vn_lora_param_size(const vn_dims* dims, int64_t* n_floats);
int  vn_lora_param_offset(const vn_dims* dims, int layer, int which, int ab, int64_t* offset, int64_t* count);
int  vn_train_enable_lora(vn_train* tr, float* lora_params, float scaling, void* stream);
/* after changing `lora_params` from outside (checkpoint load): blob <- W + s B A, then vn_train_sync                    */
int  vn_train_lora_merge(vn_train* tr, void* stream);
/* Inference-side adapter hot-swap (replaces re-running interface.py:37-46 `_load_model(ckpt, lora_ckpt)` — a whole checkpoint
 * load, merge and upload — when only the adapters change, app.py:181): blob <- base + s B A for the five LoRA'd linears of every
 * layer, on the device.  `base_blob` = the model's packed blob with the UN-merged weights (same layout; may not alias the model's
 * blob), `lora` = an adapter vector in the layout of vn_lora_param_size / _offset (absent adapters: zeros — the merge is then
 * exact, w + 0).  The model's blob (given at vn_model_create) is OVERWRITTEN in its GEMM-weight regions; the caller then rebuilds
 * the planes of the precision in use (vn_model_set_bf16x3 / vn_model_set_f16x2 / vn_model_set_bf16 — their buffers keep their
 * addresses, so captured forward graphs stay valid).  Asynchronous on `stream`.                                              */
int  vn_model_apply_lora(vn_model* m, const float* base_blob, const float* lora, float scaling, void* stream);

/* The keep-mask the kernels use at one dropout site (site 0: attention probabilities, rows = (b, h, query), cols = keys;
 * 1: attention residual, 2: GEGLU output, 3: FFN residual; rows = (b, t)); out dev u8 [rows][cols].  For parity tests. */
int  vn_dropout_keep_mask(vn_ctx* ctx, uint64_t seed, int64_t step, int layer, int site, float p, int64_t row0,
                          int64_t rows, int cols, uint8_t* out, void* stream);

/* Pure host query (no GPU needed): the relative-position bucket LUT the attention kernels index with key - query + T - 1
 * (lut_out[2 T - 1] or NULL; transformer.py:123-170) and *near_r = the half-width of the per-wave bias-gradient tables of the
 * split-plane backward: every 32 x 32 wave tile whose offsets do NOT all fall into one bucket lies within +- near_r of the diagonal. */
int  vn_attention_bwd_table_span(int T, int num_buckets, int max_distance, int32_t* lut_out, int* near_r);

/* dst [C][ldd] = transpose(src [R][C]), columns R..ldd-1 zero-filled (ldd % 4 == 0): the layout pass in front of the
 * weight-gradient GEMMs; exposed for tests and tuning.                                                              */
int  vn_transpose_f32(vn_ctx* ctx, const float* src, float* dst, int R, int C, int ldd, void* stream);

/* ---- torch's CPU random stream on the device (seeded parity mode at device speed; csrc/torch_rng.hip) ---------------
 * vn_mt19937_generate continues at::mt19937 (the engine of torch's CPU generator): state624 dev u32[624] and *pos (dev
 * int32: index of the next word to emit, 624 = block exhausted) are read and written back advanced by n words; out_raw dev
 * u32[n] receives the tempered outputs, or NULL to skip them.  The two transforms reproduce Tensor.exponential_(1) (two
 * words per element) and Tensor.uniform_(lo, hi) (one word per element) of float32 CPU tensors (transformer.py:28-30,
 * :1024-1028 consume them through multinomial / gumbel_noise_like).                                                   */
int vn_mt19937_generate(vn_ctx* ctx, uint32_t* state624, int32_t* pos, uint32_t* out_raw, int64_t n, void* stream);
/* Jump-ahead: polys dev u32 [n_targets][624] = x^J mod phi(x) per target offset J (vampnet_amd/mt_jump.py); out_states dev u32
 * [n_targets][624] receives the generator state J steps ahead of (state624, *pos), each at position 0.  vn_mt19937_generate_chunks
 * then walks n_chunks such states in parallel, chunk c writing words [c*chunk_words, min((c+1)*chunk_words, total_words)).       */
int vn_mt19937_jump(vn_ctx* ctx, const uint32_t* state624, const int32_t* pos, const uint32_t* polys, int n_targets,
                    uint32_t* out_states, void* stream);
/* Second level of a two-level plan: target b = the state poly_index[b] steps-polynomial ahead of base_states[base_index[b]] (dev u32
 * [..][624], each AT POSITION 0: the outputs of a first vn_mt19937_jump).  All the draws of a whole generate() call are then known
 * from one generator state with (steps + 1) + (chunks per step) polynomials instead of steps x chunks: one jump launch for the
 * starts of the sampling steps, one for every step's chunk starts, one walk for all chunks.                                          */
int vn_mt19937_jump_indexed(vn_ctx* ctx, const uint32_t* base_states, const int32_t* base_index, const uint32_t* polys,
                            const int32_t* poly_index, int n_targets, uint32_t* out_states, void* stream);
int vn_mt19937_generate_chunks(vn_ctx* ctx, const uint32_t* states, int n_chunks, uint32_t* out_raw, int64_t chunk_words,
                               int64_t total_words, void* stream);
int vn_torch_exponential_f32(vn_ctx* ctx, const uint32_t* raw, float* out, int64_t n, void* stream);
int vn_torch_uniform_f32(vn_ctx* ctx, const uint32_t* raw, float* out, int64_t n, float lo, float hi, void* stream);

/* Interface.build_mask on the device, RNG-exact (vampnet/interface.py:454-489, vampnet/mask.py:56-173): `raw` = the words of
 * torch's CPU mt19937 stream that the reference's draws consume, produced by vn_mt19937_generate — [0, B*C*T) linear_random's
 * bernoulli (one word per element), then the always-heads coins of periodic_mask (values unused), `roll_word` = index of the
 * randint word of the roll (-1: period == 0), `drop_word` = index of the first of the n_drop randint words of dropout.  onset:
 * optional [B][C][T] int64 mask to AND in (host-computed).  ncc / upper already normalised to [0, C].  mask: [B][C][T] int64.  */
int vn_build_mask(vn_ctx* ctx, const uint32_t* raw, const int64_t* onset, int64_t* mask, int B, int C, int T, float intensity,
                  int n_prefix, int n_suffix, int period, int width, int64_t roll_word, int64_t drop_word, int n_drop,
                  int ncc, int upper, void* stream);

/* ---- the exchange step of the batch-sharded vamp() (replaces nothing in the reference, which is single-GPU:
 * vampnet/interface.py:491-562 runs the whole batch on one device) --------------------------------------------------------------
 * The coarse and coarse-to-fine loops shard over the GPUs of a node by batch item with NO data-path collective; the one exchange is an
 * all-gather of the finished (B / world, 14, T) int64 token blocks.  RCCL is bound at run time (dlopen; VN_RCCL_LIB overrides the
 * library name), so a single-GPU host never loads it.  vn_comm_unique_id: 128 bytes made on ONE rank, passed to the others by the host
 * (vampnet_amd/interface.py broadcasts them through torch.distributed); vn_comm_create: collective over the `world` ranks, binds to
 * the context's device.  vn_allgather_tokens: every rank sends `count` int64 (equal on all ranks: the host pads the last block),
 * recv_dev = [world][count]; enqueued on `stream`, no host synchronisation.                                                       */
typedef struct vn_comm vn_comm;
int  vn_comm_unique_id(vn_ctx* ctx, uint8_t* id128);
int  vn_comm_create(vn_ctx* ctx, const uint8_t* id128, int rank, int world, vn_comm** out);
void vn_comm_destroy(vn_comm* comm);
/* what RCCL itself reports for the communicator (ncclCommCount, ncclCommUserRank): bench.py prints it as devices.rccl_nranks so that
 * a multi-GPU record shows the collective library saw N ranks, not just that N processes were started                               */
int  vn_comm_count(vn_comm* comm, int* nranks, int* rank);
int  vn_allgather_tokens(vn_comm* comm, const int64_t* send_dev, int64_t* recv_dev, int64_t count, void* stream);

/* Tuning / test hooks (vn_debug_*) are NOT part of this interface: include/vampnet_hip_debug.h.  They act on ONE vn_ctx; nothing in
 * the library is process-global, so contexts are independent of each other (each must still be driven from one stream at a time). */

#ifdef __cplusplus
}
#endif
#endif /* VAMPNET_HIP_H */
