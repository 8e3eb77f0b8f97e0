// Internal declarations of the training-step kernels (train_kernels.hip, attention_train.hip) used by train.hip.
#pragma once
#include "vn_common.h"

struct vn_adamw_args {
    float lr, beta1, beta2, eps, weight_decay;
    float bc1, bc2;      // 1 - beta^t
    float gscale;        // 1 / world_size when the gradient vector holds a SUM over ranks, else 1
    float clip;          // clip_grad_norm_ max norm (<= 0: off)
};

int vn_launch_resid_dropout(vn_ctx* ctx, const float* x_in, const float* y, float* x_out, int M, int N, const vn_drop& d,
                            hipStream_t s);
// out16 (optional, needs the mask to be on and N % 32 == 0): the result once more as TILED bf16x3 planes — the A operand of the dW / dX GEMMs
int vn_launch_dropout_bwd(vn_ctx* ctx, const float* dy, float* out, int M, int N, const vn_drop& d, hipStream_t s, uint16_t* out16 = nullptr);
int vn_launch_dropout_mask(vn_ctx* ctx, uint8_t* out, long rows, int cols, const vn_drop& d, hipStream_t s);
int vn_launch_geglu_train(vn_ctx* ctx, const float* u, const float* dg, float* out, int M, int D2, const vn_drop& d,
                          bool bwd, hipStream_t s, uint16_t* out16 = nullptr);     // out16: the result also as tiled bf16x3 planes
int vn_rmsnorm_bwd_blocks(int rows);
int vn_launch_rmsnorm_bwd(vn_ctx* ctx, const float* x, const float* w, const float* dy, const float* dres, float* dx,
                          float* dw, float* partial, int rows, int D, float eps, hipStream_t s);
int vn_launch_reduce_rows(vn_ctx* ctx, const float* partial, int nb, int C, float* out, hipStream_t s);
int vn_launch_transpose(vn_ctx* ctx, const float* src, float* dst, int R, int C, int lds_, int ldd, hipStream_t s);
// fp32 -> tiled bf16x3 planes (operands of the training GEMMs on the split-plane pipe): the matrix itself / its transpose [C][Rp]
int vn_launch_split3_tiled(vn_ctx* ctx, const float* src, uint16_t* dst, int R, int K, int lds_, hipStream_t s);
int vn_launch_transpose_split3_tiled(vn_ctx* ctx, const float* src, uint16_t* dst, int R, int C, int lds_, int Rp, hipStream_t s,
                                     uint16_t* rows16 = nullptr);   // rows16: also the tiled planes of src itself (C % 32 == 0)
int vn_launch_colsum(vn_ctx* ctx, const float* src, int R, int C, float* partial, float* out, hipStream_t s);
int vn_launch_cross_entropy(vn_ctx* ctx, float* logits, const int64_t* target, int32_t* t32, long rows, int V, float ls,
                            int32_t* n_valid, float* row_loss, float* loss, hipStream_t s);
int vn_launch_weight_norm_fold(vn_ctx* ctx, const float* g, const float* v, float* W, int rows, int D, hipStream_t s);
int vn_launch_weight_norm_bwd(vn_ctx* ctx, const float* g, const float* v, const float* dW, float* dg, float* dv, int rows,
                              int D, hipStream_t s);
int vn_embed_bwd_partial_floats(int B, int T, int C, int ld, int D);
int vn_launch_embed_bwd(vn_ctx* ctx, const float* dx, const int32_t* z, const float* tables, const float* wt, float* dtables,
                        float* dwt, float* db, float* partial, float* dlat, int B, int C, int T, int V1, int ld, int D,
                        hipStream_t s);
int vn_launch_grad_sumsq(vn_ctx* ctx, const float* g, long n, double* partial, double* out, hipStream_t s);
int vn_launch_grad_norm(vn_ctx* ctx, const float* g, long n, float gscale, double* partial, float* norm_out, hipStream_t s);
int vn_launch_adamw(vn_ctx* ctx, float* p, const float* g, float* m, float* v, long n, const vn_adamw_args& a,
                    const float* norm, hipStream_t s);

// attention_train.hip
// forward with probability dropout; also writes lse[b][h][t] = log sum_k exp(score)   (transformer.py:234-254, :250)
int vn_launch_attention_train_fwd(vn_ctx* ctx, const float* q, const float* k, const float* v, const float* relbias_full,
                                  float* out, float* lse, int B, int H, int T, const vn_drop& d, hipStream_t s);
// backward: dqkv token-major [B*T][3*H*64] (columns: dq | dk | dv, head-major inside each); delta scratch [B][H][T];
// dbias_partial: this launch's slab [B*H*ceil(T/64)][64] of per-block bucket sums of d(bias) (or NULL to skip it) — no
// atomics anywhere; vn_launch_dbias_reduce adds n_slabs consecutive slabs (the layers share the table) in a fixed order
int vn_launch_attention_bwd(vn_ctx* ctx, const float* q, const float* k, const float* v, const float* relbias_full,
                            const int32_t* lut_dev, const float* out, const float* dout, const float* lse, float* delta,
                            float* dqkv, float* dbias_partial, int B, int H, int T, int nbuckets, const vn_drop& d, hipStream_t s);
// nqb = q-blocks per (b, h) of the kernel that wrote the slabs (ceil(T / 64) for the fp32 kernel, ceil(T / 128) for the split-plane one)
int vn_launch_dbias_reduce(vn_ctx* ctx, const float* partial, float* dbias, int n_slabs, long slab_floats, int B, int H, int nqb,
                           int nbuckets, bool accumulate, hipStream_t s);

// attention_x3.hip (TRAIN instantiation of the shared-tile kernel) / attention_train_x3.hip: the same forward / backward on the split-plane
// pipe.  Operands: the planes the QKV GEMM's plane epilogue writes (qk16: q planes then, n = B H T 64 elements later, k planes; vt16: V^T
// blocked by 32 global token rows).  ws: vn_attention_x3_bwd_ws_layout(...).total uint16 elements, zero-filled once.
struct vn_ax_bwd_ws { long plane_r, plane_t, off_v16, off_do16, off_qt16, off_kt16, off_dot16, total; };
void vn_attention_x3_bwd_ws_layout(int B, int H, int T, vn_ax_bwd_ws* w);
int vn_attention_x3_near_r(const int32_t* lut_host, int T);
size_t vn_attention_x3_bwd_dq_lds(int T, int near_r);
size_t vn_attention_x3_lds_bytes(int T, int key_split, int np);
int vn_launch_attention_x3_train_fwd(vn_ctx* ctx, const uint16_t* q16, const uint16_t* k16, long plane_qk, const uint16_t* vt16, long plane_vt,
                                     const float* relbias_full, float* out, float* lse, int B, int H, int T, int cus, const vn_drop& d,
                                     hipStream_t s, uint16_t* out16 = nullptr, long plane16 = 0);   // out16: also the planes of out (e.g. tiled)
int vn_launch_attention_x3_bwd(vn_ctx* ctx, const uint16_t* qk16, long plane_qk, const uint16_t* vt16, long plane_vt, uint16_t* ws,
                               const float* relbias_full, const int32_t* lut_dev, int near_r, const float* out, const float* dout,
                               const float* lse, float* delta, float* dqkv, float* dbias_partial, int B, int H, int T, int nbuckets,
                               const vn_drop& d, hipStream_t s);
// engine.hip: fp32 q, k, v [B][H][T][64] -> the attention operands' planes, exactly as the QKV GEMM's plane epilogue writes them (tests)
int vn_launch_attn_x3_prep(vn_ctx* ctx, const float* q, const float* k, const float* v, uint16_t* qk16, long plane_qk, uint16_t* vt16,
                           long plane_vt, long heads, int H, int T, hipStream_t s);

// LoRA fine-tuning helpers (rank 8; every rank-r operand is [C][8] row-major, A stored transposed)
int vn_launch_lora_down(vn_ctx* ctx, const float* Y, int ldy, const float* P, float* H, int M, int Cn, float scale, hipStream_t s);
int vn_lora_up_partial_floats(int M, int Cn);
int vn_launch_lora_up(vn_ctx* ctx, const float* Y, int ldy, const float* H, float* G, float* partial, int M, int Cn, float scale,
                      hipStream_t s);
int vn_launch_lora_merge(vn_ctx* ctx, const float* W, const float* Bm, const float* At, float* Weff, int N, int K, float scale,
                         hipStream_t s);

int vn_launch_eval_rows(vn_ctx* ctx, const float* logits, const int64_t* target, long rows, int V, float ls, float* row_loss,
                        int32_t* rank, hipStream_t s);
