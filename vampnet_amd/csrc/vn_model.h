// Internal view of a vn_model shared by engine.hip (inference) and train.hip (training step).
#pragma once
#include "vn_common.h"

struct vn_model {
    vn_ctx* ctx;
    vn_dims d;
    const float* blob;
    int Cp, D, H, L;
    // workspace (device)
    float *x, *y, *qkv, *g, *logits, *bias_full, *psel;
    int32_t *z, *z_sampled, *sampled, *count, *lut;
    int64_t* ksched;         // device [max_steps][max_batch] per-item mask schedule of the running generate()
    // bf16 fast mode (vn_model_set_bf16): bf16 image of the weight blob (same element offsets) + bf16 GEMM A operands
    // bf16x3 mode (vn_model_set_bf16x3): blob16 holds the THREE split planes of the blob, w_plane elements apart, and the
    // A operands are written as three planes too (y16: max_rows * D apart, g16: max_rows * 2D apart)
    const uint16_t* blob16;
    long w_plane;            // 0: single-plane bf16 fast mode
    uint16_t *y16, *g16;     // sized for three planes
    // bf16x3: the GEMM weight tensors of blob16 once more as TILED planes (tensor at blob offset o -> 3 o; gemm_x3.hip reads these)
    uint16_t* w_tiled;
    // f16x2 mode (vn_model_set_f16x2): the GEMM weight tensors as tiled f16x2 planes built from the fp32 blob (tensor at blob offset
    // o -> 2 o); while the mode is on, blob16 == w_h2 and w_plane == VN_PLANES_TILED_H2
    uint16_t* w_h2;
    // attention operands as split planes (attention_x3.hip), written by the QKV GEMM epilogues: qk16 = q then k planes
    // [3][2][max_rows * D]; vt16 = blocked V^T planes [3][H * ceil(max_rows / 32) * 64 * 32], zero-filled once.  bf16x3: three exact
    // bf16 planes; f16x2: the first two planes hold the fp16 two-plane split (q / 8, k, 16 v; second plane unscaled: |v| < 4094)
    uint16_t *qk16, *vt16;
    long qk_plane, vt_plane;
    // folded RMSNorms (split-plane precisions; engine.hip forward_i32): the consumer weights (QKV, W1, classifier) hold W (.) w_norm,
    // x16 = the split planes of the raw residual stream (written by the residual GEMMs' epilogues / vn_launch_rowprep), ssq =
    // [max_rows][D / 128] sums of squares per 128-column group of the same rows
    int folded;
    uint16_t* x16;
    float* ssq;
    int bias_T;              // T the expanded bias table is currently built for (-1 = none)
    long max_rows;
    struct vn_fwd_graphs* graphs;   // captured hipGraphs of the forward pass, per (B, T, precision) (engine.hip)
};


long vn_tensor_count(const vn_dims* d, int id);
long vn_tensor_offset(const vn_dims* d, int id, int layer);
// (re)builds the per-head relative-position bias table [H][2T-1] and the key-query -> bucket LUT for length T
int vn_model_ensure_bias(vn_model* m, int T, hipStream_t s);
