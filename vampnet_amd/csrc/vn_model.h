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
    const uint16_t* blob16;
    uint16_t *y16, *g16;
    int bias_T;              // T the expanded bias table is currently built for (-1 = none)
    long max_rows;
    struct vn_fwd_graphs* graphs;   // captured hipGraphs of the forward pass, per (B, T, precision) (engine.hip)
};


long vn_tensor_count(const vn_dims* d, int id);
long vn_tensor_offset(const vn_dims* d, int id, int layer);
// (re)builds the per-head relative-position bias table [H][2T-1] and the key-query -> bucket LUT for length T
int vn_model_ensure_bias(vn_model* m, int T, hipStream_t s);
