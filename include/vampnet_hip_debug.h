/* vampnet_hip_debug.h — tuning and test hooks of libvampnet_hip.so.  NOT part of the drop-in boundary (include/vampnet_hip.h):
 * nothing here is needed to run the path, and the ablation settings produce INVALID results by design.
 *
 * Every hook takes the vn_ctx it acts on and changes that context only (state: vn_ctx::tune in csrc/vn_common.h; defaults are
 * read from the environment when the context is created).  A setter bumps the context's tuning epoch, so forward graphs that a
 * model captured under the previous setting are captured again instead of replaying the old kernels.
 */
#ifndef VAMPNET_HIP_DEBUG_H
#define VAMPNET_HIP_DEBUG_H
#include "vampnet_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* number of forward passes of this model that were served by replaying a captured hipGraph (tests)               */
int vn_debug_graph_replays(const vn_model* model, int64_t* count);

/* fp32-input MFMA GEMM (gemm_f32.hip; scripts/gemm_sweep.py): force the block tile (bm x bn in {128,64}^2; 0,0 = automatic) and,
 * with order >= 0, bits [0] tile walk (0 column-major, 1 grouped 8-row patches), [2:1] scheduler + 1 (0 auto, 1 data-parallel,
 * 2 stream-K), [15:8] stream-K start stagger + 1; order = -1 keeps walk / scheduler                                  */
int vn_debug_gemm_config(vn_ctx* ctx, int bm, int bn, int order);
/* bf16x3 GEMM (gemm_x3.hip): bm = tile height 128 / 192 / 256 (0 = by shape; the GEGLU epilogue has no 192-row form and takes 128
 * then); splitk 0 / 1 off, 2 / 4 forced, -1 = cost model; abl = ablation bits (tuning; results INVALID), <= 0 = none       */
int vn_debug_x3_config(vn_ctx* ctx, int bm, int splitk, int abl);
/* bf16x3 models of this context: 1 / 0 = always / never take the split-plane attention path (QKV GEMM with the plane epilogue +
 * attention_x3.hip), -1 = whenever its LDS image fits (default; VN_ATTN_X3)                                               */
int vn_debug_attention_x3_force(vn_ctx* ctx, int on);
/* a RESIDUAL bf16x3 GEMM that is split along K runs the RMSNorm that follows it in the layer inside its reduce pass
 * (vn_splitk_reduce_rmsnorm_kernel); 0 = keep the two kernels apart (A/B tests: both forms are bitwise equal), 1 = fuse,
 * -1 = VN_X3_FUSE_NORM / default (on)                                                                               */
int vn_debug_x3_fuse_norm(vn_ctx* ctx, int on);
/* the two forms as single ops (tests): x[rows][D] += sum of partial[s][rows][D], y16 = three split planes (plane16 elements
 * apart) of RMSNorm(x) with weight w; fused != 0 = one kernel, 0 = reduce kernel then norm kernel                      */
int vn_debug_splitk_reduce_rmsnorm(vn_ctx* ctx, const float* partial, int nsplit, float* x, const float* w, void* y16,
                                   int64_t plane16, int rows, int D, float eps, int fused, void* stream);
/* average duration (us) of `iters` launches of the bf16x3 attention kernel alone (planes prepared outside the timed region);
 * iters < 0: -iters launches on the f16x2 precision's fp16 two-plane operands                                                  */
int vn_debug_attention_x3_time(vn_ctx* ctx, const float* q, const float* k, const float* v, const float* rel_bias, float* out,
                               int B, int H, int T, int iters, float* avg_us, void* stream);
/* bf16x3 attention (attention_x3.hip; scripts/attn_probe.py): split = work decomposition (-1 by shape, 0 = 128-query blocks that
 * share their K / V^T tiles, 1 / 2 / 4 = that many key-split waves per 32-query block); lds_bytes = dynamic-LDS override of the
 * shared-tile kernel (0 = natural; sets the blocks per CU); stagger = [15:0] start delay per SIMD wave slot in units of 64 cycles, bit 16 = tail blocks keep the default wave priority (< 0 = the
 * context's default); trace_dev = uint32 [blocks][8] per-block phase-cycle sums + entry / exit time of the shared-tile kernel (NULL = none)          */
int vn_debug_attention_x3_config(vn_ctx* ctx, int split, int lds_bytes, int stagger, void* trace_dev);


/* ---- single-op test entries that ALLOCATE AND SYNCHRONISE (moved here in round 6: the boundary header promises "no allocation
 * after *_create, nothing synchronises the device") ---------------------------------------------------------------------------
 * Each call allocates its scratch (expanded bias table, operand planes, backward workspace) with the library's allocator, runs the
 * op and waits for it (hipStreamSynchronize), frees the scratch.  For parity tests and tuning only.                               */
/* The same op in the bf16x3 precision (attention_x3.hip: both products as six bf16-MFMA products of exact three-way operand
 * splits, fp32 softmax): same arguments and output as vn_attention_f32.                                                  */
int vn_attention_bf16x3(vn_ctx* ctx, const float* q, const float* k, const float* v, const float* rel_bias,
                        float* out, int B, int H, int T, int num_buckets, int max_distance, void* stream);
/* ... and in the f16x2 precision: q / 8, k and 16 v as fp16 two-plane splits (h0 = fp16(x), h1 = fp16(x - h0)), the softmax weights
 * (times 16) split the same way in registers, THREE fp16-MFMA products per step into the one accumulator (attention_x3.hip, NP = 2). */
int vn_attention_f16x2(vn_ctx* ctx, const float* q, const float* k, const float* v, const float* rel_bias,
                       float* out, int B, int H, int T, int num_buckets, int max_distance, void* stream);

/* Self-attention core (transformer.py:229-254): q,k,v dev f32 [B][H][T][64];
 * rel_bias dev f32 [num_buckets][H]; out dev f32 [B][T][H*64].                                */
int vn_attention_f32(vn_ctx* ctx, const float* q, const float* k, const float* v, const float* rel_bias,
                     float* out, int B, int H, int T, int num_buckets, int max_distance, void* stream);
/* fast-mode variant: bf16 MFMA products, fp32 softmax; out16 = bf16 [B][T][H*64] (NOT bit-exact)                      */
int vn_attention_bf16(vn_ctx* ctx, const float* q, const float* k, const float* v, const float* rel_bias,
                      void* out16, int B, int H, int T, int num_buckets, int max_distance, void* stream);

/* Training attention as a single op (tests / tuning): forward with probability dropout (keep-mask = site 0, layer 0,
 * step 1 of vn_dropout_keep_mask) writing out [B][T][H*64] and lse [B][H][T]; when `dout` is non-NULL also the backward:
 * dqkv [B*T][3*H*64] (dq | dk | dv, head-major inside each third) and dbias [num_buckets][H] ACCUMULATED into (fixed order).
 * Synchronous.  (transformer.py:234-254 and its autograd.)                                                          */
int  vn_attention_train_f32(vn_ctx* ctx, const float* q, const float* k, const float* v, const float* rel_bias,
                            float* out, float* lse, const float* dout, float* dqkv, float* dbias, int B, int H, int T,
                            int num_buckets, int max_distance, float dropout, uint64_t seed, void* stream);
/* The same op on the split-plane pipe (what the training step runs when its GEMMs do: six bf16-MFMA products of exact three-way
 * operand splits per product; the same dropout stream and the same deterministic bias gradient).                         */
int  vn_attention_train_bf16x3(vn_ctx* ctx, const float* q, const float* k, const float* v, const float* rel_bias,
                               float* out, float* lse, const float* dout, float* dqkv, float* dbias, int B, int H, int T,
                               int num_buckets, int max_distance, float dropout, uint64_t seed, void* stream);

/* ---- guard-page harness (csrc/devmem.hip; tests/test_gpu_guard.py) -------------------------------------------------------------
 * Every device allocation of the library goes through one allocator.  In guard mode a buffer is its own virtual-memory reservation
 * whose mapped pages are flanked by UNMAPPED granules: mode 1 ("end") puts the buffer's last 16-byte unit against the unmapped
 * granule behind it, mode 2 ("start") its first byte on the first byte of its first page.  An access outside the buffer is then a
 * GPU memory fault that ends the process, instead of a silent read of a neighbour.  VN_GUARD_ALLOC=end|start sets the mode at the
 * first allocation; vn_guard_mode sets it explicitly (0 = plain hipMalloc).  The modes hold for allocations made afterwards. */
int  vn_guard_mode(int mode);
/* returns the current mode; allocations = guard blocks ever made, live = mapped now (count and bytes incl. granule rounding)      */
int  vn_guard_stats(int64_t* allocations, int64_t* live, int64_t* live_bytes);
int  vn_guard_alloc(int64_t bytes, int mode, void** out);
int  vn_guard_free(void* p);
/* the harness's own known-answer test: one lane reads (write = 0, into *sink) or writes the 32-bit word at p + offset_bytes and the
 * call waits for it.  Past the end of a mode-1 block this must kill the process.                                                  */
int  vn_guard_poke(void* p, int64_t offset_bytes, int write, void* sink, void* stream);
/* the same allocator in torch.cuda.memory.CUDAPluggableAllocator's signature: every torch tensor of the process a guard block      */
void *vn_guard_torch_alloc(long size, int device, void* stream);
void vn_guard_torch_free(void* p, long size, int device, void* stream);

/* Training step: run the layers' weight-gradient GEMMs on the trainer's SIDE stream (1; the default when the trainer was created
 * with VN_TRAIN_OVERLAP unset or 1) or in the caller's stream (0); -1 = back to the state at creation.  For A/B runs and for
 * bracketing kernels without concurrency (bench.py's `roofline_serial`); call it between steps only.  Returns the state in effect. */
int  vn_debug_train_overlap(vn_train* t, int on);

#ifdef __cplusplus
}
#endif
#endif /* VAMPNET_HIP_DEBUG_H */
