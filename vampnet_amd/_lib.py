"""ctypes binding of libvampnet_hip.so (include/vampnet_hip.h).  No CPU fallback: if the HIP library is
missing or fails to load, importing/using the engine raises — the product path never degrades to torch."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VN_LIB") or os.path.join(_HERE, "libvampnet_hip.so")   # VN_LIB: A/B builds (tuning only)


class VnError(RuntimeError):
    pass


class vn_dims(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("n_heads", C.c_int32), ("d_model", C.c_int32),
                ("n_codebooks", C.c_int32), ("n_cond", C.c_int32), ("vocab", C.c_int32),
                ("latent_dim", C.c_int32), ("num_buckets", C.c_int32), ("max_distance", C.c_int32),
                ("eps", C.c_float), ("max_batch", C.c_int32), ("max_T", C.c_int32)]


class vn_sample_params(C.Structure):
    _fields_ = [("steps", C.c_int32), ("temperature", C.c_float), ("mask_temperature", C.c_float),
                ("sample_cutoff", C.c_double), ("top_p", C.c_float), ("n0_override", C.c_int64),
                ("seed", C.c_uint64), ("batch_offset", C.c_int64), ("call_batch", C.c_int32),
                ("global_batch", C.c_int32), ("step_events", C.POINTER(C.c_void_p))]


class vn_codec_op(C.Structure):
    _fields_ = [("kind", C.c_int32), ("i", C.c_int32 * 16), ("l", C.c_int64 * 2), ("f", C.c_float * 2), ("p", C.c_void_p * 12)]


class vn_codec_cfg(C.Structure):
    _fields_ = [("encoder_dim", C.c_int32), ("n_rates", C.c_int32), ("encoder_rates", C.c_int32 * 8), ("decoder_dim", C.c_int32),
                ("decoder_rates", C.c_int32 * 8), ("n_codebooks", C.c_int32), ("codebook_size", C.c_int32), ("codebook_dim", C.c_int32),
                ("latent_dim", C.c_int32)]


class vn_train_params(C.Structure):
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("grad_clip", C.c_float), ("label_smoothing", C.c_float),
                ("dropout", C.c_float), ("seed", C.c_uint64), ("step", C.c_int64), ("batch_offset", C.c_int64),
                ("world_size", C.c_int32)]


# tensor ids of the packed weight blob (enum in vampnet_hip.h)
(W_EMB_TABLES, W_EMB_WT, W_EMB_B, W_REL_BIAS, W_FINAL_NORM, W_CLS_W, W_CLS_B,
 W_NORM1, W_QKV, W_WO, W_NORM3, W_W1, W_W2) = range(13)

EPI_STORE, EPI_BIAS, EPI_RESIDUAL, EPI_GEGLU = range(4)

# every symbol include/vampnet_hip.h declares: (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "vn_ctx_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "vn_ctx_destroy": (None, [_P]),
    "vn_last_error": (C.c_char_p, [_P]),
    "vn_version": (C.c_char_p, []),
    "vn_profile_begin": (C.c_int, [_P, C.c_int]),
    "vn_profile_set_stride": (C.c_int, [_P, C.c_int]),
    "vn_profile_end": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "vn_weights_size": (C.c_int, [C.POINTER(vn_dims), C.POINTER(C.c_int64)]),
    "vn_weights_offset": (C.c_int, [C.POINTER(vn_dims), C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "vn_model_create": (C.c_int, [_P, C.POINTER(vn_dims), _P, C.POINTER(_P)]),
    "vn_model_destroy": (None, [_P]),
    "vn_model_set_bf16": (C.c_int, [_P, _P]),
    "vn_model_set_bf16x3": (C.c_int, [_P, _P, C.c_int64]),
    "vn_model_set_f16x2": (C.c_int, [_P, C.c_int]),
    "vn_saturation_flags": (C.c_int, [_P, C.POINTER(C.c_uint32), C.c_int, _P]),
    "vn_split3_f32": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, _P]),
    "vn_gemm_bf16x3": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "vn_gemm_bf16x3_tn": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "vn_split2_f16": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, C.c_int64, C.c_int, _P]),
    "vn_gemm_f16x2": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "vn_gemm_bf16": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "vn_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P]),
    "vn_generate": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.POINTER(vn_sample_params), C.POINTER(C.c_int64),
                              _P, _P, _P, _P]),
    "vn_sample_step": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.POINTER(vn_sample_params), C.POINTER(C.c_int64),
                                 _P, _P, _P, _P]),
    "vn_rmsnorm_f32": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_float, _P]),
    "vn_gemm_f32": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "vn_conv1d_f32": (C.c_int, [_P] * 9 + [C.c_int64] + [C.c_int] * 13 + [_P]),
    "vn_conv1d_bf16x3": (C.c_int, [_P, _P, C.c_int64] + [_P] * 7 + [C.c_int64] + [C.c_int] * 13 + [_P]),
    "vn_conv1d_f16x2": (C.c_int, [_P, _P, C.c_int64] + [_P] * 7 + [C.c_int64] + [C.c_int] * 13 + [_P]),
    "vn_tile_planes_bf16x3": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, C.c_int, _P]),
    "vn_dac_conv_in_f32": (C.c_int, [_P] * 7 + [C.c_int] * 3 + [_P]),
    "vn_dac_conv_out_f32": (C.c_int, [_P, _P, _P, C.c_float, _P, C.c_int, C.c_int, C.c_int, _P]),
    "vn_rvq_encode_f32": (C.c_int, [_P] * 8 + [C.c_int] * 5 + [_P]),
    "vn_rvq_decode_f32": (C.c_int, [_P] * 6 + [C.c_int] * 5 + [_P]),
    "vn_health_check": (C.c_int, [_P, _P]),
    "vn_train_param_size": (C.c_int, [C.POINTER(vn_dims), C.POINTER(C.c_int64)]),
    "vn_train_param_offset": (C.c_int, [C.POINTER(vn_dims), C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "vn_train_create": (C.c_int, [_P, _P, C.POINTER(_P)]),
    "vn_train_destroy": (None, [_P]),
    "vn_train_sync": (C.c_int, [_P, _P]),
    "vn_debug_train_overlap": (C.c_int, [_P, C.c_int]),
    "vn_train_forward_backward": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.POINTER(vn_train_params), _P, _P, _P]),
    "vn_train_forward_loss": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.POINTER(vn_train_params), _P, _P, _P]),
    "vn_train_backward": (C.c_int, [_P, C.POINTER(vn_train_params), _P, C.c_int, C.c_int, _P]),
    "vn_train_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(vn_train_params), _P, _P]),
    "vn_train_eval": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_float, _P, _P, _P]),
    "vn_train_grad_sumsq": (C.c_int, [_P, _P, C.c_int64, _P, _P]),
    "vn_train_update_shard": (C.c_int, [_P, _P, _P, _P, C.POINTER(vn_train_params), C.c_int64, C.c_int64, _P, _P]),
    "vn_train_update": (C.c_int, [_P, _P, _P, _P, C.POINTER(vn_train_params), _P, _P]),
    "vn_lora_param_size": (C.c_int, [C.POINTER(vn_dims), C.POINTER(C.c_int64)]),
    "vn_lora_param_offset": (C.c_int, [C.POINTER(vn_dims), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "vn_train_enable_lora": (C.c_int, [_P, _P, C.c_float, _P]),
    "vn_train_lora_merge": (C.c_int, [_P, _P]),
    "vn_model_apply_lora": (C.c_int, [_P, _P, _P, C.c_float, _P]),
    "vn_dropout_keep_mask": (C.c_int, [_P, C.c_uint64, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_int64, C.c_int64,
                                       C.c_int, _P, _P]),
    "vn_attention_train_f32": (C.c_int, [_P] * 10 + [C.c_int] * 5 + [C.c_float, C.c_uint64, _P]),
    "vn_attention_train_bf16x3": (C.c_int, [_P] * 10 + [C.c_int] * 5 + [C.c_float, C.c_uint64, _P]),
    "vn_transpose_f32": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "vn_mt19937_generate": (C.c_int, [_P, _P, _P, _P, C.c_int64, _P]),
    "vn_mt19937_jump": (C.c_int, [_P, _P, _P, _P, C.c_int, _P, _P]),
    "vn_mt19937_jump_indexed": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, _P, _P]),
    "vn_mt19937_generate_chunks": (C.c_int, [_P, _P, C.c_int, _P, C.c_int64, C.c_int64, _P]),
    "vn_torch_exponential_f32": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "vn_torch_uniform_f32": (C.c_int, [_P, _P, _P, C.c_int64, C.c_float, C.c_float, _P]),
    "vn_build_mask": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64,
                               C.c_int64, C.c_int, C.c_int, C.c_int, _P]),
    "vn_codec_create": (C.c_int, [_P, C.POINTER(vn_codec_op), C.c_int, C.c_int, C.POINTER(_P)]),
    "vn_codec_destroy": (None, [_P]),
    "vn_codec_weights_size": (C.c_int, [C.POINTER(vn_codec_cfg), C.POINTER(C.c_int64)]),
    "vn_codec_tensor_count": (C.c_int, [C.POINTER(vn_codec_cfg), C.POINTER(C.c_int)]),
    "vn_codec_tensor_name": (C.c_int, [C.POINTER(vn_codec_cfg), C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "vn_codec_tensor_offset": (C.c_int, [C.POINTER(vn_codec_cfg), C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "vn_codec_create_from_weights": (C.c_int, [_P, C.POINTER(vn_codec_cfg), _P, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "vn_dac_encode": (C.c_int, [_P, _P, _P, _P]),
    "vn_dac_decode": (C.c_int, [_P, _P, _P, _P]),
    "vn_preprocess_workspace": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "vn_preprocess_f32": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    _P, _P, _P]),
    "vn_comm_unique_id": (C.c_int, [_P, _P]),
    "vn_comm_create": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(_P)]),
    "vn_comm_destroy": (None, [_P]),
    "vn_allgather_tokens": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "vn_comm_count": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vn_debug_graph_replays": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "vn_debug_gemm_config": (C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    "vn_debug_attention_x3_time": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), _P]),
    "vn_debug_x3_config": (C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    "vn_debug_x3_fuse_norm": (C.c_int, [_P, C.c_int]),
    "vn_attention_bwd_table_span": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, _P]),
    "vn_debug_attention_x3_force": (C.c_int, [_P, C.c_int]),
    "vn_debug_splitk_reduce_rmsnorm": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_int, _P]),
    "vn_debug_attention_x3_config": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P]),
    "vn_attention_bf16": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "vn_attention_bf16x3": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "vn_attention_f16x2": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "vn_attention_f32": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "vn_guard_mode": (C.c_int, [C.c_int]),
    "vn_guard_stats": (C.c_int, [C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "vn_guard_alloc": (C.c_int, [C.c_int64, C.c_int, C.POINTER(_P)]),
    "vn_guard_free": (C.c_int, [_P]),
    "vn_guard_poke": (C.c_int, [_P, C.c_int64, C.c_int, _P, _P]),
    "vn_guard_torch_alloc": (_P, [C.c_long, C.c_int, _P]),
    "vn_guard_torch_free": (None, [_P, C.c_long, C.c_int, _P]),
}

_lib = None


def load():
    """dlopen the engine; raises VnError (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VnError(f"{LIB_PATH} not found: build it with `python -m vampnet_amd.build` "
                      "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise VnError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise VnError(f"{LIB_PATH} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc, ctx=None, what=""):
    if rc != 0:
        msg = load().vn_last_error(ctx).decode() if ctx else ""
        raise VnError(f"{what} failed with status {rc}: {msg}")
