"""Host-side mirror of the reference `VampNet` inference API over libvampnet_hip.so.

`VampNetModel` offers what `Interface` needs from `vampnet.modules.transformer.VampNet`
(reference transformer.py:535-946): `n_codebooks`, `n_conditioning_codebooks`, `mask_token`,
`chunk_size_s`, `generate(...)` with the same keyword names and defaults, plus `forward_codes`
(= embedding.from_codes + forward) for parity tests.  All arithmetic happens in the HIP library;
torch only owns device memory and the stream.  Nothing here imports `oracle/`.
"""
import ctypes as C
import math
import random
import warnings

import numpy as np
import torch

from . import _lib
from ._lib import VnError, vn_dims, vn_sample_params

LORA_SCALING = 1.0 / 8.0      # loralib: lora_alpha / r = 1 / 8 (transformer.py:22 LORA_R)

# The engine's default precision: "bf16x3" — every GEMM / attention operand as three bf16 planes whose sum IS the fp32 value (8 + 8 + 8
# significand bits, fp32's exponent range), six matrix-core products, fp32 accumulation: arithmetic that is not narrower than the
# reference's fp32.  "f16x2" (two fp16 planes: 22 significand bits, fp16's exponent range) is an OPT-IN fast
# mode guarded by the saturation ledger below.
DEFAULT_PRECISION = "bf16x3"


class PrecisionFallbackWarning(UserWarning):
    """precision='f16x2' could not hold a weight / an activation in fp16's range; the work was (re)done on 'bf16x3'."""


def seed_all(seed: int):
    """audiotools.util.seed as the reference calls it at transformer.py:711-712: reseeds the GLOBAL RNGs."""
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)


class Engine:
    """One vn_ctx per device."""

    def __init__(self, device="cuda:0"):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise VnError("no GPU visible: the vampnet_amd engine has no CPU path")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise VnError(f"device must be a ROCm GPU, got {device}")
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        h = C.c_void_p()
        rc = self.lib.vn_ctx_create(idx, C.byref(h))
        if rc != 0:
            raise VnError(f"vn_ctx_create({idx}) failed with status {rc}")
        self.handle = h

    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def check(self, rc, what):
        _lib.check(rc, self.handle, what)

    def version(self):
        return self.lib.vn_version().decode()

    def torch_rng(self):
        """Device-resident continuation of torch's CPU generator (rng="torch_device"), one per engine."""
        if getattr(self, "_torch_rng", None) is None:
            from .torch_rng import DeviceTorchRng
            self._torch_rng = DeviceTorchRng(self)
        return self._torch_rng

    def saturation(self, clear=True):
        """The context's saturation ledger (include/vampnet_hip.h vn_saturation_flags): (GEMM-operand planes, attention operands,
        weight planes, reserved) — non-zero where an fp16 plane writer had to clamp a value since the last clear.  Synchronises.
        The device words are per CONTEXT; flags that one consumer set aside for another (saturation_restore) are OR-ed in."""
        fl = (C.c_uint32 * 4)()
        self.check(self.lib.vn_saturation_flags(self.handle, fl, 1 if clear else 0, self.stream()), "vn_saturation_flags")
        carry = getattr(self, "_sat_carry", (0, 0, 0, 0))
        out = tuple(int(v) | int(c) for v, c in zip(fl, carry))
        if clear:
            self._sat_carry = (0, 0, 0, 0)
        return out

    def saturation_restore(self, flags):
        """Hand flags back to the ledger that a nested consumer read-and-cleared but does not own: a codec that shares the engine of an
        f16x2 model runs INSIDE that model's generate(return_signal=True) — it sets the model's words aside before its own call and
        restores them afterwards, so the model's check still sees them (and the codec does not fall back on the model's account)."""
        carry = getattr(self, "_sat_carry", (0, 0, 0, 0))
        self._sat_carry = tuple(int(c) | (1 if f else 0) for c, f in zip(carry, flags))

    def health_check(self):
        """Synchronise and raise if a stream-K GEMM ever gave up waiting for a partial tile (never expected)."""
        self.check(self.lib.vn_health_check(self.handle, self.stream()), "vn_health_check")

    def profile_begin(self, max_launches=20000, stride=1):
        """Bracket (a hash-selected 1/stride sample of) the MFMA launches with hipEvents until profile_end()."""
        self.check(self.lib.vn_profile_set_stride(self.handle, stride), "vn_profile_set_stride")
        self.check(self.lib.vn_profile_begin(self.handle, max_launches), "vn_profile_begin")

    def profile_end(self):
        """{'gemm': (launches, ms, flops, bytes), 'attention': (...), ...} since profile_begin.  The codec's convolutions come in three
        classes by what bounds them (csrc/vn_common.h VN_PROF_*): 'conv_x3' / 'conv_f32' = matrix-pipe-bound layers on the split-plane
        pipe / on the fp32-input MFMA, 'conv_hbm' = layers whose operand bytes bound them (audio-rate and k = 1 layers); 'conv1d' = all."""
        st = (C.c_double * 24)()
        self.check(self.lib.vn_profile_end(self.handle, st), "vn_profile_end")
        out = {"gemm": tuple(st[0:4]), "attention": tuple(st[4:8]), "conv_x3": tuple(st[8:12]), "gemm_bf16": tuple(st[12:16]),
               "conv_f32": tuple(st[16:20]), "conv_hbm": tuple(st[20:24])}
        out["conv1d"] = tuple(sum(v) for v in zip(out["conv_x3"], out["conv_f32"], out["conv_hbm"]))
        return out

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.vn_ctx_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ---- single kernels (tests / profiling) ---------------------------------------------------
    def rmsnorm(self, x, w, eps=1e-6):
        x = x.contiguous()
        y = torch.empty_like(x)
        rows = x.numel() // x.shape[-1]
        self.check(self.lib.vn_rmsnorm_f32(self.handle, x.data_ptr(), w.data_ptr(), y.data_ptr(), rows,
                                           x.shape[-1], eps, self.stream()), "vn_rmsnorm_f32")
        return y

    def gemm(self, a, w, bias=None, epilogue=_lib.EPI_STORE, out=None):
        """out (op)= a @ w.T ; a [M,K], w [N,K] (torch F.linear)."""
        M, K = a.shape
        N = w.shape[0]
        if out is None:
            out = torch.empty(M, N // 2 if epilogue == _lib.EPI_GEGLU else N, device=a.device, dtype=torch.float32)
        self.check(self.lib.vn_gemm_f32(self.handle, a.data_ptr(), w.data_ptr(),
                                        bias.data_ptr() if bias is not None else None, out.data_ptr(),
                                        M, N, K, epilogue, self.stream()), "vn_gemm_f32")
        return out

    def gemm_bf16(self, a16, w16, bias=None, epilogue=_lib.EPI_STORE, out=None):
        """fast-mode GEMM: a16 [M,K], w16 [N,K] torch.bfloat16 -> fp32 out (op)= a @ w.T"""
        M, K = a16.shape
        N = w16.shape[0]
        if out is None:
            out = torch.empty(M, N, device=a16.device, dtype=torch.float32)
        self.check(self.lib.vn_gemm_bf16(self.handle, a16.data_ptr(), w16.data_ptr(),
                                         bias.data_ptr() if bias is not None else None, out.data_ptr(), M, N, K, epilogue,
                                         self.stream()), "vn_gemm_bf16")
        return out

    def split3(self, x):
        """fp32 tensor -> bf16 [3, *x.shape] with planes summing to x exactly (operand format of gemm_bf16x3)."""
        x = x.contiguous()
        n = x.numel()
        out = torch.empty((3,) + tuple(x.shape), dtype=torch.bfloat16, device=x.device)
        self.check(self.lib.vn_split3_f32(self.handle, x.data_ptr(), out.data_ptr(), n, n, self.stream()), "vn_split3_f32")
        return out

    @staticmethod
    def tile3(planes):
        """[3, R, K] split planes -> the tiled layout [ceil(R/16), K/32, 3, 16, 32] of gemm_x3.hip (rows zero-padded to 16)."""
        _, R, K = planes.shape
        R16 = (R + 15) // 16 * 16
        if R16 != R:
            planes = torch.cat([planes, planes.new_zeros(3, R16 - R, K)], dim=1)
        return planes.reshape(3, R16 // 16, 16, K // 32, 32).permute(1, 3, 0, 2, 4).contiguous()

    def gemm_bf16x3(self, a3, w3, bias=None, epilogue=_lib.EPI_STORE, out=None, tiled_shape=None):
        """fp32-grade GEMM on the bf16 matrix cores: a3 [3,M,K], w3 [3,N,K] split planes -> fp32 out (op)= a @ w.T.
        tiled_shape=(M, N, K): a3 / w3 are tile3() images instead."""
        if tiled_shape is not None:
            M, N, K = tiled_shape
            if out is None:
                out = torch.empty(M, N // 2 if epilogue == _lib.EPI_GEGLU else N, device=a3.device, dtype=torch.float32)
            self.check(self.lib.vn_gemm_bf16x3(self.handle, a3.data_ptr(), -1, w3.data_ptr(), -1,
                                               bias.data_ptr() if bias is not None else None, out.data_ptr(), M, N, K, epilogue,
                                               self.stream()), "vn_gemm_bf16x3")
            return out
        _, M, K = a3.shape
        N = w3.shape[1]
        if out is None:
            out = torch.empty(M, N // 2 if epilogue == _lib.EPI_GEGLU else N, device=a3.device, dtype=torch.float32)
        self.check(self.lib.vn_gemm_bf16x3(self.handle, a3.data_ptr(), M * K, w3.data_ptr(), N * K,
                                           bias.data_ptr() if bias is not None else None, out.data_ptr(), M, N, K, epilogue,
                                           self.stream()), "vn_gemm_bf16x3")
        return out

    def gemm_bf16x3_tn(self, at_tiled, wt_tiled, M, N, tokens, out=None):
        """out [M, N] = At^T Wt with both operands token-major: at_tiled / wt_tiled = tile3() images of the split planes of At [tokens, M] /
        Wt [tokens, N] (the layout the activations of the model path already have) — the dW GEMM of training, no transposing pass."""
        if out is None:
            out = torch.empty(M, N, device=at_tiled.device, dtype=torch.float32)
        self.check(self.lib.vn_gemm_bf16x3_tn(self.handle, at_tiled.data_ptr(), wt_tiled.data_ptr(), out.data_ptr(), M, N, tokens,
                                              self.stream()), "vn_gemm_bf16x3_tn")
        return out

    def split2h(self, x, tiled=False):
        """fp32 [R, K] -> the f16x2 operand format of gemm_f16x2: float16 [2, R, K] (h0 = fp16(x), h1 = fp16((x - h0) * 2048)), or with
        tiled=True the tiled image [ceil(R/16), K/32, 2, 16, 32] (rows zero-padded to 16)."""
        x = x.contiguous()
        R, K = x.shape
        if tiled:
            R16 = (R + 15) // 16 * 16
            if R16 != R:
                x = torch.cat([x, x.new_zeros(R16 - R, K)], dim=0)
            out = torch.empty(R16 // 16, K // 32, 2, 16, 32, dtype=torch.float16, device=x.device)
            self.check(self.lib.vn_split2_f16(self.handle, x.data_ptr(), out.data_ptr(), R16, K, 0, 1, self.stream()), "vn_split2_f16")
            return out
        out = torch.empty(2, R, K, dtype=torch.float16, device=x.device)
        self.check(self.lib.vn_split2_f16(self.handle, x.data_ptr(), out.data_ptr(), R, K, R * K, 0, self.stream()), "vn_split2_f16")
        return out

    def gemm_f16x2(self, a2, w2, bias=None, epilogue=_lib.EPI_STORE, out=None, tiled_shape=None):
        """fp32-grade GEMM as three fp16 matrix-core products: a2 [2,M,K], w2 [2,N,K] (split2h) -> fp32 out (op)= a @ w.T.
        tiled_shape=(M, N, K): a2 / w2 are split2h(tiled=True) images instead."""
        if tiled_shape is not None:
            M, N, K = tiled_shape
            ap = wp = -1
        else:
            _, M, K = a2.shape
            N = w2.shape[1]
            ap, wp = M * K, N * K
        if out is None:
            out = torch.empty(M, N // 2 if epilogue == _lib.EPI_GEGLU else N, device=a2.device, dtype=torch.float32)
        self.check(self.lib.vn_gemm_f16x2(self.handle, a2.data_ptr(), ap, w2.data_ptr(), wp,
                                          bias.data_ptr() if bias is not None else None, out.data_ptr(), M, N, K, epilogue,
                                          self.stream()), "vn_gemm_f16x2")
        return out

    def attention(self, q, k, v, rel_bias, num_buckets=32, max_distance=128, precision="f32"):
        """q,k,v [B,H,T,64]; rel_bias [num_buckets,H] -> [B,T,H*64].  precision "f32": fp32-input MFMA (attention_f32.hip);
        "bf16x3": six bf16-MFMA products of exact operand splits (attention_x3.hip), same error class; "f16x2": three fp16-MFMA
        products of two-plane splits (the attention format of the f16x2 precision), same error class."""
        B, H, T, dh = q.shape
        out = torch.empty(B, T, H * dh, device=q.device, dtype=torch.float32)
        fn = {"f32": self.lib.vn_attention_f32, "bf16x3": self.lib.vn_attention_bf16x3, "f16x2": self.lib.vn_attention_f16x2}[precision]
        q, k, v, rel_bias = (t.contiguous() for t in (q, k, v, rel_bias))    # held in locals: a temporary's block is gone before the call runs
        self.check(fn(self.handle, q.data_ptr(), k.data_ptr(), v.data_ptr(),
                      rel_bias.data_ptr(), out.data_ptr(), B, H, T, num_buckets, max_distance, self.stream()),
                   "vn_attention_" + precision)
        return out


def _merge_lora(sd, key):
    """W_eff = W + (lora_B @ lora_A) * alpha / r for a lora.Linear in eval mode."""
    w = sd[key + ".weight"].float()
    a, b = sd.get(key + ".lora_A"), sd.get(key + ".lora_B")
    if a is not None and b is not None:
        w = w + (b.float() @ a.float()) * (1.0 / a.shape[0])     # loralib scaling = lora_alpha (1) / r, r = rows of lora_A
    return w


def pack_weights(lib, dims: vn_dims, sd: dict, codebooks: torch.Tensor, merge_lora: bool = True) -> torch.Tensor:
    """Builds the packed fp32 blob (layout: include/vampnet_hip.h) on the host from a reference-format
    state_dict (key names: SURVEY.md App. B) and the codec codebooks [>=C, vocab, latent].  `merge_lora=False` keeps the
    base weights un-merged (LoRA fine-tuning snapshots them and merges on the device)."""
    if not merge_lora:
        sd = {k: v for k, v in sd.items() if "lora_" not in k}
    n = C.c_int64()
    if lib.vn_weights_size(C.byref(dims), C.byref(n)) != 0:
        raise VnError("vn_weights_size rejected the dims (need d_model == 64 * n_heads, vocab 1024)")
    blob = torch.zeros(n.value, dtype=torch.float32)

    def put(tid, layer, t):
        off, cnt = C.c_int64(), C.c_int64()
        if lib.vn_weights_offset(C.byref(dims), tid, layer, C.byref(off), C.byref(cnt)) != 0:
            raise VnError(f"vn_weights_offset({tid},{layer}) failed")
        t = t.contiguous().float().reshape(-1)
        if t.numel() != cnt.value:
            raise VnError(f"tensor {tid} layer {layer}: expected {cnt.value} values, got {t.numel()}")
        blob[off.value:off.value + cnt.value] = t

    Cn, nc, V, D = dims.n_codebooks, dims.n_cond, dims.vocab, dims.d_model
    Cp = Cn - nc
    mask_rows = sd["embedding.special.MASK"].float()                       # [C, latent]
    tables = torch.cat([codebooks[:Cn].float(), mask_rows[:, None, :]], dim=1)   # [C, V+1, latent]
    put(_lib.W_EMB_TABLES, 0, tables)
    put(_lib.W_EMB_WT, 0, sd["embedding.out_proj.weight"].float().squeeze(-1).t())
    put(_lib.W_EMB_B, 0, sd["embedding.out_proj.bias"])
    put(_lib.W_REL_BIAS, 0, sd["transformer.layers.0.self_attn.relative_attention_bias.weight"])
    put(_lib.W_FINAL_NORM, 0, sd["transformer.norm.weight"])
    # classifier: fold old-style weight_norm (w = g * v / ||v||), reorder rows (p c) -> (c p)
    if "classifier.layers.0.weight_v" in sd:
        wc = torch._weight_norm(sd["classifier.layers.0.weight_v"].float(), sd["classifier.layers.0.weight_g"].float(), 0)
    else:   # new-style parametrization or already-folded weight
        wc = sd["classifier.layers.0.weight"].float()
    wc = wc.reshape(V, Cp, D).permute(1, 0, 2)
    put(_lib.W_CLS_W, 0, wc)
    put(_lib.W_CLS_B, 0, sd["classifier.layers.0.bias"].float().reshape(V, Cp).t())
    for l in range(dims.n_layers):
        p = f"transformer.layers.{l}."
        put(_lib.W_NORM1, l, sd[p + "norm_1.weight"])
        put(_lib.W_QKV, l, torch.cat([_merge_lora(sd, p + "self_attn.w_qs"), sd[p + "self_attn.w_ks.weight"].float(),
                                      _merge_lora(sd, p + "self_attn.w_vs")], dim=0))
        put(_lib.W_WO, l, _merge_lora(sd, p + "self_attn.fc"))
        put(_lib.W_NORM3, l, sd[p + "norm_3.weight"])
        w1 = _merge_lora(sd, p + "feed_forward.w_1")
        val, gate = w1[:2 * D].reshape(2 * D // 32, 32, D), w1[2 * D:].reshape(2 * D // 32, 32, D)
        put(_lib.W_W1, l, torch.stack([val, gate], dim=1))
        put(_lib.W_W2, l, _merge_lora(sd, p + "feed_forward.w_2"))
    return blob


LORA_KEYS = ("self_attn.w_qs", "self_attn.w_vs", "self_attn.fc", "feed_forward.w_1", "feed_forward.w_2")     # transformer.py:67-68, :109-114
LORA_R = 8                                                                                                     # transformer.py:22


def pack_lora_vector(lib, dims: vn_dims, sd: dict) -> torch.Tensor:
    """loralib tensors of a state_dict (name.lora_A (r, in), name.lora_B (out, r)) -> the engine's adapter vector (include/vampnet_hip.h
    vn_lora_param_size / _offset: per layer and LoRA'd linear A TRANSPOSED [in][8] then B [out][8], w_1's B rows in the packed order
    of VN_W_W1).  Linears without adapters in `sd` stay zero: their merge is w + 0."""
    n = C.c_int64()
    if lib.vn_lora_param_size(C.byref(dims), C.byref(n)) != 0:
        raise VnError("vn_lora_param_size rejected the dims")
    out = torch.zeros(n.value, dtype=torch.float32)
    D = dims.d_model
    val = torch.arange(2 * D).view(2 * D // 32, 32)
    w1_perm = torch.stack([val, val + 2 * D], 1).reshape(-1)
    off, cnt = C.c_int64(), C.c_int64()
    for l in range(dims.n_layers):
        for w, key in enumerate(LORA_KEYS):
            name = f"transformer.layers.{l}.{key}"
            a, b = sd.get(name + ".lora_A"), sd.get(name + ".lora_B")
            if a is None or b is None:
                continue
            if a.shape[0] != LORA_R or b.shape[1] != LORA_R:
                raise VnError(f"{name}: adapters of rank {a.shape[0]} (the engine merges rank-{LORA_R} adapters, transformer.py:22)")
            a, b = a.float(), b.float()
            if key == "feed_forward.w_1":
                b = b[w1_perm]
            for ab, t in ((0, a.t().contiguous()), (1, b.contiguous())):
                if lib.vn_lora_param_offset(C.byref(dims), l, w, ab, C.byref(off), C.byref(cnt)) != 0 or t.numel() != cnt.value:
                    raise VnError(f"{name}: adapter shape {tuple(t.shape)} does not fit the model")
                out[off.value:off.value + cnt.value] = t.reshape(-1)
    return out


def draw_noise_host(B, N, V, steps, sample_cutoff, b0=0, nb=None, pin=False):
    """torch-CPU noise ledger of one generate() call for a global batch B; returns the rows of items
    [b0, b0+nb): exp [steps, nb*N, V] (zeros on non-sampling steps), unif [steps, nb, N]."""
    nb = B if nb is None else nb
    exp = torch.zeros(steps, nb * N, V, dtype=torch.float32)
    unif = torch.empty(steps, nb, N, dtype=torch.float32)
    if pin:
        exp, unif = exp.pin_memory(), unif.pin_memory()
    whole = (b0 == 0 and nb == B)
    for i in range(steps):
        if (i / steps) <= sample_cutoff:                      # transformer.py:852-855
            if whole:
                exp[i].exponential_(1)
            else:
                exp[i].copy_(torch.empty(B * N, V).exponential_(1)[b0 * N:(b0 + nb) * N])
        if whole:
            unif[i].uniform_(1e-20, 1)
        else:
            unif[i].copy_(torch.zeros(B, N).uniform_(1e-20, 1)[b0:b0 + nb])
    return exp, unif


class VampNetModel:
    """Device-resident VampNet (one of coarse / c2f)."""

    def __init__(self, engine: Engine, sd: dict, codebooks: torch.Tensor, *, n_heads, n_layers, n_codebooks,
                 n_conditioning_codebooks=0, latent_dim=8, embedding_dim=1280, vocab_size=1024,
                 max_batch=8, max_T=575, chunk_size_s=10, precision=DEFAULT_PRECISION, _blob=None, **_ignored):
        self.engine = engine
        self.lib = engine.lib
        self.n_heads, self.n_layers = n_heads, n_layers
        self.n_codebooks = n_codebooks
        self.n_conditioning_codebooks = n_conditioning_codebooks
        self.n_predict_codebooks = n_codebooks - n_conditioning_codebooks
        self.latent_dim, self.embedding_dim, self.vocab_size = latent_dim, embedding_dim, vocab_size
        self.mask_token = vocab_size                     # transformer.py:576
        self.chunk_size_s = chunk_size_s
        self.dims = vn_dims(n_layers, n_heads, embedding_dim, n_codebooks, n_conditioning_codebooks, vocab_size,
                            latent_dim, 32, 128, 1e-6, max_batch, max_T)
        self.blob_base = None        # the packed blob with UN-merged weights, once adapters are (or have been) applied on the device
        lora_vec = None
        if _blob is not None:        # vampnet_amd.train.Trainer: the model lives on the prefix of the train vector
            self.blob = _blob
        elif any(".lora_" in k for k in sd):
            # loralib adapters (interface.py:37-46 + the eval-mode merge): W + (B A) / r is formed ON THE DEVICE from the un-merged
            # blob — the same kernel a later adapter swap (apply_lora) runs, so a swapped model and a freshly loaded one hold the
            # same bits
            self.blob_base = pack_weights(self.lib, self.dims, sd, codebooks, merge_lora=False).to(engine.device)
            self.blob = self.blob_base.clone()
            lora_vec = pack_lora_vector(self.lib, self.dims, sd)
        else:
            self.blob = pack_weights(self.lib, self.dims, sd, codebooks).to(engine.device)   # must outlive the vn_model
        h = C.c_void_p()
        engine.check(self.lib.vn_model_create(engine.handle, C.byref(self.dims), self.blob.data_ptr(), C.byref(h)),
                     "vn_model_create")
        self.handle = h
        self.blob16 = None
        self.blob3 = None
        self.precision = "f32"
        if lora_vec is not None:
            self._merge_lora_vector(lora_vec)
        self.set_precision(precision)

    def _merge_lora_vector(self, vec: torch.Tensor):
        """blob <- blob_base + (B A) / r on the device (vn_model_apply_lora); planes are NOT refreshed here"""
        vec = vec.to(self.device, non_blocking=False)
        self.engine.check(self.lib.vn_model_apply_lora(self.handle, self.blob_base.data_ptr(), vec.data_ptr(), 1.0 / LORA_R,
                                                       self.engine.stream()), "vn_model_apply_lora")
        torch.cuda.current_stream(self.device).synchronize()       # `vec` is dropped on return

    def apply_lora(self, lora_sd: dict = None):
        """Adapter hot-swap on a RESIDENT model (the reference reloads the whole checkpoint for this: interface.py:27-50 via
        app.py:181): merge the loralib adapters of `lora_sd` (name.lora_A / name.lora_B; None or {} = no adapters) into the un-merged
        weights on the device and rebuild the planes of the precision in use — tens of milliseconds instead of a checkpoint load,
        pack and upload.  Every buffer the model points at keeps its address (the fp32 blob is rewritten in place, the bf16x3 planes
        and the single-plane bf16 image are re-filled in place), so nothing dangles and captured forward graphs stay valid.  Memory:
        the first swap keeps one extra copy of the fp32 blob (`blob_base`, the un-merged weights: 1.3 GB for the coarse model)."""
        if self.blob_base is None:
            self.blob_base = self.blob.clone()              # no adapters merged so far: the blob IS the base
        vec = pack_lora_vector(self.lib, self.dims, lora_sd or {})
        self._merge_lora_vector(vec)
        self._has_adapters = bool(lora_sd)
        self._refresh_planes()

    def drop_adapters(self):
        """back to the plain checkpoint (what Interface.reload returns: interface.py:146-174 loads without lora_ckpt); no-op for a model
        that never carried adapters or has none merged right now"""
        if self.blob_base is not None and getattr(self, "_has_adapters", True):
            self.apply_lora({})

    def _refresh_planes(self):
        """the fp32 blob changed in place: rebuild the 16-bit images of the precision in use"""
        want = self.precision
        if want == "bf16" and self.blob16 is not None:
            self.blob16.copy_(self.blob)                    # in place: the model keeps pointing at this tensor (RNE, like .to(bfloat16))
        if want == "bf16x3" and self.blob3 is not None:
            n = self.blob.numel()
            self.engine.check(self.lib.vn_split3_f32(self.engine.handle, self.blob.data_ptr(), self.blob3.data_ptr(), n, n,
                                                     self.engine.stream()), "vn_split3_f32")
        if want != "f32":
            self.set_precision(want)

    def set_precision(self, precision: str):
        """"f32": exact-fp32 MFMA.  "bf16x3" (default): fp32-grade GEMMs evaluated as six bf16 MFMA products of exact three-way
        operand splits (same accuracy class as "f32", faster matrix pipe).  "f16x2" (opt-in fast mode, operands NARROWER than fp32):
        GEMMs as three fp16 MFMA products of two-plane splits (operand error 2^-22; half of bf16x3's matrix time; fp16's range).  A
        model whose weights do not fit fp16, or whose probe forward puts a value on the saturation ledger, is NOT run in it: the
        model falls back to "bf16x3" with a PrecisionFallbackWarning; every later generate() re-checks the ledger (see generate).
        "bf16": fast mode — GEMM operands in bf16 like the reference's own GPU path (torch.autocast(bf16), interface.py:364,428);
        NOT bit-exact."""
        if precision == "bf16":
            if self.blob16 is None:
                self.blob16 = self.blob.to(torch.bfloat16)          # same element offsets, RNE like torch autocast
            self.engine.check(self.lib.vn_model_set_bf16(self.handle, self.blob16.data_ptr()), "vn_model_set_bf16")
        elif precision == "bf16x3":
            # fp32-grade GEMMs on the bf16 matrix cores: every operand as three exact split planes (gemm_x3.hip)
            n = self.blob.numel()
            if self.blob3 is None:
                self.blob3 = torch.empty(3 * n, dtype=torch.bfloat16, device=self.device)
                self.engine.check(self.lib.vn_split3_f32(self.engine.handle, self.blob.data_ptr(), self.blob3.data_ptr(), n, n,
                                                         self.engine.stream()), "vn_split3_f32")
            self.engine.check(self.lib.vn_model_set_bf16x3(self.handle, self.blob3.data_ptr(), n), "vn_model_set_bf16x3")
        elif precision == "f16x2":
            # fp32-grade GEMMs as three fp16 matrix-core products of two-plane operand splits (gemm_x3.hip); the engine builds the
            # weight planes from the fp32 blob itself
            self.engine.saturation(clear=True)              # start from a clean ledger (synchronises: a setup call)
            self.engine.check(self.lib.vn_model_set_f16x2(self.handle, 1), "vn_model_set_f16x2")
            sat = self.engine.saturation(clear=True)        # word 2: a weight that was clamped while the planes were built
            why = f"a weight does not fit fp16 (saturation ledger {sat})" if any(sat) else self._probe_f16x2()
            if why:
                self._fall_back(why)
                return
        elif precision == "f32":
            self.engine.check(self.lib.vn_model_set_bf16(self.handle, None), "vn_model_set_bf16")
        else:
            raise ValueError("precision must be 'f32', 'f16x2', 'bf16x3' or 'bf16'")
        self.precision = precision

    def _fall_back(self, why: str):
        """precision "f16x2" cannot be trusted for this model / call: say so and move the model to "bf16x3" for good."""
        warnings.warn(f"precision='f16x2': {why}; this model runs on 'bf16x3' from here on", PrecisionFallbackWarning, stacklevel=3)
        self.set_precision("bf16x3")

    def _probe_f16x2(self):
        """An early warning at precision selection: one short forward on random codes, then the saturation ledger.  Activations depend
        on the input as well, so passing the probe proves nothing about later calls — those are covered by the ledger check after
        every generate() / forward_codes() in this precision.  Returns a reason string when the probe saturated, else None.
        VN_F16X2_PROBE=0 skips the probe."""
        import os
        if os.environ.get("VN_F16X2_PROBE", "1") == "0":
            return None
        T = max(1, min(int(self.dims.max_T), 96))
        g = torch.Generator().manual_seed(0)
        codes = torch.randint(0, self.vocab_size, (1, self.n_codebooks, T), generator=g)
        codes[:, self.n_conditioning_codebooks:, ::2] = self.vocab_size                 # MASK tokens, as inside generate()
        self._ledger_off = True                                                         # the probe reads the ledger itself
        try:
            logits = self.forward_codes(codes, layout="native")
        finally:
            self._ledger_off = False
        sat = self.engine.saturation(clear=True)
        if any(sat):
            return f"a probe forward left fp16's range (saturation ledger: operands {sat[0]}, attention {sat[1]})"
        if not bool(torch.isfinite(logits).all()):
            return "a probe forward produced non-finite logits"
        return None

    @property
    def device(self):
        return self.engine.device

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.vn_model_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ---- forward ------------------------------------------------------------------------------
    def forward_codes(self, codes: torch.Tensor, layout="reference"):
        """embedding.from_codes + forward.  codes int64 [B, C, T] (MASK = vocab).
        layout="native": [B, T, Cp, V]; "reference": [B, V, T*Cp] (transformer.py:634)."""
        B, Cn, T = codes.shape
        assert Cn == self.n_codebooks
        codes = codes.to(self.device, torch.int64).contiguous()
        logits = torch.empty(B, T, self.n_predict_codebooks, self.vocab_size, device=self.device, dtype=torch.float32)
        self.engine.check(self.lib.vn_forward(self.handle, codes.data_ptr(), B, T, logits.data_ptr(),
                                              self.engine.stream()), "vn_forward")
        if self.precision == "f16x2" and not getattr(self, "_ledger_off", False):
            sat = self.engine.saturation(clear=True)
            if any(sat):                    # a clamped value never reaches the caller: repeat the forward on bf16x3
                self._fall_back(f"a forward left fp16's range (saturation ledger: operands {sat[0]}, attention {sat[1]})")
                self.engine.check(self.lib.vn_forward(self.handle, codes.data_ptr(), B, T, logits.data_ptr(),
                                                      self.engine.stream()), "vn_forward")
        if layout == "native":
            return logits
        return logits.reshape(B, T * self.n_predict_codebooks, self.vocab_size).permute(0, 2, 1)

    # ---- sampling ------------------------------------------------------------------------------
    def _params(self, steps, temperature, mask_temperature, sample_cutoff, top_p, n0_override, seed, batch_offset=0,
                call_batch=0, global_batch=0):
        return vn_sample_params(int(steps), float(temperature), float(mask_temperature), float(sample_cutoff),
                                float(top_p) if top_p is not None else 0.0,
                                int(n0_override) if n0_override is not None else -1,
                                int(seed) & 0xFFFFFFFFFFFFFFFF, int(batch_offset), int(call_batch), int(global_batch), None)

    @staticmethod
    def mask_schedule(steps: int, n0: int):
        """floor(gamma((i+1)/steps) * N0) per step with torch's own fp32 ops (mask.py:8-9, transformer.py:831-834,903)
        so the integer schedule is bit-identical to the reference's."""
        n0_t = torch.tensor(int(n0))
        out = []
        for i in range(steps):
            r = torch.tensor((i + 1) / steps)
            g = (r * torch.pi / 2).cos().clamp(1e-10, 1.0)
            out.append(int(torch.floor(g * n0_t).long()))
        return out

    def draw_noise(self, B, T, steps, sample_cutoff, batch_offset=0, local_batch=None, pin=True, on_device=False):
        """Replays the reference's torch-CPU draw order: per step an Exp(1) tensor of shape
        (B*T*Cp, V) iff that step samples (multinomial), then U(1e-20, 1) of shape (B, T*Cp).  B is the GLOBAL
        batch; when this rank owns items [batch_offset, batch_offset + local_batch) only those rows are kept
        (every rank draws the identical global stream).  `on_device`: the same numbers produced on the
        GPU by continuing torch's mt19937 stream there (vampnet_amd/torch_rng.py) instead of drawing them on the host."""
        N = T * self.n_predict_codebooks
        V = self.vocab_size
        nb = B if local_batch is None else local_batch
        b0 = batch_offset
        if on_device:
            from .torch_rng import draw_noise_device
            return draw_noise_device(self.engine.torch_rng(), B, N, V, steps, sample_cutoff, b0, nb)
        return draw_noise_host(B, N, V, steps, sample_cutoff, b0, nb, pin)

    @torch.inference_mode()
    def generate(self, codec=None, time_steps: int = 300, _sampling_steps: int = 12, start_tokens=None,
                 temperature: float = 1.0, mask=None, mask_temperature: float = 10.5, ctrls=None, ctrl_masks=None,
                 typical_filtering=True, typical_mass=0.15, typical_min_tokens=64, top_p=None, seed: int = None,
                 sample_cutoff: float = 1.0, return_signal=True, debug=False, causal_weight: float = 0.0,
                 cfg_scale: float = 3.0, cfg_guidance: float = None, cond=None,
                 rng: str = "torch", n0_override: int = None, device_seed: int = None,
                 global_batch: int = None, batch_offset: int = 0, noise=None, call_batch: int = None):
        """Drop-in for VampNet.generate (transformer.py:686-946): see _generate for the keywords.  In precision "f16x2" the call is
        followed by a read of the context's saturation ledger (one stream synchronisation); if any fp16 plane writer had to clamp a
        value during the call, its tokens are DISCARDED, the model moves to "bf16x3" (PrecisionFallbackWarning) and the call is
        repeated from the same generator state — a saturated result never reaches the caller."""
        kw = dict(locals())
        kw.pop("self")
        if self.precision != "f16x2":
            return self._generate(**kw)
        snap = torch.get_rng_state()                       # what a seed-less call draws its noise / device seed from
        out = self._generate(**kw)
        sat = self.engine.saturation(clear=True)
        if any(sat):
            self._fall_back(f"generate() left fp16's range (saturation ledger: operands {sat[0]}, attention {sat[1]}, weights {sat[2]})")
            torch.set_rng_state(snap)
            out = self._generate(**kw)
        return out

    @torch.inference_mode()
    def _generate(self, codec=None, time_steps: int = 300, _sampling_steps: int = 12, start_tokens=None,
                  temperature: float = 1.0, mask=None, mask_temperature: float = 10.5, ctrls=None, ctrl_masks=None,
                  typical_filtering=True, typical_mass=0.15, typical_min_tokens=64, top_p=None, seed: int = None,
                  sample_cutoff: float = 1.0, return_signal=True, debug=False, causal_weight: float = 0.0,
                  cfg_scale: float = 3.0, cfg_guidance: float = None, cond=None,
                  rng: str = "torch", n0_override: int = None, device_seed: int = None,
                  global_batch: int = None, batch_offset: int = 0, noise=None, call_batch: int = None):
        """VampNet.generate (transformer.py:686-946).  Extra keywords (not in the reference):
          rng="torch"  : parity mode — noise is drawn from torch's CPU generator in the reference's order;
          rng="torch_device" : the same stream, continued on the GPU (mt19937 + the two distribution transforms run in
                         HIP; torch's generator is advanced as if the host had drawn): seed-exact at device speed;
          rng="device" : fast mode — Philox stream on the GPU (seeded by `device_seed` or the torch generator);
          n0_override  : the global batch's masked-token count when this call sees a shard;
          global_batch / batch_offset : size of the global batch and index of this shard's first item, so that
                         both RNG modes draw the noise the unsharded call would;
          n0_override may be a per-item list and `noise` a pre-drawn (exp, unif) ledger when the caller batches
                         items of several reference generate() calls (Interface.coarse_to_fine's chunks).
        `typical_filtering/typical_mass/typical_min_tokens` are accepted and have no effect, exactly like the
        reference (transformer.py:989-993 discards the filter's result).
        `cfg_guidance` (transformer.py:771-783, :845-847, :940): like the reference, the batch is doubled with an all-MASK copy
        (every codebook; mask all ones) after N0 was counted on the caller's items, all 2B items are sampled — consuming 2B items'
        worth of noise per step — and the first B are returned.  The reference computes the guided logits into a local it never
        uses (:845-847), so the logits that are sampled are the plain ones; the VALUE of cfg_guidance has no effect there or here
        (pinned against the reference: tests/test_oracle_vs_reference.py).  Needs max_batch >= 2B; not offered for sharded calls."""
        if ctrls is not None:
            raise NotImplementedError("ctrls are never used by Interface and the shipped models have no control inputs (SURVEY.md App. A.3)")
        if return_signal and not hasattr(codec, "decode_signal"):
            raise ValueError("return_signal=True (the reference's default, transformer.py:704) decodes the sampled tokens through "
                             "`codec` (transformer.py:943-944): pass a codec that can decode, or return_signal=False for tokens")
        if seed is not None:
            seed_all(seed)
        z = start_tokens
        if z is None:
            raise ValueError("start_tokens is required (the reference crashes on None too: transformer.py:731)")
        z = z.to(self.device, torch.int64).contiguous()
        B, Cn, T = z.shape
        if mask is None:
            mask = torch.ones_like(z)
            mask[:, :self.n_conditioning_codebooks, :] = 0
        mask = mask.to(self.device)
        if mask.ndim == 2:
            mask = mask[:, None, :].repeat(1, Cn, 1)
        mask = (mask != 0).to(torch.int64).contiguous()
        steps = int(_sampling_steps)
        nb_ret = B
        if cfg_guidance is not None:
            if n0_override is not None or noise is not None or call_batch or (global_batch or B) != B or batch_offset:
                raise NotImplementedError("cfg_guidance on a sharded / batched-call generate(): the reference's doubled batch puts the "
                                          "all-MASK copies of ALL items behind the last real item, which a shard cannot see")
            if 2 * B > self.dims.max_batch:
                raise ValueError(f"cfg_guidance doubles the batch (transformer.py:771-783): {2 * B} items exceed this model's "
                                 f"workspace (max_batch = {self.dims.max_batch})")
            n0_override = int(((mask != 0) | (z == self.mask_token)).sum().item())      # :766 counted BEFORE the doubling
            z = torch.cat([z, torch.full_like(z, self.mask_token)], dim=0).contiguous()
            mask = torch.cat([mask, torch.ones_like(mask)], dim=0).contiguous()
            B = 2 * B
        if n0_override is None:
            n0 = int(((mask != 0) | (z == self.mask_token)).sum().item())   # transformer.py:762-766, batch-wide
            n0_items = [n0] * B
        elif isinstance(n0_override, (list, tuple)):                         # items from different reference calls
            n0_items = [int(v) for v in n0_override]
            assert len(n0_items) == B
            n0 = n0_items[0]
        else:
            n0 = int(n0_override)
            n0_items = [n0] * B
        per_n0 = {v: self.mask_schedule(steps, v) for v in set(n0_items)}
        sched = (C.c_int64 * (steps * B))(*[per_n0[n0_items[b]][i] for i in range(steps) for b in range(B)])
        step_events = None
        if rng in ("torch", "torch_device"):
            if noise is not None:
                exp, unif = noise                                  # pre-drawn ledger [steps, B*N, V], [steps, B, N]
            elif rng == "torch_device":
                # produce the ledger on a side stream, step by step, and let the sampling loop wait per step: the serial
                # mt19937 walk (one CU) runs underneath the transformer forward of the earlier steps
                from .torch_rng import draw_noise_device
                exp, unif, step_events = draw_noise_device(self.engine.torch_rng(), global_batch or B,
                                                           T * self.n_predict_codebooks, self.vocab_size, steps, sample_cutoff,
                                                           batch_offset, B, overlap=True)
            else:
                exp, unif = self.draw_noise(global_batch or B, T, steps, sample_cutoff, batch_offset, B)
            noise_on_host = not (exp.is_cuda and unif.is_cuda)
            if noise_on_host:
                exp = exp.to(self.device, non_blocking=True)
                unif = unif.to(self.device, non_blocking=True)
            else:
                cur_s = torch.cuda.current_stream(self.device)
                exp.record_stream(cur_s)
                unif.record_stream(cur_s)
            exp_p, unif_p = exp.data_ptr(), unif.data_ptr()
            dseed = 0
        elif rng == "device":
            exp = unif = None
            exp_p = unif_p = None
            noise_on_host = False
            dseed = device_seed if device_seed is not None else int(torch.randint(0, 2 ** 62, (1,)).item())
        else:
            raise ValueError("rng must be 'torch', 'torch_device' or 'device'")
        params = self._params(steps, temperature, mask_temperature, sample_cutoff, top_p, n0, dseed, batch_offset,
                              call_batch or 0, (global_batch or 0) if call_batch else 0)
        if step_events is not None:
            ev_arr = (C.c_void_p * steps)(*[C.c_void_p(e.cuda_event) for e in step_events])
            params.step_events = C.cast(ev_arr, C.POINTER(C.c_void_p))
        out = torch.empty_like(z)
        self.engine.check(self.lib.vn_generate(self.handle, z.data_ptr(), mask.data_ptr(), B, T, C.byref(params),
                                               sched, exp_p, unif_p, out.data_ptr(), self.engine.stream()),
                          "vn_generate")
        if step_events is not None:
            self.engine.torch_rng().store_to_torch()        # waits for the side stream only; the model keeps running
        if exp is not None and (noise_on_host or not exp.is_cuda):
            # host-drawn noise travels through pinned staging buffers: keep them alive until the enqueued work has consumed them.
            # Device-drawn ledgers (rng="torch_device") are ordinary stream-ordered tensors (record_stream'd on this stream by their
            # producer): no host synchronisation — the caller goes on to enqueue the next stage (and its noise) right away
            torch.cuda.current_stream(self.device).synchronize()
        if cfg_guidance is not None:
            out = out[:nb_ret].contiguous()                                   # transformer.py:940-941
        return self.decode(out, codec) if return_signal else out

    @torch.inference_mode()
    def decode(self, z, codec):
        """VampNet.decode (transformer.py:661-684): MASK -> 0, codebook rows -> quantizer.from_latents -> codec.decode ->
        AudioSignal at codec.sample_rate.  (The reference then looks for time steps whose codebooks are ALL the mask token to
        silence them, but it searches the tensor it has just cleared of mask tokens — :668 vs :678-682 — so nothing is ever
        silenced; same here.)"""
        assert z.ndim == 3
        z = z.masked_fill(z == self.mask_token, 0)
        return codec.decode_signal(z)

    generate_batched_calls = True       # marker: generate() accepts per-item n0_override + a pre-drawn noise ledger

    @torch.inference_mode()
    def sample_step(self, z_masked, logits_native, step, steps, n0, *, temperature=1.0, mask_temperature=10.5,
                    sample_cutoff=1.0, top_p=None, exp_noise=None, unif_noise=None, device_seed=0):
        """One step of the sampling logic on given logits (teacher-forced parity tests).
        Returns (z_masked_next, sampled)."""
        z = z_masked.to(self.device, torch.int64).contiguous().clone()
        B, Cn, T = z.shape
        sampled = torch.empty_like(z)
        logits_native = logits_native.contiguous()             # held in a local: the kernels read (and top-p edits) it
        params = self._params(steps, temperature, mask_temperature, sample_cutoff, top_p, n0, device_seed)
        k = (C.c_int64 * B)(*([self.mask_schedule(steps, n0)[step]] * B))
        self.engine.check(self.lib.vn_sample_step(
            self.handle, z.data_ptr(), logits_native.data_ptr(), B, T, step, C.byref(params), k,
            exp_noise.data_ptr() if exp_noise is not None else None,
            unif_noise.data_ptr() if unif_noise is not None else None, sampled.data_ptr(), self.engine.stream()),
            "vn_sample_step")
        return z, sampled
