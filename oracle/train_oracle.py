"""CPU ORACLE for one VampNet training step — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates scripts/exp/train.py:237-304 (`train_loop`) of the reference with torch-CPU fp32 ops and
torch.autograd for the backward pass, on top of the forward restatement in vampnet_oracle.py:

    mask   = pmask.random(z, r) ; codebook_unmask ; apply_mask            (train.py:252-254, mask.py:40-54)
    z_hat  = model(embedding.from_codes(z_mask))  in train() mode         (train.py:261-265; dropout sites
             transformer.py:69/82 (FFN), :117/250 (attention probabilities), :312/347/367 (both residual branches))
    loss   = CrossEntropyLoss(label_smoothing=0.1)(z_hat, target.masked_fill(~mask, -100))   (train.py:267-278,
             conf/vampnet.yml:17)
    backward ; clip_grad_norm_(5.0) ; AdamW(lr from NoamScheduler) ; zero_grad ; scheduler.step()  (train.py:287-301,
             conf/vampnet.yml:19-22, vampnet/scheduler.py:38-46)

Dropout noise is INJECTED (a dict of {0,1} keep-masks) so the HIP engine and this oracle can be compared on the same
draw: `y = x * keep * (1/(1-p))`, which is what torch's dropout computes.

Trainable set: with the reference imported through oracle/ref_shim.py, `loralib.Linear` is a plain `nn.Linear`
(loralib is absent), so every VampNet parameter is trainable — that is the PINNED mode.
LoRA-only fine-tuning (train.py:696 `lora.mark_only_lora_as_trainable`, adapters on the five `lora.Linear`s of
transformer.py:67-68,109-114) is restated from loralib's PUBLISHED algorithm (`lora_linear`, `add_lora`,
`loss_and_grads(only_lora=True)`): PARITY UNPINNED, the dependency is not in the reference tree and no reference test
holds vectors for it.  The codec codebooks are not VampNet parameters (train.py:257) and get no gradient.

Pinned by tests/test_oracle_vs_reference.py::test_train_step_vs_reference (reference modules + torch.optim.AdamW +
clip_grad_norm_ + the reference's NoamScheduler) when /root/reference is present.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import vampnet_oracle as vo

IGNORE_INDEX = -100          # train.py:68

DROPOUT_SITES = ("attn", "res1", "ffn", "res2")


def dropout_shapes(dims, B, T):
    """Shape of the keep-mask at each dropout site of one layer (reference tensor layouts)."""
    D, H = dims["d_model"], dims["n_heads"]
    return {"attn": (H, B, T, T), "res1": (B, T, D), "ffn": (B, T, 2 * D), "res2": (B, T, D)}


def draw_dropout_masks(dims, B, T, p, generator):
    """Host-side bernoulli(1-p) keep-masks for every layer/site (used when the test does not take them from the device)."""
    out = {}
    for l in range(dims["n_layers"]):
        for site, shp in dropout_shapes(dims, B, T).items():
            out[(l, site)] = (torch.rand(shp, generator=generator) >= p).float()
    return out


def _drop(x, masks, key, p):
    if masks is None or p == 0.0:
        return x
    return x * masks[key] * (1.0 / (1.0 - p))


LORA_SCALING = 1.0 / 8.0      # loralib: lora_alpha / r with lora_alpha = 1 (default), r = LORA_R = 8 (transformer.py:22)


def lora_linear(x, params, key):
    """loralib.Linear.forward in train() mode [UNVERIFIED-DEP: loralib is absent from the reference tree; published
    semantics, SURVEY.md App. C]:  x W^T + (lora_dropout(x) A^T B^T) * scaling, lora_dropout = identity (p = 0 default).
    Falls back to a plain linear when the state_dict holds no adapters for `key`."""
    y = F.linear(x, params[key + ".weight"])
    if key + ".lora_A" in params:
        y = y + (x @ params[key + ".lora_A"].transpose(0, 1) @ params[key + ".lora_B"].transpose(0, 1)) * LORA_SCALING
    return y


def attention_train(x, params, pre, n_heads, position_bias, masks, layer, p):
    """transformer.py:211-257 with the probability dropout of :250 active."""
    B, T, D = x.shape
    dh = D // n_heads

    def heads(y):
        return y.view(B, T, n_heads, dh).permute(2, 0, 1, 3)

    q = heads(lora_linear(x, params, pre + "self_attn.w_qs"))
    k = heads(F.linear(x, params[pre + "self_attn.w_ks.weight"]))            # plain nn.Linear (transformer.py:110)
    v = heads(lora_linear(x, params, pre + "self_attn.w_vs"))
    attn = torch.einsum("hblk,hbtk->hblt", [q, k]) / np.sqrt(q.shape[-1])
    attn = attn + position_bias
    attn = torch.softmax(attn, dim=3)
    attn = _drop(attn, masks, (layer, "attn"), p)
    out = torch.einsum("hblt,hbtv->hblv", [attn, v])
    out = out.permute(1, 2, 0, 3).reshape(B, T, D)
    return lora_linear(out, params, pre + "self_attn.fc")


def forward_train(params, dims, latents, masks=None, p=0.1):
    """VampNet.forward in train() mode (transformer.py:617-639, 314-369).  `params`: dict of tensors
    (requires_grad where trainable) with the reference's state_dict names."""
    H, L = dims["n_heads"], dims["n_layers"]
    Cp = dims["n_codebooks"] - dims["n_cond"]
    x = F.conv1d(latents, params["embedding.out_proj.weight"], params["embedding.out_proj.bias"]).permute(0, 2, 1)
    T = x.shape[1]
    bias = vo.compute_bias(params["transformer.layers.0.self_attn.relative_attention_bias.weight"], T)
    for i in range(L):
        pre = f"transformer.layers.{i}."
        y = vo.rmsnorm(x, params[pre + "norm_1.weight"])
        y = attention_train(y, params, pre, H, bias, masks, i, p)
        x = x + _drop(y, masks, (i, "res1"), p)
        y = vo.rmsnorm(x, params[pre + "norm_3.weight"])
        y = lora_linear(y, params, pre + "feed_forward.w_1")
        y = _drop(vo.gated_gelu(y), masks, (i, "ffn"), p)
        y = lora_linear(y, params, pre + "feed_forward.w_2")
        x = x + _drop(y, masks, (i, "res2"), p)
    out = vo.rmsnorm(x, params["transformer.norm.weight"]).permute(0, 2, 1)
    w = torch._weight_norm(params["classifier.layers.0.weight_v"], params["classifier.layers.0.weight_g"], 0)
    out = F.conv1d(out, w, params["classifier.layers.0.bias"])
    B = out.shape[0]
    V = dims["vocab"]
    return out.view(B, V, Cp, T).permute(0, 1, 3, 2).reshape(B, V, T * Cp)


def trainable_names(sd):
    """Every VampNet parameter (see module docstring for the loralib caveat)."""
    return [k for k in sd if not k.endswith("num_batches_tracked")]


def make_training_mask(z, r, n_cond, generator=None):
    """train.py:250-254: pmask.random(z, r) -> codebook_unmask -> (applied by the caller).  r: (B,) in [0,1]."""
    probs = torch.ones_like(z, dtype=torch.float32) * vo.gamma(r)[:, None, None]
    mask = torch.bernoulli(probs, generator=generator).round().long()
    return vo.codebook_unmask(mask, n_cond)


LORA_KEYS = ("self_attn.w_qs", "self_attn.w_vs", "self_attn.fc", "feed_forward.w_1", "feed_forward.w_2")


def add_lora(sd, dims, seed=0, zero_b=False, r=8):
    """state_dict + loralib adapters for the five LoRA'd linears of every layer: lora_A (r, in) kaiming-uniform(a=sqrt 5),
    lora_B (out, r) zeros at init (loralib.Linear.reset_parameters); `zero_b=False` fills B with small random values so
    that gradient tests exercise both factors."""
    g = torch.Generator().manual_seed(seed)
    out = dict(sd)
    for l in range(dims["n_layers"]):
        for key in LORA_KEYS:
            w = sd[f"transformer.layers.{l}.{key}.weight"]
            n_out, n_in = w.shape
            bound = 1.0 / math.sqrt(n_in)              # kaiming_uniform_(a=sqrt(5)) on (r, in): bound = sqrt(6/((1+5) in))
            out[f"transformer.layers.{l}.{key}.lora_A"] = (torch.rand(r, n_in, generator=g) * 2 - 1) * bound
            out[f"transformer.layers.{l}.{key}.lora_B"] = (torch.zeros(n_out, r) if zero_b
                                                          else (torch.rand(n_out, r, generator=g) * 2 - 1) * 0.05)
    return out


def loss_and_grads(sd, dims, codebooks, z, mask, masks=None, p=0.1, label_smoothing=0.1, only_lora=False):
    """One forward/backward of train_loop (train.py:252-287).  z (B,C,T) int64 clean tokens, mask (B,C,T) {0,1}.
    `only_lora`: train.py:696 lora.mark_only_lora_as_trainable — gradients only for names containing "lora_".
    Returns (loss float tensor, grads dict name -> tensor, logits)."""
    n_cond = dims["n_cond"]
    params = {k: v.detach().clone().requires_grad_((not only_lora) or ("lora_" in k)) for k, v in sd.items()}
    z_mask, mask = vo.apply_mask(z, mask, dims["vocab"])
    latents = vo.from_codes(params, codebooks, z_mask)
    z_hat = forward_train(params, dims, latents, masks, p)
    target = vo.codebook_flatten(z[:, n_cond:, :])
    flat_mask = vo.codebook_flatten(mask[:, n_cond:, :])
    t_masked = target.masked_fill(~flat_mask.bool(), IGNORE_INDEX)
    loss = F.cross_entropy(z_hat, t_masked, label_smoothing=label_smoothing, ignore_index=IGNORE_INDEX)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in params.items()
             if (not only_lora) or ("lora_" in k)}
    return loss.detach(), grads, z_hat.detach()


def noam_lr(step, d_model, factor=2.0, warmup=10000):
    """vampnet/scheduler.py:38-46 evaluated at `steps = step` (step >= 1)."""
    return factor * (d_model ** (-0.5) * min(step ** (-0.5), step * warmup ** (-1.5)))


def clip_and_adamw(sd, grads, state, lr, grad_clip=5.0, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
    """clip_grad_norm_(params, grad_clip) (train.py:296-298) followed by one torch.optim.AdamW step (defaults of
    torch 2.x: betas (0.9,0.999), eps 1e-8, weight_decay 1e-2, decoupled decay).  `state`: dict with "step" and per-name
    "m"/"v" tensors (created on first use).  Returns (new_sd, grad_norm)."""
    names = list(grads)
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(grads[k]) for k in names]))
    coef = torch.clamp(grad_clip / (total + 1e-6), max=1.0)
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    b1, b2 = betas
    new_sd = {}
    for k in names:
        g = grads[k] * coef
        m = state.setdefault(("m", k), torch.zeros_like(g))
        v = state.setdefault(("v", k), torch.zeros_like(g))
        w = sd[k] * (1.0 - lr * weight_decay)
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** t
        bc2 = 1 - b2 ** t
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        new_sd[k] = w.addcdiv(m, denom, value=-(lr / bc1))
    return new_sd, total


def train_flops(dims, T):
    """Algorithmic FLOPs of one training step per batch item: forward + 2x for the backward GEMMs (dX and dW of every
    linear, dQ/dK/dV/dP of attention), 2*M*N*K convention; element-wise work excluded."""
    return 3.0 * vo.forward_flops(dims, T)
