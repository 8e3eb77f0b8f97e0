"""Deterministic synthetic weights / inputs for the VampNet hot path (benchmark + test data; no model logic here).

No checkpoints exist in this image (they live on the HF hub), so every
parity test and the benchmark run on seeded random-init weights of the real architecture.
The generator is numpy's PCG64 (`default_rng`) so the same (dims, seed) gives bit-identical
tensors on every machine, independent of torch's CPU RNG vectorisation.

Key names and shapes follow the reference state_dict (SURVEY.md App. B;
/root/reference/vampnet/modules/transformer.py:535-604, layers.py:104-132):

  embedding.special.MASK (C, 8); embedding.out_proj.{weight (D, 8C, 1), bias (D)}
  transformer.layers.{i}.norm_1.weight (D); .self_attn.{w_qs,w_ks,w_vs,fc}.weight (D, D)
  transformer.layers.0.self_attn.relative_attention_bias.weight (32, H)
  transformer.layers.{i}.norm_3.weight (D); .feed_forward.w_1.weight (4D, D); .w_2.weight (D, 2D)
  transformer.norm.weight (D)
  classifier.layers.0.{weight_g (V*Cp,1,1), weight_v (V*Cp, D, 1), bias (V*Cp)}

Scales follow PyTorch's default inits (Linear/Conv: U(+-1/sqrt(fan_in)); Embedding/MASK: N(0,1)),
except that norm weights and weight_g are perturbed away from their trivial defaults
(ones / ||v||) so that a kernel which forgets to apply them fails parity.
"""
import math

import numpy as np
import torch

COARSE_DIMS = dict(n_heads=20, n_layers=20, n_codebooks=4, n_cond=0, latent_dim=8,
                   d_model=1280, vocab=1024)
C2F_DIMS = dict(n_heads=20, n_layers=16, n_codebooks=14, n_cond=4, latent_dim=8,
                d_model=1280, vocab=1024)
# small shapes the CPU oracle finishes in milliseconds (unit tests, golden fixtures)
# (d_model / n_heads = 64 like the shipped configs: the HIP attention kernel is specialised for d_head 64)
TINY_COARSE_DIMS = dict(n_heads=4, n_layers=2, n_codebooks=4, n_cond=0, latent_dim=8,
                        d_model=256, vocab=1024)
TINY_C2F_DIMS = dict(n_heads=4, n_layers=2, n_codebooks=14, n_cond=4, latent_dim=8,
                     d_model=256, vocab=1024)


def _uniform(rng, shape, bound):
    return torch.from_numpy(rng.uniform(-bound, bound, size=shape).astype(np.float32))


def _normal(rng, shape, std=1.0):
    return torch.from_numpy((rng.standard_normal(size=shape) * std).astype(np.float32))


def synth_state_dict(dims: dict, seed: int = 0, heavy_tail=None) -> dict:
    """heavy_tail: None = PyTorch-default-like scales (activations O(1)); "in_range" / "saturating" = the same model re-shaped into
    what TRAINED T5-style stacks look like (see _heavy_tail): outlier residual channels, large norm gains, log-normal row scales —
    with every value still inside fp16's range ("in_range") or with a GEGLU unit beyond 65504 and attention values beyond 4094
    ("saturating": what precision "f16x2" must detect and hand to "bf16x3")."""
    sd = _synth_state_dict(dims, seed)
    if heavy_tail:
        _heavy_tail(sd, dims, seed, heavy_tail)
    return sd


# outlier residual channels of the heavy-tailed variants (indices < 256 so that the tiny test dims work too)
HEAVY_CHANNELS = (5, 77, 200)


def _heavy_tail(sd, dims, seed, kind):
    """Re-shape a default-init state_dict into a "trained-like" one, in place (numpy PCG64 stream of its own):
      * log-normal scales on the rows of every attention / FFN matrix (sigma 0.4);
      * three OUTLIER residual channels: the embedding bias puts +-A there for every token (A = 1e3 "in_range", 1e4 "saturating"),
        i.e. after RMSNorm those channels sit at ~ +-20 and everything else at ~ 1 / (A sqrt(3 / D));
      * norm gains up to 30 (all norms: a random 2 % of the channels between 5 and 30; the outlier channels get gain 30 in layer 0
        -> normalised outliers of ~ 600);
      * layer 0: one GEGLU unit whose value and gate both read outlier channel 0 with weight w_g -> |p1 gelu(p2)| ~ (600 w_g)^2, and one
        attention value channel reading it with weight w_v -> |v| ~ 600 w_v.  "in_range": ~3e4 and ~3e3 (inside fp16 / inside the
        x16 attention operand); "saturating": ~3.6e5 and ~6e3 (beyond both).  The consumers of those two units (a W2 column, an fc
        column) are scaled down so the rest of the model keeps seeing O(1) updates: the point is the operand range, not chaos."""
    assert kind in ("in_range", "saturating"), kind
    rng = np.random.default_rng(1000003 * (seed + 1) + (1 if kind == "saturating" else 0))
    D, L = dims["d_model"], dims["n_layers"]
    A = 1e4 if kind == "saturating" else 1e3
    y_out = 30.0 * math.sqrt(D / 3.0)                      # normalised outlier magnitude at gain 30 (620 at D = 1280)
    w_g = math.sqrt(3.6e5 if kind == "saturating" else 3.0e4) / y_out        # |p1 gelu(p2)| = (y_out w_g)^2
    w_v = (6.0e3 if kind == "saturating" else 3.0e3) / y_out                 # |v| = y_out w_v
    c0 = HEAVY_CHANNELS[0]
    for i in range(L):
        p = f"transformer.layers.{i}."
        for name in ("self_attn.w_qs", "self_attn.w_ks", "self_attn.w_vs", "self_attn.fc", "feed_forward.w_1", "feed_forward.w_2"):
            w = sd[p + name + ".weight"]
            w *= torch.from_numpy(np.exp(0.4 * rng.standard_normal(w.shape[0])).astype(np.float32))[:, None]
        for nm in ("norm_1", "norm_3"):
            g = sd[p + nm + ".weight"]
            pick = rng.random(D) < 0.02
            g[torch.from_numpy(pick)] = torch.from_numpy(rng.uniform(5.0, 30.0, int(pick.sum())).astype(np.float32))
    g = sd["transformer.norm.weight"]
    pick = rng.random(D) < 0.02
    g[torch.from_numpy(pick)] = torch.from_numpy(rng.uniform(5.0, 30.0, int(pick.sum())).astype(np.float32))
    bias = sd["embedding.out_proj.bias"]
    for k, c in enumerate(HEAVY_CHANNELS):
        bias[c] = A * (1.0 if k % 2 == 0 else -1.0)
    p = "transformer.layers.0."
    for nm in ("norm_1", "norm_3"):
        for c in HEAVY_CHANNELS:
            sd[p + nm + ".weight"][c] = 30.0
    # the GEGLU unit: value row j and gate row 2D + j of w_1 read outlier channel c0 (positive gate: gelu ~ identity)
    j = 11
    w1 = sd[p + "feed_forward.w_1.weight"]
    w1[j].zero_(); w1[2 * D + j].zero_()
    w1[j, c0] = w_g; w1[2 * D + j, c0] = w_g
    sd[p + "feed_forward.w_2.weight"][:, j] *= 1e-5
    # the attention value channel (head 0, d = 3)
    wv = sd[p + "self_attn.w_vs.weight"]
    wv[3].zero_(); wv[3, c0] = w_v
    sd[p + "self_attn.fc.weight"][:, 3] *= 1e-3


def _synth_state_dict(dims: dict, seed: int = 0) -> dict:
    rng = np.random.default_rng(seed)
    D, H, L = dims["d_model"], dims["n_heads"], dims["n_layers"]
    C, nc, ld, V = dims["n_codebooks"], dims["n_cond"], dims["latent_dim"], dims["vocab"]
    Cp = C - nc
    sd = {}
    sd["embedding.special.MASK"] = _normal(rng, (C, ld))
    sd["embedding.out_proj.weight"] = _uniform(rng, (D, C * ld, 1), 1.0 / np.sqrt(C * ld))
    sd["embedding.out_proj.bias"] = _uniform(rng, (D,), 1.0 / np.sqrt(C * ld))
    for i in range(L):
        p = f"transformer.layers.{i}."
        sd[p + "norm_1.weight"] = 1.0 + _uniform(rng, (D,), 0.1)
        for name in ("w_qs", "w_ks", "w_vs", "fc"):
            sd[p + f"self_attn.{name}.weight"] = _uniform(rng, (D, D), 1.0 / np.sqrt(D))
        if i == 0:
            sd[p + "self_attn.relative_attention_bias.weight"] = _normal(rng, (32, H))
        sd[p + "norm_3.weight"] = 1.0 + _uniform(rng, (D,), 0.1)
        sd[p + "feed_forward.w_1.weight"] = _uniform(rng, (4 * D, D), 1.0 / np.sqrt(D))
        sd[p + "feed_forward.w_2.weight"] = _uniform(rng, (D, 2 * D), 1.0 / np.sqrt(2 * D))
    sd["transformer.norm.weight"] = 1.0 + _uniform(rng, (D,), 0.1)
    v = _uniform(rng, (V * Cp, D, 1), 1.0 / np.sqrt(D))
    sd["classifier.layers.0.weight_v"] = v
    sd["classifier.layers.0.weight_g"] = (
        v.norm(dim=(1, 2), keepdim=True) * (1.0 + _uniform(rng, (V * Cp, 1, 1), 0.1)))
    sd["classifier.layers.0.bias"] = _uniform(rng, (V * Cp,), 1.0 / np.sqrt(D))
    return sd


def synth_codebooks(n_codebooks: int = 14, vocab: int = 1024, latent_dim: int = 8,
                    seed: int = 1234) -> torch.Tensor:
    """Stand-in for codec.quantizer.quantizers[i].codebook.weight (reference layers.py:145)."""
    rng = np.random.default_rng(seed)
    return _normal(rng, (n_codebooks, vocab, latent_dim))


def synth_codes(batch: int, n_codebooks: int = 14, T: int = 575, vocab: int = 1024,
                seed: int = 7) -> torch.Tensor:
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.integers(0, vocab, size=(batch, n_codebooks, T), dtype=np.int64))


class SynthCodec:
    """What Interface reads from a codec on the vamp() path when no real DAC checkpoint exists
    (reference layers.py:145: quantizer.quantizers[i].codebook.weight; interface.py:176-189: hop_length, sample_rate)."""

    class _Q:
        def __init__(self, w):
            self.codebook = type("CB", (), {"weight": w})()

    def __init__(self, codebooks, hop_length=768, sample_rate=44100):
        self.quantizer = type("RVQ", (), {"quantizers": [SynthCodec._Q(codebooks[i]) for i in range(codebooks.shape[0])]})()
        self.hop_length, self.sample_rate = hop_length, sample_rate


def model_kwargs(dims):
    """dims dict -> VampNet constructor kwargs (reference transformer.py:535-552)."""
    return dict(n_heads=dims["n_heads"], n_layers=dims["n_layers"], n_codebooks=dims["n_codebooks"],
                n_conditioning_codebooks=dims["n_cond"], latent_dim=dims["latent_dim"],
                embedding_dim=dims["d_model"], vocab_size=dims["vocab"])


# ----------------------------------------------------------------------------- DAC codec
DAC_DEFAULT_CFG = dict(encoder_dim=64, encoder_rates=[2, 4, 8, 12], latent_dim=None, decoder_dim=1536,
                       decoder_rates=[12, 8, 4, 2], n_codebooks=14, codebook_size=1024, codebook_dim=8,
                       sample_rate=44100)
# shapes the CPU finishes in milliseconds (channel counts stay multiples of 32 like every real layer)
DAC_TINY_CFG = dict(encoder_dim=32, encoder_rates=[2, 4], latent_dim=None, decoder_dim=128,
                    decoder_rates=[4, 2], n_codebooks=4, codebook_size=1024, codebook_dim=8, sample_rate=44100)


def synth_dac_state_dict(cfg, seed=0):
    """Seeded synthetic codec weights (numpy PCG64).  PyTorch-default-like scales; snake alphas around 1;
    weight_g perturbed from ||v|| so the weight-norm fold is exercised."""
    rng = np.random.default_rng(seed)
    sd = {}

    def U(shape, b):
        return torch.from_numpy(rng.uniform(-b, b, size=shape).astype(np.float32))

    def conv(key, cout, cin, k, transposed=False):
        shape = (cin, cout, k) if transposed else (cout, cin, k)
        fan_in = (cout if transposed else cin) * k
        v = U(shape, 1.0 / math.sqrt(fan_in))
        sd[key + ".weight_v"] = v
        sd[key + ".weight_g"] = v.norm(dim=(1, 2), keepdim=True) * (1.0 + U((shape[0], 1, 1), 0.1))
        sd[key + ".bias"] = U((cout,), 1.0 / math.sqrt(fan_in))

    def alpha(key, c):
        sd[key + ".alpha"] = 1.0 + U((1, c, 1), 0.3)

    def res(p, c):
        alpha(p + ".block.0", c); conv(p + ".block.1", c, c, 7); alpha(p + ".block.2", c); conv(p + ".block.3", c, c, 1)

    d = cfg["encoder_dim"]
    conv("encoder.block.0", d, 1, 7)
    for i, s in enumerate(cfg["encoder_rates"]):
        d *= 2
        p = f"encoder.block.{1 + i}"
        for j in range(3):
            res(f"{p}.block.{j}", d // 2)
        alpha(p + ".block.3", d // 2)
        conv(p + ".block.4", d, d // 2, 2 * s)
    n = len(cfg["encoder_rates"])
    L = cfg["latent_dim"] or cfg["encoder_dim"] * (2 ** len(cfg["encoder_rates"]))
    alpha(f"encoder.block.{n + 1}", d)
    conv(f"encoder.block.{n + 2}", L, d, 3)
    for i in range(cfg["n_codebooks"]):
        p = f"quantizer.quantizers.{i}"
        conv(p + ".in_proj", cfg["codebook_dim"], L, 1)
        conv(p + ".out_proj", L, cfg["codebook_dim"], 1)
        sd[p + ".codebook.weight"] = torch.from_numpy(
            rng.standard_normal((cfg["codebook_size"], cfg["codebook_dim"])).astype(np.float32))
    D = cfg["decoder_dim"]
    conv("decoder.model.0", D, L, 7)
    for i, s in enumerate(cfg["decoder_rates"]):
        cin, cout = D // 2 ** i, D // 2 ** (i + 1)
        p = f"decoder.model.{1 + i}"
        alpha(p + ".block.0", cin)
        conv(p + ".block.1", cout, cin, 2 * s, transposed=True)
        for j in range(3):
            res(f"{p}.block.{2 + j}", cout)
    n = len(cfg["decoder_rates"])
    alpha(f"decoder.model.{n + 1}", cout)
    conv(f"decoder.model.{n + 2}", 1, cout, 7)
    return sd
