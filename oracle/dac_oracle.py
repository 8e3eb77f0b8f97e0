"""CPU ORACLE for the DAC / `lac.LAC` codec used by Interface.encode / Interface.decode — TEST INFRASTRUCTURE.

**PARITY UNPINNED.**  The codec is a third-party dependency that is absent from /root/reference
(`lac @ git+https://github.com/hugofloresgarcia/lac.git`, unpinned: requirements.txt:6, setup.py:32) and
no checkpoint is available offline.  Nothing in the reference's tree pins results at this boundary.
This file restates the *published* Descript Audio Codec design the fork follows (SURVEY.md App. D:
dac/model/dac.py + dac/nn/quantize.py + dac/nn/layers.py of descript-audio-codec 1.0), in torch-CPU fp32,
anchored on the reference's call sites:
    interface.py:215  codec.preprocess(samples, sr)          -> right-pad to a multiple of hop_length
    interface.py:223  codec.encode(samples, sr)["codes"]     -> (B, n_codebooks, T) int64
    transformer.py:671-675  codec.decode(codec.quantizer.from_latents(from_codes(z))[0])["audio"]
    layers.py:145     codec.quantizer.quantizers[i].codebook.weight  (1024 x 8)
Hyper-parameters (encoder_dim, encoder_rates, latent_dim, decoder_dim, decoder_rates, n_codebooks,
codebook_size, codebook_dim, sample_rate) are read from the checkpoint's metadata kwargs when one exists;
hop 768 / 44.1 kHz / 14 codebooks of 1024 x 8 are inferred from the reference.

State-dict key names follow descript-audio-codec (old-style weight_norm: *.weight_g / *.weight_v / *.bias):
  encoder.block.0                                  WNConv1d(1, d, 7, pad 3)
  encoder.block.{1+i}.block.{0,1,2}.block.{0,2}.alpha ; .block.{1,3}  ResidualUnit(dil 1,3,9): snake, conv7, snake, conv1
  encoder.block.{1+i}.block.3.alpha ; .block.4     Snake1d ; WNConv1d(d/2, d, 2s, stride s, pad ceil(s/2))
  encoder.block.{n+1}.alpha ; encoder.block.{n+2}  Snake1d ; WNConv1d(d, latent, 3, pad 1)
  quantizer.quantizers.{i}.{in_proj,out_proj}, .codebook.weight
  decoder.model.0                                  WNConv1d(latent, D, 7, pad 3)
  decoder.model.{1+i}.block.0.alpha ; .block.1     Snake1d ; WNConvTranspose1d(Din, Dout, 2s, stride s, pad ceil(s/2))
  decoder.model.{1+i}.block.{2,3,4}.block.*        ResidualUnit(dil 1,3,9)
  decoder.model.{n+1}.alpha ; decoder.model.{n+2}  Snake1d ; WNConv1d(Dlast, 1, 7, pad 3) ; tanh
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# configs and the seeded synthetic-weight generator live in the product's data-generator module (no arithmetic there);
# re-exported here so tests keep one import
from vampnet_amd.synth import DAC_DEFAULT_CFG, DAC_TINY_CFG, synth_dac_state_dict  # noqa: E402,F401


def latent_dim(cfg):
    return cfg["latent_dim"] or cfg["encoder_dim"] * (2 ** len(cfg["encoder_rates"]))


def hop_length(cfg):
    return int(np.prod(cfg["encoder_rates"]))


def wn(sd, key):
    """fold old-style weight_norm: w = g * v / ||v|| (norm over all dims but 0)"""
    return torch._weight_norm(sd[key + ".weight_v"], sd[key + ".weight_g"], 0)


def snake(x, alpha):
    """vampnet/modules/layers.py:12-18 (same op the codec uses): x + (alpha + 1e-9)^-1 * sin(alpha x)^2"""
    return x + (alpha + 1e-9).reciprocal() * torch.sin(alpha * x).pow(2)


def residual_unit(sd, p, x, dilation):
    pad = ((7 - 1) * dilation) // 2
    y = snake(x, sd[p + ".block.0.alpha"])
    y = F.conv1d(y, wn(sd, p + ".block.1"), sd[p + ".block.1.bias"], dilation=dilation, padding=pad)
    y = snake(y, sd[p + ".block.2.alpha"])
    y = F.conv1d(y, wn(sd, p + ".block.3"), sd[p + ".block.3.bias"])
    return x + y


def encoder(sd, cfg, audio):
    x = F.conv1d(audio, wn(sd, "encoder.block.0"), sd["encoder.block.0.bias"], padding=3)
    for i, s in enumerate(cfg["encoder_rates"]):
        p = f"encoder.block.{1 + i}"
        for j, d in enumerate((1, 3, 9)):
            x = residual_unit(sd, f"{p}.block.{j}", x, d)
        x = snake(x, sd[p + ".block.3.alpha"])
        x = F.conv1d(x, wn(sd, p + ".block.4"), sd[p + ".block.4.bias"], stride=s, padding=math.ceil(s / 2))
    n = len(cfg["encoder_rates"])
    x = snake(x, sd[f"encoder.block.{n + 1}.alpha"])
    return F.conv1d(x, wn(sd, f"encoder.block.{n + 2}"), sd[f"encoder.block.{n + 2}.bias"], padding=1)


def decode_latents(sd, i, latents):
    """VectorQuantize.decode_latents: nearest codebook row by cosine (L2-normalised) distance."""
    B, D, T = latents.shape
    enc = latents.permute(0, 2, 1).reshape(B * T, D)
    cb = sd[f"quantizer.quantizers.{i}.codebook.weight"]
    enc_n, cb_n = F.normalize(enc), F.normalize(cb)
    dist = enc_n.pow(2).sum(1, keepdim=True) - 2 * enc_n @ cb_n.t() + cb_n.pow(2).sum(1, keepdim=True).t()
    idx = (-dist).max(1)[1].view(B, T)
    z_q = F.embedding(idx, cb).transpose(1, 2)
    return z_q, idx


def rvq_encode(sd, cfg, z):
    """ResidualVectorQuantize.forward (eval, all quantizers): returns (z_q, codes (B, n, T))."""
    residual, z_q, codes = z, 0, []
    for i in range(cfg["n_codebooks"]):
        p = f"quantizer.quantizers.{i}"
        z_e = F.conv1d(residual, wn(sd, p + ".in_proj"), sd[p + ".in_proj.bias"])
        zq_i, idx = decode_latents(sd, i, z_e)
        zq_i = z_e + (zq_i - z_e)                                  # straight-through estimator, eval arithmetic kept
        zq_i = F.conv1d(zq_i, wn(sd, p + ".out_proj"), sd[p + ".out_proj.bias"])
        z_q = z_q + zq_i
        residual = residual - zq_i
        codes.append(idx)
    return z_q, torch.stack(codes, dim=1)


def from_codes(sd, cfg, codes):
    """quantizer.from_latents(embedding.from_codes(z)) as VampNet.decode calls it (transformer.py:671-672):
    z_q = sum_i out_proj_i(codebook_i[code_i])  (re-quantising exact codebook rows is idempotent)."""
    z_q = 0
    for i in range(codes.shape[1]):
        p = f"quantizer.quantizers.{i}"
        z_p = F.embedding(codes[:, i], sd[p + ".codebook.weight"]).transpose(1, 2)
        z_q = z_q + F.conv1d(z_p, wn(sd, p + ".out_proj"), sd[p + ".out_proj.bias"])
    return z_q


def decoder(sd, cfg, z):
    x = F.conv1d(z, wn(sd, "decoder.model.0"), sd["decoder.model.0.bias"], padding=3)
    for i, s in enumerate(cfg["decoder_rates"]):
        p = f"decoder.model.{1 + i}"
        x = snake(x, sd[p + ".block.0.alpha"])
        x = F.conv_transpose1d(x, wn(sd, p + ".block.1"), sd[p + ".block.1.bias"], stride=s, padding=math.ceil(s / 2))
        for j, d in enumerate((1, 3, 9)):
            x = residual_unit(sd, f"{p}.block.{2 + j}", x, d)
    n = len(cfg["decoder_rates"])
    x = snake(x, sd[f"decoder.model.{n + 1}.alpha"])
    x = F.conv1d(x, wn(sd, f"decoder.model.{n + 2}"), sd[f"decoder.model.{n + 2}.bias"], padding=3)
    return torch.tanh(x)


def preprocess(cfg, audio):
    """DAC.preprocess (interface.py:215): right-pad to a multiple of hop_length."""
    hop = hop_length(cfg)
    L = audio.shape[-1]
    return F.pad(audio, (0, math.ceil(L / hop) * hop - L))


@torch.inference_mode()
def encode(sd, cfg, audio):
    """audio (B,1,L) -> codes (B, n_codebooks, L/hop) int64   (interface.py:223)"""
    return rvq_encode(sd, cfg, encoder(sd, cfg, preprocess(cfg, audio)))[1]


@torch.inference_mode()
def decode(sd, cfg, codes):
    """codes (B, n, T) -> audio (B,1,T*hop)   (transformer.py:669-675, MASK already replaced by 0)"""
    return decoder(sd, cfg, from_codes(sd, cfg, codes))
