"""Shared helpers for the `-m gpu` parity tests (HIP engine vs the CPU oracle)."""
import numpy as np
import torch

from oracle import vampnet_oracle as O, weights as W


class SynthCodec:
    """What Interface reads from the codec on the vamp() path (layers.py:145; interface.py:176-189)."""

    class _Q:
        def __init__(self, w):
            self.codebook = type("CB", (), {"weight": w})()

    def __init__(self, codebooks, hop_length=768, sample_rate=44100):
        self.quantizer = type("RVQ", (), {"quantizers": [SynthCodec._Q(codebooks[i]) for i in range(codebooks.shape[0])]})()
        self.hop_length, self.sample_rate = hop_length, sample_rate


def model_kwargs(dims):
    return dict(n_heads=dims["n_heads"], n_layers=dims["n_layers"], n_codebooks=dims["n_codebooks"],
                n_conditioning_codebooks=dims["n_cond"], latent_dim=dims["latent_dim"],
                embedding_dim=dims["d_model"], vocab_size=dims["vocab"])


def to_native(logits_ref, Cp):
    """reference layout (B, V, T*Cp) -> engine layout (B, T, Cp, V)."""
    B, V, N = logits_ref.shape
    return logits_ref.permute(0, 2, 1).reshape(B, N // Cp, Cp, V).contiguous()


def sample_margins(logits_bnv, exp, temperature, sample):
    """relative top-2 margin of the decision each row makes (oracle side)."""
    if sample:
        p = torch.softmax(logits_bnv / temperature, -1).reshape(-1, logits_bnv.shape[-1])
        s = p / exp
    else:
        s = logits_bnv.reshape(-1, logits_bnv.shape[-1])
    top = s.topk(2, -1).values
    return ((top[:, 0] - top[:, 1]) / top[:, 0].abs().clamp_min(1e-30)).reshape(logits_bnv.shape[:-1])
