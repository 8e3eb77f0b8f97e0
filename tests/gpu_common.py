"""Shared helpers for the `-m gpu` parity tests (HIP engine vs the CPU oracle)."""
import numpy as np
import torch

from oracle import vampnet_oracle as O, weights as W


from vampnet_amd.synth import SynthCodec, model_kwargs  # noqa: F401,E402


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
