"""Shim loader for the *reference* Python (test infrastructure, this container only).

The reference package (/root/reference/vampnet) cannot be imported as-is: its
dependencies `audiotools`, `loralib`, `lac`, `librosa` are not installed
(SURVEY.md §8(c)).  This module registers ~40 lines of stub modules in
``sys.modules`` so that the reference's own files
  vampnet/modules/{transformer,layers,activations}.py, vampnet/mask.py,
  vampnet/util.py, vampnet/interface.py
import and run UNMODIFIED on CPU.  It is used only by ``oracle/make_golden.py``
and by CPU tests that are skipped when /root/reference is absent (the GPU box
has no /root/reference).  Nothing in the product (vampnet_amd/) imports this.
"""
import importlib
import os
import random
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get("VAMPNET_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "vampnet", "modules"))


class _AudioSignal:
    """Minimal stand-in: the hot path only constructs it in VampNet.decode."""

    def __init__(self, audio, sample_rate=44100):
        self.samples = audio
        self.audio_data = audio
        self.sample_rate = sample_rate


def _install_stubs():
    if "audiotools" in sys.modules and getattr(sys.modules["audiotools"], "_vn_stub", False):
        return
    at = types.ModuleType("audiotools")
    at._vn_stub = True
    ml = types.ModuleType("audiotools.ml")

    class BaseModel(nn.Module):
        INTERN = []
        EXTERN = []

        @property
        def device(self):
            return next(self.parameters()).device

    ml.BaseModel = BaseModel
    util = types.ModuleType("audiotools.util")

    def seed(s):
        # descript-audiotools util.seed: torch + numpy + python RNGs [UNVERIFIED-DEP]
        torch.manual_seed(s)
        np.random.seed(s)
        random.seed(s)

    util.seed = seed
    at.ml, at.util, at.AudioSignal = ml, util, _AudioSignal
    sys.modules["audiotools"] = at
    sys.modules["audiotools.ml"] = ml
    sys.modules["audiotools.util"] = util

    lora = types.ModuleType("loralib")

    class Linear(nn.Linear):
        def __init__(self, i, o, r=0, lora_alpha=1, **kw):
            super().__init__(i, o, bias=kw.get("bias", True))

    lora.Linear = Linear
    sys.modules["loralib"] = lora

    for name in ("librosa", "lac", "lac.model", "lac.model.lac"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["lac.model.lac"].LAC = object

    pkg = types.ModuleType("vampnet")
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "vampnet")]
    sys.modules["vampnet"] = pkg


def load_reference():
    """Returns a namespace with the reference's modules (transformer, layers, mask, util, interface)."""
    if not reference_available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")
    _install_stubs()
    ns = types.SimpleNamespace()
    ns.transformer = importlib.import_module("vampnet.modules.transformer")
    ns.layers = importlib.import_module("vampnet.modules.layers")
    ns.activations = importlib.import_module("vampnet.modules.activations")
    ns.mask = importlib.import_module("vampnet.mask")
    ns.util = importlib.import_module("vampnet.util")
    # interface.py imports .beats (-> librosa stub, wavebeat): stub beats wholesale
    beats = types.ModuleType("vampnet.beats")
    beats.WaveBeat = object
    sys.modules.setdefault("vampnet.beats", beats)
    ns.interface = importlib.import_module("vampnet.interface")
    return ns


class FakeCodec(nn.Module):
    """Stand-in for lac.LAC exposing only what the hot path touches
    (reference call sites: layers.py:145, interface.py:179-189)."""

    def __init__(self, codebooks: torch.Tensor, hop_length=768, sample_rate=44100):
        super().__init__()
        n = codebooks.shape[0]
        self.quantizer = nn.Module()
        self.quantizer.quantizers = nn.ModuleList()
        for i in range(n):
            q = nn.Module()
            q.codebook = nn.Embedding(codebooks.shape[1], codebooks.shape[2])
            q.codebook.weight.data.copy_(codebooks[i])
            self.quantizer.quantizers.append(q)
        self.hop_length = hop_length
        self.sample_rate = sample_rate


def build_reference_model(ns, dims: dict, state_dict: dict):
    """Instantiate the reference VampNet with `dims` and load our synthetic state_dict."""
    m = ns.transformer.VampNet(
        n_heads=dims["n_heads"], n_layers=dims["n_layers"], n_codebooks=dims["n_codebooks"],
        n_conditioning_codebooks=dims["n_cond"], latent_dim=dims["latent_dim"],
        embedding_dim=dims["d_model"], vocab_size=dims["vocab"], flash_attn=False)
    missing, unexpected = m.load_state_dict(state_dict, strict=True), None
    m.eval()
    return m


def build_reference_interface(ns, coarse, c2f, codec, device="cpu"):
    """object.__new__(Interface) with fake codec (SURVEY.md App. E probe 7)."""
    itf = object.__new__(ns.interface.Interface)
    nn.Module.__init__(itf)
    itf.codec = codec
    itf.coarse = coarse
    itf.c2f = c2f
    coarse.chunk_size_s = 10
    if c2f is not None:
        c2f.chunk_size_s = 3
    itf.device = device
    itf.beat_tracker = None
    itf.loudness = -24.0
    return itf
