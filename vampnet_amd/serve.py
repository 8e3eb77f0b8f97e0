"""Request adapter for the reference's `vamp` endpoint — no web framework, just the wire contract.

The reference serves `Interface.vamp` through a gradio app whose `api_name="vamp"` endpoint takes 18 positional
inputs (app.py:660-711, `_inputs`) and returns two audio outputs; the unloop Max/OSC client packs its OSC message into
exactly those keyword names (unloop/client.py:116-186).  This module keeps both orders verbatim and runs the body of
`_vamp_internal` (app.py:120-263) against the HIP `Interface`, so that whichever transport a deployment puts in front
(gradio, FastAPI, OSC) binds `VampService.api_vamp` with the reference's argument list and existing clients keep working.

Out of scope here (documented errors, not silent fallbacks): `pitch_shift_amt != 0` (torch_pitch_shift is absent),
`beat_mask_ms > 0` without an injected `interface.beat_tracker` (WaveBeat itself is not rebuilt) and the HF-hub model zoo behind `load_finetuned` —
`model_choice` resolves through a local `{name: (coarse_ckpt, c2f_ckpt)}` registry instead.
"""
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from .codec import AudioSignal
from .engine import seed_all

# app.py:660-679 (`_inputs`) — the positional order of both the UI button and the `api_name="vamp"` endpoint
VAMP_ARG_ORDER = (
    "input_audio", "sampletemp", "top_p", "periodic_p", "dropout", "stretch_factor", "onset_mask_width",
    "typical_filtering", "typical_mass", "typical_min_tokens", "seed", "model_choice", "n_mask_codebooks",
    "pitch_shift_amt", "sample_cutoff", "sampling_steps", "beat_mask_ms", "num_feedback_steps",
)

# unloop/client.py:119-136 — field order of the OSC message the Max patch sends
UNLOOP_OSC_ORDER = (
    "query_id", "client_type", "audio_path", "model_choice", "periodic_p", "dropout", "seed", "looplength_ms",
    "typical_filter", "typical_mass", "typical_min_tokens", "upper_codebook_mask", "onset_mask_width",
    "sampling_steps", "temperature", "top_p", "beat_mask_ms", "num_feedback_steps",
)

BATCH_SIZE = 2               # app.py:231 `batch_size=2`; unloop/client.py:105 reads the first two outputs
CHUNK_SIZE_S = 10.0          # app.py:219


def unloop_to_request(osc_args: Sequence) -> Dict:
    """OSC argument list -> the keyword set `client.submit(...)` sends to `/vamp` (unloop/client.py:160-180): the client
    pins stretch_factor=1, pitch_shift_amt=0, sample_cutoff=1.0 and renames four fields.  `input_audio` is the path."""
    if len(osc_args) != len(UNLOOP_OSC_ORDER):
        raise ValueError(f"expected {len(UNLOOP_OSC_ORDER)} OSC fields, got {len(osc_args)}")
    a = dict(zip(UNLOOP_OSC_ORDER, osc_args))
    return dict(
        input_audio=str(a["audio_path"]), sampletemp=a["temperature"], top_p=a["top_p"], periodic_p=a["periodic_p"],
        dropout=a["dropout"], stretch_factor=1, onset_mask_width=a["onset_mask_width"],
        typical_filtering=bool(a["typical_filter"]), typical_mass=a["typical_mass"],
        typical_min_tokens=a["typical_min_tokens"], seed=a["seed"], model_choice=a["model_choice"],
        n_mask_codebooks=a["upper_codebook_mask"], pitch_shift_amt=0, sample_cutoff=1.0,
        sampling_steps=a["sampling_steps"], beat_mask_ms=int(a["beat_mask_ms"]),
        num_feedback_steps=a["num_feedback_steps"],
    )


def crop_to_loop(sig: AudioSignal, looplength_ms: float) -> AudioSignal:
    """unloop/client.py:148-158: keep the first `looplength_ms` of the (48 kHz-forced) recording."""
    end = int((looplength_ms * sig.sample_rate) / 1000)
    return AudioSignal(sig.samples[..., :end], sig.sample_rate)


def _to_signal(input_audio) -> AudioSignal:
    """gradio hands `(sample_rate, int ndarray)` (app.py:171-174: divide by the integer type's max); a path is read as
    16-bit PCM; an AudioSignal passes through."""
    if input_audio is None:
        raise ValueError("no input audio received!")                       # app.py:169-170 (gr.Error)
    if isinstance(input_audio, AudioSignal):
        return input_audio.clone()
    if isinstance(input_audio, (str, bytes)) or hasattr(input_audio, "__fspath__"):
        return AudioSignal.from_wav(input_audio)
    sr, x = input_audio
    x = np.asarray(x)
    if np.issubdtype(x.dtype, np.integer):
        x = x / np.iinfo(x.dtype).max
    x = np.asarray(x, dtype=np.float32)
    if x.ndim == 2:                                                         # gradio: (samples, channels)
        x = x.T
    return AudioSignal(torch.from_numpy(np.ascontiguousarray(x)), sr)


def to_output(sig: AudioSignal, item: int) -> Tuple[int, np.ndarray]:
    """app.py:34-35 applied to `sig[item]`: (sample_rate, mono float samples)."""
    return sig.sample_rate, sig.samples[item, 0].detach().cpu().numpy()


class VampService:
    """`_vamp_internal` (app.py:120-263) bound to one HIP Interface.

    models: {model_choice: (coarse_ckpt, c2f_ckpt or None)} — local stand-in for the HF model zoo of
    `Interface.load_finetuned` (interface.py:134-174); an unknown name raises KeyError, `None`/"default" keeps the
    weights that are loaded."""

    def __init__(self, interface, models: Optional[Dict[str, Tuple[str, Optional[str]]]] = None,
                 chunk_size_s: float = CHUNK_SIZE_S):
        self.interface = interface
        self.models = dict(models or {})
        self.chunk_size_s = chunk_size_s
        self.last_seed = None
        self.last_mask = None

    # -- the two entry points the reference binds (app.py:267-344) ------------------------------
    def vamp(self, *args, **kwargs):
        return self._run(self._bind(args, kwargs), api=False)

    def api_vamp(self, *args, **kwargs):
        return self._run(self._bind(args, kwargs), api=True)

    def unloop(self, osc_args: Sequence):
        """One unloop OSC request end to end (client side crop + the `/vamp` call)."""
        req = unloop_to_request(osc_args)
        sig = AudioSignal.from_wav(req["input_audio"]).to_mono()
        sig.sample_rate = 48000                                             # unloop/client.py:145 ("HOT PATCH")
        req["input_audio"] = crop_to_loop(sig, dict(zip(UNLOOP_OSC_ORDER, osc_args))["looplength_ms"])
        return self._run(req, api=True)

    @staticmethod
    def _bind(args, kwargs) -> Dict:
        if len(args) > len(VAMP_ARG_ORDER):
            raise TypeError(f"vamp takes {len(VAMP_ARG_ORDER)} positional arguments but {len(args)} were given")
        req = dict(zip(VAMP_ARG_ORDER, args))
        for k, v in kwargs.items():
            if k not in VAMP_ARG_ORDER:
                raise TypeError(f"vamp got an unexpected keyword argument '{k}'")
            if k in req:
                raise TypeError(f"vamp got multiple values for argument '{k}'")
            req[k] = v
        missing = [k for k in VAMP_ARG_ORDER if k not in req]
        if missing:
            raise TypeError(f"vamp missing required arguments: {missing}")
        return req

    def _select_model(self, model_choice):
        if model_choice in (None, "", "default") and model_choice not in self.models:
            return
        coarse, c2f = self.models[model_choice]
        self.interface.reload(coarse_ckpt=coarse, c2f_ckpt=c2f)            # no-op when already loaded

    def _run(self, r: Dict, api: bool):
        itf = self.interface
        seed = int(r["seed"])
        _seed = seed if seed > 0 else int(torch.randint(0, 2 ** 32, (1,)).item())          # app.py:164-166
        seed_all(_seed)                                                                    # at.util.seed, app.py:167
        self.last_seed = _seed
        sig = _to_signal(r["input_audio"]).to_mono()                                       # app.py:171-174
        loudness = sig.loudness()
        sig = itf._preprocess(sig)
        self._select_model(r["model_choice"])                                              # app.py:180
        if r["pitch_shift_amt"] != 0:
            raise NotImplementedError("pitch_shift_amt needs torch_pitch_shift (absent in this image)")
        if r["beat_mask_ms"] > 0 and getattr(itf, "beat_tracker", None) is None:
            raise NotImplementedError("beat_mask_ms needs a beat tracker: set interface.beat_tracker to an object with "
                                      "extract_beats(signal) (the reference's WaveBeat model is out of scope, DESIGN.md §8)")
        codes = itf.encode(sig)
        mask = itf.build_mask(codes, sig=sig, periodic_prompt=r["periodic_p"],
                              onset_mask_width=r["onset_mask_width"], _dropout=r["dropout"],
                              upper_codebook_mask=r["n_mask_codebooks"])                   # app.py:198-205
        if r["beat_mask_ms"] > 0:                                                          # app.py:206-217
            beat = itf.make_beat_mask(sig, after_beat_s=r["beat_mask_ms"] / 1000.0)
            mask = torch.min(mask, beat.to(mask.device))                                   # pmask.mask_and
            mask = mask.clone()
            mask[:, int(r["n_mask_codebooks"]):, :] = 1                                    # pmask.codebook_mask
        itf.set_chunk_size(self.chunk_size_s)
        top_p = r["top_p"]
        if top_p is not None and not top_p > 0:                                            # app.py:221-226
            top_p = None
        codes, mask_z = itf.vamp(
            codes, mask, batch_size=BATCH_SIZE, feedback_steps=r["num_feedback_steps"],
            _sampling_steps=r["sampling_steps"], time_stretch_factor=r["stretch_factor"], return_mask=True,
            temperature=r["sampletemp"], typical_filtering=r["typical_filtering"], typical_mass=r["typical_mass"],
            typical_min_tokens=r["typical_min_tokens"], top_p=top_p, seed=_seed, sample_cutoff=r["sample_cutoff"],
        )                                                                                  # app.py:228-243
        self.last_mask = mask
        out = itf.decode(codes)
        out = out.normalize(loudness.expand(out.batch_size))                               # app.py:247
        if api:
            return to_output(out, 0), to_output(out, 1)                                    # app.py:262
        return to_output(out, 0), to_output(out, 1), mask[0].cpu()                         # (mask image -> the mask itself)
