"""`Interface` — drop-in twin of the reference's `vampnet.interface.Interface` (interface.py:54-575) whose
encode / build_mask / vamp / coarse_vamp / coarse_to_fine / decode run on the MI355X engine.

Same method names, argument meaning, defaults, return shapes and error behaviour (asserts) as the
reference; tokens and masks are torch LongTensors on `self.device`.  The control flow here is host
orchestration (chunking, padding, stitching); every FLOP is in libvampnet_hip.so.

Additions that the reference does not have (all optional, defaults reproduce the reference):
  * `rng="torch" | "device"`: parity mode replays torch's CPU generator stream (bit-exact tokens vs the
    CPU reference, tie-audit caveat in DESIGN.md); "device" uses the in-kernel Philox stream (fast mode).
  * batch sharding over the GPUs of a node: construct with `process_group=` (one process per GPU,
    torch.distributed "nccl" = RCCL).  `vamp()` then takes the GLOBAL batch on every rank, runs the
    coarse + c2f loops on its contiguous slice of batch items and finishes with ONE all-gather of the
    (B,14,T) token tensor.  The batch-wide N0 quirk (transformer.py:766) is honoured by
    computing N0 on the global tensors, which every rank holds — no data-path collective.
"""
import functools
import math
import os
from pathlib import Path

import torch

from . import masks
from .checkpoint import load_model_checkpoint, load_tensor_dict, validate_vampnet_state_dict
from .engine import DEFAULT_PRECISION, Engine, VampNetModel


def _load_checkpoint(path):
    """`VampNet.load(location=Path(ckpt), map_location="cpu", strict=False)` (interface.py:34): audiotools dict checkpoint or
    torch.package archive -> (state_dict, constructor kwargs); formats and the trust rule in vampnet_amd/checkpoint.py."""
    sd, kw = load_model_checkpoint(path, package_name="VampNet", kwarg_keys=_MODEL_KEYS)
    validate_vampnet_state_dict(sd, kw)          # a clear error for a file of another architecture, before anything is packed
    return sd, kw


def _load_lora(sd, lora_ckpt):
    """interface.py:37-46: `model.load_state_dict(torch.load(lora_ckpt), strict=False)`.  The reference asks on stdin
    whether to go on when the file is missing; a service cannot — VN_LORA_MISSING_OK=1 answers "y", anything else aborts
    with the reference's exception text."""
    if not Path(lora_ckpt).exists():
        import os
        if os.environ.get("VN_LORA_MISSING_OK") == "1":
            return
        raise Exception(f"lora checkpoint {lora_ckpt} does not exist. aborting")
    sd.update(load_tensor_dict(lora_ckpt))


_MODEL_KEYS = ("n_heads", "n_layers", "n_codebooks", "n_conditioning_codebooks", "latent_dim", "embedding_dim",
               "vocab_size")
_DEFAULT_KW = dict(n_heads=20, n_layers=16, n_codebooks=9, n_conditioning_codebooks=0, latent_dim=8,
                   embedding_dim=1280, vocab_size=1024)          # VampNet.__init__ defaults, transformer.py:536-545


def _on_engine_stream(fn):
    """Run a public entry point on the Interface's own (non-default) HIP stream when the caller sits on the legacy default
    stream: the engine replays the forward pass as a captured hipGraph, and the default stream cannot be captured.  The
    side stream waits for the caller's stream first and the caller's stream waits for it afterwards, so the call looks
    synchronous-in-stream-order exactly like the reference's torch ops."""
    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        st = getattr(self, "_stream", None)
        if st is None or getattr(self, "_in_stream", False):
            return fn(self, *args, **kwargs)
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream != 0:                       # the caller already chose a capturable stream
            return fn(self, *args, **kwargs)
        st.wait_stream(cur)
        self._in_stream = True
        try:
            with torch.cuda.stream(st):
                out = fn(self, *args, **kwargs)
        finally:
            self._in_stream = False
        cur.wait_stream(st)
        for t in (out if isinstance(out, tuple) else (out,)):
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(cur)
        return out
    return wrapper


def signal_concat(audio_signals: list):
    """interface.py:19-25: concatenate AudioSignals along time."""
    from .codec import AudioSignal
    return AudioSignal(torch.cat([x.audio_data for x in audio_signals], dim=-1), sample_rate=audio_signals[0].sample_rate)


def _codec_codebooks(codec):
    """codec.quantizer.quantizers[i].codebook.weight, the only codec state the sampling loop reads (layers.py:145)."""
    return torch.stack([q.codebook.weight.detach().float().cpu() for q in codec.quantizer.quantizers])


class Interface:
    def __init__(self, coarse_ckpt: str = None, coarse_lora_ckpt: str = None, coarse2fine_ckpt: str = None,
                 coarse2fine_lora_ckpt: str = None, codec_ckpt: str = None,
                 wavebeat_ckpt: str = None, device: str = "cuda:0", coarse_chunk_size_s: int = 10,
                 coarse2fine_chunk_size_s: int = 3, compile=True, *, codec=None, max_batch: int = 8,
                 rng: str = "torch", process_group=None, precision: str = DEFAULT_PRECISION, exchange: str = None):
        assert codec_ckpt is not None or codec is not None, "must provide a codec checkpoint"
        assert coarse_ckpt is not None, "must provide a coarse checkpoint"
        if codec is None:
            from .codec import DacCodec
            codec = DacCodec.load(codec_ckpt, device=device)
        csd, ckw = _load_checkpoint(coarse_ckpt)
        if coarse_lora_ckpt is not None:
            _load_lora(csd, coarse_lora_ckpt)
        fsd = fkw = None
        if coarse2fine_ckpt is not None:
            fsd, fkw = _load_checkpoint(coarse2fine_ckpt)
            if coarse2fine_lora_ckpt is not None:
                _load_lora(fsd, coarse2fine_lora_ckpt)
        self._init(codec, csd, ckw, fsd, fkw, device, coarse_chunk_size_s, coarse2fine_chunk_size_s, max_batch, rng,
                   process_group, precision, exchange)
        self.coarse_path = Path(coarse_ckpt)
        self.c2f_path = Path(coarse2fine_ckpt) if coarse2fine_ckpt is not None else None
        self.codec_path = Path(codec_ckpt) if codec_ckpt is not None else None
        self._resident_put("coarse", self.coarse_path, self.coarse)
        if self.c2f is not None:
            self._resident_put("c2f", self.c2f_path, self.c2f)

    # ---- resident models: what a hot swap costs -------------------------------------------------------------------------
    # The reference app swaps models per request (app.py:181 -> load_finetuned -> reload, interface.py:134-174): every swap there is a
    # checkpoint read + .to(device).  Here a model that was loaded before STAYS packed, uploaded and split on the device (HBM is 288 GB;
    # a coarse model with its planes and workspace is ~6 GB, ~7.3 GB once an adapter swap keeps its un-merged blob), keyed by (role, file
    # path, mtime, size): swapping back to it is a dictionary lookup.  The least recently used entries beyond `max_resident` per role
    # (VN_RESIDENT_MODELS, default 4) are dropped.  An entry IS the plain checkpoint: reload() takes no adapter argument in the reference
    # either (interface.py:146-174 -> _load_model(ckpt) without lora_ckpt), so a resident model that carries adapters — from the
    # constructor's *_lora_ckpt or from load_lora() — has them removed on the device when a reload() hands it out again (tens of ms).
    @staticmethod
    def _ckpt_key(path):
        p = Path(path)
        try:
            st = p.stat()
            return (str(p.resolve()), st.st_mtime_ns, st.st_size)
        except OSError:
            return (str(p), 0, 0)

    def _resident_put(self, role, path, model):
        import collections
        import os
        if not hasattr(self, "_resident"):
            self._resident = {"coarse": collections.OrderedDict(), "c2f": collections.OrderedDict()}
            self.max_resident = max(1, int(os.environ.get("VN_RESIDENT_MODELS", "4")))
        d = self._resident[role]
        key = self._ckpt_key(path)
        d[key] = model
        d.move_to_end(key)
        while len(d) > self.max_resident:
            d.popitem(last=False)

    def _resident_get(self, role, path):
        d = getattr(self, "_resident", {}).get(role)
        if not d:
            return None
        key = self._ckpt_key(path)
        m = d.get(key)
        if m is not None:
            d.move_to_end(key)
        return m

    def load_lora(self, coarse_lora_ckpt: str = None, coarse2fine_lora_ckpt: str = None):
        """Swap the loralib ADAPTERS of the resident models (the constructor's `coarse_lora_ckpt` / `coarse2fine_lora_ckpt`,
        interface.py:55-67 + :37-46) without touching the base checkpoints: the adapters are merged into the un-merged weights on the
        device and the planes rebuilt (VampNetModel.apply_lora) — tens of milliseconds.  "" removes a model's adapters."""
        for model, ck in ((self.coarse, coarse_lora_ckpt), (self.c2f, coarse2fine_lora_ckpt)):
            if ck is None or model is None:
                continue
            sd = {}
            if ck != "":
                _load_lora(sd, ck)
            model.apply_lora(sd)

    @classmethod
    def from_state_dicts(cls, codec, coarse_sd, coarse_kwargs, c2f_sd=None, c2f_kwargs=None, device="cuda:0",
                         coarse_chunk_size_s=10, coarse2fine_chunk_size_s=3, max_batch=8, rng="torch",
                         process_group=None, precision=DEFAULT_PRECISION, exchange=None):
        """Build from in-memory reference-format state_dicts (what the checkpoints hold)."""
        self = object.__new__(cls)
        self._init(codec, coarse_sd, coarse_kwargs, c2f_sd, c2f_kwargs, device, coarse_chunk_size_s,
                   coarse2fine_chunk_size_s, max_batch, rng, process_group, precision, exchange)
        self.coarse_path = self.c2f_path = self.codec_path = None
        return self

    def _init(self, codec, csd, ckw, fsd, fkw, device, coarse_chunk_s, c2f_chunk_s, max_batch, rng, process_group,
              precision=DEFAULT_PRECISION, exchange=None):
        self.precision = precision
        self.codec = codec
        self.device = torch.device(device)
        self.engine = Engine(device)
        self._stream = torch.cuda.Stream(self.engine.device)      # see _on_engine_stream
        self._in_stream = False
        self.loudness = -24.0
        self.beat_tracker = None
        # build_mask on the GPU (draws included, RNG-exact: masks.build_mask_device) instead of the host twin of vampnet/mask.py;
        # both give the same mask and leave torch's CPU generator at the same position
        self.mask_on_device = os.environ.get("VN_MASK_ON_DEVICE", "1") != "0"     # VN_MASK_ON_DEVICE=0: the host twin (A/B)
        self.rng = rng
        self.max_batch = max_batch
        self.pg = process_group
        self.exchange_log = None          # set to a list to have _allgather_batch record (start, end) events
        self._call_idx = 0
        if process_group is not None:
            import torch.distributed as dist
            self.rank, self.world = dist.get_rank(process_group), dist.get_world_size(process_group)
        else:
            self.rank, self.world = 0, 1
        # the exchange step: "torch" (default) = torch.distributed's all_gather_into_tensor on the group ("nccl" IS RCCL on ROCm; gloo in the
        # CPU tests); "c_abi" = the library's own RCCL communicator behind vn_allgather_tokens (include/vampnet_hip.h: what a host without
        # torch.distributed binds) — the process group is then only the channel that carries the 128-byte unique id to the ranks
        self.exchange = exchange or os.environ.get("VN_EXCHANGE", "torch")
        if self.exchange not in ("torch", "c_abi"):
            raise ValueError("exchange must be 'torch' or 'c_abi'")
        self._comm = None
        if process_group is not None and self.exchange == "c_abi":
            self._comm = self._make_comm()
        self._codebooks = _codec_codebooks(codec)
        self.coarse = self._make_model(csd, ckw, coarse_chunk_s)
        # the coarse-to-fine chunks of one coarse chunk are batched into one launch: size its workspace for them
        self.c2f = self._make_model(fsd, fkw, c2f_chunk_s, self._c2f_max_batch(coarse_chunk_s, c2f_chunk_s)) if fsd is not None else None

    @property
    def effective_precision(self):
        """the precision each model ACTUALLY runs in: an opt-in "f16x2" model that met a value outside fp16's range has moved itself
        to "bf16x3" (engine.PrecisionFallbackWarning)"""
        return {"coarse": self.coarse.precision, "c2f": self.c2f.precision if self.c2f is not None else None}

    def _c2f_max_batch(self, coarse_chunk_s, c2f_chunk_s):
        """workspace rows of the c2f model: every coarse-to-fine chunk of one coarse chunk, for every item, in one launch"""
        return self.max_batch * math.ceil(self.s2t(coarse_chunk_s) / self.s2t(c2f_chunk_s))

    def _make_model(self, sd, kw, chunk_s, max_batch=None):
        kwargs = dict(_DEFAULT_KW)
        kwargs.update({k: v for k, v in (kw or {}).items() if k in _MODEL_KEYS})
        return VampNetModel(self.engine, sd, self._codebooks, max_batch=max_batch or self.max_batch,
                            max_T=self.s2t(chunk_s), chunk_size_s=chunk_s, precision=self.precision, **kwargs)

    # ---- reference API that needs the network / other models -----------------------------------
    @classmethod
    def default(cls):
        raise RuntimeError("Interface.default() downloads checkpoints from the HF hub (vampnet/__init__.py:20-47); "
                           "pass local checkpoint paths instead")

    # name -> (coarse_ckpt, c2f_ckpt): the local stand-in for the HF-hub model zoo behind vampnet.list_finetuned /
    # download_finetuned (vampnet/__init__.py:20-47); fill it with Interface.register_model
    model_zoo = {}

    @classmethod
    def register_model(cls, name: str, coarse_ckpt: str, c2f_ckpt: str = None):
        cls.model_zoo[name] = (coarse_ckpt, c2f_ckpt)

    @classmethod
    def available_models(cls):
        """interface.py:128-131: the fine-tuned model names + "default"."""
        return [n for n in cls.model_zoo if n != "default"] + ["default"]

    def load_finetuned(self, name: str):
        """interface.py:134-144 with the registry instead of a download; "default" without a registered path keeps the
        weights that are loaded."""
        assert name in self.available_models(), f"{name} is not a valid model name"
        if name not in self.model_zoo:
            if name == "default":
                return
            raise RuntimeError("load_finetuned() needs the HF hub; register local checkpoints with Interface.register_model")
        coarse, c2f = self.model_zoo[name]
        self.reload(coarse_ckpt=coarse, c2f_ckpt=c2f)

    def reload(self, coarse_ckpt: str = None, c2f_ckpt: str = None):
        """Hot-swap weights (interface.py:146-174); no-op when the path is already loaded."""
        if coarse_ckpt is not None and self.coarse_path != Path(coarse_ckpt):
            hit = self._resident_get("coarse", coarse_ckpt)
            if hit is not None:                            # loaded before and still resident: nothing to read, pack, upload or split
                hit.chunk_size_s = self.coarse.chunk_size_s
                hit.drop_adapters()
                self.coarse = hit
            else:
                sd, kw = _load_checkpoint(coarse_ckpt)
                self.coarse = self._make_model(sd, kw, self.coarse.chunk_size_s)
                self._resident_put("coarse", coarse_ckpt, self.coarse)
            self.coarse_path = Path(coarse_ckpt)
        if c2f_ckpt is not None and self.c2f_path != Path(c2f_ckpt):
            chunk_s = self.c2f.chunk_size_s if self.c2f is not None else 3
            hit = self._resident_get("c2f", c2f_ckpt)
            if hit is not None:
                hit.chunk_size_s = chunk_s
                hit.drop_adapters()
                self.c2f = hit
            else:
                sd, kw = _load_checkpoint(c2f_ckpt)
                self.c2f = self._make_model(sd, kw, chunk_s, self._c2f_max_batch(self.coarse.chunk_size_s, chunk_s))
                self._resident_put("c2f", c2f_ckpt, self.c2f)
            self.c2f_path = Path(c2f_ckpt)

    # ---- unit conversion (interface.py:176-189) -------------------------------------------------
    def s2t(self, seconds: float):
        return masks.seconds_to_tokens(seconds, self.codec.sample_rate, self.codec.hop_length)

    def s2t2s(self, seconds: float):
        return self.t2s(self.s2t(seconds))

    def t2s(self, tokens: int):
        return tokens * self.codec.hop_length / self.codec.sample_rate

    def to(self, device):
        if torch.device(device) != self.device:
            raise RuntimeError("the engine is bound to its GPU at construction; build a new Interface instead")
        return self

    def set_chunk_size(self, chunk_size_s: float):
        """interface.py:324-325 — note the coarse workspace was sized for the construction-time chunk."""
        if self.s2t(chunk_size_s) > self.coarse.dims.max_T:
            raise ValueError("chunk larger than the workspace allocated at construction")
        self.coarse.chunk_size_s = chunk_size_s

    # ---- codec --------------------------------------------------------------------------------
    def _preprocess(self, signal):
        """interface.py:206-217: resample to the codec rate, mono, loudness-normalise to self.loudness, peak-limit, pad."""
        if not hasattr(self.codec, "preprocess_signal"):
            raise RuntimeError("this codec object cannot preprocess audio (synthetic-codebook stand-in)")
        return self.codec.preprocess_signal(signal, loudness=self.loudness)

    @torch.inference_mode()
    def encode(self, signal):
        """interface.py:219-224 (incl. _preprocess :206-217) — delegated to the codec object."""
        if not hasattr(self.codec, "encode_signal"):
            raise RuntimeError("this codec object cannot encode audio (synthetic-codebook stand-in)")
        return self.codec.encode_signal(signal, loudness=self.loudness).to(self.device)

    @torch.inference_mode()
    def decode(self, z: torch.Tensor):
        """interface.py:203-204 -> VampNet.decode (transformer.py:661-684): MASK -> 0, codes -> audio."""
        if not hasattr(self.codec, "decode_signal"):
            raise RuntimeError("this codec object cannot decode audio (synthetic-codebook stand-in)")
        assert z.ndim == 3
        z = z.masked_fill(z == self.coarse.mask_token, 0)
        return self.codec.decode_signal(z)          # AudioSignal(audio (B,1,T*hop), codec.sample_rate)

    # ---- beats (interface.py:226-321): `beat_tracker` is any object with extract_beats(signal) -> (beats_s, downbeats_s);
    # the reference's WaveBeat model is out of scope, the mask arithmetic around it is not
    def snap_to_beats(self, signal):
        assert getattr(self, "beat_tracker", None) is not None, "No beat tracker loaded"
        beats, downbeats = self.beat_tracker.extract_beats(signal)
        samples_begin = int(beats[0] * signal.sample_rate)
        samples_end = int(beats[-1] * signal.sample_rate)
        return signal.clone().trim(samples_begin, signal.length - samples_end)

    def make_beat_mask(self, signal, before_beat_s: float = 0.0, after_beat_s: float = 0.02, mask_downbeats: bool = True,
                       mask_upbeats: bool = True, downbeat_downsample_factor: int = None, beat_downsample_factor: int = None,
                       dropout: float = 0.0, invert: bool = True):
        assert self.beat_tracker is not None, "No beat tracker loaded"
        beats, downbeats = self.beat_tracker.extract_beats(signal)
        n_cb = self.c2f.n_codebooks if self.c2f is not None else self.coarse.n_codebooks
        return masks.beat_mask(beats, downbeats, signal.duration, self.codec.sample_rate, self.codec.hop_length, n_cb,
                               before_beat_s, after_beat_s, mask_downbeats, mask_upbeats, downbeat_downsample_factor,
                               beat_downsample_factor, dropout, invert).to(self.device)

    def visualize_codes(self, z: torch.Tensor):
        """interface.py:564-573 (needs matplotlib, like the reference)."""
        import matplotlib.pyplot as plt
        fig = plt.figure(figsize=(10, 7))
        fig.add_subplot(2, 1, 1)
        plt.imshow(z[0].cpu().numpy(), aspect="auto", origin="lower", cmap="tab20", interpolation="none")
        plt.title("codes")
        plt.ylabel("codebook index")

    # ---- masks --------------------------------------------------------------------------------
    def build_mask(self, z: torch.Tensor, sig=None, rand_mask_intensity: float = 1.0, prefix_s: float = 0.0,
                   suffix_s: float = 0.0, periodic_prompt: int = 7, periodic_prompt_width: int = 1,
                   onset_mask_width: int = 0, _dropout: float = 0.0, upper_codebook_mask: int = 3, ncc: int = 0):
        """interface.py:454-489.  RNG consumption identical to the reference on CPU (vampnet_amd/masks.py)."""
        onset = None
        if onset_mask_width > 0:
            assert sig is not None, "must provide a signal to use onset mask"
            onset = masks.onset_mask(sig, z, self, width=onset_mask_width)
        if getattr(self, "mask_on_device", False):   # same mask, same generator position, no host loop (csrc/elementwise.hip: vn_build_mask_kernel)
            return masks.build_mask_device(self.engine, z, rand_mask_intensity=rand_mask_intensity, n_prefix=self.s2t(prefix_s),
                                           n_suffix=self.s2t(suffix_s), periodic_prompt=periodic_prompt,
                                           periodic_prompt_width=periodic_prompt_width, onset_mask=onset, dropout=_dropout,
                                           upper_codebook_mask=upper_codebook_mask, ncc=ncc)
        return masks.build_mask(z, rand_mask_intensity=rand_mask_intensity, n_prefix=self.s2t(prefix_s),
                                n_suffix=self.s2t(suffix_s), periodic_prompt=periodic_prompt,
                                periodic_prompt_width=periodic_prompt_width, onset_mask=onset, dropout=_dropout,
                                upper_codebook_mask=upper_codebook_mask, ncc=ncc)

    # ---- batch sharding helpers ---------------------------------------------------------------
    def _shard(self, B):
        """Contiguous block partition of B batch items over the ranks: item b -> rank b // ceil(B / world)."""
        per = math.ceil(B / self.world)
        b0 = min(self.rank * per, B)
        return b0, min(b0 + per, B)

    def _generate(self, model, start_tokens, mask, gen_fn=None, **kwargs):
        """model.generate on this rank's slice of the (global) batch with the global N0."""
        rng = kwargs.pop("rng", self.rng)
        if rng == "device" and kwargs.get("device_seed") is not None:
            # one independent Philox stream per generate() call of a vamp(): chunks / stages must not share noise
            kwargs["device_seed"] = (int(kwargs["device_seed"]) * 0x9E3779B97F4A7C15 + self._call_idx) & (2 ** 64 - 1)
            self._call_idx += 1
        elif rng != "device":
            kwargs.pop("device_seed", None)
        if self.world == 1:
            if gen_fn is not None:
                return gen_fn(codec=self.codec, start_tokens=start_tokens, mask=mask, return_signal=False, **kwargs)
            return model.generate(codec=self.codec, start_tokens=start_tokens, mask=mask, return_signal=False,
                                  rng=rng, **kwargs)
        B = start_tokens.shape[0]
        b0, b1 = self._shard(B)
        if mask is None:
            mask = torch.ones_like(start_tokens)
            mask[:, :model.n_conditioning_codebooks, :] = 0
        n0 = int(((mask != 0) | (start_tokens == model.mask_token)).sum().item())        # global, transformer.py:766
        # rows of other ranks: a MASK-free stand-in (masked slots -> token 0, others keep the input) so that the
        # next stage's global N0 = #(mask | token == MASK) equals what the real tokens would give
        out = torch.where(mask != 0, torch.zeros_like(start_tokens), start_tokens)
        if b1 > b0:
            out[b0:b1] = model.generate(codec=self.codec, start_tokens=start_tokens[b0:b1], mask=mask[b0:b1],
                                        return_signal=False, rng=rng, n0_override=n0, global_batch=B,
                                        batch_offset=b0, **kwargs)
        return out          # rows outside [b0, b1) are stand-ins until vamp()'s final all-gather

    def _generate_calls(self, model, starts, masks_, **kwargs):
        """Several reference generate() calls of identical shape (the coarse-to-fine chunks, interface.py:360-374)
        executed as one device batch.  Call c keeps its own batch-wide N0 (transformer.py:766), its own slice of the
        torch noise stream (drawn call after call, in the reference's order) or its own Philox stream."""
        rng = kwargs.pop("rng", self.rng)
        dseed = kwargs.pop("device_seed", None)
        if kwargs.get("seed") is not None:
            raise ValueError("seed= reseeds per call; batched calls take the ambient generator (c2f never passes seed)")
        steps = int(kwargs.get("_sampling_steps", 12))
        cutoff = kwargs.get("sample_cutoff", 1.0)
        nC = len(starts)
        Bg = starts[0].shape[0]
        b0, b1 = self._shard(Bg)
        nb = b1 - b0
        n0s, exps, unifs, zs, ms = [], [], [], [], []
        noise = None
        if rng == "torch_device":
            # seed-exact at device speed: the words of ALL these calls are generated on the generator's side stream, straight into the
            # batch's ledger, BEFORE anything below asks the device for a value (the N0 counts wait for the previous stage) — the
            # draw shapes are static, so the c2f stage's noise is produced underneath the coarse stage's forwards
            from .torch_rng import draw_noise_device_calls
            noise = draw_noise_device_calls(self.engine.torch_rng(), nC, Bg, starts[0].shape[-1] * model.n_predict_codebooks,
                                            model.vocab_size, steps, cutoff, b0, nb)
        for st, mk in zip(starts, masks_):
            if mk is None:
                mk = torch.ones_like(st)
                mk[:, :model.n_conditioning_codebooks, :] = 0
            n0 = int(((mk != 0) | (st == model.mask_token)).sum().item())              # GLOBAL batch of this call
            n0s += [n0] * nb
            if rng == "torch":
                e, u = model.draw_noise(Bg, st.shape[-1], steps, cutoff, b0, nb, pin=False)
                exps.append(e)
                unifs.append(u)
            zs.append(st[b0:b1])
            ms.append(mk[b0:b1])
        if rng == "torch":
            N = unifs[0].shape[-1]
            exp = torch.stack([e.view(steps, nb, N, -1) for e in exps], dim=1).reshape(steps, nC * nb * N, -1)
            unif = torch.stack(unifs, dim=1).reshape(steps, nC * nb, N)
            pin = torch.cuda.is_available() and not exp.is_cuda
            noise = (exp.pin_memory(), unif.pin_memory()) if pin else (exp, unif)
        elif dseed is not None:
            dseed = (int(dseed) * 0x9E3779B97F4A7C15 + self._call_idx) & (2 ** 64 - 1)
            self._call_idx += nC
        outs = []
        if nb > 0:
            res = model.generate(codec=self.codec, start_tokens=torch.cat(zs).contiguous(), mask=torch.cat(ms).contiguous(),
                                 return_signal=False, rng=rng, n0_override=n0s, noise=noise, device_seed=dseed,
                                 batch_offset=b0, call_batch=nb, global_batch=Bg, **kwargs)
            res = res.view(nC, nb, *res.shape[1:])
        for c, (st, mk) in enumerate(zip(starts, masks_)):
            if mk is None:
                mk = torch.ones_like(st)
                mk[:, :model.n_conditioning_codebooks, :] = 0
            out = torch.where(mk != 0, torch.zeros_like(st), st)        # stand-in rows for other ranks' items
            if nb > 0:
                out[b0:b1] = res[c]
            outs.append(out)
        return outs

    def _make_comm(self):
        """vn_comm over the ranks of the process group: rank 0 makes the RCCL unique id, the group carries it to the others"""
        import ctypes as C
        import torch.distributed as dist
        eng = self.engine
        ident = (C.c_uint8 * 128)()
        if self.rank == 0:
            eng.check(eng.lib.vn_comm_unique_id(eng.handle, ident), "vn_comm_unique_id")
        box = [bytes(ident)]
        dist.broadcast_object_list(box, src=dist.get_global_rank(self.pg, 0), group=self.pg)
        ident = (C.c_uint8 * 128).from_buffer_copy(box[0])
        h = C.c_void_p()
        eng.check(eng.lib.vn_comm_create(eng.handle, ident, self.rank, self.world, C.byref(h)), "vn_comm_create")
        return h

    def __del__(self):
        try:
            if getattr(self, "_comm", None):
                self.engine.lib.vn_comm_destroy(self._comm)
                self._comm = None
        except Exception:
            pass

    def _allgather_batch(self, z):
        """The single exchange step: every rank contributes its block of batch items (RCCL all-gather over xGMI).  Runs whenever a
        process group was given — also a one-rank group, so that the device collective (padding to `per * world` rows included) is the
        same code on one GPU as on eight.  `exchange_log` (a list, or None): (start, end) torch.cuda.Event pairs of every exchange,
        for bench.py's `exchange_ms`."""
        if self.pg is None:
            return z
        import torch.distributed as dist
        B = z.shape[0]
        per = math.ceil(B / self.world)
        b0, b1 = self._shard(B)
        log = getattr(self, "exchange_log", None)
        if log is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        local = torch.zeros((per,) + tuple(z.shape[1:]), dtype=z.dtype, device=z.device)
        local[:b1 - b0] = z[b0:b1]
        full = torch.empty((per * self.world,) + tuple(z.shape[1:]), dtype=z.dtype, device=z.device)
        if getattr(self, "_comm", None) is not None:          # the library's own RCCL communicator, enqueued on the current stream (no torch collective)
            assert z.dtype == torch.int64 and z.is_cuda
            self.engine.check(self.engine.lib.vn_allgather_tokens(self._comm, local.data_ptr(), full.data_ptr(), local.numel(),
                                                                  self.engine.stream()), "vn_allgather_tokens")
        else:
            dist.all_gather_into_tensor(full, local, group=self.pg)
        out = full[:B].contiguous()
        if log is not None:
            ev1.record()
            log.append((ev0, ev1))
        return out

    # ---- the hot path --------------------------------------------------------------------------
    @torch.inference_mode()
    @_on_engine_stream
    def coarse_to_fine(self, z: torch.Tensor, mask: torch.Tensor = None, return_mask: bool = False, **kwargs):
        """interface.py:328-380: pad T to a multiple of the c2f chunk (z <- 0, mask <- 1), condition on the
        coarse codebooks, generate each chunk independently, trim."""
        assert self.c2f is not None, "No coarse2fine model loaded"
        c2f = self.c2f
        length = z.shape[-1]
        chunk_len = self.s2t(c2f.chunk_size_s)
        n_chunks = math.ceil(length / chunk_len)
        pad = (-length) % chunk_len
        if pad:
            z = torch.nn.functional.pad(z, (0, pad))
            mask = torch.nn.functional.pad(mask, (0, pad), value=1) if mask is not None else None
        missing = c2f.n_codebooks - z.shape[1]
        if missing > 0:
            z = torch.cat([z, torch.zeros(z.shape[0], missing, z.shape[-1], dtype=torch.long, device=z.device)], dim=1)
        if mask is not None:
            mask = mask.clone()
            mask[:, :c2f.n_conditioning_codebooks, :] = 0
        chunks = [z[:, :, i * chunk_len:(i + 1) * chunk_len] for i in range(n_chunks)]
        mchunks = [mask[:, :, i * chunk_len:(i + 1) * chunk_len] if mask is not None else None for i in range(n_chunks)]
        if n_chunks > 1 and hasattr(c2f, "generate_batched_calls"):
            # the chunks are independent generate() calls of equal length: run them as ONE device batch of
            # n_chunks*B items (4x larger GEMMs), each item keeping its own call's N0 and noise stream
            b0, b1 = self._shard(z.shape[0])
            grp = max(1, c2f.dims.max_batch // max(1, b1 - b0))         # calls per launch that fit the workspace
            pieces = []
            for g0 in range(0, n_chunks, grp):
                pieces += self._generate_calls(c2f, chunks[g0:g0 + grp], mchunks[g0:g0 + grp], time_steps=chunk_len,
                                               cfg_guidance=None, **dict(kwargs))
        else:
            pieces = [self._generate(c2f, ch.contiguous(), m.contiguous() if m is not None else None,
                                     time_steps=chunk_len, cfg_guidance=None, **kwargs) for ch, m in zip(chunks, mchunks)]
        fine_z = torch.cat(pieces, dim=-1)
        if return_mask:
            return (fine_z[:, :, :length].clone(),
                    masks.apply_mask(fine_z, mask, c2f.mask_token)[0][:, :, :length].clone())
        return fine_z[:, :, :length].clone()

    @torch.inference_mode()
    @_on_engine_stream
    def coarse_vamp(self, z, mask, return_mask=False, gen_fn=None, **kwargs):
        """interface.py:383-452: coarse codebooks only, 10 s chunks generated independently; the first and last
        timestep of every chunk that has any unmasked token are forced unmasked (:407-413)."""
        coarse = self.coarse
        nC = coarse.n_codebooks
        cz = z[:, :nC, :].clone()
        mask = mask[:, :nC, :]
        chunk_len = self.s2t(coarse.chunk_size_s)
        n_chunks = math.ceil(cz.shape[-1] / chunk_len)
        masked_parts, vamped_parts = [], []
        for i in range(n_chunks):
            sl = slice(i * chunk_len, (i + 1) * chunk_len)
            chunk, mchunk = cz[:, :, sl], mask[:, :, sl]
            if bool(torch.any(mchunk == 0)):
                mchunk = mchunk.clone()
                mchunk[:, :, 0] = 0
                mchunk[:, :, -1] = 0
            chunk_masked, mchunk = masks.apply_mask(chunk, mchunk, coarse.mask_token)
            masked_parts.append(chunk_masked)
            vamped_parts.append(self._generate(coarse, chunk_masked.contiguous(), mchunk.contiguous(), gen_fn=gen_fn,
                                               time_steps=chunk_len, **kwargs))
        cz_masked = torch.cat(masked_parts, dim=-1)
        c_vamp = torch.cat(vamped_parts, dim=-1)
        c_vamp = torch.cat([c_vamp, z[:, nC:, :]], dim=1)
        if return_mask:
            return c_vamp, cz_masked
        return c_vamp

    @_on_engine_stream
    def vamp(self, codes: torch.Tensor, mask: torch.Tensor, batch_size: int = 1, feedback_steps: int = 1,
             time_stretch_factor: int = 1, return_mask: bool = False, **kwargs):
        """interface.py:491-562.  `kwargs` reach only the coarse stage; c2f always runs 2 steps at temperature 1
        (SURVEY.md §0 fact 6)."""
        rng = kwargs.get("rng", self.rng)
        self._call_idx = 0
        z = codes.to(self.device).expand(batch_size, -1, -1)
        mask = mask.to(self.device).expand(batch_size, -1, -1)
        if time_stretch_factor > 1:
            z = z.repeat_interleave(time_stretch_factor, dim=-1)
            mask = mask.repeat_interleave(time_stretch_factor, dim=-1)
            added = torch.ones_like(mask)
            added[:, :, ::time_stretch_factor] = 0
            mask = (mask.bool() | added.bool()).long()
        zv = z
        for i in range(feedback_steps):
            zv, mask_z = self.coarse_vamp(zv, mask=mask, return_mask=True, **kwargs)
            mask_z = mask_z.roll(shifts=(i + 1) % feedback_steps, dims=-1)
        if zv.shape[1] < z.shape[1]:
            zv = torch.cat([zv, z[:, self.coarse.n_codebooks:, :]], dim=1)
        zv, fine_mask = self.coarse_to_fine(zv, mask=mask, typical_filtering=True, _sampling_steps=2,
                                            return_mask=True, rng=rng, device_seed=kwargs.get("device_seed"))
        zv = self._allgather_batch(zv)
        if self.world > 1:      # masked view of the gathered tokens (apply_mask is elementwise)
            full_mask = mask.clone()
            full_mask[:, :self.c2f.n_conditioning_codebooks, :] = 0
            fine_mask = masks.apply_mask(zv, full_mask, self.c2f.mask_token)[0]
        mask_z = torch.cat([mask_z[:, :self.coarse.n_codebooks, :], fine_mask[:, self.coarse.n_codebooks:, :]], dim=1)
        if return_mask:
            return zv, mask_z.cpu()
        return zv
