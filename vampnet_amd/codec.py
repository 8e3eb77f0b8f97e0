"""DAC / `lac.LAC` codec on the MI355X engine — what `Interface.encode` / `Interface.decode` call
(reference call sites: interface.py:215,223; transformer.py:661-684; layers.py:145).

PARITY UNPINNED: the codec's source (`lac`, an unpinned git dependency) and weights are not available; this
implements the published Descript-Audio-Codec layer graph and is tested against
oracle/dac_oracle.py on seeded synthetic weights.  Hyper-parameters come from the checkpoint's metadata kwargs.

Host side = layer graph + weight packing (weight-norm folded, channels-last tap-major conv weights, transposed
convolutions split into `stride` polyphase 2-tap convolutions); every FLOP runs in libvampnet_hip.so:
vn_conv1d_f32 (exact-f32 MFMA implicit GEMM with fused bias / residual / next-layer Snake), vn_dac_conv_in/out,
vn_rvq_encode/decode.  Activations are channels-last [B][T][C] fp32 on the device.
"""
import math
import os
from pathlib import Path

import numpy as np
import torch

from . import _lib
from .engine import DEFAULT_PRECISION, Engine

DEFAULT_CFG = dict(encoder_dim=64, encoder_rates=[2, 4, 8, 12], latent_dim=None, decoder_dim=1536,
                   decoder_rates=[12, 8, 4, 2], n_codebooks=14, codebook_size=1024, codebook_dim=8, sample_rate=44100)


def normalize_codec_kwargs(kw: dict) -> dict:
    """metadata.kwargs of a DAC-family checkpoint -> the subset DacCodec consumes."""
    cfg = {k: v for k, v in (kw or {}).items() if k in DEFAULT_CFG}
    cd = cfg.get("codebook_dim")
    if isinstance(cd, (list, tuple)):
        if len(set(cd)) != 1:
            raise ValueError(f"per-level codebook_dim {list(cd)} is not supported (one width for all levels)")
        cfg["codebook_dim"] = int(cd[0])
    for k in ("encoder_rates", "decoder_rates"):
        if k in cfg:
            cfg[k] = [int(r) for r in cfg[k]]
    return cfg


def validate_codec_state_dict(sd: dict, cfg: dict):
    """Shape check of a codec state_dict against its kwargs before any kernel sees it: a clear error instead of a KeyError
    deep in the packer when the file belongs to another architecture (weight-normed convs hold weight_g / weight_v)."""
    def shape(key):
        for k in (key + ".weight_v", key + ".weight"):
            if k in sd:
                return tuple(sd[k].shape)
        raise ValueError(f"codec checkpoint has no tensor {key}.weight[_v]: not a DAC-family state_dict")
    ed, n = int(cfg["encoder_dim"]), int(cfg["n_codebooks"])
    if shape("encoder.block.0") != (ed, 1, 7):
        raise ValueError(f"encoder stem is {shape('encoder.block.0')}, kwargs say encoder_dim={ed}")
    latent = cfg["latent_dim"] or ed * 2 ** len(cfg["encoder_rates"])
    levels = sorted({int(k.split(".")[2]) for k in sd if k.startswith("quantizer.quantizers.")})
    if levels != list(range(n)):
        raise ValueError(f"checkpoint holds quantizer levels {levels}, kwargs say n_codebooks={n}")
    for i in range(n):
        cb = tuple(sd[f"quantizer.quantizers.{i}.codebook.weight"].shape)
        if cb != (int(cfg["codebook_size"]), int(cfg["codebook_dim"])):
            raise ValueError(f"codebook {i} is {cb}, kwargs say ({cfg['codebook_size']}, {cfg['codebook_dim']})")
        if shape(f"quantizer.quantizers.{i}.in_proj")[:2] != (int(cfg["codebook_dim"]), latent):
            raise ValueError(f"quantizer {i} in_proj is {shape(f'quantizer.quantizers.{i}.in_proj')}, latent_dim should be {latent}")
    if shape("decoder.model.0")[:2] != (int(cfg["decoder_dim"]), latent):
        raise ValueError(f"decoder stem is {shape('decoder.model.0')}, kwargs say decoder_dim={cfg['decoder_dim']}, latent {latent}")


def codec_cfg_struct(cfg: dict):
    """kwargs of a DAC-family codec -> the vn_codec_cfg a C host hands to vn_codec_create_from_weights (include/vampnet_hip.h)"""
    cfg = dict(DEFAULT_CFG, **(cfg or {}))
    c = _lib.vn_codec_cfg()
    c.encoder_dim, c.decoder_dim = int(cfg["encoder_dim"]), int(cfg["decoder_dim"])
    c.n_rates = len(cfg["encoder_rates"])
    assert len(cfg["decoder_rates"]) == c.n_rates <= 8
    for i, (a, b) in enumerate(zip(cfg["encoder_rates"], cfg["decoder_rates"])):
        c.encoder_rates[i], c.decoder_rates[i] = int(a), int(b)
    c.n_codebooks, c.codebook_size, c.codebook_dim = int(cfg["n_codebooks"]), int(cfg["codebook_size"]), int(cfg["codebook_dim"])
    c.latent_dim = int(cfg["latent_dim"] or 0)
    return c


def pack_codec_blob(lib, cfg_struct, sd: dict) -> torch.Tensor:
    """The flat fp32 blob vn_codec_create_from_weights reads (host tensor): every tensor the library names (vn_codec_tensor_name) copied
    from the state_dict under that name, weight-norm folded (`<conv>.weight` = g v / ||v|| where the file holds weight_g / weight_v) —
    the only arithmetic a C host has to do itself; layouts stay PyTorch's."""
    import ctypes as C
    n, cnt = C.c_int64(), C.c_int()
    if lib.vn_codec_weights_size(C.byref(cfg_struct), C.byref(n)) or lib.vn_codec_tensor_count(C.byref(cfg_struct), C.byref(cnt)):
        raise ValueError("vn_codec_weights_size rejected the configuration")
    blob = torch.zeros(n.value, dtype=torch.float32)
    name = C.create_string_buffer(256)
    off, num = C.c_int64(), C.c_int64()
    for i in range(cnt.value):
        if lib.vn_codec_tensor_name(C.byref(cfg_struct), i, name, 256, C.byref(off), C.byref(num)):
            raise ValueError(f"vn_codec_tensor_name({i}) failed")
        key = name.value.decode()
        if key.endswith(".weight") and key not in sd:
            t = _fold(sd, key[:-len(".weight")])
        else:
            t = sd[key].float()
        if t.numel() != num.value:
            raise ValueError(f"codec tensor {key}: the checkpoint holds {tuple(t.shape)}, the configuration expects {num.value} values")
        blob[off.value:off.value + num.value] = t.contiguous().reshape(-1)
    return blob


class AudioSignal:
    """Minimal stand-in for audiotools.AudioSignal (not installed): samples (B, C, T) + sample_rate."""

    def __init__(self, samples, sample_rate=44100):
        if isinstance(samples, np.ndarray):
            samples = torch.from_numpy(samples)
        if samples.ndim == 1:
            samples = samples[None, None]
        elif samples.ndim == 2:
            samples = samples[None]
        self.samples = samples.float()
        self.sample_rate = int(sample_rate)

    @property
    def audio_data(self):
        return self.samples

    @property
    def duration(self):
        return self.samples.shape[-1] / self.sample_rate

    def cpu(self):
        return AudioSignal(self.samples.cpu(), self.sample_rate)

    def to(self, device):
        return AudioSignal(self.samples.to(device), self.sample_rate)

    def clone(self):
        return AudioSignal(self.samples.clone(), self.sample_rate)

    @property
    def batch_size(self):
        return self.samples.shape[0]

    @property
    def length(self):
        return self.samples.shape[-1]

    def trim(self, before: int, after: int):
        """Drop `before` samples from the start and `after` from the end (audiotools' `AudioSignal.trim`)."""
        end = self.samples.shape[-1] - after if after else self.samples.shape[-1]
        self.samples = self.samples[..., before:end]
        return self

    def to_mono(self):
        self.samples = self.samples.mean(dim=1, keepdim=True)
        return self

    def loudness(self):
        """Integrated loudness (LUFS) per batch item, tensor (B,) — audiotools' `AudioSignal.loudness()` [UNVERIFIED-DEP]."""
        x = self.samples.detach().float().cpu().numpy()
        return torch.tensor([integrated_loudness(x[b], self.sample_rate) for b in range(x.shape[0])])

    def normalize(self, db=-24.0):
        """Scale every item to `db` LUFS (scalar or per-item tensor) — audiotools' `normalize` [UNVERIFIED-DEP]."""
        target = torch.as_tensor(db, dtype=torch.float32).reshape(-1).expand(self.batch_size)
        have = self.loudness()
        gain = torch.where(have > -70.0, target - have, torch.zeros(self.batch_size))      # silence stays silent
        self.samples = self.samples * (10.0 ** (gain / 20.0)).to(self.samples.device)[:, None, None]
        return self

    def write(self, path):
        import wave
        x = (self.samples[0].clamp(-1, 1).cpu().numpy().T * 32767.0).astype("<i2")
        with wave.open(str(path), "wb") as f:
            f.setnchannels(x.shape[1]); f.setsampwidth(2); f.setframerate(self.sample_rate)
            f.writeframes(x.tobytes())
        return self

    @classmethod
    def from_wav(cls, path):
        import wave
        with wave.open(str(path), "rb") as f:
            n, ch, sr, sw = f.getnframes(), f.getnchannels(), f.getframerate(), f.getsampwidth()
            raw = f.readframes(n)
        assert sw == 2, "16-bit PCM only"
        x = np.frombuffer(raw, dtype="<i2").reshape(-1, ch).T.astype(np.float32) / 32768.0
        return cls(torch.from_numpy(x.copy())[None], sr)


def _k_weighting(sr):
    """ITU-R BS.1770 K-weighting biquads (pre-filter shelf + RLB high-pass), bilinear-designed for `sr`."""
    f0, G, Q = 1681.974450955533, 3.999843853973347, 0.7071752369554196
    K = math.tan(math.pi * f0 / sr)
    Vh = 10 ** (G / 20)
    Vb = Vh ** 0.4996667741545416
    a0 = 1 + K / Q + K * K
    b1 = [(Vh + Vb * K / Q + K * K) / a0, 2 * (K * K - Vh) / a0, (Vh - Vb * K / Q + K * K) / a0]
    a1 = [1.0, 2 * (K * K - 1) / a0, (1 - K / Q + K * K) / a0]
    f0, Q = 38.13547087602444, 0.5003270373238773
    K = math.tan(math.pi * f0 / sr)
    a0 = 1 + K / Q + K * K
    b2 = [1.0, -2.0, 1.0]
    a2 = [1.0, 2 * (K * K - 1) / a0, (1 - K / Q + K * K) / a0]
    return (b1, a1), (b2, a2)


def kweight_state_power(sr, n):
    """The cascade of the two K-weighting biquads (direct form II transposed, as scipy.signal.lfilter evaluates them) as a linear
    system s' = A s + B x with s = (z1a, z2a, z1b, z2b): returns A^n (4 x 4, float64) — what lets the device evaluate the IIR filters
    chunk by chunk in parallel (csrc/preprocess.hip)."""
    (b1, a1), (b2, a2) = _k_weighting(sr)

    def step(s):            # one sample with x = 0
        y1 = s[0]
        n0, n1 = -a1[1] * y1 + s[1], -a1[2] * y1
        y2 = b2[0] * y1 + s[2]
        return [n0, n1, b2[1] * y1 - a2[1] * y2 + s[3], b2[2] * y1 - a2[2] * y2]
    A = np.array([step([1.0 if i == j else 0.0 for i in range(4)]) for j in range(4)], dtype=np.float64).T
    return np.linalg.matrix_power(A, int(n))


def integrated_loudness(x, sr):
    """BS.1770-4 gated integrated loudness (LUFS) of x (C, T) on the host — what audiotools' `normalize(-24)`
    measures [UNVERIFIED-DEP]."""
    from scipy.signal import lfilter
    x = np.asarray(x, dtype=np.float64)
    for b, a in _k_weighting(sr):
        x = lfilter(b, a, x, axis=-1)
    blk, hop = int(0.4 * sr), int(0.1 * sr)
    if x.shape[-1] < blk:
        x = np.pad(x, ((0, 0), (0, blk - x.shape[-1])))
    n = 1 + (x.shape[-1] - blk) // hop
    z = np.stack([np.mean(x[:, i * hop:i * hop + blk] ** 2, axis=-1) for i in range(n)], axis=-1)   # (C, n)
    lk = -0.691 + 10 * np.log10(np.maximum(z.sum(0), 1e-12))
    keep = lk > -70.0
    if not keep.any():
        return -70.0
    rel = -0.691 + 10 * np.log10(np.maximum(z[:, keep].mean(-1).sum(), 1e-12)) - 10.0
    keep2 = keep & (lk > rel)
    if not keep2.any():
        return -70.0
    return float(-0.691 + 10 * np.log10(np.maximum(z[:, keep2].mean(-1).sum(), 1e-12)))


class _Quantizer:
    """Exposes codec.quantizer.quantizers[i].codebook.weight like lac.LAC (read at layers.py:145)."""

    class _Q:
        def __init__(self, w):
            self.codebook = type("Codebook", (), {"weight": w})()

    def __init__(self, codebooks):
        self.quantizers = [_Quantizer._Q(codebooks[i]) for i in range(codebooks.shape[0])]


def _fold(sd, key):
    if key + ".weight_v" in sd:
        return torch._weight_norm(sd[key + ".weight_v"].float(), sd[key + ".weight_g"].float(), 0)
    return sd[key + ".weight"].float()


_FAKE_BASE = 1 << 60            # "addresses" of program buffers while a program is being recorded (never dereferenced)
_PTR_IN, _PTR_OUT = 1, 2        # include/vampnet_hip.h VN_CODEC_PTR_IN / VN_CODEC_PTR_OUT


class _VBuf:
    """A buffer of a codec program under construction: shape / dtype like a tensor, an id instead of memory."""
    __slots__ = ("rec", "bid", "shape", "dtype")

    def __init__(self, rec, bid, shape, dtype):
        self.rec, self.bid, self.shape, self.dtype = rec, bid, tuple(int(v) for v in shape), dtype

    def numel(self):
        return int(np.prod(self.shape)) if self.shape else 1

    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        shape = list(shape)
        if -1 in shape:
            k = shape.index(-1)
            rest = int(np.prod([v for j, v in enumerate(shape) if j != k])) or 1
            shape[k] = self.numel() // rest
        assert int(np.prod(shape)) == self.numel()
        return _VBuf(self.rec, self.bid, shape, self.dtype)

    def contiguous(self):
        return self

    def data_ptr(self):
        return self.rec.ptr_of(self.bid)


class _Recorder:
    """Records the launches of one codec direction as a program (include/vampnet_hip.h vn_codec_op) and plans ONE arena for its
    buffers from their live ranges: a buffer's bytes are handed to later buffers once its last reader has been recorded."""

    def __init__(self):
        self.bufs = []           # dict(nbytes, first, last, io, pinned)
        self.ops = []            # (kind, [(type, value), ...])

    def new(self, shape, dtype, io=None, pinned=False):
        nbytes = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        self.bufs.append(dict(nbytes=nbytes, first=None, last=None, io=io, pinned=pinned))
        return _VBuf(self, len(self.bufs) - 1, shape, dtype)

    def ptr_of(self, bid):
        io = self.bufs[bid]["io"]
        return {None: _FAKE_BASE + bid, "in": _PTR_IN, "out": _PTR_OUT}[io]

    def emit(self, kind, typed_args):
        k = len(self.ops)
        for t, v in typed_args:
            if t == "p" and v is not None and v >= _FAKE_BASE:
                b = self.bufs[v - _FAKE_BASE]
                b["first"] = k if b["first"] is None else b["first"]
                b["last"] = k
        self.ops.append((kind, typed_args))

    def plan(self, align=256):
        """offsets of the buffers in one arena: first-fit over the blocks freed by buffers whose last use has passed"""
        order = sorted((b["first"], i) for i, b in enumerate(self.bufs) if b["first"] is not None and b["io"] is None)
        live, free, end, off = [], [], 0, {}
        for first, i in order:
            for item in [it for it in live if it[0] < first]:          # its last reader ran before this buffer's first op
                live.remove(item)
                free.append((item[1], item[2]))
            need = -(-self.bufs[i]["nbytes"] // align) * align
            free.sort()
            slot = next((j for j, (o, sz) in enumerate(free) if sz >= need), None)
            if slot is None:
                o, end = end, end + need
            else:
                o, sz = free.pop(slot)
                if sz > need:
                    free.append((o + need, sz - need))
            off[i] = o
            last = len(self.ops) if self.bufs[i]["pinned"] else self.bufs[i]["last"]
            live.append((last, o, need))
        return off, end


class _Act:
    """A channels-last activation [B][T][C] in the format(s) its consumer reads: `f32` for the fp32-input MFMA convolution
    (conv1d_f32.hip), `p16` = three exact split bf16 planes [3][B*T][C] for the bf16x3 pipe (gemm_x3.hip's implicit-GEMM mode)."""
    __slots__ = ("f32", "p16")

    def __init__(self, f32=None, p16=None):
        self.f32, self.p16 = f32, p16


class DacCodec:
    # A convolution runs on the bf16x3 pipe (six bf16-MFMA products of exact operand splits: the transformer's GEMM kernel with the
    # A rows gathered per tap) where it is MFMA-bound there.  Measured per layer on the 44.1 kHz configuration at B = 8
    # (profiles/history/r03_codec_kernel_trace_*.txt): k = 7 and strided / transposed convolutions with >= 128 output channels run 26-39 %
    # faster than on the fp32-input MFMA kernel; the k = 1 tails below 512 channels are HBM-bound and LOSE (planes are 6 bytes per
    # element against 4), and 96 / 192 output channels fill only 75 % of the 128-wide tile.  Rule: >= 128 output channels and
    # K x (tile efficiency) >= 512, K = taps * C_in.  The 64- and 96-channel audio-rate blocks, the small k = 1 tails and the
    # 1-channel stem / head stay on the fp32 kernels; every epilogue writes its Snake output in the format its consumer reads.
    X3_MIN_COUT, X3_MIN_WORK = 128, 512
    # 96 / 192-channel layers (channels-on-rows form): taps x C_in from which the layer runs on the split-plane pipe; csrc/codec_plan.hip reads
    # the same variable
    X3_MIN_WORK_CONVT = int(os.environ.get("VN_CODEC_CONVT_MIN_WORK", "512"))

    def __init__(self, sd: dict, cfg: dict = None, device="cuda:0", engine: Engine = None, precision: str = DEFAULT_PRECISION):
        cfg = dict(DEFAULT_CFG, **(cfg or {}))
        self.cfg = cfg
        if precision not in ("f16x2", "bf16x3", "f32"):
            raise ValueError("precision must be 'bf16x3' (default: exact three-way bf16 splits on the matrix cores), 'f16x2' (opt-in fast "
                             "mode, fp16's range) or 'f32'")
        self.engine = engine or Engine(device)
        self.lib = self.engine.lib
        self.device = self.engine.device
        self.sample_rate = int(cfg["sample_rate"])
        self.hop_length = int(np.prod(cfg["encoder_rates"]))
        self.latent_dim = cfg["latent_dim"] or cfg["encoder_dim"] * 2 ** len(cfg["encoder_rates"])
        self.n_codebooks = cfg["n_codebooks"]
        self._sd = sd                              # kept (host tensors): a precision fallback re-lays the weights
        # one C call per direction (vn_dac_encode / vn_dac_decode over a recorded program); VN_CODEC_EAGER=1: one ctypes call per layer
        import os
        self.use_program = os.environ.get("VN_CODEC_EAGER", "0") != "1"
        self._rec = None
        self._programs = {}
        if precision == "f16x2":
            self.engine.saturation(clear=True)
        self._build(precision)
        if precision == "f16x2":
            sat = self.engine.saturation(clear=True)
            if any(sat):                           # a weight that was clamped while its fp16 planes were built
                self._fall_back(f"a codec weight does not fit fp16 (saturation ledger {sat})")

    def _fall_back(self, why: str):
        """precision "f16x2" met a value outside fp16's range: say so and re-lay the codec on "bf16x3" for good."""
        import warnings
        from .engine import PrecisionFallbackWarning
        warnings.warn(f"DacCodec precision='f16x2': {why}; the codec runs on 'bf16x3' from here on", PrecisionFallbackWarning, stacklevel=3)
        self._build("bf16x3")

    def _guarded(self, fn, what):
        """run fn(); in precision "f16x2" read the saturation ledger afterwards and, if an activation was clamped, repeat on bf16x3.
        The ledger is per context: whatever it holds BEFORE the call belongs to somebody else (an f16x2 model on the same engine whose
        generate(return_signal=True) decodes through this codec) — it is set aside first and handed back afterwards."""
        if self.precision != "f16x2":
            return fn()
        foreign = self.engine.saturation(clear=True)
        try:
            out = fn()
            sat = self.engine.saturation(clear=True)
            if any(sat):
                self._fall_back(f"{what} left fp16's range (saturation ledger: operands {sat[0]})")
                out = fn()
                self.engine.saturation(clear=True)         # (bf16x3 writers do not report; keep the ledger clean of this call)
        finally:
            self.engine.saturation_restore(foreign)
        return out

    def _build(self, precision):
        """(re)lay the weights for `precision` (device fp32 tensors + the split-plane images of the layers that run on the matrix-core pipe)"""
        sd, cfg = self._sd, self.cfg
        self.precision = precision
        self._drop_programs()                      # they point at the previous weight images
        dev = self.device

        def conv(key):          # Conv1d (Cout, Cin, k) -> [Cout][k][Cin]
            w = _fold(sd, key)
            c = dict(w=w.permute(0, 2, 1).contiguous().to(dev), b=sd[key + ".bias"].float().to(dev),
                     cout=w.shape[0], cin=w.shape[1], k=w.shape[2])
            if self._on_x3(c):
                c["w16"] = self._tile_planes(c["w"].reshape(c["cout"], -1))
            return c

        def convT(key, s):      # ConvTranspose1d (Cin, Cout, 2s) -> per phase r: [Cout][2][Cin] = w[ci][co][r + jj*s]
            w = _fold(sd, key)
            ph = torch.stack([torch.stack([w[:, :, r], w[:, :, r + s]], dim=0).permute(2, 0, 1) for r in range(s)])
            c = dict(w=ph.contiguous().to(dev), b=sd[key + ".bias"].float().to(dev), cout=w.shape[1], cin=w.shape[0], s=s)
            if self._on_x3(c, taps=2):
                c["w16"] = [self._tile_planes(c["w"][r].reshape(c["cout"], -1)) for r in range(s)]
            return c

        def alpha(key):
            return sd[key + ".alpha"].float().reshape(-1).contiguous().to(dev)

        def res(p):
            return dict(a1=alpha(p + ".block.0"), c7=conv(p + ".block.1"), a2=alpha(p + ".block.2"), c1=conv(p + ".block.3"))

        e = {}
        w0 = _fold(sd, "encoder.block.0")
        e["stem"] = dict(w=w0.reshape(w0.shape[0], 7).contiguous().to(dev), b=sd["encoder.block.0.bias"].float().to(dev),
                         c=w0.shape[0])
        e["blocks"] = []
        for i, s in enumerate(cfg["encoder_rates"]):
            p = f"encoder.block.{1 + i}"
            e["blocks"].append(dict(res=[res(f"{p}.block.{j}") for j in range(3)], a=alpha(p + ".block.3"),
                                    down=conv(p + ".block.4"), s=s))
        n = len(cfg["encoder_rates"])
        e["a_out"] = alpha(f"encoder.block.{n + 1}")
        e["out"] = conv(f"encoder.block.{n + 2}")
        self.enc = e
        q = "quantizer.quantizers."
        L = self.latent_dim
        self.rvq = dict(
            win=torch.stack([_fold(sd, f"{q}{i}.in_proj").reshape(-1, L) for i in range(self.n_codebooks)]).contiguous().to(dev),
            bin=torch.stack([sd[f"{q}{i}.in_proj.bias"].float() for i in range(self.n_codebooks)]).contiguous().to(dev),
            cb=torch.stack([sd[f"{q}{i}.codebook.weight"].float() for i in range(self.n_codebooks)]).contiguous().to(dev),
            wout=torch.stack([_fold(sd, f"{q}{i}.out_proj").reshape(L, -1) for i in range(self.n_codebooks)]).contiguous().to(dev),
            bout=torch.stack([sd[f"{q}{i}.out_proj.bias"].float() for i in range(self.n_codebooks)]).contiguous().to(dev))
        self.quantizer = _Quantizer(torch.stack([sd[f"{q}{i}.codebook.weight"].float() for i in range(self.n_codebooks)]))
        d = {"in": conv("decoder.model.0"), "blocks": []}
        for i, s in enumerate(cfg["decoder_rates"]):
            p = f"decoder.model.{1 + i}"
            d["blocks"].append(dict(a=alpha(p + ".block.0"), up=convT(p + ".block.1", s),
                                    res=[res(f"{p}.block.{2 + j}") for j in range(3)], s=s))
        n = len(cfg["decoder_rates"])
        d["a_out"] = alpha(f"decoder.model.{n + 1}")
        wl = _fold(sd, f"decoder.model.{n + 2}")                                   # (1, C, 7) -> [7][C]
        d["head_w"] = wl[0].t().contiguous().to(dev)
        d["head_b"] = float(sd[f"decoder.model.{n + 2}.bias"].float().item())
        self.dec = d

    @classmethod
    def load(cls, path, device="cuda:0", engine=None, trusted=False):
        """`DAC.load(Path(codec_ckpt))` (interface.py:70): audiotools dict checkpoint or torch.package archive
        (vampnet_amd/checkpoint.py).  `metadata.kwargs` of a published DAC / lac checkpoint carries more keys than the conv
        stacks need (`quantizer_dropout`, ...): unknown keys are dropped, `codebook_dim` may be an int or a per-level list."""
        from .checkpoint import load_model_checkpoint
        sd, kw = load_model_checkpoint(path, package_name="DAC", kwarg_keys=tuple(DEFAULT_CFG), trusted=trusted)
        cfg = normalize_codec_kwargs(kw)
        validate_codec_state_dict(sd, dict(DEFAULT_CFG, **cfg))
        return cls(sd, cfg, device=device, engine=engine)

    # ---- kernels ------------------------------------------------------------------------------
    def _on_x3(self, c, taps=None):
        taps = c.get("k", 2) if taps is None else taps
        cout = c["cout"]
        eff = cout / (128.0 * math.ceil(cout / 128))                   # fraction of the 128-wide column tiles that is real output
        if self.precision == "bf16x3" and cout in (96, 192) and c["cin"] % 32 == 0:
            # round 6: these run with the output channels on the tile's ROW axis (gemm_x3.hip CONVT: exactly one 96- / 192-row tile, nothing
            # multiplies padding) — the same rule in csrc/codec_plan.hip (on_x3)
            return taps * c["cin"] >= self.X3_MIN_WORK_CONVT
        return (self.precision in ("bf16x3", "f16x2") and cout >= self.X3_MIN_COUT and cout % 16 == 0 and c["cin"] % 32 == 0
                and taps * c["cin"] * eff >= self.X3_MIN_WORK)

    def _fmt(self, c, taps=None):
        """input format of convolution c: "x3" (split planes) or "f32" """
        return "x3" if self._on_x3(c, taps) else "f32"

    def _tile_planes(self, w2d):
        """fp32 [rows][K] on the device -> the tiled split planes the bf16x3 kernel reads weights in (3 * rows * K bf16)"""
        eng = self.engine
        rows, K = w2d.shape
        if self.precision == "f16x2":
            return eng.split2h(w2d.contiguous(), tiled=True).reshape(-1)       # [rows / 16][K / 32][2][16][32] fp16
        planes = eng.split3(w2d.contiguous())
        tiled = torch.empty(3 * rows * K, dtype=torch.bfloat16, device=self.device)
        eng.check(self.lib.vn_tile_planes_bf16x3(eng.handle, planes.data_ptr(), rows * K, tiled.data_ptr(), rows, K, eng.stream()),
                  "vn_tile_planes_bf16x3")
        return tiled

    # ---- launches: issued now (eager) or recorded into a program (the same code path builds both) ------------------------------
    MAX_PROGRAMS = 8         # cached (direction, batch, length, precision) programs; the oldest goes first
    _OP_KIND = {"vn_conv1d_f32": 0, "vn_conv1d_bf16x3": 1, "vn_conv1d_f16x2": 2, "vn_dac_conv_in_f32": 3, "vn_dac_conv_out_f32": 4,
                "vn_rvq_encode_f32": 5, "vn_rvq_decode_f32": 6, "vn_split3_f32": 7, "vn_split2_f16": 8}      # VN_CODEC_OP_*

    def _new(self, shape, dtype=torch.float32, io=None):
        """an activation buffer: a tensor (eager) or a program buffer (recording)"""
        if self._rec is not None:
            return self._rec.new(shape, dtype, io)
        return torch.empty(*shape, device=self.device, dtype=dtype)

    def _emit(self, name, *args):
        """one launch of a single-layer entry point (arguments between ctx and stream)"""
        if self._rec is None:
            eng = self.engine
            eng.check(getattr(self.lib, name)(eng.handle, *args, eng.stream()), name)
            return
        import ctypes as C
        code = {C.c_void_p: "p", C.c_int: "i", C.c_int64: "l", C.c_float: "f"}
        types = _lib.SYMBOLS[name][1][1:-1]
        assert len(types) == len(args), name
        self._rec.emit(self._OP_KIND[name], [(code[t], a) for t, a in zip(types, args)])

    def _drop_programs(self):
        for prog in getattr(self, "_programs", {}).values():
            self.lib.vn_codec_destroy(prog["handle"])
        self._programs = {}

    def __del__(self):
        try:
            self._drop_programs()
        except Exception:
            pass

    def _program(self, direction, B, n):
        """the recorded program of one direction for batch B and length n (samples / tokens) in the current precision"""
        import ctypes as C
        key = (direction, B, n, self.precision)
        if key in self._programs:
            return self._programs[key]
        rec = _Recorder()
        self._rec = rec
        try:
            if direction == 0:
                res = self._encode_layers(rec.new((B, n), torch.float32, io="in"), B, n)
                rec.bufs[res["z"].bid]["pinned"] = True                      # the latents stay readable after the call
                z = res["z"]
            else:
                self._decode_layers(rec.new((B, self.n_codebooks, n), torch.int64, io="in"), B, self.n_codebooks, n)
                z = None
        finally:
            self._rec = None
        off, total = rec.plan()
        arena = torch.empty(max(total, 256), dtype=torch.uint8, device=self.device)
        base = arena.data_ptr()
        ops = (_lib.vn_codec_op * len(rec.ops))()
        for k, (kind, typed) in enumerate(rec.ops):
            op, cnt = ops[k], dict(p=0, i=0, l=0, f=0)
            op.kind = kind
            for t, v in typed:
                j = cnt[t]
                cnt[t] += 1
                if t == "p":
                    v = None if v is None else (base + off[v - _FAKE_BASE] if v >= _FAKE_BASE else v)
                    op.p[j] = v
                else:
                    getattr(op, t)[j] = v
        h = C.c_void_p()
        self.engine.check(self.lib.vn_codec_create(self.engine.handle, ops, len(rec.ops), direction, C.byref(h)), "vn_codec_create")
        prog = dict(handle=h, arena=arena, n_ops=len(rec.ops), arena_bytes=total,
                    z=(off[z.bid], z.shape) if z is not None else None)
        while len(self._programs) >= self.MAX_PROGRAMS:       # every program owns an arena: keep the most recent shapes only
            old = self._programs.pop(next(iter(self._programs)))
            self.lib.vn_codec_destroy(old["handle"])
        self._programs[key] = prog
        return prog

    def _empty_planes(self, rows, cols):
        """planar split planes of a [rows][cols] activation in this codec's format: three bf16 or two fp16 planes"""
        if self.precision == "f16x2":
            return self._new((2, rows, cols), torch.float16)
        return self._new((3, rows, cols), torch.bfloat16)

    def _planes(self, x):
        """fp32 [B][T][C] -> _Act with split planes (the few places where a producer outside the conv stack feeds the bf16x3 pipe)"""
        x2d = x.reshape(-1, x.shape[-1])
        R, K = x2d.shape
        out = self._empty_planes(R, K)
        if self.precision == "f16x2":
            self._emit("vn_split2_f16", x2d.data_ptr(), out.data_ptr(), R, K, R * K, 0)
        else:
            self._emit("vn_split3_f32", x2d.data_ptr(), out.data_ptr(), R * K, R * K)
        return _Act(f32=x, p16=out)

    def _conv(self, x, c, *, T_in, T_rows, T_out, phase=None, taps=None, in_stride=1, dil=1, pad=0, out_stride=1, out_off=0,
              resid=None, alpha=None, want_raw=True, s_fmt=None, act=0, out=None):
        """One convolution launch.  x: _Act holding the format this convolution reads.  Returns (raw fp32 result or None, _Act of
        snake(result, alpha) in the format(s) `s_fmt` names: "f32", "x3", "both" or None).  `out` = (raw, _Act) buffers to write
        into (the phases of a transposed convolution share theirs)."""
        eng = self.engine
        cout, cin = c["cout"], c["cin"]
        taps = c.get("k", 2) if taps is None else taps
        x3 = self._on_x3(c, taps)
        src = x.p16 if x3 else x.f32
        assert src is not None, "producer / consumer format mismatch in the codec graph"
        B = src.shape[0] if not x3 else src.shape[1] // T_in
        y, sn = out if out is not None else (None, None)
        if want_raw and y is None:
            y = self._new((B, T_out, cout))
        if s_fmt is not None and sn is None:
            assert alpha is not None
            sn = _Act(self._new((B, T_out, cout)) if s_fmt in ("f32", "both") else None,
                      self._empty_planes(B * T_out, cout) if s_fmt in ("x3", "both") else None)
        p = lambda t: t.data_ptr() if t is not None else None
        y2, y216 = (sn.f32, sn.p16) if sn is not None else (None, None)
        plane = B * T_out * cout
        h2 = self.precision == "f16x2"
        if x3:
            w16 = c["w16"] if phase is None else c["w16"][phase]
            self._emit("vn_conv1d_f16x2" if h2 else "vn_conv1d_bf16x3",
                       src.data_ptr(), src.shape[1] * src.shape[2], w16.data_ptr(), c["b"].data_ptr(), p(resid), p(alpha),
                       p(y), p(y2), p(y216), plane, B, T_in, T_rows, T_out, cin, cout, taps, in_stride, dil, pad, out_stride, out_off, act)
        else:
            w = c["w"] if phase is None else c["w"][phase]
            self._emit("vn_conv1d_f32", src.data_ptr(), w.data_ptr(), c["b"].data_ptr(), p(resid), p(alpha), p(y), p(y2), p(y216),
                       -plane if h2 else plane,        # the fp32 kernel writes two fp16 planes when the stride is negative
                       B, T_in, T_rows, T_out, cin, cout, taps, in_stride, dil, pad, out_stride, out_off, act)
        return y, sn

    def _res_unit(self, x, s, r, dil, alpha_next, T, next_fmt):
        """ResidualUnit: x + conv1(snake(conv7(snake(x)))).  x = raw fp32 residual stream, s = snake(x) as the k = 7 convolution
        reads it; returns (x', snake(x', alpha_next) in `next_fmt`)."""
        _, hs = self._conv(s, r["c7"], T_in=T, T_rows=T, T_out=T, dil=dil, pad=3 * dil, alpha=r["a2"], want_raw=False,
                           s_fmt=self._fmt(r["c1"]))
        return self._conv(hs, r["c1"], T_in=T, T_rows=T, T_out=T, resid=x, alpha=alpha_next, s_fmt=next_fmt)

    # ---- reference API ------------------------------------------------------------------------
    def preprocess(self, audio_data, sample_rate=None):
        """DAC.preprocess (interface.py:215): right-pad to a multiple of hop_length; returns (audio, length)."""
        if sample_rate is not None:
            assert sample_rate == self.sample_rate
        length = audio_data.shape[-1]
        pad = math.ceil(length / self.hop_length) * self.hop_length - length
        return torch.nn.functional.pad(audio_data, (0, pad)), length

    @torch.inference_mode()
    def encode(self, audio_data, sample_rate=None):
        """codec.encode(samples, sr) (interface.py:223): audio (B,1,L), L % hop == 0 -> {"codes": (B,n,T) int64, "z": ...}"""
        return self._guarded(lambda: self._encode(audio_data), "encode()")

    def _encode(self, audio_data):
        x = audio_data.to(self.device, torch.float32)
        B, ch, L = x.shape
        assert ch == 1 and L % self.hop_length == 0, "mono audio padded by preprocess() expected"
        x = x.reshape(B, L).contiguous()
        if not self.use_program:
            return self._encode_layers(x, B, L)
        prog = self._program(0, B, L)                       # ONE C call: vn_dac_encode walks the recorded layer loop
        codes = torch.empty(B, self.n_codebooks, L // self.hop_length, device=self.device, dtype=torch.int64)
        self.engine.check(self.lib.vn_dac_encode(prog["handle"], x.data_ptr(), codes.data_ptr(), self.engine.stream()), "vn_dac_encode")
        prog["arena"].record_stream(torch.cuda.current_stream(self.device))   # an evicted program's arena must outlive this stream's kernels
        zo, zshape = prog["z"]
        nz = int(np.prod(zshape)) * 4
        return {"codes": codes, "z": prog["arena"][zo:zo + nz].view(torch.float32).view(*zshape).clone()}  # owned (the arena is re-used)

    def _encode_layers(self, x, B, L):
        """the encoder's layer loop: x = [B][L] samples -> {"codes", "z"}; issues (or records) one launch per layer"""
        e = self.enc
        C0 = e["stem"]["c"]
        cur = self._new((B, L, C0))
        s0 = self._new((B, L, C0))
        self._emit("vn_dac_conv_in_f32", x.data_ptr(), e["stem"]["w"].data_ptr(), e["stem"]["b"].data_ptr(),
                   e["blocks"][0]["res"][0]["a1"].data_ptr(), cur.data_ptr(), s0.data_ptr(), B, L, C0)
        s = _Act(f32=s0) if self._fmt(e["blocks"][0]["res"][0]["c7"]) == "f32" else self._planes(s0)
        T = L
        nb = len(e["blocks"])
        for bi, blk in enumerate(e["blocks"]):
            for j, dil in enumerate((1, 3, 9)):
                a_next = blk["res"][j + 1]["a1"] if j < 2 else blk["a"]
                nxt = self._fmt(blk["res"][j + 1]["c7"]) if j < 2 else self._fmt(blk["down"])
                cur, s = self._res_unit(cur, s, blk["res"][j], dil, a_next, T, nxt)
            st = blk["s"]
            pad = math.ceil(st / 2)
            T_out = (T + 2 * pad - 2 * st) // st + 1
            a_next = e["blocks"][bi + 1]["res"][0]["a1"] if bi + 1 < nb else e["a_out"]
            nxt = self._fmt(e["blocks"][bi + 1]["res"][0]["c7"]) if bi + 1 < nb else self._fmt(e["out"])
            cur, s = self._conv(s, blk["down"], T_in=T, T_rows=T_out, T_out=T_out, in_stride=st, pad=pad, alpha=a_next,
                                want_raw=bi + 1 < nb, s_fmt=nxt)
            T = T_out
        z, _ = self._conv(s, e["out"], T_in=T, T_rows=T, T_out=T, pad=1)
        codes = self._new((B, self.n_codebooks, T), torch.int64, io="out")
        r = self.rvq
        self._emit("vn_rvq_encode_f32", z.data_ptr(), r["win"].data_ptr(), r["bin"].data_ptr(), r["cb"].data_ptr(), r["wout"].data_ptr(),
                   r["bout"].data_ptr(), codes.data_ptr(), B, T, self.latent_dim, self.n_codebooks, self.cfg["codebook_size"])
        return {"codes": codes, "z": z}

    @torch.inference_mode()
    def decode_codes(self, codes):
        """codes (B,n,T) -> audio (B,1,T*hop): codec.decode(codec.quantizer.from_latents(from_codes(z))[0])["audio"]
        (transformer.py:669-675)."""
        return self._guarded(lambda: self._decode_codes(codes), "decode()")

    def _decode_codes(self, codes):
        codes = codes.to(self.device, torch.int64).contiguous()
        B, n, T = codes.shape
        if not self.use_program or n != self.n_codebooks:
            return self._decode_layers(codes, B, n, T)
        prog = self._program(1, B, T)                       # ONE C call: vn_dac_decode walks the recorded layer loop
        audio = torch.empty(B, 1, T * self.hop_length, device=self.device, dtype=torch.float32)
        self.engine.check(self.lib.vn_dac_decode(prog["handle"], codes.data_ptr(), audio.data_ptr(), self.engine.stream()), "vn_dac_decode")
        prog["arena"].record_stream(torch.cuda.current_stream(self.device))
        return audio

    def _decode_layers(self, codes, B, n, T):
        """the decoder's layer loop: codes [B][n][T] -> audio [B][1][T * hop]; issues (or records) one launch per layer / phase"""
        d, r = self.dec, self.rvq
        zq = self._new((B, T, self.latent_dim))
        self._emit("vn_rvq_decode_f32", codes.data_ptr(), r["cb"].data_ptr(), r["wout"].data_ptr(), r["bout"].data_ptr(), zq.data_ptr(),
                   B, T, self.latent_dim, n, self.cfg["codebook_size"])
        zin = _Act(f32=zq) if self._fmt(d["in"]) == "f32" else self._planes(zq)
        _, s = self._conv(zin, d["in"], T_in=T, T_rows=T, T_out=T, pad=3, alpha=d["blocks"][0]["a"], want_raw=False,
                          s_fmt=self._fmt(d["blocks"][0]["up"], taps=2))
        nb = len(d["blocks"])
        cur = None
        for bi, blk in enumerate(d["blocks"]):
            st, up = blk["s"], blk["up"]
            pad = math.ceil(st / 2)
            T_out = (T - 1) * st - 2 * pad + 2 * st
            f0 = self._fmt(blk["res"][0]["c7"])
            y = self._new((B, T_out, up["cout"]))
            y2 = _Act(self._new((B, T_out, up["cout"])) if f0 == "f32" else None,
                      self._empty_planes(B * T_out, up["cout"]) if f0 == "x3" else None)
            for ph in range(st):        # polyphase: output rows t = t'*st + ph - pad read x[t'] and x[t'-1]
                self._conv(s, up, phase=ph, taps=2, T_in=T, T_rows=T + 1, T_out=T_out, in_stride=1, dil=-1, pad=0,
                           out_stride=st, out_off=ph - pad, alpha=blk["res"][0]["a1"], s_fmt=f0, out=(y, y2))
            cur, s, T = y, y2, T_out
            for j, dil in enumerate((1, 3, 9)):
                if j < 2:
                    a_next, nxt = blk["res"][j + 1]["a1"], self._fmt(blk["res"][j + 1]["c7"])
                elif bi + 1 < nb:
                    a_next, nxt = d["blocks"][bi + 1]["a"], self._fmt(d["blocks"][bi + 1]["up"], taps=2)
                else:
                    a_next, nxt = d["a_out"], "f32"                  # the 1-channel head reads fp32
                cur, s = self._res_unit(cur, s, blk["res"][j], dil, a_next, T, nxt)
        s = s.f32
        audio = self._new((B, T), torch.float32, io="out")
        self._emit("vn_dac_conv_out_f32", s.data_ptr(), d["head_w"].data_ptr(), d["head_b"], audio.data_ptr(), B, T, s.shape[-1])
        return audio.reshape(B, 1, T)

    def decode(self, z_or_codes):
        return {"audio": self.decode_codes(z_or_codes)}

    # ---- Interface helpers (interface.py:206-224, 203-204) --------------------------------------
    def encode_signal(self, signal, loudness=-24.0):
        """Interface._preprocess + encode: resample -> mono -> loudness-normalise -> peak-limit -> pad -> codes."""
        sig = self.preprocess_signal(signal, loudness)
        return self.encode(sig.samples, self.sample_rate)["codes"]

    def preprocess_signal(self, signal, loudness=-24.0):
        """Interface._preprocess (interface.py:206-217): clone -> resample(codec rate) -> to_mono -> normalize(loudness) ->
        ensure_max_of_audio(1.0) -> codec.preprocess (right-pad to the hop); returns a new AudioSignal whose samples live ON THE DEVICE:
        the loudness measurement (BS.1770-4, float64), the gain, the peak limit and the pad run in csrc/preprocess.hip
        (vn_preprocess_f32).  Resampling (only when the input is not at the codec's rate) stays on the host (scipy).
        VN_PREPROCESS_HOST=1 / preprocess_signal_host: the all-host twin the tests compare against.  Parity with audiotools: unpinned."""
        import ctypes as C
        import os
        if os.environ.get("VN_PREPROCESS_HOST") == "1":
            return self.preprocess_signal_host(signal, loudness)
        # what the device kernels do not take (csrc/preprocess.hip): a codec rate at which a 400 ms block is not four 100 ms hops
        # (int(0.4 sr) != 4 (sr // 10): any rate that is not a multiple of 10), and an empty signal — the host twin handles both
        if int(0.4 * self.sample_rate) != 4 * (self.sample_rate // 10) or signal.samples.shape[-1] == 0:
            out = self.preprocess_signal_host(signal, loudness)
            return AudioSignal(out.samples.to(self.device), out.sample_rate)
        x, sr = signal.samples.float(), signal.sample_rate
        if sr != self.sample_rate:
            from scipy.signal import resample_poly
            g = math.gcd(sr, self.sample_rate)
            x = torch.from_numpy(resample_poly(x.cpu().numpy(), self.sample_rate // g, sr // g, axis=-1).astype(np.float32))
        x = x.to(self.device).mean(dim=1).contiguous()                               # to_mono: (B, T)
        B, T = x.shape
        Tp = math.ceil(T / self.hop_length) * self.hop_length
        kw = getattr(self, "_kw_cache", None)
        if kw is None:
            (b1, a1), (b2, a2) = _k_weighting(self.sample_rate)
            kw = ((C.c_double * 12)(*b1, *a1, *b2, *a2), (C.c_double * 16)(*kweight_state_power(self.sample_rate, self.sample_rate // 10).reshape(-1)))
            self._kw_cache = kw
        nbytes = C.c_int64()
        self.engine.check(self.lib.vn_preprocess_workspace(B, T, self.sample_rate, C.byref(nbytes)), "vn_preprocess_workspace")
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=self.device)
        y = torch.empty(B, 1, Tp, dtype=torch.float32, device=self.device)
        self.engine.check(self.lib.vn_preprocess_f32(self.engine.handle, x.data_ptr(), y.data_ptr(), B, T, Tp, self.sample_rate, float(loudness),
                                                     kw[0], kw[1], ws.data_ptr(), None, self.engine.stream()), "vn_preprocess_f32")
        ws.record_stream(torch.cuda.current_stream(self.device))
        return AudioSignal(y, self.sample_rate)

    def preprocess_signal_host(self, signal, loudness=-24.0):
        """the same chain entirely on the host (numpy / scipy): the twin of the device path"""
        x, sr = signal.samples.float().cpu().clone(), signal.sample_rate
        if sr != self.sample_rate:
            from scipy.signal import resample_poly
            g = math.gcd(sr, self.sample_rate)
            x = torch.from_numpy(resample_poly(x.numpy(), self.sample_rate // g, sr // g, axis=-1).astype(np.float32))
        x = x.mean(dim=1, keepdim=True)                                           # to_mono
        for b in range(x.shape[0]):                                               # normalize(loudness)
            lufs = integrated_loudness(x[b].numpy(), self.sample_rate)
            if lufs > -70.0:
                x[b] *= 10 ** ((loudness - lufs) / 20)
        peak = x.abs().amax(dim=(1, 2), keepdim=True)                             # ensure_max_of_audio(1.0)
        x = torch.where(peak > 1.0, x / peak.clamp_min(1e-12), x)
        x, _ = self.preprocess(x, self.sample_rate)
        return AudioSignal(x, self.sample_rate)

    def decode_signal(self, codes):
        return AudioSignal(self.decode_codes(codes), self.sample_rate)
