"""Host-side mask construction for `Interface.build_mask` (reference: vampnet/mask.py, interface.py:454-489).

Masks are tiny int64 tensors (B, 14, T), so they are built on the host with vectorised index
arithmetic; what matters for seeded parity is that torch's *CPU* generator is consumed in exactly the
reference's order and shapes (SURVEY.md §7 "RNG replay"):
  linear_random : one bernoulli over a float tensor of z's shape                     (mask.py:70)
  periodic_mask : one bernoulli(ones(w)) per prompt column per batch item            (mask.py:120)
                  one randint(0, period, (1,)) for the roll                          (mask.py:128)
  dropout       : one randint(0, T, (int(T*p),)) — also when it is empty             (mask.py:170)
All draws are made on the CPU generator regardless of where `z` lives; the result is moved to z.device.
"""
import math

import numpy as np
import torch


def seconds_to_tokens(seconds, sample_rate: int, hop_length: int):
    """Interface.s2t (interface.py:176-181): an int for a scalar, a float64 array of ceilings for a numpy array (beat times)."""
    if isinstance(seconds, np.ndarray):
        return np.ceil(seconds * sample_rate / hop_length)
    return math.ceil(seconds * sample_rate / hop_length)


def gamma(r: torch.Tensor) -> torch.Tensor:
    """Cosine masking schedule (mask.py:8-9)."""
    return (r * torch.pi / 2).cos().clamp(1e-10, 1.0)


def apply_mask(x: torch.Tensor, mask: torch.Tensor, mask_token: int):
    """x where mask == 0, mask_token where mask == 1 (mask.py:24-38; same assertions)."""
    assert mask.ndim == 3, f"mask must be (batch, n_codebooks, seq), but got {mask.ndim}"
    assert mask.shape == x.shape, f"mask must be same shape as x, but got {mask.shape} and {x.shape}"
    assert mask.dtype == torch.long, f"mask must be long dtype, but got {mask.dtype}"
    assert not bool(torch.any(mask > 1)), "mask must be binary"
    assert not bool(torch.any(mask < 0)), "mask must be binary"
    return torch.where(mask.bool(), torch.full_like(x, mask_token), x), mask


def random_intensity_mask(shape, intensity: float) -> torch.Tensor:
    """linear_random (mask.py:56-73): Bernoulli(intensity) per element."""
    p = torch.full(tuple(shape), 1.0, dtype=torch.float32) * torch.tensor(float(intensity), dtype=torch.float32)
    return torch.bernoulli(p).round().long()


def inpaint_mask(shape, n_prefix: int, n_suffix: int) -> torch.Tensor:
    """inpaint (mask.py:75-99): keep (0) the first n_prefix and last n_suffix timesteps."""
    m = torch.ones(tuple(shape), dtype=torch.long)
    if n_prefix > 0:
        m[..., :n_prefix] = 0
    if n_suffix > 0:
        m[..., -n_suffix:] = 0
    return m


def periodic_prompt_mask(shape, period: int, width: int = 1, random_roll: bool = False) -> torch.Tensor:
    """periodic_mask (mask.py:101-131): un-mask a window of `width` around every `period`-th timestep."""
    B, _, T = shape
    m = torch.ones(tuple(shape), dtype=torch.long)
    if period == 0:
        return m
    centers = torch.arange(0, T, period)
    lo = (centers - width // 2).clamp(min=0)
    hi = (centers + width // 2).clamp(max=T - 1) + 1
    keep = torch.zeros(T + 1, dtype=torch.long)
    keep.index_add_(0, lo, torch.ones_like(lo))
    keep.index_add_(0, hi, -torch.ones_like(hi))
    keep = keep.cumsum(0)[:T] > 0
    m[:, :, keep] = 0
    for _ in range(B):                       # RNG parity: the reference flips an always-heads coin per column
        for a, b in zip(lo.tolist(), hi.tolist()):
            torch.bernoulli(torch.ones(b - a))
    if random_roll:
        offset = int(torch.randint(0, period, (1,)).item())
        m = torch.roll(m, offset, dims=-1)
    return m


def timestep_dropout(mask: torch.Tensor, p: float) -> torch.Tensor:
    """dropout (mask.py:162-173): re-mask int(T*p) random timesteps (with replacement)."""
    T = mask.shape[-1]
    idx = torch.randint(0, T, (int(T * p),))
    out = mask.clone().long()
    out[:, :, idx] = 1
    return out


def onset_mask_from_samples(samples, sample_rate: int, z: torch.Tensor, hop_length: int, width: int = 1) -> torch.Tensor:
    """onset_mask (mask.py:205-228): un-mask [idx - width, idx + width) around every detected onset frame of the FIRST item /
    channel of the signal.  Python slice semantics are kept on purpose: an onset closer than `width` frames to the start
    gives a negative slice start, i.e. (usually) an empty slice, exactly like the reference.  The detector restates
    librosa's published algorithm (vampnet_amd/onsets.py) [UNVERIFIED-DEP: parity unpinned]."""
    from .onsets import onset_detect
    y = samples[0][0].detach().cpu().numpy() if isinstance(samples, torch.Tensor) else samples[0][0]
    mask = torch.ones(tuple(z.shape), dtype=torch.long)
    for idx in onset_detect(y, sample_rate, hop_length, backtrack=True):
        idx = int(idx)
        mask[:, :, idx - width:idx + width] = 0
    return mask


def beat_mask(beats, downbeats, duration_s: float, sample_rate: int, hop_length: int, n_codebooks: int,
              before_beat_s: float = 0.0, after_beat_s: float = 0.02, mask_downbeats: bool = True, mask_upbeats: bool = True,
              downbeat_downsample_factor: int = None, beat_downsample_factor: int = None, dropout: float = 0.0,
              invert: bool = True) -> torch.Tensor:
    """Interface.make_beat_mask (interface.py:241-321) after the beat tracker: `beats` / `downbeats` are its times in seconds
    (numpy arrays).  Marks [beat - before, beat + after) token windows (one bernoulli(1 - dropout) draw per window from the
    torch CPU generator, in the reference's order: up-beats, then down-beats), clamps, inverts (1 = masked everywhere
    except around the beats) and repeats over the codebooks.  Returns (1, n_codebooks, T) int64 on the CPU."""
    s2t = lambda x: seconds_to_tokens(x, sample_rate, hop_length)
    beats_z, downbeats_z = s2t(np.asarray(beats)), s2t(np.asarray(downbeats))
    bz = torch.tensor(beats_z)
    beats_z = bz[~torch.isin(bz, torch.tensor(downbeats_z))].tolist()          # a down-beat is not also an up-beat
    downbeats_z = downbeats_z.tolist()
    mask = torch.zeros(s2t(duration_s))
    mask_b4, mask_after = s2t(before_beat_s), s2t(after_beat_s)
    for name, f in (("beat", beat_downsample_factor), ("downbeat", downbeat_downsample_factor)):
        if f is not None and f < 1:
            raise ValueError("mask_beat_downsample_factor must be >= 1 or None")
    beats_z = beats_z[::beat_downsample_factor or 1]
    downbeats_z = downbeats_z[::downbeat_downsample_factor or 1]

    def mark(indices):
        for idx in indices:
            a, b = int(idx - mask_b4), int(idx + mask_after)
            m = torch.ones(mask[a:b].shape[0])                               # python slice semantics, incl. negative starts
            m = m * torch.bernoulli(m * (1 - dropout)).long()
            mask[a:b] = m

    if mask_upbeats:
        mark(beats_z)
    if mask_downbeats:
        mark(downbeats_z)
    mask = mask.clamp(0, 1)
    if invert:
        mask = 1 - mask
    return mask[None, None, :].bool().long().repeat(1, n_codebooks, 1)


def build_mask(z: torch.Tensor, *, rand_mask_intensity=1.0, n_prefix=0, n_suffix=0, periodic_prompt=7,
               periodic_prompt_width=1, onset_mask=None, dropout=0.0, upper_codebook_mask=3, ncc=0) -> torch.Tensor:
    """Composition of Interface.build_mask (interface.py:454-489): AND of the random, inpaint, periodic (rolled)
    and optional onset masks, then dropout, conditioning-codebook un-mask, upper-codebook mask."""
    shape = tuple(z.shape)
    m = random_intensity_mask(shape, rand_mask_intensity)
    m = torch.minimum(m, inpaint_mask(shape, n_prefix, n_suffix))
    m = torch.minimum(m, periodic_prompt_mask(shape, periodic_prompt, periodic_prompt_width, random_roll=True))
    if onset_mask is not None:
        m = torch.minimum(m, onset_mask.to("cpu").long())
    m = timestep_dropout(m, dropout)
    if ncc is not None:
        m[:, :ncc, :] = 0
    m[:, int(upper_codebook_mask):, :] = 1
    return m.to(z.device)


def _slice_start(v, C):
    """python slice semantics of mask[:, v:, :] / mask[:, :v, :] for an int bound -> index in [0, C]."""
    v = int(v)
    return max(C + v, 0) if v < 0 else min(v, C)


def mask_word_ledger(B: int, C: int, T: int, period: int, width: int, dropout: float):
    """Where the reference's build_mask draws sit in torch's CPU mt19937 stream, in 32-bit words from the current position:
    [0, B C T) linear_random's bernoulli; then one word per always-heads coin of periodic_mask (B items x the clipped windows);
    then the roll's randint (absent when period == 0); then dropout's int(T p) randints.
    -> (roll_word or -1, drop_word, n_drop, n_words)."""
    n_lin = B * C * T
    n_coin = 0
    if period > 0:
        centers = torch.arange(0, T, period)
        lo = (centers - width // 2).clamp(min=0)
        hi = (centers + width // 2).clamp(max=T - 1) + 1
        n_coin = B * int((hi - lo).sum())
    roll_word = n_lin + n_coin if period > 0 else -1
    n_drop = int(T * dropout)
    drop_word = n_lin + n_coin + (1 if period > 0 else 0)
    return roll_word, drop_word, n_drop, drop_word + n_drop


_warned_host_fallback = False


def build_mask_device(engine, z: torch.Tensor, *, rand_mask_intensity=1.0, n_prefix=0, n_suffix=0, periodic_prompt=7,
                      periodic_prompt_width=1, onset_mask=None, dropout=0.0, upper_codebook_mask=3, ncc=0) -> torch.Tensor:
    """build_mask with every draw and the whole composition on the GPU (csrc/elementwise.hip: vn_build_mask_kernel): torch's CPU
    generator is handed to the device (vampnet_amd/torch_rng.py), the words the reference's draws consume — B C T for
    linear_random, one per always-heads coin of periodic_mask, one for the roll, int(T p) for dropout — are produced there, and
    the advanced generator is written back, so the mask AND the generator position equal the host path's / the reference's
    (tests/test_gpu_model.py::test_build_mask_on_device).  `onset_mask`, when given, is the host-computed onset mask."""
    import ctypes as C_
    B, C, T = (int(v) for v in z.shape)
    dev = engine.device
    period, width = int(periodic_prompt), int(periodic_prompt_width)
    if period < 0 or width < 0:
        raise ValueError("periodic_prompt and periodic_prompt_width must be >= 0")
    p = float(torch.tensor(float(rand_mask_intensity), dtype=torch.float32))
    if not 0.0 <= p <= 1.0:
        raise RuntimeError("Expected p_in >= 0 && p_in <= 1")                 # torch.bernoulli's own check
    roll_word, drop_word, n_drop, n_words = mask_word_ledger(B, C, T, period, width, dropout)
    rng = engine.torch_rng()
    rng._producer = torch.cuda.current_stream(dev)
    try:
        rng.load_from_torch("mask")       # checks (once) that this torch build draws bernoulli / randint the way the kernel does
    except RuntimeError as e:
        global _warned_host_fallback
        if not _warned_host_fallback:
            import warnings
            warnings.warn(f"build_mask: {e}")
            _warned_host_fallback = True
        return build_mask(z, rand_mask_intensity=rand_mask_intensity, n_prefix=n_prefix, n_suffix=n_suffix,
                          periodic_prompt=periodic_prompt, periodic_prompt_width=periodic_prompt_width, onset_mask=onset_mask,
                          dropout=dropout, upper_codebook_mask=upper_codebook_mask, ncc=ncc)
    raw = torch.empty(n_words, dtype=torch.int32, device=dev)
    rng._gen(raw.data_ptr(), n_words)
    onset = None
    if onset_mask is not None:
        onset = onset_mask.to(dev, torch.int64).expand(B, C, T).contiguous()
    mask = torch.empty(B, C, T, dtype=torch.int64, device=dev)
    C_up = _slice_start(upper_codebook_mask, C)
    C_ncc = 0 if ncc is None else _slice_start(ncc, C)
    engine.check(engine.lib.vn_build_mask(engine.handle, raw.data_ptr(), onset.data_ptr() if onset is not None else None,
                                          mask.data_ptr(), B, C, T, C_.c_float(p), int(n_prefix), int(n_suffix), period, width,
                                          roll_word, drop_word, n_drop, C_ncc, C_up, engine.stream()), "vn_build_mask")
    rng.store_to_torch()
    return mask if z.device == mask.device else mask.to(z.device)


# ---- the reference module's own names and signatures (vampnet/mask.py), so that `from vampnet_amd import masks as pmask` serves
# its callers (app.py:206-217, scripts/exp/train.py:250-254).  Same arithmetic, same torch-CPU RNG consumption; pinned against
# the reference in tests/test_oracle_vs_reference.py::test_mask_module_names_bitwise.
_gamma = gamma


def _invgamma(y):
    """mask.py:11-14"""
    if not torch.is_tensor(y):
        y = torch.tensor(y)[None]
    return 2 * y.acos() / torch.pi


def _batch(v, n):
    """util.scalar_to_batch_tensor (util.py:6-7)"""
    return torch.tensor(v).repeat(n)


def full_mask(x: torch.Tensor):
    assert x.ndim == 3, "x must be (batch, n_codebooks, seq)"
    return torch.ones_like(x).long()


def empty_mask(x: torch.Tensor):
    assert x.ndim == 3, "x must be (batch, n_codebooks, seq)"
    return torch.zeros_like(x).long()


def random(x: torch.Tensor, r):
    """mask.py:40-54 (the training mask, train.py:250): Bernoulli(gamma(r)) per element, r per batch item."""
    assert x.ndim == 3, "x must be (batch, n_codebooks, seq)"
    if not isinstance(r, torch.Tensor):
        r = _batch(r, x.shape[0]).to(x.device)
    probs = torch.ones_like(x) * gamma(r)[:, None, None]
    return torch.bernoulli(probs).round().long()


def linear_random(x: torch.Tensor, r):
    """mask.py:56-73"""
    assert x.ndim == 3, "x must be (batch, n_codebooks, seq)"
    if not isinstance(r, torch.Tensor):
        r = _batch(r, x.shape[0]).to(x.device).float()[:, None, None]
    probs = torch.ones_like(x).float().expand(x.shape[0], x.shape[1], -1) * r
    return torch.bernoulli(probs).round().long()


def inpaint(x: torch.Tensor, n_prefix, n_suffix):
    """mask.py:75-99: scalar or per-item prefix / suffix lengths."""
    assert n_prefix is not None
    assert n_suffix is not None
    mask = full_mask(x)
    if n_prefix > 0:
        if not isinstance(n_prefix, torch.Tensor):
            n_prefix = _batch(n_prefix, x.shape[0])
        for i, n in enumerate(n_prefix):
            if n > 0:
                mask[i, :, :n] = 0
    if n_suffix > 0:
        if not isinstance(n_suffix, torch.Tensor):
            n_suffix = _batch(n_suffix, x.shape[0])
        for i, n in enumerate(n_suffix):
            if n > 0:
                mask[i, :, -n:] = 0
    return mask


def periodic_mask(x: torch.Tensor, period, width: int = 1, random_roll=False):
    """mask.py:101-131.  A scalar period takes the vectorised path above; a tensor period is walked like the reference (which
    only accepts one element, and then only marks the batch items it enumerates)."""
    if not isinstance(period, torch.Tensor):
        return periodic_prompt_mask(tuple(x.shape), int(period), width, random_roll).to(x.device)
    mask = full_mask(x)
    if period == 0:                     # like the reference: ambiguous (raises) for a tensor with more than one element
        return mask
    T = mask.shape[-1]
    for i, factor in enumerate(period):
        if factor == 0:
            continue
        for j in range(T):
            if j % factor == 0:
                a, b = max(0, j - width // 2), min(T - 1, j + width // 2) + 1
                torch.bernoulli(torch.ones(b - a))                   # the reference's always-heads coin (RNG parity)
                mask[i, :, a:b] = 0
    if random_roll:
        mask = torch.roll(mask, int(torch.randint(0, period[0], (1,)).item()), dims=-1)
    return mask


def codebook_unmask(mask: torch.Tensor, n_conditioning_codebooks):
    if n_conditioning_codebooks is None:
        return mask
    mask = mask.clone()
    mask[:, :n_conditioning_codebooks, :] = 0
    return mask


def codebook_mask(mask: torch.Tensor, val1: int, val2: int = None):
    mask = mask.clone()
    mask[:, val1:, :] = 1
    return mask


def mask_and(mask1: torch.Tensor, mask2: torch.Tensor):
    assert mask1.shape == mask2.shape, "masks must be same shape"
    return torch.min(mask1, mask2)


def mask_or(mask1: torch.Tensor, mask2: torch.Tensor):
    assert mask1.shape == mask2.shape, f"masks must be same shape, but got {mask1.shape} and {mask2.shape}"
    assert mask1.max() <= 1, "mask1 must be binary"
    assert mask2.max() <= 1, "mask2 must be binary"
    assert mask1.min() >= 0, "mask1 must be binary"
    assert mask2.min() >= 0, "mask2 must be binary"
    return (mask1 + mask2).clamp(0, 1)


def dropout(mask: torch.Tensor, p: float):
    return timestep_dropout(mask, p)


def time_stretch_mask(x: torch.Tensor, stretch_factor: int):
    assert stretch_factor >= 1, "stretch factor must be >= 1"
    T = x.shape[-1]
    x = x.repeat_interleave(stretch_factor, dim=-1)[:, :, :T]
    return periodic_mask(x, stretch_factor, width=1)


def onset_mask(sig, z: torch.Tensor, interface, width: int = 1):
    """mask.py:205-228 (signature of the reference; the detector is vampnet_amd/onsets.py)."""
    return onset_mask_from_samples(sig.samples, sig.sample_rate, z, interface.codec.hop_length, width).to(z.device)
