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


def onset_mask(samples, sample_rate: int, z: torch.Tensor, hop_length: int, width: int = 1) -> torch.Tensor:
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
