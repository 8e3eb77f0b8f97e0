"""CPU tests of the Interface twin's host orchestration (chunking, padding, edge un-mask, stitching, batch sharding)
with an oracle-backed stand-in for the device model — the HIP engine itself is covered by `-m gpu` tests.
Includes the world_size-2 gloo test of the batch-shard + all-gather protocol."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from oracle import vampnet_oracle as O, weights as W
from tests.gpu_common import SynthCodec
from vampnet_amd.engine import draw_noise_host, seed_all
from vampnet_amd.interface import Interface


class _Dims(dict):
    max_batch = 16


class OracleBackedModel:
    """Same surface as vampnet_amd.engine.VampNetModel.generate, computing with the CPU oracle.  It consumes the
    product's own noise ledger (draw_noise_host) and honours n0_override / global_batch / batch_offset, so the
    sharding protocol is exercised end to end."""

    def __init__(self, sd, dims, cb, chunk_size_s):
        self.sd, self.dims, self.cb = sd, dims, cb
        self.n_codebooks, self.n_conditioning_codebooks = dims["n_codebooks"], dims["n_cond"]
        self.n_predict_codebooks = self.n_codebooks - self.n_conditioning_codebooks
        self.mask_token = dims["vocab"]
        self.chunk_size_s = chunk_size_s

    generate_batched_calls = True

    def draw_noise(self, B, T, steps, sample_cutoff, batch_offset=0, local_batch=None, pin=True):
        return draw_noise_host(B, T * self.n_predict_codebooks, 1024, steps, sample_cutoff, batch_offset,
                               B if local_batch is None else local_batch, False)

    def generate(self, codec=None, start_tokens=None, mask=None, _sampling_steps=12, temperature=1.0,
                 mask_temperature=10.5, seed=None, sample_cutoff=1.0, rng="torch", n0_override=None,
                 global_batch=None, batch_offset=0, noise=None, call_batch=None, **_):
        if seed is not None:
            seed_all(seed)
        B, Cn, T = start_tokens.shape
        if isinstance(n0_override, (list, tuple)):
            # several reference calls batched by Interface._generate_calls: un-batch and run them one by one with
            # the ledger slices the product assembled (checks its interleaving against the sequential oracle)
            nb = call_batch
            exp, unif = noise
            N = T * self.n_predict_codebooks
            outs = []
            for c in range(B // nb):
                sl = slice(c * nb, (c + 1) * nb)
                nz = [dict(exp=exp[i, c * nb * N:(c + 1) * nb * N] if (i / _sampling_steps) <= sample_cutoff else None,
                           unif=unif[i, sl]) for i in range(_sampling_steps)]
                outs.append(O.generate(self.sd, self.dims, self.cb, start_tokens[sl], mask[sl],
                                       sampling_steps=_sampling_steps, temperature=temperature,
                                       mask_temperature=mask_temperature, sample_cutoff=sample_cutoff,
                                       n0_override=n0_override[c * nb], noise=nz))
            return torch.cat(outs)
        if mask is None:
            mask = torch.ones_like(start_tokens)
            mask[:, :self.n_conditioning_codebooks] = 0
        N = T * self.n_predict_codebooks
        exp, unif = draw_noise_host(global_batch or B, N, 1024, _sampling_steps, sample_cutoff, batch_offset, B)
        noise = [dict(exp=exp[i] if (i / _sampling_steps) <= sample_cutoff else None, unif=unif[i])
                 for i in range(_sampling_steps)]
        return O.generate(self.sd, self.dims, self.cb, start_tokens, mask, sampling_steps=_sampling_steps,
                          temperature=temperature, mask_temperature=mask_temperature, sample_cutoff=sample_cutoff,
                          n0_override=n0_override, noise=noise)


def make_interface(pg=None, c2f_max_batch=16):
    cb = W.synth_codebooks()
    csd, fsd = W.synth_state_dict(W.TINY_COARSE_DIMS, 0), W.synth_state_dict(W.TINY_C2F_DIMS, 1)
    itf = object.__new__(Interface)
    itf.codec = SynthCodec(cb)
    itf.device = torch.device("cpu")
    itf.rng, itf.pg, itf._call_idx, itf.beat_tracker, itf.loudness = "torch", pg, 0, None, -24.0
    if pg is not None:
        import torch.distributed as dist
        itf.rank, itf.world = dist.get_rank(pg), dist.get_world_size(pg)
    else:
        itf.rank, itf.world = 0, 1
    itf.coarse = OracleBackedModel(csd, W.TINY_COARSE_DIMS, cb, 10)
    itf.c2f = OracleBackedModel(fsd, W.TINY_C2F_DIMS, cb, 3)
    itf.c2f.dims = _Dims(W.TINY_C2F_DIMS)            # dict for the oracle + .max_batch like vn_dims
    itf.c2f.dims.max_batch = c2f_max_batch
    return itf, O.OracleModels(csd, W.TINY_COARSE_DIMS, fsd, W.TINY_C2F_DIMS, cb)


@pytest.mark.parametrize("T", [100, 575, 600, 1200])
@pytest.mark.parametrize("B,kw", [(1, dict(seed=0, _sampling_steps=3)),
                                  (3, dict(seed=1, _sampling_steps=2, sample_cutoff=0.5, temperature=0.7))])
def test_vamp_orchestration_matches_oracle(T, B, kw):
    itf, models = make_interface()
    z = W.synth_codes(1, 14, T, seed=6)
    torch.manual_seed(3)
    mask = itf.build_mask(z)
    ref, ref_m = O.vamp(models, z, mask, batch_size=B, return_mask=True, **kw)
    got, got_m = itf.vamp(z, mask, batch_size=B, return_mask=True, **kw)
    assert torch.equal(ref, got) and torch.equal(ref_m, got_m)


@pytest.mark.parametrize("c2f_max_batch", [1, 3, 5, 64])
def test_c2f_chunk_batching_groups(c2f_max_batch):
    """coarse_to_fine batches its chunks into as many launches as the c2f workspace allows (1 = sequential)."""
    itf, models = make_interface(c2f_max_batch=c2f_max_batch)
    z = W.synth_codes(1, 14, 800, seed=6)
    torch.manual_seed(3)
    mask = itf.build_mask(z)
    ref = O.vamp(models, z, mask, batch_size=2, seed=4, _sampling_steps=2)
    got = itf.vamp(z, mask, batch_size=2, seed=4, _sampling_steps=2)
    assert torch.equal(ref, got)


def test_vamp_time_stretch_and_feedback():
    itf, models = make_interface()
    z = W.synth_codes(1, 14, 90, seed=2)
    torch.manual_seed(1)
    mask = itf.build_mask(z, periodic_prompt=3)
    kw = dict(batch_size=2, time_stretch_factor=2, feedback_steps=2, seed=5, _sampling_steps=2)
    ref = O.vamp(models, z, mask, **kw)
    got = itf.vamp(z, mask, **kw)
    assert got.shape == (2, 14, 180) and torch.equal(ref, got)


def test_coarse_vamp_and_c2f_separately():
    itf, models = make_interface()
    z = W.synth_codes(2, 14, 300, seed=8)
    torch.manual_seed(2)
    mask = itf.build_mask(z, periodic_prompt=5, upper_codebook_mask=4)
    a, am = O.coarse_vamp(models, z, mask, return_mask=True, seed=3, _sampling_steps=2)
    b, bm = itf.coarse_vamp(z, mask, return_mask=True, seed=3, _sampling_steps=2)
    assert torch.equal(a, b) and torch.equal(am, bm)
    torch.manual_seed(9)
    c = O.coarse_to_fine(models, a, mask=mask, _sampling_steps=2)
    torch.manual_seed(9)
    d = itf.coarse_to_fine(b, mask=mask, _sampling_steps=2)
    assert torch.equal(c, d)


# ---------------------------------------------------------------------------------------- 2-rank gloo
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, T, q, c2f_max_batch=16):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    itf, _ = make_interface(dist.group.WORLD, c2f_max_batch=c2f_max_batch)
    z = W.synth_codes(B, 14, T, seed=6)
    torch.manual_seed(3)
    mask = itf.build_mask(z)
    out = itf.vamp(z, mask, batch_size=B, seed=11, _sampling_steps=3)
    q.put((rank, out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B,T,c2f_max_batch", [(4, 200, 16), (3, 200, 16), (16, 120, 8)])
def test_batch_shard_two_ranks_gloo(B, T, c2f_max_batch):
    """Every rank holds the global batch, computes its contiguous block of items with the GLOBAL N0 and the global
    noise stream, then one all-gather: the result on every rank equals the unsharded oracle vamp()
    (B = 3: ragged shards 2 + 1; B = 16 with a c2f workspace of 8: the per-rank shape of BASELINE configs[3] — 8 items per rank,
    the coarse-to-fine chunks of a rank's items cut into launches of at most 8 — at a length the CPU finishes in seconds)."""
    _, models = make_interface()
    z = W.synth_codes(B, 14, T, seed=6)
    torch.manual_seed(3)
    mask = O.build_mask(z)
    ref = O.vamp(models, z, mask, batch_size=B, seed=11, _sampling_steps=3)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, T, q, c2f_max_batch)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        assert (got[r] == ref.numpy()).all(), f"rank {r} differs from the unsharded result"


def test_model_registry_beats_and_onset_surface(monkeypatch):
    """The rest of the reference Interface surface (interface.py:128-144, :226-321, :454-489): the model-zoo calls over a
    local registry, snap_to_beats / make_beat_mask over an injected tracker, build_mask with the onset and beat masks."""
    import numpy as np
    from vampnet_amd.codec import AudioSignal
    itf, _ = make_interface()
    monkeypatch.setattr(Interface, "model_zoo", {})
    assert Interface.available_models() == ["default"]
    itf.load_finetuned("default")                                   # nothing registered: keeps the loaded weights
    with pytest.raises(AssertionError):
        itf.load_finetuned("opera")
    Interface.register_model("opera", "/m/opera/coarse.pth", "/m/opera/c2f.pth")
    assert Interface.available_models() == ["opera", "default"]
    calls = []
    itf.reload = lambda coarse_ckpt=None, c2f_ckpt=None: calls.append((coarse_ckpt, c2f_ckpt))
    itf.load_finetuned("opera")
    assert calls == [("/m/opera/coarse.pth", "/m/opera/c2f.pth")]
    # s2t on arrays (beat times) like the reference
    assert itf.s2t(np.array([0.0, 0.5, 1.0])).tolist() == [0.0, 29.0, 58.0] and itf.s2t(0.5) == 29
    sr = itf.codec.sample_rate
    sig = AudioSignal(torch.zeros(1, 1, 3 * sr), sr)
    with pytest.raises(AssertionError):
        itf.make_beat_mask(sig)

    class Tracker:
        def extract_beats(self, signal):
            return np.array([0.5, 1.0, 1.5, 2.0, 2.5]), np.array([0.5, 2.5])

    itf.beat_tracker = Tracker()
    cut = itf.snap_to_beats(sig)
    assert cut.length == int(2.5 * sr) - int(0.5 * sr) and sig.length == 3 * sr          # trimmed copy, input untouched
    torch.manual_seed(0)
    m = itf.make_beat_mask(sig, after_beat_s=0.05)
    T = itf.s2t(3.0)
    assert m.shape == (1, 14, T) and m.dtype == torch.long
    open_cols = torch.nonzero(m[0, 0] == 0).flatten().tolist()
    w = itf.s2t(0.05)
    want = sorted({c for b in (0.5, 1.0, 1.5, 2.0, 2.5) for c in range(int(itf.s2t(b)), int(itf.s2t(b)) + w)})
    assert open_cols == want and torch.equal(m[0, 0], m[0, 13])
    # app.py:207-216: mask_and(build_mask, beat mask) — the caller re-masks the upper codebooks afterwards (codebook_mask)
    z = W.synth_codes(1, 14, T, seed=2)
    torch.manual_seed(1)
    bm = itf.build_mask(z, periodic_prompt=0)
    both = torch.min(bm, m)
    assert torch.equal(both, m) and bool((bm[:, 3:] == 1).all())
    # onset mask through build_mask
    y = np.zeros(3 * sr, dtype=np.float32)
    rng = np.random.default_rng(0)
    for t0 in (0.7, 1.9):
        i = int(t0 * sr)
        y[i:i + 4000] += (0.5 * rng.standard_normal(4000) * np.exp(-np.arange(4000) / 800.0)).astype(np.float32)
    y += 1e-4 * rng.standard_normal(len(y)).astype(np.float32)
    torch.manual_seed(1)
    om = itf.build_mask(z, sig=AudioSignal(torch.from_numpy(y)[None, None], sr), periodic_prompt=0, onset_mask_width=2)
    cols = torch.nonzero(om[0, 0] == 0).flatten().tolist()
    assert 4 <= len(cols) <= 8 and all(abs(c - itf.s2t(0.7)) <= 6 or abs(c - itf.s2t(1.9)) <= 6 for c in cols)
