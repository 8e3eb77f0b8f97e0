"""torch's global CPU generator, continued on the GPU (seeded parity mode at device speed).

`rng="torch"` replays the reference's noise by drawing it with torch on the host (0.7 s per clip: 2.4 M `exponential_`
samples per item per sampling step).  at::CPUGeneratorImpl is an mt19937 whose state torch exposes
(`torch.get_rng_state()`), and both distributions the sampling loop uses are simple functions of its sequential output
(vampnet_amd/csrc/torch_rng.hip), so the engine can continue the SAME stream on the device: the state is handed over
before a `generate()` call and the advanced state is written back afterwards — torch's generator ends exactly where
the reference's would, and later host-side draws (e.g. the next `build_mask`) see the reference's stream.

State blob layout (legacy THGeneratorState inside CPUGeneratorImplState, 5056 bytes, verified in tests):
  u64 seed | i32 left | i32 seeded | u64 next | u64 state[624] | normal-distribution cache (40 bytes, untouched here)
at::mt19937 emits state[next++] after `if (--left == 0) next_state()`, i.e. left == 625 - next, "block exhausted" is
(left, next) = (1, 624) or, right after seeding, (1, 0).
"""
import ctypes as C

import numpy as np
import torch

_STATE_OFF, _N = 24, 624


def parse_torch_rng_state(blob: torch.Tensor):
    """-> (state uint32[624], pos) with pos = index of the next word to emit, 624 = regenerate first."""
    b = blob.cpu().numpy()
    if b.size != 5056:
        raise ValueError(f"unexpected torch CPU RNG state size {b.size} (torch build with a different generator layout)")
    left = int(b[8:12].view(np.int32)[0])
    nxt = int(b[16:24].view(np.uint64)[0])
    state = b[_STATE_OFF:_STATE_OFF + _N * 8].view(np.uint64).astype(np.uint32)
    pos = _N if left == 1 else nxt
    if left != 1 and left != _N + 1 - nxt:
        raise ValueError(f"inconsistent mt19937 state: left={left}, next={nxt}")
    return state, pos


def patch_torch_rng_state(blob: torch.Tensor, state: np.ndarray, pos: int) -> torch.Tensor:
    """Inverse of parse_torch_rng_state on a copy of `blob` (seed, normal cache untouched)."""
    b = blob.cpu().numpy().copy()
    b[8:12].view(np.int32)[0] = _N + 1 - pos
    b[16:24].view(np.uint64)[0] = pos
    b[_STATE_OFF:_STATE_OFF + _N * 8].view(np.uint64)[:] = state.astype(np.uint64)
    return torch.from_numpy(b)


class DeviceTorchRng:
    """mt19937 state of torch's default CPU generator living in device memory between load_from_torch / store_to_torch."""

    def __init__(self, engine):
        self.engine, self.lib = engine, engine.lib
        dev = engine.device
        self.state = torch.zeros(_N, dtype=torch.int32, device=dev)       # uint32 bit patterns
        self.pos = torch.zeros(1, dtype=torch.int32, device=dev)
        self._raw = None
        self._blob = None

    def _scratch(self, n_words):
        if self._raw is None or self._raw.numel() < n_words:
            self._raw = torch.empty(n_words, dtype=torch.int32, device=self.engine.device)
        return self._raw

    def _with_state_copy(self, fn):
        """Run fn() on a copy of torch's CURRENT generator state loaded into the device stream, then restore the device state."""
        state, pos = parse_torch_rng_state(torch.get_rng_state())
        keep_state, keep_pos = self.state.clone(), self.pos.clone()
        try:
            self.state.copy_(torch.from_numpy(state.view(np.int32)))
            self.pos.fill_(pos)
            return fn()
        finally:
            self.state.copy_(keep_state)
            self.pos.copy_(keep_pos)

    def self_check(self, kind: str = "noise"):
        """Once per process and per kind: the device continuation rests on facts about THIS torch build's CPU generator —
        kind "noise" (the sampling loop): exponential_ = float(-log1p(-u53)) of consecutive word pairs, uniform_ = the 24-bit
        formula of one word; kind "mask" (build_mask): bernoulli(p tensor) = (24 bits of one word) * 2^-24 < p, randint = one
        word % range — all strictly sequential (tests/test_host_logic.py pins them where the CPU suite runs).  A build that
        vectorises one of them differently (e.g. an MKL/VSL path) would silently part from the reference, so compare a few dozen
        device samples with torch's own from a copy of the current generator.  Raises RuntimeError on a mismatch (the caller
        decides: the mask path falls back to the host twin, the noise path has rng='torch')."""
        if kind in DeviceTorchRng._checked:
            return
        if kind in DeviceTorchRng._failed:                 # a failed check is remembered too: every later call falls back at once
            raise RuntimeError(DeviceTorchRng._failed[kind])
        g = torch.Generator()
        g.set_state(torch.get_rng_state())
        dev = self.engine.device
        if kind == "noise":
            want_e = torch.empty(48).exponential_(1, generator=g)
            want_u = torch.empty(48).uniform_(1e-20, 1.0, generator=g)
            got_e, got_u = self._with_state_copy(lambda: (self.exponential_(torch.empty(48, device=dev)).cpu(),
                                                          self.uniform_(torch.empty(48, device=dev), 1e-20, 1.0).cpu()))
            ok = torch.equal(got_e, want_e) and torch.equal(got_u, want_u)
            msg = ("exponential_ / uniform_ differ: use rng='torch' (host-drawn noise) instead of 'torch_device'")
        elif kind == "mask":
            # the draws of vampnet/mask.py: torch.bernoulli(p tensor) (linear_random, periodic_mask's coins), torch.randint (roll,
            # dropout columns) — in the word arithmetic of vn_build_mask_kernel (csrc/elementwise.hip)
            p = torch.tensor([0.37, 0.5, 0.93, 1.0] * 12)
            want_b = torch.bernoulli(p, generator=g)
            want_r = torch.cat([torch.randint(0, r, (1,), generator=g) for r in (7, 575, 173, 13, 1000, 3) * 2])

            def words():
                raw = torch.empty(48 + 12, dtype=torch.int32, device=dev)
                self._gen(raw.data_ptr(), raw.numel())
                return raw.cpu().numpy().view(np.uint32)
            w = self._with_state_copy(words)
            u24 = (w[:48] & np.uint32(0xFFFFFF)).astype(np.float32) * np.float32(2.0 ** -24)
            got_b = torch.from_numpy((u24 < p.numpy()).astype(np.float32))
            got_r = torch.from_numpy((w[48:].astype(np.int64) % np.array([7, 575, 173, 13, 1000, 3] * 2, dtype=np.int64)))
            ok = torch.equal(got_b, want_b) and torch.equal(got_r, want_r)
            msg = ("bernoulli(p tensor) / randint differ: build_mask runs on the host twin instead (VN_MASK_ON_DEVICE=0 selects "
                   "it explicitly)")
        else:
            raise ValueError(kind)
        if not ok:
            DeviceTorchRng._failed[kind] = ("this torch build's CPU generator does not follow the sequential mt19937 formulas the "
                                            "device RNG continues — " + msg)
            raise RuntimeError(DeviceTorchRng._failed[kind])
        DeviceTorchRng._checked.add(kind)

    _checked = set()
    _failed = {}

    def load_from_torch(self, kind: str = "noise"):
        if getattr(self, "_producer", None) is None:
            self._producer = torch.cuda.current_stream(self.engine.device)
        self.self_check(kind)
        self._blob = torch.get_rng_state()
        state, pos = parse_torch_rng_state(self._blob)
        self.state.copy_(torch.from_numpy(state.view(np.int32)))
        self.pos.fill_(pos)

    def side_stream(self):
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(self.engine.device)
        return self._side

    def store_to_torch(self):
        """Synchronises with the stream that produced the noise, then advances torch's generator to where the device
        stream stands."""
        # download on the stream that PRODUCED the noise (set by draw_noise_device): in overlap mode that is the side stream —
        # the caller's stream may hold a long generate() queue — otherwise the current stream; a synchronous copy on any other
        # stream would read the state before the mt19937 kernels have advanced it
        prod = getattr(self, "_producer", None) or torch.cuda.current_stream(self.engine.device)
        with torch.cuda.stream(prod):
            state = self.state.cpu().numpy().view(np.uint32)
            pos = int(self.pos.item())
        torch.set_rng_state(patch_torch_rng_state(self._blob, state, pos))

    def _gen(self, raw_ptr, n):
        self.engine.check(self.lib.vn_mt19937_generate(self.engine.handle, self.state.data_ptr(), self.pos.data_ptr(), raw_ptr, n,
                                                       self.engine.stream()), "vn_mt19937_generate")

    def skip(self, n_words: int):
        if n_words > 0:
            self._gen(None, n_words)

    def exponential_(self, out: torch.Tensor):
        """out.exponential_(1) of a contiguous float32 tensor, from the device-resident stream (2 words per element)."""
        n = out.numel()
        raw = self._scratch(2 * n)
        self._gen(raw.data_ptr(), 2 * n)
        self.engine.check(self.lib.vn_torch_exponential_f32(self.engine.handle, raw.data_ptr(), out.data_ptr(), n,
                                                            self.engine.stream()), "vn_torch_exponential_f32")
        return out

    CHUNK_WORDS = 1 << 21            # words per parallel chunk of the jump-ahead path (1.4 ms of serial walk each)

    def exponential_block_(self, out: torch.Tensor, lead_words: int, total_words: int, chunk_words: int = None):
        """The `out.numel()` exponentials that start `lead_words` into a block of `total_words` stream words, advancing the
        generator by the WHOLE block (a rank's rows of a sharded batch: the rest of the block belongs to other ranks).
        Large blocks use the jump-ahead path (vampnet_amd/mt_jump.py): the generator states at the chunk starts and at the
        block end are computed with one workgroup each, then all chunks are walked in parallel."""
        from . import mt_jump
        chunk = self.CHUNK_WORDS if chunk_words is None else int(chunk_words)
        n = out.numel()
        length = 2 * n
        assert 0 <= lead_words and lead_words + length <= total_words
        if length < 2 * chunk:
            self.skip(lead_words)
            self.exponential_(out)
            self.skip(total_words - lead_words - length)
            return out
        nc = -(-length // chunk)
        key = (lead_words, length, total_words, chunk)
        cache = self.__dict__.setdefault("_poly_cache", {})
        if key not in cache:
            offs = [lead_words + c * chunk for c in range(nc)] + [total_words]
            polys = np.stack([mt_jump.jump_poly_words(o) for o in offs])
            cache[key] = torch.from_numpy(polys.view(np.int32)).to(self.engine.device)
        polys = cache[key]
        states = torch.empty(nc + 1, _N, dtype=torch.int32, device=self.engine.device)
        eng, st = self.engine, self.engine.stream()
        eng.check(self.lib.vn_mt19937_jump(eng.handle, self.state.data_ptr(), self.pos.data_ptr(), polys.data_ptr(), nc + 1,
                                           states.data_ptr(), st), "vn_mt19937_jump")
        raw = self._scratch(length)
        eng.check(self.lib.vn_mt19937_generate_chunks(eng.handle, states.data_ptr(), nc, raw.data_ptr(), chunk, length, st),
                  "vn_mt19937_generate_chunks")
        eng.check(self.lib.vn_torch_exponential_f32(eng.handle, raw.data_ptr(), out.data_ptr(), n, st), "vn_torch_exponential_f32")
        self.state.copy_(states[nc])                       # the block end, at position 0
        self.pos.zero_()
        return out

    def uniform_(self, out: torch.Tensor, lo: float, hi: float):
        n = out.numel()
        raw = self._scratch(n)
        self._gen(raw.data_ptr(), n)
        self.engine.check(self.lib.vn_torch_uniform_f32(self.engine.handle, raw.data_ptr(), out.data_ptr(), n, lo, hi,
                                                        self.engine.stream()), "vn_torch_uniform_f32")
        return out


RAW_WORDS_PER_ROUND = 160 << 20     # raw mt19937 words held at once by draw_units (640 MB): the exponentials of a call are produced in rounds


def draw_units(rng: DeviceTorchRng, units, B, N, V, b0, nb):
    """All the draws of one or more generate() calls from ONE generator state, on the current stream.

    `units` = the sampling steps in the reference's draw order, each (exp_dst or None, unif_dst): a step consumes
    `exponential_` over (B*N, V) when it samples (two words per element; rows [b0*N, (b0+nb)*N) land in exp_dst) and then
    `uniform_` over (B, N) (rows [b0, b0+nb) land in unif_dst).  Every offset is known before anything runs (static shapes), so the plan
    is two jump launches and two walks for the whole call instead of one jump + one walk per step, each waiting for the previous
    step's end state:
      level 1: the generator states at the start of every unit and at the end of the call, from (state, pos) — one polynomial per unit;
      level 2: from each unit's state, its chunk starts and the start of its uniform block — the SAME polynomials for every unit
               (vn_mt19937_jump_indexed);
      walks  : every chunk of every unit in one launch per round (RAW_WORDS_PER_ROUND), the uniform blocks in one more.
    The rng kernels of the per-step form were active during 95 ms of a 245 ms vamp() on ~19 CUs and slowed the GEMMs they shared CUs
    with by 5 ms in all (profiles/r05_rng_kernel_time.txt); this form has the same work done in a few milliseconds of the side stream.
    The generator is left at the end of the last unit."""
    from . import mt_jump
    eng, lib, dev = rng.engine, rng.lib, rng.engine.device
    st = eng.stream()
    E, U = 2 * B * N * V, B * N
    length, lead = 2 * nb * N * V, 2 * b0 * N * V
    chunk = rng.CHUNK_WORDS if length >= rng.CHUNK_WORDS else max(1024, -(-length // 1024) * 1024)
    nc = max(1, -(-length // chunk))
    samp = [u for u, (e, _) in enumerate(units) if e is not None]
    n_units, n_samp = len(units), len(samp)
    key = ("units", tuple(e is not None for e, _ in units), B, N, V, b0, nb, chunk)
    cache = rng.__dict__.setdefault("_plan_cache", {})
    if key not in cache:
        starts, off = [], 0
        for e, _ in units:
            starts.append(off)
            off += (E if e is not None else 0) + U
        # the end of the call as (state array one block back, position = what is left of it): torch's generator cannot hold "position 0 of a
        # fresh block" (at::mt19937 regenerates and consumes in one step: left = 625 is rejected by set_state)
        end_base = max(off - _N, 0)
        polys1 = np.stack([mt_jump.jump_poly_words(o) for o in starts + [end_base]])
        rel = [lead + k * chunk for k in range(nc)] + [E + b0 * N, b0 * N]      # chunk starts; uniform start of a sampling / non-sampling unit
        polys2 = np.stack([mt_jump.jump_poly_words(o) for o in rel])
        base_idx = [u for u in samp for _ in range(nc)] + list(range(n_units))
        poly_idx = [k for _ in samp for k in range(nc)] + [nc if units[u][0] is not None else nc + 1 for u in range(n_units)]
        as_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
        cache[key] = (as_dev(polys1), as_dev(polys2), torch.tensor(base_idx, dtype=torch.int32, device=dev),
                      torch.tensor(poly_idx, dtype=torch.int32, device=dev), off - end_base)
    polys1, polys2, base_idx, poly_idx, end_pos = cache[key]
    states1 = torch.empty(n_units + 1, _N, dtype=torch.int32, device=dev)
    eng.check(lib.vn_mt19937_jump(eng.handle, rng.state.data_ptr(), rng.pos.data_ptr(), polys1.data_ptr(), n_units + 1,
                                  states1.data_ptr(), st), "vn_mt19937_jump")
    if nb == 0 or N == 0:
        # a rank without items (world > batch): nothing to draw, but torch's generator still has to end where the unsharded call leaves
        # it — the level-1 jump alone gives that state (the chunk / uniform walks would be launches of zero words, which the C entries reject)
        rng.state.copy_(states1[n_units])
        rng.pos.fill_(end_pos)
        return
    n2 = n_samp * nc + n_units
    states2 = torch.empty(n2, _N, dtype=torch.int32, device=dev)
    eng.check(lib.vn_mt19937_jump_indexed(eng.handle, states1.data_ptr(), base_idx.data_ptr(), polys2.data_ptr(), poly_idx.data_ptr(), n2,
                                          states2.data_ptr(), st), "vn_mt19937_jump_indexed")
    per_round = max(1, RAW_WORDS_PER_ROUND // (nc * chunk))
    for r0 in range(0, n_samp, per_round):
        g = min(per_round, n_samp - r0)
        raw = rng._scratch(g * nc * chunk)
        eng.check(lib.vn_mt19937_generate_chunks(eng.handle, states2[r0 * nc:].data_ptr(), g * nc, raw.data_ptr(), chunk, g * nc * chunk, st),
                  "vn_mt19937_generate_chunks")
        for j in range(g):
            dst = units[samp[r0 + j]][0]
            assert dst.is_contiguous() and dst.numel() * 2 == length
            eng.check(lib.vn_torch_exponential_f32(eng.handle, raw[j * nc * chunk:].data_ptr(), dst.data_ptr(), length // 2, st),
                      "vn_torch_exponential_f32")
    nu = nb * N
    raw_u = torch.empty(n_units * nu, dtype=torch.int32, device=dev)
    eng.check(lib.vn_mt19937_generate_chunks(eng.handle, states2[n_samp * nc:].data_ptr(), n_units, raw_u.data_ptr(), nu, n_units * nu, st),
              "vn_mt19937_generate_chunks")
    for u, (_, udst) in enumerate(units):
        assert udst.is_contiguous() and udst.numel() == nu
        eng.check(lib.vn_torch_uniform_f32(eng.handle, raw_u[u * nu:].data_ptr(), udst.data_ptr(), nu, 1e-20, 1.0, st), "vn_torch_uniform_f32")
    rng.state.copy_(states1[n_units])
    rng.pos.fill_(end_pos)


def draw_noise_device(rng: DeviceTorchRng, B, N, V, steps, sample_cutoff, b0=0, nb=None, overlap=False):
    """Device twin of engine.draw_noise_host: the same ledger (exp [steps, nb*N, V], zeros on non-sampling steps; unif
    [steps, nb, N]) for items [b0, b0+nb) of a global batch B, produced from — and advancing — torch's CPU generator.

    overlap=False: produced on the current stream, torch's generator updated before returning -> (exp, unif).
    overlap=True : produced on the rng's side stream with one event per step -> (exp, unif, events); the caller makes its
    consumer wait on events[i] and calls rng.store_to_torch() AFTER enqueueing its own work.  The side stream does NOT wait for the
    caller's stream: the draw shapes are static and every earlier use of the generator ended with
    store_to_torch(), which returns only when its mt19937 kernels have run — so the words of this call are generated underneath
    whatever the caller's stream still has queued (the previous stage's forwards)."""
    nb = B if nb is None else nb
    dev = rng.engine.device
    cur = torch.cuda.current_stream(dev)
    side = rng.side_stream() if overlap else cur
    events = []
    rng._producer = side
    with torch.cuda.stream(side):
        exp = torch.zeros(steps, nb * N, V, dtype=torch.float32, device=dev)
        unif = torch.empty(steps, nb, N, dtype=torch.float32, device=dev)
        rng.load_from_torch()
        if overlap:
            # the whole call's draws as one plan (draw_units): ready a few milliseconds into the first forward; every step waits on the one event
            draw_units(rng, [(exp[i] if (i / steps) <= sample_cutoff else None, unif[i]) for i in range(steps)], B, N, V, b0, nb)
            ev = torch.cuda.Event()
            ev.record(side)
            events = [ev] * steps
        else:
            for i in range(steps):
                if (i / steps) <= sample_cutoff:                      # transformer.py:852-855
                    rng.exponential_block_(exp[i], 2 * b0 * N * V, 2 * B * N * V)
                rng.skip(b0 * N)
                rng.uniform_(unif[i], 1e-20, 1.0)
                rng.skip((B - b0 - nb) * N)
            rng.store_to_torch()
    if overlap:
        exp.record_stream(cur)
        unif.record_stream(cur)
        return exp, unif, events
    return exp, unif


def draw_noise_device_calls(rng: DeviceTorchRng, n_calls, B, N, V, steps, sample_cutoff, b0=0, nb=None):
    """The noise of `n_calls` consecutive reference generate() calls of identical shape (Interface.coarse_to_fine's chunks,
    interface.py:360-374), drawn call after call in the reference's order, written STRAIGHT into the ledger of the one device batch
    that runs them: exp [steps, n_calls * nb * N, V], unif [steps, n_calls * nb, N] (call c = rows [c nb, (c + 1) nb) of every step).
    Produced on the rng's side stream without waiting for the caller's stream (see draw_noise_device): while the coarse stage still
    runs, the c2f stage's words are already being generated.  The caller's stream waits for the ledger (one event); torch's
    generator is advanced before returning (the host waits for the side stream only)."""
    nb = B if nb is None else nb
    dev = rng.engine.device
    cur = torch.cuda.current_stream(dev)
    side = rng.side_stream()
    rng._producer = side
    with torch.cuda.stream(side):
        exp = torch.zeros(steps, n_calls * nb * N, V, dtype=torch.float32, device=dev)
        unif = torch.empty(steps, n_calls * nb, N, dtype=torch.float32, device=dev)
        rng.load_from_torch()
        draw_units(rng, [(exp[i, c * nb * N:(c + 1) * nb * N] if (i / steps) <= sample_cutoff else None, unif[i, c * nb:(c + 1) * nb])
                         for c in range(n_calls) for i in range(steps)], B, N, V, b0, nb)
        done = torch.cuda.Event()
        done.record(side)
        rng.store_to_torch()
    exp.record_stream(cur)
    unif.record_stream(cur)
    cur.wait_event(done)
    return exp, unif
