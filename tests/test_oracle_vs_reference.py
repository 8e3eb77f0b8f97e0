"""Pins oracle/vampnet_oracle.py against the reference's OWN Python, imported unmodified through
oracle/ref_shim.py.  Runs only where /root/reference exists (this container); the GPU box relies on
the committed fixtures in tests/golden/ (tests/test_oracle_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import ref_shim, vampnet_oracle as O, weights as W

pytestmark = [pytest.mark.reference,
              pytest.mark.skipif(not ref_shim.reference_available(), reason="/root/reference absent")]


@pytest.fixture(scope="module")
def ns():
    return ref_shim.load_reference()


@pytest.fixture(scope="module")
def tiny(ns):
    cb = W.synth_codebooks()
    csd, fsd = W.synth_state_dict(W.TINY_COARSE_DIMS, 0), W.synth_state_dict(W.TINY_C2F_DIMS, 1)
    coarse = ref_shim.build_reference_model(ns, W.TINY_COARSE_DIMS, csd)
    c2f = ref_shim.build_reference_model(ns, W.TINY_C2F_DIMS, fsd)
    codec = ref_shim.FakeCodec(cb)
    itf = ref_shim.build_reference_interface(ns, coarse, c2f, codec)
    models = O.OracleModels(csd, W.TINY_COARSE_DIMS, fsd, W.TINY_C2F_DIMS, cb)
    return dict(coarse=coarse, c2f=c2f, codec=codec, itf=itf, models=models, cb=cb, csd=csd, fsd=fsd)


def test_bucket_table_known_answers(ns):
    """SURVEY.md App. B table."""
    rel = torch.arange(-130, 131)
    got = O.relative_position_bucket(rel)
    attn = ns.transformer.MultiHeadRelativeAttention(4, 64)
    assert torch.equal(got, attn._relative_position_bucket(rel))
    expect = {0: 0, -1: 1, -7: 7, -8: 8, -11: 8, -12: 9, -15: 9, -16: 10, -22: 10, -23: 11, -31: 11,
              -32: 12, -45: 12, -46: 13, -63: 13, -64: 14, -90: 14, -91: 15, -130: 15,
              1: 17, 7: 23, 8: 24, 11: 24, 12: 25, 16: 26, 23: 27, 32: 28, 46: 29, 64: 30, 90: 30, 91: 31}
    for r, b in expect.items():
        assert got[r + 130].item() == b, (r, got[r + 130].item(), b)
    assert 16 not in got.tolist()


@pytest.mark.parametrize("which,T,B", [("coarse", 50, 2), ("c2f", 37, 1)])
def test_forward_bitwise(tiny, which, T, B):
    dims = W.TINY_COARSE_DIMS if which == "coarse" else W.TINY_C2F_DIMS
    sd = tiny["csd"] if which == "coarse" else tiny["fsd"]
    model = tiny[which]
    codes = W.synth_codes(B, dims["n_codebooks"], T, seed=3)
    codes[:, :, ::3] = 1024
    with torch.inference_mode():
        lat_ref = model.embedding.from_codes(codes, tiny["codec"])
        ref, acts_ref = model(lat_ref, return_activations=True)
        lat = O.from_codes(sd, tiny["cb"], codes)
        out, acts = O.forward(sd, dims, lat, return_activations=True)
    assert torch.equal(lat, lat_ref)
    assert torch.equal(acts, acts_ref)
    assert torch.equal(out, ref)


def test_multinomial_is_exp_race():
    """multinomial(1) == argmax(p / Exp(1)) with identical stream position."""
    torch.manual_seed(5)
    p = torch.rand(300, 1024).softmax(-1)
    torch.manual_seed(11)
    a = p.multinomial(1).squeeze(1)
    after_a = torch.rand(4)
    torch.manual_seed(11)
    q = torch.empty_like(p).exponential_(1)
    b = (p / q).argmax(-1)
    after_b = torch.rand(4)
    assert torch.equal(a, b) and torch.equal(after_a, after_b)


GEN_CASES = [
    dict(B=1, T=50, kw=dict(_sampling_steps=6, seed=0)),
    dict(B=3, T=41, kw=dict(_sampling_steps=5, seed=1, temperature=0.8, mask_temperature=7.0)),
    dict(B=1, T=50, kw=dict(_sampling_steps=6, seed=2, sample_cutoff=-1, mask_temperature=0.0)),  # true greedy
    dict(B=2, T=33, kw=dict(_sampling_steps=4, seed=3, top_p=0.9)),
    dict(B=1, T=50, kw=dict(_sampling_steps=4, seed=4, temperature=1e-8)),
    dict(B=2, T=29, kw=dict(_sampling_steps=3, seed=5, temperature=0.0, sample_cutoff=0.5)),
    # cfg_guidance: the reference doubles the batch with an all-MASK copy and then discards the guided logits (transformer.py:845-847)
    dict(B=2, T=31, kw=dict(_sampling_steps=4, seed=6, cfg_guidance=3.0)),
    dict(B=1, T=40, kw=dict(_sampling_steps=3, seed=7, cfg_guidance=0.5, temperature=0.9, top_p=0.95)),
]


@pytest.mark.parametrize("case", GEN_CASES)
@pytest.mark.parametrize("which", ["coarse", "c2f"])
def test_generate_bitwise(tiny, which, case):
    dims = W.TINY_COARSE_DIMS if which == "coarse" else W.TINY_C2F_DIMS
    sd = tiny["csd"] if which == "coarse" else tiny["fsd"]
    model = tiny[which]
    B, T = case["B"], case["T"]
    z = W.synth_codes(B, dims["n_codebooks"], T, seed=9)
    torch.manual_seed(123)
    mask = (torch.rand(B, dims["n_codebooks"], T) < 0.7).long()
    mask[:, :dims["n_cond"]] = 0
    ref = model.generate(codec=tiny["codec"], start_tokens=z.clone(), mask=mask.clone(),
                         return_signal=False, typical_filtering=True, **case["kw"])
    got = O.generate(sd, dims, tiny["cb"], z.clone(), mask.clone(), **O._gen_kwargs(dict(case["kw"])))
    assert torch.equal(ref, got)
    # RNG stream position identical afterwards
    torch.manual_seed(77)
    model.generate(codec=tiny["codec"], start_tokens=z.clone(), mask=mask.clone(), return_signal=False,
                   **{k: v for k, v in case["kw"].items() if k != "seed"})
    a = torch.rand(3)
    torch.manual_seed(77)
    O.generate(sd, dims, tiny["cb"], z.clone(), mask.clone(),
               **O._gen_kwargs({k: v for k, v in case["kw"].items() if k != "seed"}))
    assert torch.equal(a, torch.rand(3))


def test_typical_filtering_is_noop(tiny):
    dims, model = W.TINY_COARSE_DIMS, tiny["coarse"]
    z = W.synth_codes(1, 4, 40, seed=2)
    mask = torch.ones_like(z)
    a = model.generate(codec=tiny["codec"], start_tokens=z, mask=mask, return_signal=False,
                       _sampling_steps=4, seed=0, typical_filtering=True)
    b = model.generate(codec=tiny["codec"], start_tokens=z, mask=mask, return_signal=False,
                       _sampling_steps=4, seed=0, typical_filtering=False)
    assert torch.equal(a, b)


def test_batch_schedule_quirk(tiny):
    """S=6, T=50, C=4, period-7 prompt: masked counts after each step."""
    z = W.synth_codes(1, 4, 50, seed=2)
    m = O.periodic_mask(z, 7, 1, random_roll=False)
    for B, expect in ((1, [162, 145, 118, 83, 43, 0]), (4, [167, 166, 165, 164, 163, 0])):
        trace = []
        O.generate(tiny["csd"], W.TINY_COARSE_DIMS, tiny["cb"], z.expand(B, -1, -1), m.expand(B, -1, -1),
                   sampling_steps=6, seed=0, trace=trace)
        counts = [int((t["z_out"][0] == 1024).sum()) for t in trace]
        assert counts == expect, (B, counts)


@pytest.mark.parametrize("kw", [dict(), dict(periodic_prompt=5, upper_codebook_mask=2, _dropout=0.1),
                                dict(rand_mask_intensity=0.8, prefix_s=0.2, suffix_s=0.1, periodic_prompt=0),
                                dict(periodic_prompt=13, periodic_prompt_width=3, ncc=1)])
def test_build_mask_bitwise(tiny, kw):
    z = W.synth_codes(2, 14, 120, seed=4)
    for seed in (0, 1, 2):
        torch.manual_seed(seed)
        ref = tiny["itf"].build_mask(z, **kw)
        a = torch.rand(2)
        torch.manual_seed(seed)
        got = O.build_mask(z, **kw)
        assert torch.equal(ref, got) and torch.equal(a, torch.rand(2))


def test_rms_mask_artifact():
    """scratch/rms_mask.txt: saved build_mask(periodic_prompt=7, upper_codebook_mask=3) (roll 0), 14 x 100."""
    import os
    path = os.path.join(ref_shim.REFERENCE_ROOT, "scratch", "rms_mask.txt")
    ref = torch.from_numpy(np.loadtxt(path).astype(np.int64))
    z = torch.zeros(1, 14, ref.shape[1], dtype=torch.long)
    m = O.codebook_mask(O.periodic_mask(z, 7, 1, random_roll=False), 3)[0]
    assert torch.equal(m, ref)


@pytest.mark.parametrize("B,kw", [(1, dict(seed=0, _sampling_steps=4)),
                                  (2, dict(seed=1, _sampling_steps=3, temperature=0.9)),
                                  (1, dict(seed=2, _sampling_steps=4, sample_cutoff=-1, mask_temperature=0.0))])
def test_vamp_bitwise(tiny, B, kw):
    """Whole Interface.vamp() (coarse chunks + c2f chunks) incl. T not a multiple of either chunk."""
    itf, models = tiny["itf"], tiny["models"]
    T = 600          # > one coarse chunk (575): exercises chunking + edge un-mask; c2f pads 600 -> 692
    z = W.synth_codes(1, 14, T, seed=6)
    torch.manual_seed(3)
    mask = O.build_mask(z)
    ref, ref_mask = itf.vamp(z, mask, batch_size=B, return_mask=True, **kw)
    got, got_mask = O.vamp(models, z, mask, batch_size=B, return_mask=True, **kw)
    assert torch.equal(ref, got)
    assert torch.equal(ref_mask, got_mask)


# ----------------------------------------------------------------------------- training step
def _injected_dropout(model, masks, p):
    """Replace every nn.Dropout of the reference stack by `x * keep * 1/(1-p)` with the injected keep-mask.
    Per layer the reference calls: self_attn.dropout (probabilities), layer.dropout (res1), feed_forward.drop,
    layer.dropout (res2) — transformer.py:250, :347, :82, :367."""
    import types
    for i, layer in enumerate(model.transformer.layers):
        calls = {"n": 0}

        def attn_drop(self, x, i=i):
            return x * masks[(i, "attn")] * (1.0 / (1.0 - p))

        def ffn_drop(self, x, i=i):
            return x * masks[(i, "ffn")] * (1.0 / (1.0 - p))

        def res_drop(self, x, i=i, calls=calls):
            site = "res1" if calls["n"] % 2 == 0 else "res2"
            calls["n"] += 1
            return x * masks[(i, site)] * (1.0 / (1.0 - p))

        layer.self_attn.dropout.forward = types.MethodType(attn_drop, layer.self_attn.dropout)
        layer.feed_forward.drop.forward = types.MethodType(ffn_drop, layer.feed_forward.drop)
        layer.dropout.forward = types.MethodType(res_drop, layer.dropout)


@pytest.mark.parametrize("which,p", [("coarse", 0.1), ("c2f", 0.1), ("coarse", 0.0)])
def test_train_step_vs_reference(ns, tiny, which, p):
    """scripts/exp/train.py:237-304 restated: loss, every gradient, grad norm and the AdamW/Noam update are checked
    against the reference's modules driven by torch.optim.AdamW + clip_grad_norm_ + vampnet.scheduler.NoamScheduler."""
    import importlib
    from oracle import train_oracle as TO
    sched_mod = importlib.import_module("vampnet.scheduler")
    dims = W.TINY_COARSE_DIMS if which == "coarse" else W.TINY_C2F_DIMS
    sd = tiny["csd"] if which == "coarse" else tiny["fsd"]
    model = ref_shim.build_reference_model(ns, dims, sd)
    model.train()
    B, T = 2, 24
    z = W.synth_codes(B, dims["n_codebooks"], T, seed=5)
    g = torch.Generator().manual_seed(3)
    r = torch.tensor([0.3, 0.8])
    mask = TO.make_training_mask(z, r, dims["n_cond"], generator=g)
    # the reference's own mask pipeline on the same generator state gives the same mask (mask.py:40-54)
    torch.manual_seed(9)
    m_ref = ns.mask.codebook_unmask(ns.mask.random(z, r), dims["n_cond"])
    torch.manual_seed(9)
    assert torch.equal(m_ref, TO.make_training_mask(z, r, dims["n_cond"]))
    masks = TO.draw_dropout_masks(dims, B, T, p, torch.Generator().manual_seed(4)) if p > 0 else None
    if masks is not None:
        _injected_dropout(model, masks, p)
    else:
        for mod in model.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0

    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    sched = sched_mod.NoamScheduler(opt, d_model=dims["d_model"], factor=2.0, warmup=10000)
    sched.step()                                                      # train.py:655-ish: lr is set before the first step
    state = {}
    cur = {k: v.clone() for k, v in sd.items()}
    for it in range(2):
        z_mask, mk = ns.mask.apply_mask(z, mask, model.mask_token)
        lat = model.embedding.from_codes(z_mask, tiny["codec"])
        z_hat = model(lat)
        target = ns.util.codebook_flatten(z[:, dims["n_cond"]:, :])
        flat_mask = ns.util.codebook_flatten(mk[:, dims["n_cond"]:, :])
        t_masked = target.masked_fill(~flat_mask.bool(), -100)
        loss_ref = torch.nn.CrossEntropyLoss(label_smoothing=0.1)(z_hat, t_masked)
        loss_ref.backward()
        gref = {k: v.grad.clone() for k, v in model.named_parameters()}
        norm_ref = torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)

        loss, grads, logits = TO.loss_and_grads(cur, dims, tiny["cb"], z, mask, masks, p)
        assert torch.equal(logits, z_hat.detach())
        assert torch.equal(loss, loss_ref.detach())
        assert set(grads) == set(gref)
        for k in gref:
            assert torch.equal(grads[k], gref[k]), k
        lr = TO.noam_lr(it + 1, dims["d_model"])
        assert lr == opt.param_groups[0]["lr"]
        cur, norm = TO.clip_and_adamw(cur, grads, state, lr)
        opt.step()
        opt.zero_grad()
        sched.step()
        torch.testing.assert_close(norm, norm_ref, rtol=1e-6, atol=0)
        for k, v in model.state_dict().items():
            torch.testing.assert_close(cur[k], v, rtol=1e-6, atol=1e-9, msg=k)


class _FakeTracker:
    """Stands in for the WaveBeat tracker (out of scope): fixed beat / down-beat times in seconds."""
    def __init__(self, beats, downbeats):
        self.beats, self.downbeats = np.asarray(beats, dtype=np.float64), np.asarray(downbeats, dtype=np.float64)

    def extract_beats(self, signal):
        return self.beats, self.downbeats


class _Sig:
    def __init__(self, duration):
        self.duration = duration


@pytest.mark.parametrize("kw", [dict(), dict(after_beat_s=0.1, before_beat_s=0.05), dict(dropout=0.4, after_beat_s=0.08),
                                dict(mask_downbeats=False, beat_downsample_factor=2), dict(mask_upbeats=False, invert=False),
                                dict(downbeat_downsample_factor=2, after_beat_s=0.3)])
def test_make_beat_mask_bitwise(tiny, kw):
    """Interface.make_beat_mask (interface.py:241-321) with an injected tracker: the product's host arithmetic
    (vampnet_amd/masks.py::beat_mask) equals the reference's own method, incl. its bernoulli draws per beat window, the
    down-beats-are-not-up-beats rule, python slice semantics at the clip start and the 14-codebook repeat."""
    from vampnet_amd import masks as M
    beats = np.array([0.01, 0.52, 1.03, 1.49, 2.0, 2.51, 3.02, 3.55, 4.01, 4.6, 5.2, 5.9])
    downbeats = beats[::4]
    itf = tiny["itf"]
    itf.beat_tracker = _FakeTracker(beats, downbeats)
    try:
        for seed in (0, 1):
            torch.manual_seed(seed)
            ref = itf.make_beat_mask(_Sig(6.2), **kw)
            a = torch.rand(2)
            torch.manual_seed(seed)
            got = M.beat_mask(beats, downbeats, 6.2, itf.codec.sample_rate, itf.codec.hop_length, 14, **kw)
            assert ref.dtype == got.dtype and torch.equal(ref, got) and torch.equal(a, torch.rand(2))
    finally:
        itf.beat_tracker = None


def test_mask_module_names_bitwise(ns):
    """vampnet_amd.masks exposes vampnet/mask.py's function names and signatures: each equals the reference's function bit for
    bit AND leaves torch's CPU generator at the same position (the functions are called in the same order under one seed)."""
    from vampnet_amd import masks as M
    R = ns.mask if hasattr(ns, "mask") else None
    if R is None:
        import importlib
        R = importlib.import_module("vampnet.mask")
    x = W.synth_codes(3, 14, 97, seed=8)
    r = torch.tensor([0.1, 0.5, 0.9])

    def both(f):
        torch.manual_seed(5)
        a = f(R)
        ta = torch.rand(3)
        torch.manual_seed(5)
        b = f(M)
        tb = torch.rand(3)
        assert torch.equal(ta, tb), "RNG position differs"
        if isinstance(a, tuple):
            assert all(torch.equal(u, v) and u.dtype == v.dtype for u, v in zip(a, b))
        else:
            assert a.dtype == b.dtype and torch.equal(a, b)
        return b

    both(lambda P: P._gamma(r))
    both(lambda P: P._invgamma(0.3))
    both(lambda P: P._invgamma(torch.tensor([0.2, 0.9])))
    both(lambda P: P.full_mask(x))
    both(lambda P: P.empty_mask(x))
    both(lambda P: P.random(x, r))
    both(lambda P: P.random(x, 0.7))
    both(lambda P: P.linear_random(x, 0.4))
    both(lambda P: P.linear_random(x, torch.tensor([0.2, 0.5, 1.0])[:, None, None]))
    both(lambda P: P.inpaint(x, 5, 9))
    for P in (R, M):                       # per-item tensors: `if n_prefix > 0` is ambiguous in the reference too
        with pytest.raises(RuntimeError):
            P.inpaint(x, torch.tensor([0, 3, 7]), torch.tensor([2, 0, 11]))
    both(lambda P: P.inpaint(x, 0, 0))
    both(lambda P: P.periodic_mask(x, 7, 1, random_roll=True))
    both(lambda P: P.periodic_mask(x, 5, 3, random_roll=False))
    both(lambda P: P.periodic_mask(x, 0))
    both(lambda P: P.periodic_mask(x, torch.tensor([6]), 2, random_roll=True))     # 1-element tensor: only item 0 is marked
    for P in (R, M):
        with pytest.raises(RuntimeError):
            P.periodic_mask(x, torch.tensor([3, 0, 8]), 2)
    m1 = both(lambda P: P.periodic_mask(x, 4))
    m2 = both(lambda P: P.linear_random(x, 0.5))
    both(lambda P: P.mask_and(m1, m2))
    both(lambda P: P.mask_or(m1, m2))
    both(lambda P: P.codebook_unmask(m2, 4))
    both(lambda P: P.codebook_unmask(m2, None))
    both(lambda P: P.codebook_mask(m1, 3))
    both(lambda P: P.dropout(m1, 0.2))
    both(lambda P: P.dropout(m1, 0.0))
    both(lambda P: P.time_stretch_mask(x, 3))
    both(lambda P: P.apply_mask(x, m2, 1024))


def test_util_module_names_bitwise(ns):
    import importlib
    from vampnet_amd import util as U
    R = importlib.import_module("vampnet.util")
    z = W.synth_codes(3, 14, 41, seed=1)
    f = R.codebook_flatten(z)
    assert torch.equal(U.codebook_flatten(z), f)
    assert torch.equal(U.codebook_unflatten(f, 14), R.codebook_unflatten(f, n_c=14)) and torch.equal(U.codebook_unflatten(f, 14), z)
    for v in (3, 0.25):
        a, b = R.scalar_to_batch_tensor(v, 5), U.scalar_to_batch_tensor(v, 5)
        assert a.dtype == b.dtype and torch.equal(a, b)
