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
