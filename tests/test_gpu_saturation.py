"""-m gpu: the opt-in precision "f16x2" never hands a caller a clamped result.

fp16 planes hold |x| < 65504 (attention values: |v| < 4094, they carry a factor 16).  Every fp16 plane writer of the library records a
value it had to clamp on the context's saturation ledger (vn_saturation_flags); the host side reads the ledger after every generate() /
forward / codec call made in this precision and repeats a call that saturated on "bf16x3" (engine.PrecisionFallbackWarning).  The
tests drive that with "trained-like" synthetic models (vampnet_amd/synth.py heavy_tail=): outlier residual channels, norm gains up to
30, log-normal row scales — once with every operand still inside fp16's range ("in_range": f16x2 must STAY f16x2 and match the oracle)
and once with a GEGLU unit at 3.6e5 and attention values at 6e3 ("saturating": f16x2 must notice and end on bf16x3's tokens).
bf16x3 — the engine's default — must match the oracle on both."""
import os
import warnings

import pytest
import torch

from oracle import vampnet_oracle as O, weights as W
from tests.gpu_common import SynthCodec, model_kwargs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from vampnet_amd.engine import Engine
    return Engine("cuda:0")


# ---------------------------------------------------------------------------------------- the ledger itself
def test_ledger_words(eng):
    """each word is set by its class of writer, is sticky until read with clear, and stays zero for values that fit"""
    eng.saturation(clear=True)
    x = torch.randn(64, 128, device="cuda") * 1000.0
    eng.split2h(x)
    assert eng.saturation(clear=False) == (0, 0, 0, 0)
    x[3, 5] = 7.0e4
    eng.split2h(x)                                                           # the weight / operand builder
    assert eng.saturation(clear=False) == (0, 0, 1, 0)
    assert eng.saturation(clear=True) == (0, 0, 1, 0)                        # sticky until cleared
    assert eng.saturation(clear=True) == (0, 0, 0, 0)
    x[3, 5] = float("nan")
    eng.split2h(x, tiled=True)
    assert eng.saturation(clear=True) == (0, 0, 1, 0)                        # NaN does not fit either
    # attention operands: q / 8, k, 16 v
    B, H, T = 1, 2, 40
    q, k, v = (torch.randn(B, H, T, 64, device="cuda") for _ in range(3))
    bias = torch.randn(32, H, device="cuda")
    eng.attention(q, k, v, bias, precision="f16x2")
    assert eng.saturation(clear=True) == (0, 0, 0, 0)
    v2 = v.clone()
    v2[0, 1, 7, 9] = 4.2e3                                                   # 16 v = 67200 > 65504
    eng.attention(q, k, v2, bias, precision="f16x2")
    assert eng.saturation(clear=True) == (0, 1, 0, 0)
    eng.attention(q, k, v2, bias, precision="bf16x3")                        # the default precision has no such limit
    assert eng.saturation(clear=True) == (0, 0, 0, 0)


def test_codec_on_the_models_engine_leaves_the_models_flags_alone(eng):
    """The ledger is per context.  A DacCodec built on a model's engine runs INSIDE that model's generate(return_signal=True): whatever
    the ledger holds when the codec call starts is the model's — the codec must neither take it for its own (a spurious fall-back to
    bf16x3) nor clear it (the model's check after the call would see zeros and return clamped tokens)."""
    from oracle import dac_oracle as D
    from vampnet_amd.codec import DacCodec
    from vampnet_amd.engine import PrecisionFallbackWarning
    cfg = D.DAC_TINY_CFG
    codec = DacCodec(D.synth_dac_state_dict(cfg, 0), cfg, engine=eng, precision="f16x2")
    assert codec.precision == "f16x2"
    codes = torch.randint(0, 1024, (2, cfg["n_codebooks"], 24))
    clean = codec.decode_codes(codes).clone()
    eng.saturation(clear=True)
    x = torch.randn(64, 128, device="cuda")
    x[3, 5] = 7.0e4
    eng.split2h(x)                                                           # "the model" leaves a flag behind (word 2 here)
    with warnings.catch_warnings():
        warnings.simplefilter("error", PrecisionFallbackWarning)
        audio = codec.decode_codes(codes)                                    # the codec's own call: in range
        enc = codec.encode(torch.zeros(1, 1, 4 * codec.hop_length, device="cuda"))
    assert codec.precision == "f16x2" and torch.equal(audio, clean)
    assert enc["codes"].shape[-1] == 4
    assert eng.saturation(clear=False) == (0, 0, 1, 0)                       # still there for its owner
    assert eng.saturation(clear=True) == (0, 0, 1, 0)
    assert eng.saturation(clear=True) == (0, 0, 0, 0)


def test_ledger_gemm_epilogue_and_norm(eng):
    """the plane-writing producers of the model path: RMSNorm rows and the GEGLU epilogue (direct and LDS-staged forms)"""
    from vampnet_amd import _lib
    from vampnet_amd.engine import VampNetModel
    dims = W.TINY_COARSE_DIMS
    cb = W.synth_codebooks()
    sd = {k: v.clone() for k, v in W.synth_state_dict(dims, 0).items()}
    os.environ["VN_F16X2_PROBE"] = "0"
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("error")                                    # building it must not warn: the weights fit
            m = VampNetModel(eng, sd, cb, max_batch=1, max_T=64, precision="f16x2", **model_kwargs(dims))
    finally:
        del os.environ["VN_F16X2_PROBE"]
    assert m.precision == "f16x2"
    codes = W.synth_codes(1, dims["n_codebooks"], 48, seed=5)
    m._ledger_off = True                     # skip forward_codes' own ledger handling: the test reads the words itself
    eng.saturation(clear=True)
    m.forward_codes(codes)
    assert eng.saturation(clear=True) == (0, 0, 0, 0)
    # norm gain 1e5 on one channel -> a normalised row value beyond 65504 (RMSNorm writer)
    sd2 = {k: v.clone() for k, v in sd.items()}
    sd2["transformer.layers.1.norm_3.weight"][17] = 1.0e5
    os.environ["VN_F16X2_PROBE"] = "0"
    try:
        m2 = VampNetModel(eng, sd2, cb, max_batch=1, max_T=64, precision="f16x2", **model_kwargs(dims))
    finally:
        del os.environ["VN_F16X2_PROBE"]
    m2._ledger_off = True
    eng.saturation(clear=True)
    m2.forward_codes(codes)
    assert eng.saturation(clear=True)[0] == 1


# ---------------------------------------------------------------------------------------- tiny heavy-tailed models
def _tiny(kind):
    cb = W.synth_codebooks()
    return cb, W.synth_state_dict(W.TINY_COARSE_DIMS, 0, heavy_tail=kind), W.synth_state_dict(W.TINY_C2F_DIMS, 1, heavy_tail=kind)


@pytest.mark.parametrize("kind", ["in_range", "saturating"])
@pytest.mark.parametrize("precision", ["bf16x3", "f16x2", "f32"])
def test_tiny_heavy_tail_generate_vs_oracle(eng, kind, precision):
    """generate() of a trained-like tiny model, seeded, vs the oracle.  f16x2 / saturating is built with the probe OFF so that the
    GENERATE-time ledger check is what has to catch it (the probe path is covered below)."""
    from vampnet_amd.engine import PrecisionFallbackWarning, VampNetModel
    dims = W.TINY_COARSE_DIMS
    cb, sd, _ = _tiny(kind)
    z = W.synth_codes(2, 4, 96, seed=8)
    mask = O.codebook_mask(O.periodic_mask(z, 5, 1), 3)
    ref = O.generate(sd, dims, cb, z, mask, sampling_steps=6, seed=11)
    os.environ["VN_F16X2_PROBE"] = "0"
    try:
        model = VampNetModel(eng, sd, cb, max_batch=2, max_T=96, precision=precision, **model_kwargs(dims))
    finally:
        del os.environ["VN_F16X2_PROBE"]
    assert model.precision == precision
    if precision == "f16x2" and kind == "saturating":
        with pytest.warns(PrecisionFallbackWarning, match="generate"):
            got = model.generate(return_signal=False, start_tokens=z, mask=mask, _sampling_steps=6, seed=11).cpu()
        assert model.precision == "bf16x3"
    else:
        with warnings.catch_warnings():
            warnings.simplefilter("error", PrecisionFallbackWarning)
            got = model.generate(return_signal=False, start_tokens=z, mask=mask, _sampling_steps=6, seed=11).cpu()
        assert model.precision == precision
    same = (got == ref).float().mean().item()
    print(f"tiny heavy-tail [{kind}, {precision}]: token agreement {same:.6f}")
    assert torch.equal(got, ref)


def test_tiny_saturating_probe_and_unseeded_rerun(eng):
    """(1) with the probe on, a saturating model never starts in f16x2; (2) a generate() WITHOUT seed= that saturates is repeated from
    the generator state it started from: same tokens as a bf16x3 model run from that state, and torch's generator ends where one
    call leaves it"""
    from vampnet_amd.engine import PrecisionFallbackWarning, VampNetModel
    dims = W.TINY_COARSE_DIMS
    cb, sd, _ = _tiny("saturating")
    with pytest.warns(PrecisionFallbackWarning, match="probe forward"):
        m = VampNetModel(eng, sd, cb, max_batch=2, max_T=96, precision="f16x2", **model_kwargs(dims))
    assert m.precision == "bf16x3"
    z = W.synth_codes(2, 4, 96, seed=9)
    mask = O.codebook_mask(O.periodic_mask(z, 7, 1), 3)
    os.environ["VN_F16X2_PROBE"] = "0"
    try:
        mh = VampNetModel(eng, sd, cb, max_batch=2, max_T=96, precision="f16x2", **model_kwargs(dims))
    finally:
        del os.environ["VN_F16X2_PROBE"]
    for rng in ("torch", "device"):
        mh.set_precision("bf16x3")
        torch.manual_seed(77)
        want = mh.generate(return_signal=False, start_tokens=z, mask=mask, _sampling_steps=4, rng=rng).cpu()
        end_state = torch.get_rng_state()
        os.environ["VN_F16X2_PROBE"] = "0"
        try:
            mh.set_precision("f16x2")
        finally:
            del os.environ["VN_F16X2_PROBE"]
        assert mh.precision == "f16x2"
        torch.manual_seed(77)
        with pytest.warns(PrecisionFallbackWarning):
            got = mh.generate(return_signal=False, start_tokens=z, mask=mask, _sampling_steps=4, rng=rng).cpu()
        assert torch.equal(got, want), rng
        assert torch.equal(torch.get_rng_state(), end_state), rng


def test_tiny_heavy_tail_interface_vamp(eng):
    """the whole Interface.vamp() on saturating tiny models asked for in f16x2: both stages end on bf16x3, tokens == the oracle"""
    from vampnet_amd.engine import PrecisionFallbackWarning
    from vampnet_amd.interface import Interface
    cb, csd, fsd = _tiny("saturating")
    z = W.synth_codes(1, 14, 200, seed=6)
    with pytest.warns(PrecisionFallbackWarning):
        itf = Interface.from_state_dicts(SynthCodec(cb), csd, model_kwargs(W.TINY_COARSE_DIMS), fsd, model_kwargs(W.TINY_C2F_DIMS),
                                         device="cuda:0", max_batch=2, precision="f16x2")
    assert itf.effective_precision == {"coarse": "bf16x3", "c2f": "bf16x3"}
    torch.manual_seed(3)
    mask = itf.build_mask(z)
    got = itf.vamp(z, mask, batch_size=2, seed=1, _sampling_steps=4).cpu()
    ref = O.vamp(O.OracleModels(csd, W.TINY_COARSE_DIMS, fsd, W.TINY_C2F_DIMS, cb), z, mask, batch_size=2, seed=1, _sampling_steps=4)
    assert torch.equal(got, ref)


# ---------------------------------------------------------------------------------------- full size
_ORACLE = {}


@pytest.mark.parametrize("kind", ["in_range", "saturating"])
@pytest.mark.parametrize("precision", ["bf16x3", "f16x2"])
def test_full_size_heavy_tail_vamp_vs_oracle(eng, kind, precision):
    """Interface.vamp() (12 coarse steps + coarse-to-fine) on the FULL-SIZE coarse (333 M) and c2f (275 M) architectures with
    trained-like weights, batch 1, seeded: all 14 x 575 tokens equal the oracle's.  bf16x3 must pass as it is; f16x2 must pass
    while staying f16x2 ("in_range") or notice and end on bf16x3 ("saturating") — never differ silently."""
    from vampnet_amd.engine import PrecisionFallbackWarning
    from vampnet_amd.interface import Interface
    cb = W.synth_codebooks()
    csd, fsd = W.synth_state_dict(W.COARSE_DIMS, 0, heavy_tail=kind), W.synth_state_dict(W.C2F_DIMS, 1, heavy_tail=kind)
    z = W.synth_codes(1, 14, 575, seed=2)
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        itf = Interface.from_state_dicts(SynthCodec(cb), csd, model_kwargs(W.COARSE_DIMS), fsd, model_kwargs(W.C2F_DIMS),
                                         max_batch=1, precision=precision)
        torch.manual_seed(0)
        mask = itf.build_mask(z)
        got = itf.vamp(z, mask, batch_size=1, _sampling_steps=12, seed=0).cpu()
    fell = [w for w in wlist if issubclass(w.category, PrecisionFallbackWarning)]
    if kind not in _ORACLE:
        _ORACLE[kind] = O.vamp(O.OracleModels(csd, W.COARSE_DIMS, fsd, W.C2F_DIMS, cb), z, mask, batch_size=1, _sampling_steps=12, seed=0)
    ref = _ORACLE[kind]
    same = (got == ref).float().mean().item()
    print(f"full-size heavy-tail vamp [{kind}, {precision}]: token agreement {same:.6f}; effective {itf.effective_precision}; "
          f"{len(fell)} fallback warning(s)")
    if precision == "f16x2" and kind == "saturating":
        assert fell and itf.effective_precision == {"coarse": "bf16x3", "c2f": "bf16x3"}
    elif precision == "f16x2":
        assert not fell and itf.effective_precision == {"coarse": "f16x2", "c2f": "f16x2"}
    else:
        assert not fell
    assert torch.equal(got, ref), f"agreement {same:.6f}"
