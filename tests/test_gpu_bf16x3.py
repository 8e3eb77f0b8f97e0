"""-m gpu: the split-plane precisions of gemm_x3.hip — "bf16x3" (fp32-grade GEMMs as six bf16-MFMA products of exact three-way
operand splits) and "f16x2" (three fp16-MFMA products of two-plane splits, second accumulator) — are held to the SAME parity bars as
the exact-fp32 MFMA path: the test bodies are the ones of tests/test_gpu_model.py, run on models switched to that precision —
logits vs the CPU oracle and the reference's frozen probes at the fp32 tolerances, tokens bit-identical to the oracle and to the
reference's golden tokens."""
import pytest
import torch

from oracle import vampnet_oracle as O, weights as W
from tests import test_gpu_model as TM
from tests.gpu_common import SynthCodec, model_kwargs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from vampnet_amd.engine import Engine
    return Engine("cuda:0")


SPLIT_PRECISIONS = ["bf16x3", "f16x2"]


@pytest.fixture(scope="module", params=SPLIT_PRECISIONS)
def prec(request):
    return request.param


@pytest.fixture(scope="module")
def tiny3(eng, prec):
    from vampnet_amd.engine import VampNetModel
    cb = W.synth_codebooks()
    csd, fsd = W.synth_state_dict(W.TINY_COARSE_DIMS, 0), W.synth_state_dict(W.TINY_C2F_DIMS, 1)
    kw = dict(precision=prec)
    coarse = VampNetModel(eng, csd, cb, max_batch=4, max_T=575, **kw, **model_kwargs(W.TINY_COARSE_DIMS))
    c2f = VampNetModel(eng, fsd, cb, max_batch=4, max_T=173, **kw, **model_kwargs(W.TINY_C2F_DIMS))
    return dict(cb=cb, csd=csd, fsd=fsd, coarse=coarse, c2f=c2f,
                models=O.OracleModels(csd, W.TINY_COARSE_DIMS, fsd, W.TINY_C2F_DIMS, cb))


@pytest.fixture(scope="module")
def itf3(tiny3, prec):
    from vampnet_amd.interface import Interface
    return Interface.from_state_dicts(SynthCodec(tiny3["cb"]), tiny3["csd"], model_kwargs(W.TINY_COARSE_DIMS),
                                      tiny3["fsd"], model_kwargs(W.TINY_C2F_DIMS), device="cuda:0", max_batch=4,
                                      precision=prec)


@pytest.mark.parametrize("which,B,T", [("coarse", 2, 50), ("coarse", 1, 575), ("c2f", 3, 37), ("c2f", 1, 173), ("coarse", 1, 1)])
def test_forward_tiny_vs_oracle(tiny3, which, B, T):
    TM.test_forward_tiny_vs_oracle(tiny3, which, B, T)


def test_forward_tiny_vs_golden(tiny3):
    TM.test_forward_tiny_vs_golden(tiny3)


@pytest.mark.parametrize("which", ["coarse", "c2f"])
@pytest.mark.parametrize("case", TM.GEN_CASES)
def test_generate_vs_oracle(tiny3, which, case):
    TM.test_generate_vs_oracle(tiny3, which, case)


def test_generate_and_vamp_vs_golden(tiny3, itf3):
    TM.test_generate_vs_golden(tiny3)
    TM.test_interface_vamp_vs_golden(itf3)


@pytest.mark.parametrize("name,dims,T,seed", [("coarse", W.COARSE_DIMS, 575, 0), ("c2f", W.C2F_DIMS, 173, 1)])
def test_forward_full_size_vs_reference_probe(eng, prec, name, dims, T, seed, monkeypatch):
    """Full-size models against the REFERENCE's frozen logits, same bars as the exact-fp32 path; then the two precisions
    against each other on the same weights."""
    from vampnet_amd import engine as E
    made = []
    orig = E.VampNetModel

    def make(*a, **k):
        m = orig(*a, **k)
        m.set_precision(prec)
        made.append(m)
        return m

    monkeypatch.setattr(E, "VampNetModel", make)
    TM.test_forward_full_size_vs_reference_probe(eng, name, dims, T, seed)
    model = made[0]
    codes = W.synth_codes(1, dims["n_codebooks"], T, seed=11)
    codes[:, dims["n_cond"]:, 1::2] = 1024
    a = model.forward_codes(codes, layout="native")
    model.set_precision("f32")
    b = model.forward_codes(codes, layout="native")
    d = (a - b).abs().max().item()
    print(f"{name}: max |logit({prec}) - logit(f32 mfma)| = {d:.3e}")
    assert d <= TM.LOGIT_ATOL_FULL


@pytest.mark.parametrize("dims,T,B", [(W.TINY_COARSE_DIMS, 200, 3), (W.TINY_C2F_DIMS, 173, 2)])
def test_fused_splitk_reduce_rmsnorm_is_bitwise_the_two_kernel_form(eng, prec, dims, T, B):
    """A RESIDUAL GEMM that is split along K runs the next RMSNorm inside its reduce pass (vn_splitk_reduce_rmsnorm_kernel): the
    logits must not move by a bit against the reduce kernel followed by the norm kernel (same summation order, same row math).
    Split-K forced to 2 so that the small shapes take the path the B = 8 model shapes take by the cost model."""
    from vampnet_amd.engine import VampNetModel
    cb = W.synth_codebooks()
    m = VampNetModel(eng, W.synth_state_dict(dims, 3), cb, max_batch=4, max_T=256, precision=prec, **model_kwargs(dims))
    codes = W.synth_codes(B, dims["n_codebooks"], T, seed=5)
    codes[:, dims["n_cond"]:, ::3] = 1024
    lib = eng.lib
    try:
        lib.vn_debug_x3_config(eng.handle, 0, 2, -1)
        lib.vn_debug_x3_fuse_norm(eng.handle, 1)
        a = m.forward_codes(codes, layout="native").clone()
        a2 = m.forward_codes(codes, layout="native").clone()
        lib.vn_debug_x3_fuse_norm(eng.handle, 0)
        b = m.forward_codes(codes, layout="native").clone()
        b2 = m.forward_codes(codes, layout="native").clone()
        lib.vn_debug_x3_config(eng.handle, 0, 0, -1)            # no split at all: the residual epilogue + the stand-alone norm
        c = m.forward_codes(codes, layout="native").clone()
    finally:
        lib.vn_debug_x3_config(eng.handle, 0, -1, -1)
        lib.vn_debug_x3_fuse_norm(eng.handle, -1)
    print(f"fused vs two-kernel logits: max |d| = {(a - b).abs().max().item():.3e}; run to run: fused {(a - a2).abs().max().item():.3e}, "
          f"two-kernel {(b - b2).abs().max().item():.3e}")
    assert torch.equal(a, a2) and torch.equal(b, b2)
    assert torch.equal(a, b)
    assert (a - c).abs().max().item() <= 2e-5       # split vs unsplit k-order: fp32 re-association only
    sd = W.synth_state_dict(dims, 3)
    ref = O.forward(sd, dims, O.from_codes(sd, cb, codes))
    lib.vn_debug_x3_config(eng.handle, 0, 2, -1)
    try:
        got = m.forward_codes(codes).cpu()          # reference layout, fused form
    finally:
        lib.vn_debug_x3_config(eng.handle, 0, -1, -1)
    assert (got - ref).abs().max().item() <= TM.LOGIT_ATOL_TINY


@pytest.mark.parametrize("dims,B,T", [(W.TINY_COARSE_DIMS, 3, 200), (W.TINY_C2F_DIMS, 2, 173), (W.TINY_COARSE_DIMS, 1, 33)])
def test_split_plane_attention_path_at_every_tile_height(eng, prec, dims, B, T):
    """The QKV GEMM with the plane epilogue (q x 1/8 and k planes head-major, V^T transposed through the LDS image into the blocked
    layout) + attention_x3.hip, forced on for a small model, at the three tile heights of gemm_x3.hip (192 rows: 2 x 4 wave
    grid, 64-row epilogue images): logits bitwise equal across the heights, equal to the oracle at the tiny-model tolerance, and
    within fp32 noise of the fp32-attention path."""
    from vampnet_amd.engine import VampNetModel
    cb = W.synth_codebooks()
    sd = W.synth_state_dict(dims, 4)
    m = VampNetModel(eng, sd, cb, max_batch=4, max_T=256, precision=prec, **model_kwargs(dims))
    codes = W.synth_codes(B, dims["n_codebooks"], T, seed=6)
    codes[:, dims["n_cond"]:, ::2] = 1024
    lib = eng.lib
    outs = {}
    heights = (128, 192, 256) if prec == "f16x2" else (128, 192, 256, 96)      # 96 rows: the k-split tile, bf16x3 operands only
    try:
        lib.vn_debug_attention_x3_force(eng.handle, 1)
        for bm in heights:
            lib.vn_debug_x3_config(eng.handle, bm, -1, -1)
            outs[bm] = m.forward_codes(codes).clone()
        lib.vn_debug_attention_x3_force(eng.handle, 0)              # fp32-attention path: the QKV GEMM's fp32 head-major scatter epilogue
        plains = {}
        for bm in heights:
            lib.vn_debug_x3_config(eng.handle, bm, -1, -1)
            plains[bm] = m.forward_codes(codes).clone()
        lib.vn_debug_x3_config(eng.handle, 0, -1, -1)
        plain = m.forward_codes(codes).clone()
        assert torch.equal(plains[128], plains[256])
        # 192 rows: bf16x3 runs W1 + GEGLU on it through the tile-image epilogue (value and gate columns of a 96 x 32 wave tile live in
        # different waves) — the same sums, but gelu's expression is contracted differently there: fp32 noise, not bits; f16x2 keeps
        # 128 rows for the GEGLU launch and stays bitwise
        if prec == "f16x2":
            assert torch.equal(plains[128], plains[192])
        assert (plains[192] - plains[128]).abs().max().item() <= TM.LOGIT_ATOL_TINY
        # the by-shape choice mixes heights per GEMM (the 96-row tile adds its two k-halves in its epilogue): fp32 noise from any one
        assert (plain - plains[128]).abs().max().item() <= TM.LOGIT_ATOL_TINY
    finally:
        lib.vn_debug_attention_x3_force(eng.handle, -1)
        lib.vn_debug_x3_config(eng.handle, 0, -1, -1)
    assert torch.equal(outs[128], outs[256])
    if prec == "f16x2":
        assert torch.equal(outs[128], outs[192])
    assert (outs[192] - outs[128]).abs().max().item() <= TM.LOGIT_ATOL_TINY
    if 96 in outs:
        d = (outs[96] - outs[128]).abs().max().item()
        print(f"96-row k-split tile vs 128 rows: logits max |d| = {d:.3e}")
        assert d <= TM.LOGIT_ATOL_TINY and (plains[96] - plains[128]).abs().max().item() <= TM.LOGIT_ATOL_TINY
    ref = O.forward(sd, dims, O.from_codes(sd, cb, codes))
    assert (outs[192].cpu() - ref).abs().max().item() <= TM.LOGIT_ATOL_TINY
    assert (outs[192] - plain).abs().max().item() <= TM.LOGIT_ATOL_TINY


def test_f16x2_falls_back_on_weights_beyond_fp16_range(eng):
    """fp16 planes cannot hold |w| >= 65504: a model with such a weight asked for in f16x2 says so (PrecisionFallbackWarning, from the
    saturation ledger's weight word) and runs on bf16x3 — same logits as a model built on bf16x3 directly"""
    from vampnet_amd.engine import PrecisionFallbackWarning, VampNetModel
    cb = W.synth_codebooks()
    sd = {k: v.clone() for k, v in W.synth_state_dict(W.TINY_COARSE_DIMS, 0).items()}
    key = next(k for k, v in sd.items() if k.endswith("w_1.weight"))
    sd[key][0, 0] = 1.0e5
    with pytest.warns(PrecisionFallbackWarning, match="weight"):
        mh = VampNetModel(eng, sd, cb, max_batch=1, max_T=64, precision="f16x2", **model_kwargs(W.TINY_COARSE_DIMS))
    assert mh.precision == "bf16x3"
    m = VampNetModel(eng, sd, cb, max_batch=1, max_T=64, precision="bf16x3", **model_kwargs(W.TINY_COARSE_DIMS))
    codes = W.synth_codes(1, W.TINY_COARSE_DIMS["n_codebooks"], 40, seed=2)
    ref = m.forward_codes(codes)
    assert torch.isfinite(ref).all() and torch.equal(mh.forward_codes(codes), ref)


def test_f16x2_probe_falls_back_on_activations_beyond_fp16_range(eng):
    """weights inside fp16's range whose ACTIVATIONS leave it (a feed-forward scaled by 3e3: GEGLU outputs ~ 1e7): the probe forward
    at precision selection puts them on the saturation ledger and the model moves to bf16x3, which runs it at fp32 grade"""
    from vampnet_amd.engine import PrecisionFallbackWarning, VampNetModel
    cb = W.synth_codebooks()
    sd = {k: v.clone() for k, v in W.synth_state_dict(W.TINY_COARSE_DIMS, 0).items()}
    for k in sd:
        if k.endswith("w_1.weight"):
            sd[k] *= 3.0e3
    assert max(float(v.abs().max()) for v in sd.values()) < 6.0e4
    with pytest.warns(PrecisionFallbackWarning, match="probe forward"):
        m = VampNetModel(eng, sd, cb, max_batch=1, max_T=64, precision="f16x2", **model_kwargs(W.TINY_COARSE_DIMS))
    assert m.precision == "bf16x3"
    codes = W.synth_codes(1, W.TINY_COARSE_DIMS["n_codebooks"], 40, seed=2)
    ref = O.forward(sd, W.TINY_COARSE_DIMS, O.from_codes(sd, cb, codes))
    got = m.forward_codes(codes).cpu()
    assert torch.isfinite(got).all() and (got - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("precision", SPLIT_PRECISIONS)
def test_folded_norm_matches_unfolded_and_oracle(precision):
    """The split-plane models fold every RMSNorm into its consumer GEMM (engine.hip forward_folded: y W^T = r (.) (x (W (.) w)^T), the
    residual GEMMs' epilogues write the planes and the sums of squares of the new rows).  A context created with VN_FOLD_NORM=0 runs
    the norm kernels instead: both forms agree with each other and with the oracle at the fp32 tolerance, on a shape that takes the
    one-round GEMM path and on one that splits the residual GEMMs along K (their reduce pass then writes planes + sums: vn_launch_rowprep),
    and the tile heights of the folded form are bitwise equal to each other."""
    import os
    from vampnet_amd.engine import Engine, VampNetModel
    dims = dict(W.TINY_COARSE_DIMS, n_layers=3)
    cb, sd = W.synth_codebooks(), W.synth_state_dict(dict(W.TINY_COARSE_DIMS, n_layers=3), 4)
    sd["transformer.layers.1.norm_3.weight"] *= 3.0                       # a norm weight far from 1: a forgotten fold would show
    eng_f = Engine("cuda:0")
    os.environ["VN_FOLD_NORM"] = "0"
    try:
        eng_u = Engine("cuda:0")
    finally:
        del os.environ["VN_FOLD_NORM"]
    os.environ["VN_FOLD_X16ONLY"] = "0"
    try:
        eng_x = Engine("cuda:0")          # folded, but the residual stream also kept as an fp32 image (what f16x2 always does)
    finally:
        del os.environ["VN_FOLD_X16ONLY"]
    mf = VampNetModel(eng_f, sd, cb, max_batch=4, max_T=575, precision=precision, **model_kwargs(dims))
    mu = VampNetModel(eng_u, sd, cb, max_batch=4, max_T=575, precision=precision, **model_kwargs(dims))
    mx = VampNetModel(eng_x, sd, cb, max_batch=4, max_T=575, precision=precision, **model_kwargs(dims))
    for B, T in ((4, 575), (1, 575), (2, 37), (1, 1)):
        codes = W.synth_codes(B, 4, T, seed=12)
        codes[:, :, ::3] = 1024
        ref = TM.to_native(O.forward(sd, dims, O.from_codes(sd, cb, codes)), 4)
        a, b = mf.forward_codes(codes, layout="native").cpu(), mu.forward_codes(codes, layout="native").cpu()
        # bf16x3 planes sum to the fp32 value exactly: keeping the residual stream in the planes alone changes no bit
        assert torch.equal(mx.forward_codes(codes, layout="native").cpu(), a)
        print(f"[{precision}] B={B} T={T}: folded vs oracle {(a - ref).abs().max():.3e}, unfolded vs oracle {(b - ref).abs().max():.3e}, "
              f"folded vs unfolded {(a - b).abs().max():.3e}")
        assert (a - ref).abs().max().item() <= TM.LOGIT_ATOL_TINY and (b - ref).abs().max().item() <= TM.LOGIT_ATOL_TINY
    codes = W.synth_codes(3, 4, 300, seed=13)
    outs = {}
    try:
        for bm in (128, 192, 256):
            eng_f.lib.vn_debug_x3_config(eng_f.handle, bm, -1, -1)
            outs[bm] = mf.forward_codes(codes).clone()
        for sk in (2, 4):                                                  # forced k-splits of the residual GEMMs: the reduce pass writes planes + sums
            eng_f.lib.vn_debug_x3_config(eng_f.handle, 128, sk, -1)
            outs[f"sk{sk}"] = mf.forward_codes(codes).clone()
    finally:
        eng_f.lib.vn_debug_x3_config(eng_f.handle, 0, -1, -1)
    assert torch.equal(outs[128], outs[256])
    if precision == "f16x2":
        assert torch.equal(outs[128], outs[192])
    assert (outs[192] - outs[128]).abs().max().item() <= TM.LOGIT_ATOL_TINY      # bf16x3: W1 + GEGLU of the 192-row tile = the image epilogue
    if precision == "bf16x3":                                             # the 96-row k-split tile: every folded-norm epilogue through its image form
        try:
            eng_f.lib.vn_debug_x3_config(eng_f.handle, 96, -1, -1)
            o96 = mf.forward_codes(codes).clone()
        finally:
            eng_f.lib.vn_debug_x3_config(eng_f.handle, 0, -1, -1)
        d = (o96 - outs[128]).abs().max().item()
        print(f"[{precision}] 96-row tile vs 128 rows: {d:.3e}")
        assert d <= TM.LOGIT_ATOL_TINY
    for sk in (2, 4):
        d = (outs[f"sk{sk}"] - outs[128]).abs().max().item()
        print(f"[{precision}] forced split {sk} vs one round: {d:.3e}")
        assert d <= TM.LOGIT_ATOL_TINY
