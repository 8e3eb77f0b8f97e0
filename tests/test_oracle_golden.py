"""Oracle vs the committed golden fixtures (reference outputs frozen by oracle/make_golden.py).
Runs on CPU anywhere (no /root/reference needed)."""
import ast
import os

import numpy as np
import pytest
import torch

from oracle import vampnet_oracle as O, weights as W

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def rng_fingerprint(seed):
    torch.manual_seed(seed)
    e = torch.empty(4, 1024).exponential_(1)
    u = torch.zeros(2, 100).uniform_(1e-20, 1)
    return np.array([e.double().sum().item(), u.double().sum().item()])


def require_same_rng(fp, seed):
    if not np.array_equal(rng_fingerprint(seed), fp):
        pytest.skip("torch CPU RNG stream differs from the machine that froze the fixtures")


@pytest.fixture(scope="module")
def tiny():
    cb = W.synth_codebooks()
    csd, fsd = W.synth_state_dict(W.TINY_COARSE_DIMS, 0), W.synth_state_dict(W.TINY_C2F_DIMS, 1)
    return dict(cb=cb, csd=csd, fsd=fsd,
                models=O.OracleModels(csd, W.TINY_COARSE_DIMS, fsd, W.TINY_C2F_DIMS, cb))


def test_forward_tiny(tiny):
    g = load("forward_tiny.npz")
    for name, sd, dims in (("coarse", tiny["csd"], W.TINY_COARSE_DIMS), ("c2f", tiny["fsd"], W.TINY_C2F_DIMS)):
        codes = torch.from_numpy(g[f"{name}_codes"].astype(np.int64))
        logits, acts = O.forward(sd, dims, O.from_codes(sd, tiny["cb"], codes), return_activations=True)
        # same torch build + same ops -> bitwise here; allow fp32 re-association across CPUs
        np.testing.assert_allclose(logits.numpy(), g[f"{name}_logits"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(acts[-1].numpy(), g[f"{name}_act_last"], rtol=0, atol=2e-5)


def test_generate_tiny(tiny):
    g = load("generate_tiny.npz")
    for idx, m in enumerate(g["meta"]):
        which, B, T, kw = ast.literal_eval(str(m))
        require_same_rng(g[f"case{idx}_rngfp"], kw["seed"])
        sd, dims = (tiny["csd"], W.TINY_COARSE_DIMS) if which == "coarse" else (tiny["fsd"], W.TINY_C2F_DIMS)
        z = torch.from_numpy(g[f"case{idx}_z"].astype(np.int64))
        mask = torch.from_numpy(g[f"case{idx}_mask"].astype(np.int64))
        out = O.generate(sd, dims, tiny["cb"], z, mask, **O._gen_kwargs(kw))
        assert np.array_equal(out.numpy(), g[f"case{idx}_out"].astype(np.int64)), m


def test_build_mask():
    g = load("build_mask.npz")
    torch.manual_seed(0)
    fp = np.array([torch.bernoulli(torch.full((64,), 0.5)).sum().item(), torch.randint(0, 7, (1,)).item()])
    if not np.array_equal(fp, g["rngfp_bernoulli"]):
        pytest.skip("torch CPU RNG stream differs")
    z = W.synth_codes(2, 14, 120, seed=4)
    for i, m in enumerate(g["meta"]):
        kw = ast.literal_eval(str(m))
        for seed in (0, 1):
            torch.manual_seed(seed)
            assert np.array_equal(O.build_mask(z, **kw).numpy(), g[f"kw{i}_seed{seed}"].astype(np.int64))


def test_vamp_tiny(tiny):
    g = load("vamp_tiny.npz")
    z = torch.from_numpy(g["z"].astype(np.int64))
    mask = torch.from_numpy(g["mask"].astype(np.int64))
    for i, m in enumerate(g["meta"]):
        B, kw = ast.literal_eval(str(m))
        require_same_rng(g[f"case{i}_rngfp"], kw["seed"])
        out, mz = O.vamp(tiny["models"], z, mask, batch_size=B, return_mask=True, **kw)
        assert np.array_equal(out.numpy(), g[f"case{i}_out"].astype(np.int64))
        assert np.array_equal(mz.numpy(), g[f"case{i}_maskz"].astype(np.int64))


def test_misc_tables():
    g = load("misc.npz")
    rel = torch.from_numpy(g["bucket_rel"])
    assert np.array_equal(O.relative_position_bucket(rel).numpy(), g["bucket"].astype(np.int64))
    np.testing.assert_array_equal(O.gamma(torch.from_numpy(g["gamma_r"])).numpy(), g["gamma"])
    assert O.s2t(10) == 575 and O.s2t(3) == 173
    assert O.forward_flops(W.COARSE_DIMS, 575) == pytest.approx(416.76e9, rel=1e-4)
    assert O.forward_flops(W.C2F_DIMS, 173) == pytest.approx(97.74e9, rel=1e-4)


@pytest.mark.parametrize("name,dims,T,seed", [("c2f", W.C2F_DIMS, 173, 1)])
def test_forward_full_size_probe(name, dims, T, seed):
    """Real-size model (c2f: 275 M params, ~0.3 s on 8 cores) vs the reference's frozen logits.
    Tolerance = fp32 re-association class (1.9e-6 between thread counts)."""
    g = load("forward_full.npz")
    cb = W.synth_codebooks()
    sd = W.synth_state_dict(dims, seed)
    codes = W.synth_codes(1, dims["n_codebooks"], T, seed=11)
    codes[:, dims["n_cond"]:, 1::2] = 1024
    with torch.inference_mode():
        lg = O.forward(sd, dims, O.from_codes(sd, cb, codes))[0].T
    np.testing.assert_allclose(lg[g[f"{name}_rows"]].numpy(), g[f"{name}_logits_rows"], rtol=0, atol=5e-5)
    am = lg.argmax(-1).numpy()
    bad = np.nonzero(am != g[f"{name}_argmax"])[0]
    assert all(g[f"{name}_gap"][b] < 2e-5 for b in bad), "argmax flip outside the near-tie band"


@pytest.mark.parametrize("name", ["coarse", "c2f"])
def test_train_step_golden(name):
    """oracle/train_oracle.py against one training step of the REFERENCE's own modules (train.py:237-304) frozen by
    oracle/make_golden.py: loss, global gradient norm, per-parameter gradient statistics, four full gradient tensors
    and the AdamW-updated w_2 of layer 1.  Bitwise on the machine that froze them; 2e-5 relative elsewhere (BLAS)."""
    from oracle import train_oracle as TO
    g = load("train_tiny.npz")
    dims = W.TINY_COARSE_DIMS if name == "coarse" else W.TINY_C2F_DIMS
    sd, cb = W.synth_state_dict(dims, 0 if name == "coarse" else 1), W.synth_codebooks()
    B, T, p = 2, 24, 0.1
    z = W.synth_codes(B, dims["n_codebooks"], T, seed=5)
    mask = TO.make_training_mask(z, torch.tensor([0.3, 0.8]), dims["n_cond"], generator=torch.Generator().manual_seed(3))
    masks = TO.draw_dropout_masks(dims, B, T, p, torch.Generator().manual_seed(4))
    loss, grads, _ = TO.loss_and_grads(sd, dims, cb, z, mask, masks, p)
    assert abs(loss.item() - float(g[f"{name}_loss"])) <= 2e-6 * abs(float(g[f"{name}_loss"]))
    names = [str(n) for n in g[f"{name}_names"]]
    assert set(names) == set(grads)
    for i, k in enumerate(names):
        am = float(g[f"{name}_grad_absmax"][i])
        assert abs(grads[k].abs().max().item() - am) <= 2e-5 * max(am, 1e-12), k
    for key in g.files:
        if key.startswith(f"{name}_grad::"):
            k = key.split("::", 1)[1]
            ref = torch.from_numpy(g[key])
            assert (grads[k] - ref).abs().max().item() <= 2e-5 * ref.abs().max().item(), k
    lr = TO.noam_lr(1, dims["d_model"])
    assert lr == float(g[f"{name}_lr"])
    new, norm = TO.clip_and_adamw(sd, grads, {}, lr)
    assert abs(norm.item() - float(g[f"{name}_grad_norm"])) <= 1e-4 * float(g[f"{name}_grad_norm"])
    w2 = torch.from_numpy(g[f"{name}_w2_after"])
    assert (new["transformer.layers.1.feed_forward.w_2.weight"] - w2).abs().max().item() <= 1e-9 + 2.1 * lr
