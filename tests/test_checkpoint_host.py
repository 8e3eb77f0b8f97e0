"""CPU tests of vampnet_amd/checkpoint.py — the reader behind `VampNet.load` / `DAC.load` (interface.py:27-50,70; audiotools
`BaseModel.load`: torch.package archive first, `{"state_dict", "metadata": {"kwargs"}}` dict second) — of the trust rule for
files that execute code, and of the codec kwargs / state_dict validation."""
import os
import pathlib

import pytest
import torch

from oracle import weights as W
from vampnet_amd import checkpoint as CK
from vampnet_amd.synth import model_kwargs


def _sd():
    return W.synth_state_dict(W.TINY_COARSE_DIMS, 0)


def test_dict_checkpoint_with_kwargs_variants(tmp_path):
    sd = _sd()
    kw = dict(model_kwargs(W.TINY_COARSE_DIMS), dropout=0.1, r_cond_dim=0, noise_mode="mask", max_seq_len=1024,
              num_reg_tokens=0, flash_attn=False)                      # VampNet.__init__ carries more keys than the engine needs
    torch.save({"state_dict": sd, "metadata": {"kwargs": kw, "version": "0.0.1"}}, tmp_path / "coarse.pth")
    got, gkw = CK.load_model_checkpoint(tmp_path / "coarse.pth")
    assert gkw == kw and list(got) == list(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    torch.save({"state_dict": sd}, tmp_path / "bare.pth")             # no metadata at all: constructor defaults apply
    assert CK.load_model_checkpoint(tmp_path / "bare.pth")[1] == {}
    torch.save({"weights": sd}, tmp_path / "other.pth")
    with pytest.raises(ValueError, match="state_dict"):
        CK.load_model_checkpoint(tmp_path / "other.pth")
    with pytest.raises(FileNotFoundError):
        CK.load_model_checkpoint(tmp_path / "missing.pth")


def test_untrusted_pickles_are_refused(tmp_path, monkeypatch):
    """weights_only=True reads tensors + primitives; a checkpoint that needs arbitrary unpickling is refused unless trusted."""
    sd = _sd()
    torch.save({"state_dict": sd, "metadata": {"kwargs": {"n_heads": 4}, "path": pathlib.Path("/x")}}, tmp_path / "obj.pth")
    monkeypatch.delenv("VN_TRUST_CHECKPOINTS", raising=False)
    with pytest.raises(PermissionError, match="weights_only"):
        CK.load_model_checkpoint(tmp_path / "obj.pth")
    assert CK.load_model_checkpoint(tmp_path / "obj.pth", trusted=True)[1] == {"n_heads": 4}
    monkeypatch.setenv("VN_TRUST_CHECKPOINTS", "1")
    assert CK.load_model_checkpoint(tmp_path / "obj.pth")[1] == {"n_heads": 4}
    torch.save({"a.lora_A": torch.ones(2, 3)}, tmp_path / "lora.pth")
    assert torch.equal(CK.load_tensor_dict(tmp_path / "lora.pth")["a.lora_A"], torch.ones(2, 3))


@pytest.mark.filterwarnings("ignore::UserWarning")
def test_torch_package_branch(tmp_path, monkeypatch):
    """`BaseModel.load` tries `_load_package` first: <Class>/<Class>.pth (+ .metadata) inside a torch.package archive."""
    from torch import package
    from tests import pkg_model
    sd, kw = _sd(), model_kwargs(W.TINY_COARSE_DIMS)
    model = pkg_model.VampNet(sd, **kw)
    with package.PackageExporter(str(tmp_path / "coarse.pth")) as pe:
        pe.extern(["torch.**"])
        pe.intern("tests.**")
        pe.save_pickle("VampNet", "VampNet.pth", model)
        pe.save_pickle("VampNet", "VampNet.metadata", {"kwargs": kw})
    with package.PackageExporter(str(tmp_path / "nometa.pth")) as pe:
        pe.extern(["torch.**"])
        pe.intern("tests.**")
        pe.save_pickle("VampNet", "VampNet.pth", model)
    assert CK.is_torch_package(tmp_path / "coarse.pth")
    torch.save({"state_dict": sd}, tmp_path / "dict.pth")
    assert not CK.is_torch_package(tmp_path / "dict.pth")            # torch.save also writes zip files
    monkeypatch.delenv("VN_TRUST_CHECKPOINTS", raising=False)
    with pytest.raises(PermissionError, match="executes the code"):
        CK.load_model_checkpoint(tmp_path / "coarse.pth")
    got, gkw = CK.load_model_checkpoint(tmp_path / "coarse.pth", trusted=True)
    assert gkw == kw and list(got) == list(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    # no metadata inside: constructor kwargs are read off the module's attributes
    from vampnet_amd.interface import _MODEL_KEYS
    _, akw = CK.load_model_checkpoint(tmp_path / "nometa.pth", kwarg_keys=_MODEL_KEYS, trusted=True)
    assert akw == {k: kw[k] for k in _MODEL_KEYS}
    with pytest.raises(ValueError, match="no .* model inside"):
        CK.load_model_checkpoint(_wrong_package(tmp_path), trusted=True)


def _wrong_package(tmp_path):
    from torch import package
    with package.PackageExporter(str(tmp_path / "wrong.pth")) as pe:
        pe.extern(["torch.**"])
        pe.save_pickle("Something", "Something.pth", {"x": 1})
    return tmp_path / "wrong.pth"


def test_codec_kwargs_and_state_dict_validation():
    """A published DAC checkpoint's kwargs (descript-audio-codec 44 kHz: rates 2·4·8·8, 9 codebooks, quantizer_dropout, ...) is
    accepted and reduced to what the conv stacks need; a state_dict of another architecture fails with a message."""
    from oracle import dac_oracle as D
    from vampnet_amd.codec import DEFAULT_CFG, normalize_codec_kwargs, validate_codec_state_dict
    dac44 = dict(encoder_dim=64, encoder_rates=[2, 4, 8, 8], latent_dim=None, decoder_dim=1536, decoder_rates=[8, 8, 4, 2],
                 n_codebooks=9, codebook_size=1024, codebook_dim=8, quantizer_dropout=0.5, sample_rate=44100)
    cfg = normalize_codec_kwargs(dac44)
    assert "quantizer_dropout" not in cfg and cfg["encoder_rates"] == [2, 4, 8, 8] and cfg["n_codebooks"] == 9
    assert normalize_codec_kwargs(dict(dac44, codebook_dim=[8] * 9))["codebook_dim"] == 8
    with pytest.raises(ValueError, match="per-level"):
        normalize_codec_kwargs(dict(dac44, codebook_dim=[8, 16]))
    tiny = dict(D.DAC_TINY_CFG)
    sd = D.synth_dac_state_dict(tiny, 1)
    validate_codec_state_dict(sd, dict(DEFAULT_CFG, **tiny))
    with pytest.raises(ValueError, match="n_codebooks"):
        validate_codec_state_dict(sd, dict(DEFAULT_CFG, **dict(tiny, n_codebooks=tiny["n_codebooks"] + 1)))
    with pytest.raises(ValueError, match="encoder_dim"):
        validate_codec_state_dict(sd, dict(DEFAULT_CFG, **dict(tiny, encoder_dim=2 * tiny["encoder_dim"])))
    with pytest.raises(ValueError, match="not a DAC-family"):
        validate_codec_state_dict(_sd(), dict(DEFAULT_CFG, **tiny))


@pytest.mark.filterwarnings("ignore::UserWarning")
def test_parity_script_on_synthetic_checkpoints(tmp_path):
    """scripts/parity_real_ckpt.py (BASELINE configs[0] against the reference on real checkpoints) end to end on stand-ins: a
    torch.package coarse checkpoint + a LoRA file, a dict c2f checkpoint, a codec dict checkpoint, tokens from an npy file; the
    engine side replaced by the oracle (dry run).  Also pins merge_lora_state_dict against the oracle's own forward."""
    import importlib.util
    import numpy as np
    from torch import package
    from oracle import dac_oracle as D, vampnet_oracle as O
    from tests import pkg_model
    from vampnet_amd.checkpoint import merge_lora_state_dict
    csd, fsd = W.synth_state_dict(W.TINY_COARSE_DIMS, 0), W.synth_state_dict(W.TINY_C2F_DIMS, 1)
    model = pkg_model.VampNet(csd, **model_kwargs(W.TINY_COARSE_DIMS))
    with package.PackageExporter(str(tmp_path / "coarse.pth")) as pe:
        pe.extern(["torch.**"])
        pe.intern("tests.**")
        pe.save_pickle("VampNet", "VampNet.pth", model)
        pe.save_pickle("VampNet", "VampNet.metadata", {"kwargs": model_kwargs(W.TINY_COARSE_DIMS)})
    torch.save({"state_dict": fsd, "metadata": {"kwargs": model_kwargs(W.TINY_C2F_DIMS)}}, tmp_path / "c2f.pth")
    g = torch.Generator().manual_seed(3)
    key = "transformer.layers.1.feed_forward.w_2"
    lora = {key + ".lora_A": torch.randn(8, csd[key + ".weight"].shape[1], generator=g) * 0.05,
            key + ".lora_B": torch.randn(csd[key + ".weight"].shape[0], 8, generator=g) * 0.05}
    torch.save(lora, tmp_path / "lora.pth")
    merged = merge_lora_state_dict({**csd, **lora})
    assert not any("lora_" in k for k in merged)
    assert torch.equal(merged[key + ".weight"], csd[key + ".weight"] + (lora[key + ".lora_B"] @ lora[key + ".lora_A"]) / 8.0)
    assert all(torch.equal(merged[k], v) for k, v in csd.items() if k != key + ".weight")
    cfg = dict(D.DAC_TINY_CFG, n_codebooks=14)
    torch.save({"state_dict": D.synth_dac_state_dict(cfg, 1), "metadata": {"kwargs": cfg}}, tmp_path / "codec.pth")
    np.save(tmp_path / "z.npy", W.synth_codes(1, 14, 90, seed=4).numpy())
    spec = importlib.util.spec_from_file_location("parity_real_ckpt", os.path.join(os.path.dirname(__file__), "..", "scripts", "parity_real_ckpt.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv = ["--coarse", str(tmp_path / "coarse.pth"), "--coarse-lora", str(tmp_path / "lora.pth"), "--c2f", str(tmp_path / "c2f.pth"),
            "--codec", str(tmp_path / "codec.pth"), "--tokens", str(tmp_path / "z.npy"), "--engine", "oracle", "--steps", "3", "--seeds", "0", "1"]
    with pytest.raises(PermissionError):
        mod.main(argv)                                     # the torch.package checkpoint needs --trusted
    assert mod.main(argv + ["--trusted"]) == 0


def test_lora_merge_is_loralibs_eval_formula():
    """merge_lora_state_dict against the public loralib.Linear arithmetic (loralib/layers.py, fan_in_fan_out=False): in train mode
    the layer computes  x W^T + (x A^T B^T) * (lora_alpha / r);  `train(False)` folds  W += (B @ A) * scaling  once.  The merged
    weight must reproduce the train-mode output, with r read from the adapters (not a constant): ranks 8 (the reference's LORA_R)
    and 4."""
    from vampnet_amd.checkpoint import lora_scaling, merge_lora_state_dict
    g = torch.Generator().manual_seed(0)
    for r in (8, 4):
        W = torch.randn(48, 32, generator=g)
        A, B = torch.randn(r, 32, generator=g), torch.randn(48, r, generator=g)
        sd = {"lin.weight": W, "lin.lora_A": A, "lin.lora_B": B, "other.weight": torch.randn(5, 5, generator=g)}
        assert lora_scaling(sd) == 1.0 / r
        merged = merge_lora_state_dict(sd)
        assert set(merged) == {"lin.weight", "other.weight"} and torch.equal(merged["other.weight"], sd["other.weight"])
        x = torch.randn(7, 32, generator=g)
        train_mode = torch.nn.functional.linear(x, W) + (x @ A.t() @ B.t()) * (1.0 / r)        # loralib Linear.forward, not merged
        assert torch.allclose(torch.nn.functional.linear(x, merged["lin.weight"]), train_mode, atol=1e-5, rtol=1e-5)
        assert torch.equal(merged["lin.weight"], W + (B @ A) * (1.0 / r))                       # loralib Linear.train(False)
    with pytest.raises(ValueError):
        merge_lora_state_dict({"a.weight": torch.zeros(4, 4), "a.lora_A": torch.zeros(2, 4), "a.lora_B": torch.zeros(4, 2),
                               "b.weight": torch.zeros(4, 4), "b.lora_A": torch.zeros(3, 4), "b.lora_B": torch.zeros(4, 3)})
    with pytest.raises(ValueError):
        merge_lora_state_dict({"a.weight": torch.zeros(4, 4), "a.lora_A": torch.zeros(2, 5), "a.lora_B": torch.zeros(4, 2)})


def test_vampnet_state_dict_validation():
    """validate_vampnet_state_dict: a synthetic state_dict of the real layout passes against its kwargs and fails with a message
    naming the tensor when the kwargs belong to another architecture, a tensor is missing, or adapters are inconsistent."""
    from vampnet_amd import synth as W
    from vampnet_amd.checkpoint import validate_vampnet_state_dict
    from vampnet_amd.synth import model_kwargs
    dims = W.TINY_COARSE_DIMS
    sd, kw = W.synth_state_dict(dims, 0), model_kwargs(dims)
    rep = validate_vampnet_state_dict(sd, kw)
    assert rep["n_lora_pairs"] == 0 and rep["lora_rank"] is None and rep["kwargs"]["n_layers"] == dims["n_layers"]
    with pytest.raises(ValueError, match="n_layers|layers"):
        validate_vampnet_state_dict(sd, dict(kw, n_layers=dims["n_layers"] - 1))
    with pytest.raises(ValueError, match="embedding.special.MASK"):
        validate_vampnet_state_dict(sd, dict(kw, n_codebooks=kw["n_codebooks"] + 1))
    broken = dict(sd)
    del broken["transformer.layers.0.feed_forward.w_2.weight"]
    with pytest.raises(ValueError, match="w_2"):
        validate_vampnet_state_dict(broken, kw)
    D = dims["d_model"]
    lora = dict(sd)
    lora["transformer.layers.0.self_attn.w_qs.lora_A"] = torch.zeros(8, D)
    with pytest.raises(ValueError, match="only one of"):
        validate_vampnet_state_dict(lora, kw)
    lora["transformer.layers.0.self_attn.w_qs.lora_B"] = torch.zeros(D, 8)
    assert validate_vampnet_state_dict(lora, kw)["lora_rank"] == 8
    lora["transformer.layers.0.self_attn.w_ks.lora_A"] = torch.zeros(8, D)
    lora["transformer.layers.0.self_attn.w_ks.lora_B"] = torch.zeros(D, 8)
    with pytest.raises(ValueError, match="only adapts"):
        validate_vampnet_state_dict(lora, kw)
