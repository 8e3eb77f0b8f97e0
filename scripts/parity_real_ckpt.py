#!/usr/bin/env python
"""BASELINE configs[0] on REAL checkpoints, one command (needs the files the reference downloads from the HF hub,
vampnet/__init__.py:20-47 — absent from this image, so this script has only ever run on synthetic stand-ins, see
tests/test_checkpoint_host.py::test_parity_script_on_synthetic_checkpoints):

    python scripts/parity_real_ckpt.py --coarse coarse.pth --c2f c2f.pth --codec codec.pth [--wav assets/example.wav | --tokens z.npy]
        [--coarse-lora lora.pth] [--seeds 0 1 2] [--engine hip|oracle] [--trusted]

What it does: reads the checkpoints with the product loader (audiotools dict or torch.package, LoRA adapters merged the way
loralib's eval() merges them), builds BOTH sides from the same merged weights —
  * reference: the reference's own modules (/root/reference through oracle/ref_shim.py: vampnet.interface.Interface.vamp on the
    CPU, fp32; loralib.Linear is nn.Linear there, hence the merged weights) or, when /root/reference is absent, the oracle;
  * engine: vampnet_amd.Interface on the GPU (--engine hip, default) or the CPU oracle (--engine oracle: a dry run of this
    script's plumbing, NOT a product path) —
builds the mask of configs[0] (periodic_prompt=7, upper_codebook_mask=3) and compares the tokens of
  greedy:      vamp(seed=s, sample_cutoff=-1, mask_temperature=0)      (coarse stage argmax; c2f stage is seeded-stochastic, fact 6)
  stochastic:  vamp(seed=s)
for every seed.  Exit status 0 iff every token of every case is equal.  Input tokens come from --tokens (npy, (1,14,T) int) or
from --wav through the engine's DAC encoder (codec parity itself is unpinned: both sides get the SAME tokens either way)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_KEYS = ("n_heads", "n_layers", "n_codebooks", "n_conditioning_codebooks", "latent_dim", "embedding_dim", "vocab_size")
_DEFAULT = dict(n_heads=20, n_layers=16, n_codebooks=9, n_conditioning_codebooks=0, latent_dim=8, embedding_dim=1280, vocab_size=1024)


def _dims(kw):
    k = dict(_DEFAULT, **{a: b for a, b in (kw or {}).items() if a in _KEYS})
    return dict(n_heads=k["n_heads"], n_layers=k["n_layers"], n_codebooks=k["n_codebooks"], n_cond=k["n_conditioning_codebooks"],
                latent_dim=k["latent_dim"], d_model=k["embedding_dim"], vocab=k["vocab_size"])


def load_models(args):
    from vampnet_amd.checkpoint import (load_model_checkpoint, load_tensor_dict, lora_scaling, merge_lora_state_dict,
                                        validate_vampnet_state_dict)
    from vampnet_amd.codec import DEFAULT_CFG, normalize_codec_kwargs, validate_codec_state_dict
    out = {}
    for name, path, lora in (("coarse", args.coarse, args.coarse_lora), ("c2f", args.c2f, args.c2f_lora)):
        sd, kw = load_model_checkpoint(path, package_name="VampNet", kwarg_keys=_KEYS, trusted=args.trusted)
        if lora:
            sd = {**sd, **load_tensor_dict(lora, trusted=args.trusted)}
        rep = validate_vampnet_state_dict(sd, kw)         # keys / shapes / adapter pairs against metadata.kwargs, before any packing
        out[name] = (merge_lora_state_dict(sd), _dims(kw))
        print(f"{name}: {path}: {rep['n_tensors']} tensors OK against kwargs {rep['kwargs']}; {rep['n_lora_pairs']} LoRA pairs "
              f"(rank {rep['lora_rank']}, scaling {lora_scaling(sd):g}) merged")
    csd, ckw = load_model_checkpoint(args.codec, package_name="DAC", kwarg_keys=tuple(DEFAULT_CFG), trusted=args.trusted)
    validate_codec_state_dict(csd, dict(DEFAULT_CFG, **normalize_codec_kwargs(ckw)))
    print(f"codec: {args.codec}: {len(csd)} tensors OK against kwargs {normalize_codec_kwargs(ckw)}")
    n = 1 + max(int(k.split(".")[2]) for k in csd if k.startswith("quantizer.quantizers."))
    out["codebooks"] = torch.stack([csd[f"quantizer.quantizers.{i}.codebook.weight"].float() for i in range(n)])
    return out


def reference_side(models):
    """-> (label, vamp(z, mask, **kw) on the CPU in fp32)"""
    from oracle import ref_shim, vampnet_oracle as O                # the checker
    (csd, cd), (fsd, fd), cb = models["coarse"], models["c2f"], models["codebooks"]
    if ref_shim.reference_available():
        ns = ref_shim.load_reference()
        itf = ref_shim.build_reference_interface(ns, ref_shim.build_reference_model(ns, cd, csd),
                                                 ref_shim.build_reference_model(ns, fd, fsd), ref_shim.FakeCodec(cb))
        return "reference modules (ref_shim)", lambda z, m, **kw: itf.vamp(z, m, **kw)
    om = O.OracleModels(csd, cd, fsd, fd, cb)
    return "oracle (reference absent)", lambda z, m, **kw: O.vamp(om, z, m, **kw)


def engine_side(args, models):
    if args.engine == "oracle":
        from oracle import vampnet_oracle as O
        (csd, cd), (fsd, fd), cb = models["coarse"], models["c2f"], models["codebooks"]
        om = O.OracleModels(csd, cd, fsd, fd, cb)
        return "CPU oracle (dry run)", None, lambda z, m, **kw: O.vamp(om, z, m, **kw)
    from vampnet_amd.interface import Interface
    itf = Interface(coarse_ckpt=args.coarse, coarse_lora_ckpt=args.coarse_lora, coarse2fine_ckpt=args.c2f,
                    coarse2fine_lora_ckpt=args.c2f_lora, codec_ckpt=args.codec, device="cuda:0", max_batch=1,
                    precision=args.precision, rng="torch")
    return f"HIP engine ({args.precision})", itf, lambda z, m, **kw: itf.vamp(z, m, **kw).cpu()


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--coarse", required=True)
    ap.add_argument("--c2f", required=True)
    ap.add_argument("--codec", required=True)
    ap.add_argument("--coarse-lora")
    ap.add_argument("--c2f-lora")
    ap.add_argument("--wav")
    ap.add_argument("--tokens")
    ap.add_argument("--seeds", type=int, nargs="+", default=[0])
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--engine", choices=["hip", "oracle"], default="hip")
    ap.add_argument("--precision", choices=["f32", "bf16x3", "f16x2"], default="bf16x3",
                    help="bf16x3 = the engine's default (exact three-way bf16 splits: not narrower than fp32).  f16x2 is the opt-in fast "
                         "mode: real checkpoints are the first chance to see activation ranges of a TRAINED model in it — a value beyond "
                         "fp16's range puts the model back on bf16x3 with a PrecisionFallbackWarning (the saturation ledger), it never "
                         "changes tokens silently")
    ap.add_argument("--trusted", action="store_true", help="allow torch.package archives / full unpickling (they execute code)")
    args = ap.parse_args(argv)
    with torch.no_grad():
        return _run(args)


def _run(args):
    models = load_models(args)
    elabel, itf, eng_vamp = engine_side(args, models)
    if args.tokens:
        z = torch.from_numpy(np.load(args.tokens).astype(np.int64))
    elif args.wav:
        assert itf is not None, "--wav needs the HIP engine's codec (use --tokens for a dry run)"
        from vampnet_amd.codec import AudioSignal
        z = itf.encode(AudioSignal.from_wav(args.wav)).cpu()          # 16-bit PCM wav (assets/example.wav is one)
    else:
        raise SystemExit("give --tokens or --wav")
    assert z.ndim == 3 and z.shape[0] == 1, z.shape
    rlabel, ref_vamp = reference_side(models)
    print(f"tokens {tuple(z.shape)}; reference side: {rlabel}; engine side: {elabel}")
    from oracle import vampnet_oracle as O
    bad = 0
    for seed in args.seeds:
        torch.manual_seed(seed)
        mask = O.build_mask(z, periodic_prompt=7, upper_codebook_mask=3)
        for name, kw in (("greedy", dict(sample_cutoff=-1, mask_temperature=0.0)), ("stochastic", {})):
            ref = ref_vamp(z, mask, batch_size=1, seed=seed, _sampling_steps=args.steps, **kw)
            got = eng_vamp(z, mask, batch_size=1, seed=seed, _sampling_steps=args.steps, **kw)
            same = (got == ref)
            print(f"seed {seed} {name:10s}: {int(same.sum())}/{same.numel()} tokens equal"
                  + ("" if bool(same.all()) else f"; first mismatch at (codebook, t) = {tuple(int(v) for v in (~same[0]).nonzero()[0])}"))
            bad += int((~same).sum())
    print("PARITY OK" if bad == 0 else f"PARITY FAILED: {bad} tokens differ")
    return 0 if bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
