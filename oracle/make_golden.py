"""Generates tests/golden/*.npz by running the REFERENCE's own Python (through oracle/ref_shim.py)
in this container.  TEST INFRASTRUCTURE.  Re-run:  python -m oracle.make_golden

The reference ships no golden vectors, and /root/reference does not travel to the GPU
box, so these fixtures are the reference's outputs frozen on seeded synthetic weights
(vampnet_amd/synth.py: numpy PCG64 -> identical on every machine).  Each generate/vamp fixture also
stores a checksum of the first torch-CPU noise draw so a consumer can tell "RNG stream differs on
this machine" from "algorithm differs".
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_shim, vampnet_oracle as O, weights as W  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def rng_fingerprint(seed):
    torch.manual_seed(seed)
    e = torch.empty(4, 1024).exponential_(1)
    u = torch.zeros(2, 100).uniform_(1e-20, 1)
    return np.array([e.double().sum().item(), u.double().sum().item()])


def main():
    os.makedirs(OUT, exist_ok=True)
    ns = ref_shim.load_reference()
    torch.set_num_threads(8)
    cb = W.synth_codebooks()
    csd, fsd = W.synth_state_dict(W.TINY_COARSE_DIMS, 0), W.synth_state_dict(W.TINY_C2F_DIMS, 1)
    coarse = ref_shim.build_reference_model(ns, W.TINY_COARSE_DIMS, csd)
    c2f = ref_shim.build_reference_model(ns, W.TINY_C2F_DIMS, fsd)
    codec = ref_shim.FakeCodec(cb)
    itf = ref_shim.build_reference_interface(ns, coarse, c2f, codec)

    # --- forward (tiny), both stages ---------------------------------------------------------
    fw = {}
    for name, model, dims in (("coarse", coarse, W.TINY_COARSE_DIMS), ("c2f", c2f, W.TINY_C2F_DIMS)):
        codes = W.synth_codes(2, dims["n_codebooks"], 24, seed=3)
        codes[:, dims["n_cond"]:, ::3] = 1024
        with torch.inference_mode():
            logits, acts = model(model.embedding.from_codes(codes, codec), return_activations=True)
        fw[f"{name}_codes"] = codes.numpy().astype(np.int16)
        fw[f"{name}_logits"] = logits.numpy()
        fw[f"{name}_act_last"] = acts[-1].numpy()
    np.savez_compressed(os.path.join(OUT, "forward_tiny.npz"), **fw)

    # --- generate (tiny) ---------------------------------------------------------------------
    gen = {}
    cases = [
        ("coarse", 1, 50, dict(_sampling_steps=6, seed=0)),
        ("coarse", 3, 41, dict(_sampling_steps=5, seed=1, temperature=0.8, mask_temperature=7.0)),
        ("coarse", 1, 50, dict(_sampling_steps=6, seed=2, sample_cutoff=-1, mask_temperature=0.0)),
        ("coarse", 2, 33, dict(_sampling_steps=4, seed=3, top_p=0.9)),
        ("c2f", 2, 37, dict(_sampling_steps=2, seed=4)),
        ("c2f", 1, 173, dict(_sampling_steps=2, seed=5, sample_cutoff=-1, mask_temperature=0.0)),
    ]
    meta = []
    for idx, (which, B, T, kw) in enumerate(cases):
        model, dims = (coarse, W.TINY_COARSE_DIMS) if which == "coarse" else (c2f, W.TINY_C2F_DIMS)
        z = W.synth_codes(B, dims["n_codebooks"], T, seed=9 + idx)
        torch.manual_seed(123 + idx)
        mask = (torch.rand(B, dims["n_codebooks"], T) < 0.7).long()
        mask[:, :dims["n_cond"]] = 0
        out = model.generate(codec=codec, start_tokens=z.clone(), mask=mask.clone(), return_signal=False,
                             typical_filtering=True, **kw)
        gen[f"case{idx}_z"] = z.numpy().astype(np.int16)
        gen[f"case{idx}_mask"] = mask.numpy().astype(np.int8)
        gen[f"case{idx}_out"] = out.numpy().astype(np.int16)
        gen[f"case{idx}_rngfp"] = rng_fingerprint(kw["seed"])
        meta.append(repr((which, B, T, kw)))
    gen["meta"] = np.array(meta)
    np.savez_compressed(os.path.join(OUT, "generate_tiny.npz"), **gen)

    # --- build_mask --------------------------------------------------------------------------
    bm = {}
    z = W.synth_codes(2, 14, 120, seed=4)
    kws = [dict(), dict(periodic_prompt=5, upper_codebook_mask=2, _dropout=0.1),
           dict(rand_mask_intensity=0.8, prefix_s=0.2, suffix_s=0.1, periodic_prompt=0),
           dict(periodic_prompt=13, periodic_prompt_width=3, ncc=1)]
    for i, kw in enumerate(kws):
        for seed in (0, 1):
            torch.manual_seed(seed)
            bm[f"kw{i}_seed{seed}"] = itf.build_mask(z, **kw).numpy().astype(np.int8)
    bm["meta"] = np.array([repr(k) for k in kws])
    torch.manual_seed(0)
    bm["rngfp_bernoulli"] = np.array([torch.bernoulli(torch.full((64,), 0.5)).sum().item(),
                                      torch.randint(0, 7, (1,)).item()])
    np.savez_compressed(os.path.join(OUT, "build_mask.npz"), **bm)

    # --- vamp end to end (tiny models, real chunk arithmetic: T = 600 > 575) ------------------
    vp = {}
    z = W.synth_codes(1, 14, 600, seed=6)
    torch.manual_seed(3)
    mask = itf.build_mask(z)
    vp["z"], vp["mask"] = z.numpy().astype(np.int16), mask.numpy().astype(np.int8)
    vcases = [(1, dict(seed=0, _sampling_steps=4)), (2, dict(seed=1, _sampling_steps=3, temperature=0.9)),
              (1, dict(seed=2, _sampling_steps=4, sample_cutoff=-1, mask_temperature=0.0))]
    for i, (B, kw) in enumerate(vcases):
        out, mz = itf.vamp(z, mask, batch_size=B, return_mask=True, **kw)
        vp[f"case{i}_out"] = out.numpy().astype(np.int16)
        vp[f"case{i}_maskz"] = mz.numpy().astype(np.int16)
        vp[f"case{i}_rngfp"] = rng_fingerprint(kw["seed"])
    vp["meta"] = np.array([repr(c) for c in vcases])
    np.savez_compressed(os.path.join(OUT, "vamp_tiny.npz"), **vp)

    # --- bucket table, schedule counts, chunk arithmetic ----------------------
    attn = ns.transformer.MultiHeadRelativeAttention(4, 64)
    rel = torch.arange(-600, 601)
    misc = dict(bucket_rel=rel.numpy(), bucket=attn._relative_position_bucket(rel).numpy().astype(np.int8),
                gamma_r=np.linspace(0, 1, 37).astype(np.float32))
    misc["gamma"] = ns.mask._gamma(torch.from_numpy(misc["gamma_r"])).numpy()
    np.savez_compressed(os.path.join(OUT, "misc.npz"), **misc)

    # --- full-size coarse + c2f forward: sparse probes of the logits ---------------------------
    full = {}
    for name, dims, T, seed in (("coarse", W.COARSE_DIMS, 575, 0), ("c2f", W.C2F_DIMS, 173, 1)):
        sd = W.synth_state_dict(dims, seed)
        model = ref_shim.build_reference_model(ns, dims, sd)
        codes = W.synth_codes(1, dims["n_codebooks"], T, seed=11)
        codes[:, dims["n_cond"]:, 1::2] = 1024
        with torch.inference_mode():
            logits = model(model.embedding.from_codes(codes, codec))      # (1, V, T*Cp)
        lg = logits[0].T.contiguous()                                      # (T*Cp, V)
        rows = np.linspace(0, lg.shape[0] - 1, 24).astype(np.int64)
        top2 = lg.topk(2, dim=-1).values
        full[f"{name}_rows"] = rows
        full[f"{name}_logits_rows"] = lg[rows].numpy()
        full[f"{name}_argmax"] = lg.argmax(-1).numpy().astype(np.int16)
        full[f"{name}_gap"] = (top2[:, 0] - top2[:, 1]).numpy()
        full[f"{name}_rowsum"] = lg.double().sum(-1).numpy()
        del model, sd
    np.savez_compressed(os.path.join(OUT, "forward_full.npz"), **full)

    # --- one training step of the REFERENCE modules (train.py:237-304) on the tiny models, dropout masks injected ----
    import importlib
    import types
    from oracle import train_oracle as TO
    sched_mod = importlib.import_module("vampnet.scheduler")
    tr = {}
    for name, dims, seed in (("coarse", W.TINY_COARSE_DIMS, 0), ("c2f", W.TINY_C2F_DIMS, 1)):
        sd = W.synth_state_dict(dims, seed)
        model = ref_shim.build_reference_model(ns, dims, sd)
        model.train()
        B, T, p = 2, 24, 0.1
        z = W.synth_codes(B, dims["n_codebooks"], T, seed=5)
        r = torch.tensor([0.3, 0.8])
        mask = TO.make_training_mask(z, r, dims["n_cond"], generator=torch.Generator().manual_seed(3))
        masks = TO.draw_dropout_masks(dims, B, T, p, torch.Generator().manual_seed(4))
        for i, layer in enumerate(model.transformer.layers):       # inject the keep-masks (call order: attn, res1, ffn, res2)
            calls = {"n": 0}
            layer.self_attn.dropout.forward = types.MethodType(lambda self, x, i=i: x * masks[(i, "attn")] * (1.0 / (1.0 - p)), layer.self_attn.dropout)
            layer.feed_forward.drop.forward = types.MethodType(lambda self, x, i=i: x * masks[(i, "ffn")] * (1.0 / (1.0 - p)), layer.feed_forward.drop)

            def res_drop(self, x, i=i, calls=calls):
                site = "res1" if calls["n"] % 2 == 0 else "res2"
                calls["n"] += 1
                return x * masks[(i, site)] * (1.0 / (1.0 - p))
            layer.dropout.forward = types.MethodType(res_drop, layer.dropout)
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
        sched = sched_mod.NoamScheduler(opt, d_model=dims["d_model"], factor=2.0, warmup=10000)
        sched.step()
        z_mask, mk = ns.mask.apply_mask(z, mask, model.mask_token)
        z_hat = model(model.embedding.from_codes(z_mask, codec))
        target = ns.util.codebook_flatten(z[:, dims["n_cond"]:, :])
        flat = ns.util.codebook_flatten(mk[:, dims["n_cond"]:, :])
        loss = torch.nn.CrossEntropyLoss(label_smoothing=0.1)(z_hat, target.masked_fill(~flat.bool(), -100))
        loss.backward()
        gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
        names = [k for k, _ in model.named_parameters()]
        tr[f"{name}_loss"] = np.float32(loss.item())
        tr[f"{name}_grad_norm"] = np.float32(gnorm.item())
        tr[f"{name}_lr"] = np.float64(opt.param_groups[0]["lr"])
        tr[f"{name}_names"] = np.array(names)
        tr[f"{name}_grad_absmax"] = np.array([pp.grad.abs().max().item() for pp in model.parameters()], np.float32)
        tr[f"{name}_grad_sum"] = np.array([pp.grad.double().sum().item() for pp in model.parameters()], np.float64)
        for k in ("transformer.layers.1.feed_forward.w_2.weight", "transformer.layers.0.self_attn.w_qs.weight",
                  "embedding.special.MASK", "transformer.layers.0.self_attn.relative_attention_bias.weight"):
            tr[f"{name}_grad::{k}"] = dict(model.named_parameters())[k].grad.numpy().copy()
        opt.step()
        tr[f"{name}_w2_after"] = dict(model.named_parameters())["transformer.layers.1.feed_forward.w_2.weight"].detach().numpy().copy()
        del model
    np.savez_compressed(os.path.join(OUT, "train_tiny.npz"), **tr)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
