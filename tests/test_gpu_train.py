"""`-m gpu` parity tests of the TRAINING step (BASELINE config #5): the HIP engine's
forward-with-dropout / cross-entropy / backward / clip + AdamW against oracle/train_oracle.py (torch CPU fp32 autograd,
itself pinned bitwise to the reference's modules by tests/test_oracle_vs_reference.py::test_train_step_vs_reference).

Dropout: the engine's counter-based keep-masks are read back through the C ABI (vn_dropout_keep_mask) and INJECTED into
the oracle, so both sides see the same noise.  Tolerances (fp32, different summation orders; stated per assert):
logits 2e-5 abs, loss 1e-5 rel, every gradient tensor 1e-4 of its own max-abs, updated parameters: see _check_update."""
import os

import pytest
import torch

from oracle import train_oracle as TO, vampnet_oracle as O, weights as W

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from vampnet_amd.engine import Engine
    return Engine("cuda:0")


def _trainer(engine, dims, sd, cb, **kw):
    from vampnet_amd.train import Trainer
    from vampnet_amd.synth import model_kwargs
    return Trainer(engine, sd, cb, **model_kwargs(dims), **kw)


def _masks_from_device(tr, dims, B, T, step, p):
    if p == 0:
        return None
    out = {}
    for l in range(dims["n_layers"]):
        for site in TO.DROPOUT_SITES:
            out[(l, site)] = tr.dropout_keep_mask(l, site, B, T, step=step, p=p).cpu()
    return out


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _check_update(new, ref, lr):
    """Adam's first steps move every element by ~lr * sign(g): an element whose gradient is ~0 can legitimately flip
    (|delta| up to 2 lr) when g differs in the last bits, so: all elements within 2.1 lr, 99.9 % within 0.02 lr."""
    tot = bad = 0
    for k, v in ref.items():
        d = (new[k].reshape(v.shape) - v).abs()
        assert d.max().item() <= 2.1 * lr + 1e-7, (k, d.max().item())
        tot += d.numel()
        bad += int((d > 0.02 * lr + 1e-7).sum())
    assert bad <= 1e-3 * tot, (bad, tot)


def _check_norm(norm, norm_o, grads_o):
    """The engine reduces the squared norm in double; torch's fp32 vector_norm (what clip_grad_norm_ uses) is itself only
    good to ~4e-4 on the 2.6 M-element classifier gradient, so: 1e-5 against the double norm, 1e-3 against torch's."""
    exact = torch.sqrt(sum(g.double().pow(2).sum() for g in grads_o.values())).item()
    assert abs(norm.item() - exact) < 1e-5 * exact, (norm.item(), exact)
    assert abs(norm.item() - norm_o.item()) < 1e-3 * norm_o.item()


CASES = [
    ("coarse", 2, 40, 0.1),
    ("coarse", 3, 37, 0.0),      # ragged T (not a multiple of anything), no dropout
    ("c2f", 2, 33, 0.1),         # conditioning codebooks: targets only on the 10 predicted ones
]


@pytest.mark.parametrize("which,B,T,p", CASES)
def test_train_step_vs_oracle(engine, which, B, T, p):
    dims = W.TINY_COARSE_DIMS if which == "coarse" else W.TINY_C2F_DIMS
    sd = W.synth_state_dict(dims, 0 if which == "coarse" else 1)
    # give the LoRA-free synthetic weights a non-trivial MASK row / bias table gradient path
    cb = W.synth_codebooks()
    tr = _trainer(engine, dims, sd, cb, max_batch=B, max_T=T, dropout=p, seed=11, use_noam=False, lr=1e-3)
    z = W.synth_codes(B, dims["n_codebooks"], T, seed=5)
    r = torch.linspace(0.2, 0.9, B)
    mask = TO.make_training_mask(z, r, dims["n_cond"], generator=torch.Generator().manual_seed(3))
    z_mask, target = tr.make_batch(z, mask=mask)

    cur = {k: v.clone() for k, v in sd.items()}
    state = {}
    for it in range(2):
        step = it + 1
        masks = _masks_from_device(tr, dims, B, T, step, p)
        if masks is not None and it == 0:
            keep = torch.cat([m.flatten() for m in masks.values()]).mean().item()
            assert abs(keep - (1 - p)) < 5e-3, keep                   # Bernoulli(1-p) keep rate
        loss_o, grads_o, logits_o = TO.loss_and_grads(cur, dims, cb, z, mask, masks, p)

        logits = tr.forward(z_mask, step=step).cpu()
        lerr = (logits - logits_o).abs().max().item()
        print(f'logits max abs err {lerr:.2e}')
        assert lerr < 2e-5
        loss = tr.forward_backward(z_mask, target, step=step).cpu()
        assert abs(loss.item() - loss_o.item()) < 1e-5 * abs(loss_o.item())
        grads = tr.export(tr.grads)
        worst = 0.0
        for k, g_o in grads_o.items():
            g = grads[k].reshape(g_o.shape)
            e = _rel(g, g_o) if g_o.abs().max() > 0 else float(g.abs().max())
            worst = max(worst, e)
            assert e < 1e-4, (k, e)
        print(f"{which} step {step}: loss {loss.item():.6f} (oracle {loss_o.item():.6f}), worst grad rel err {worst:.2e}")

        cur, norm_o = TO.clip_and_adamw(cur, grads_o, state, 1e-3)
        norm = tr.update().cpu()
        _check_norm(norm, norm_o, grads_o)
        new = tr.state_dict()
        _check_update(new, cur, 1e-3)
        cur = {k: new[k].reshape(v.shape).clone() for k, v in cur.items()}   # next step: same parameters on both sides
    engine.health_check()


def test_codec_codebooks_are_not_trained(engine):
    """The codec's codebook rows inside VN_W_EMB_TABLES are not VampNet parameters (train.py:257): untouched by updates,
    no weight decay; only the MASK rows move."""
    from vampnet_amd import _lib
    dims = W.TINY_COARSE_DIMS
    sd, cb = W.synth_state_dict(dims, 0), W.synth_codebooks()
    tr = _trainer(engine, dims, sd, cb, max_batch=2, max_T=24, dropout=0.1)
    before = tr._tensor(tr.params, _lib.W_EMB_TABLES).clone()
    z = W.synth_codes(2, 4, 24, seed=1)
    for _ in range(3):
        tr.step(z, r=torch.tensor([0.5, 0.7]))
    after = tr._tensor(tr.params, _lib.W_EMB_TABLES)
    V, ld = dims["vocab"], dims["latent_dim"]
    b, a = before.view(4, V + 1, ld), after.view(4, V + 1, ld)
    assert torch.equal(b[:, :V], a[:, :V])
    assert not torch.equal(b[:, V], a[:, V])


def test_overfit_fixed_batch(engine):
    """Loss on a fixed batch goes down under the reference's optimiser settings (sanity of the whole loop), and the
    inference path (generate/forward on the same blob) sees the updated weights."""
    dims = W.TINY_COARSE_DIMS
    sd, cb = W.synth_state_dict(dims, 0), W.synth_codebooks()
    tr = _trainer(engine, dims, sd, cb, max_batch=4, max_T=48, dropout=0.0, use_noam=False, lr=1e-3)
    z = W.synth_codes(4, 4, 48, seed=2)
    mask = TO.make_training_mask(z, torch.full((4,), 0.6), 0, generator=torch.Generator().manual_seed(1))
    z_mask, target = tr.make_batch(z, mask=mask)
    l0_logits = tr.model.forward_codes(z_mask).clone()
    losses = []
    for _ in range(30):
        tr.forward_backward(z_mask, target)
        losses.append(tr.loss.item())
        tr.update()
    assert losses[-1] < 0.7 * losses[0], losses[::5]
    # eval-mode forward == train-mode forward with p = 0, on the updated parameters
    ev = tr.model.forward_codes(z_mask)
    tv = tr.forward(z_mask)
    assert (ev - tv).abs().max().item() < 2e-5
    assert (ev - l0_logits).abs().max().item() > 1e-3
    # and equals the oracle run on the exported state_dict
    new = tr.state_dict()
    lo = O.forward(new, dims, O.from_codes(new, cb, z_mask.cpu()))
    assert (ev.cpu() - lo).abs().max().item() < 5e-5


def test_train_step_is_deterministic_and_shard_invariant(engine):
    """Same inputs, same step -> bitwise same loss; the dropout stream is indexed by GLOBAL batch item, so item 1 of a
    batch draws the mask a rank holding only that item (batch_offset = 1) draws."""
    dims = W.TINY_COARSE_DIMS
    sd, cb = W.synth_state_dict(dims, 0), W.synth_codebooks()
    tr = _trainer(engine, dims, sd, cb, max_batch=2, max_T=32, dropout=0.1, seed=5)
    z = W.synth_codes(2, 4, 32, seed=9)
    mask = TO.make_training_mask(z, torch.tensor([0.4, 0.8]), 0, generator=torch.Generator().manual_seed(2))
    z_mask, target = tr.make_batch(z, mask=mask)
    a = tr.forward_backward(z_mask, target).clone()
    ga = tr.grads.clone()
    b = tr.forward_backward(z_mask, target).clone()
    assert torch.equal(a, b)
    # every gradient is bitwise reproducible run to run — including the shared relative-position table, whose gradient is
    # reduced through per-wave tables and per-block slots in a fixed order (no floating-point atomics in the step)
    assert torch.equal(ga, tr.grads)
    full = tr.dropout_keep_mask(1, "attn", 2, 32)
    tr2 = _trainer(engine, dims, sd, cb, max_batch=1, max_T=32, dropout=0.1, seed=5, batch_offset=1)
    part = tr2.dropout_keep_mask(1, "attn", 1, 32)
    assert torch.equal(full[:, 1:2], part)


@pytest.mark.parametrize("which,p", [("coarse", 0.0), ("c2f", 0.0), ("coarse", 0.1)])
def test_full_size_train_step_vs_oracle(engine, which, p):
    """Full-size models (333 M / 275 M parameters), one sequence: loss and every gradient against CPU autograd (with the
    engine's own keep-masks injected when dropout is on); then one clipped AdamW update."""
    dims = W.COARSE_DIMS if which == "coarse" else W.C2F_DIMS
    T = 575 if which == "coarse" else 173
    sd = W.synth_state_dict(dims, 0 if which == "coarse" else 1)
    cb = W.synth_codebooks()
    tr = _trainer(engine, dims, sd, cb, max_batch=1, max_T=T, dropout=p, seed=9)
    z = W.synth_codes(1, dims["n_codebooks"], T, seed=4)
    mask = TO.make_training_mask(z, torch.tensor([0.7]), dims["n_cond"], generator=torch.Generator().manual_seed(6))
    z_mask, target = tr.make_batch(z, mask=mask)
    loss = tr.forward_backward(z_mask, target).cpu()
    torch.set_num_threads(max(1, min(32, torch.get_num_threads())))
    masks = _masks_from_device(tr, dims, 1, T, 1, p)
    loss_o, grads_o, _ = TO.loss_and_grads(sd, dims, cb, z, mask, masks, p)
    del masks
    assert abs(loss.item() - loss_o.item()) < 1e-5 * abs(loss_o.item())
    grads = tr.export(tr.grads)
    worst = ("", 0.0)
    for k, g_o in grads_o.items():
        e = _rel(grads[k].reshape(g_o.shape), g_o)
        if e > worst[1]:
            worst = (k, e)
        assert e < 2e-4, (k, e)
    print(f"{which} full size (dropout {p}): loss {loss.item():.6f}, worst grad rel err {worst[1]:.2e} at {worst[0]}")
    state = {}
    lr = TO.noam_lr(1, dims["d_model"])                  # conf/vampnet.yml:21-22 -> 5.6e-8 at step 1
    new_o, norm_o = TO.clip_and_adamw(sd, grads_o, state, lr)
    norm = tr.update().cpu()
    assert tr.last_lr == lr and tr.steps == 1
    _check_norm(norm, norm_o, grads_o)
    new = tr.state_dict()
    for k, v in new_o.items():                            # at this lr the step is below fp32 resolution of most weights
        assert (new[k].reshape(v.shape) - v).abs().max().item() < 2e-7, k
    engine.health_check()


# ----------------------------------------------------------------------------- LoRA-only fine-tuning (parity unpinned)
@pytest.mark.parametrize("which,B,T,p", [("coarse", 2, 40, 0.1), ("c2f", 2, 33, 0.0)])
def test_lora_finetune_step_vs_oracle(engine, which, B, T, p):
    """train.py:696 mark_only_lora_as_trainable: only lora_A / lora_B of the five LoRA'd linears per layer get gradients and
    updates; base weights, norms, embedding, classifier stay frozen.  Oracle = loralib's published forward restated in
    oracle/train_oracle.py (dependency absent from the reference: parity unpinned).  Tolerances as in
    test_train_step_vs_oracle."""
    dims = W.TINY_COARSE_DIMS if which == "coarse" else W.TINY_C2F_DIMS
    sd = TO.add_lora(W.synth_state_dict(dims, 0 if which == "coarse" else 1), dims, seed=3)
    cb = W.synth_codebooks()
    tr = _trainer(engine, dims, sd, cb, max_batch=B, max_T=T, dropout=p, seed=2, use_noam=False, lr=1e-3, only_lora=True)
    # load/export round trip of the adapters
    back = tr.lora_state_dict()
    for k, v in back.items():
        assert torch.equal(v, sd[k]), k
    z = W.synth_codes(B, dims["n_codebooks"], T, seed=8)
    mask = TO.make_training_mask(z, torch.linspace(0.3, 0.9, B), dims["n_cond"], generator=torch.Generator().manual_seed(4))
    z_mask, target = tr.make_batch(z, mask=mask)
    cur, state = {k: v.clone() for k, v in sd.items()}, {}
    for it in range(2):
        step = it + 1
        masks = _masks_from_device(tr, dims, B, T, step, p)
        loss_o, grads_o, logits_o = TO.loss_and_grads(cur, dims, cb, z, mask, masks, p, only_lora=True)
        assert set(grads_o) == set(back)
        logits = tr.forward(z_mask, step=step).cpu()
        assert (logits - logits_o).abs().max().item() < 2e-5
        loss = tr.forward_backward(z_mask, target, step=step).cpu()
        assert abs(loss.item() - loss_o.item()) < 1e-5 * abs(loss_o.item())
        grads = tr.export_lora(tr.grads)
        worst = 0.0
        for k, g_o in grads_o.items():
            e = _rel(grads[k], g_o)
            worst = max(worst, e)
            assert e < 1e-4, (k, e)
        print(f"lora {which} step {step}: loss {loss.item():.6f}, worst grad rel err {worst:.2e}")
        new_lora, norm_o = TO.clip_and_adamw({k: cur[k] for k in grads_o}, grads_o, state, 1e-3)
        norm = tr.update().cpu()
        _check_norm(norm, norm_o, grads_o)
        got = tr.lora_state_dict()
        _check_update(got, new_lora, 1e-3)
        cur.update({k: got[k].clone() for k in new_lora})
    # frozen parameters did not move; the merged blob equals W + B A / 8
    full = tr.state_dict()
    for k, v in sd.items():
        if "lora_" not in k:
            assert torch.equal(full[k], v), k
    merged = tr.export(tr.params)
    k0 = "transformer.layers.1.feed_forward.w_1"
    want = sd[k0 + ".weight"] + (full[k0 + ".lora_B"] @ full[k0 + ".lora_A"]) * 0.125
    assert (merged[k0 + ".weight"] - want).abs().max().item() < 1e-6
    # inference on the fine-tuned model == Interface-style load of base + lora.pth (merge at pack time)
    from vampnet_amd.engine import VampNetModel
    from vampnet_amd.synth import model_kwargs
    ref_model = VampNetModel(engine, full, cb, **model_kwargs(dims), max_batch=B, max_T=T, precision="f32")
    assert (ref_model.forward_codes(z_mask) - tr.model.forward_codes(z_mask)).abs().max().item() < 2e-5
    engine.health_check()


# ----------------------------------------------------------------------------- validation + checkpoint / resume
def test_evaluate_matches_reference_metrics(engine):
    """val_loop (train.py:327-377): loss and the eight accuracy metrics of _metrics (train.py:184-215) computed by the
    reference's own formulas on the oracle's eval-mode logits."""
    from einops import rearrange
    dims = W.TINY_C2F_DIMS
    sd, cb = W.synth_state_dict(dims, 1), W.synth_codebooks()
    tr = _trainer(engine, dims, sd, cb, max_batch=4, max_T=30, dropout=0.1)
    z = W.synth_codes(4, 14, 30, seed=12)
    r = torch.tensor([0.1, 0.45, 0.6, 0.95])
    mask = TO.make_training_mask(z, r, 4, generator=torch.Generator().manual_seed(8))
    out = tr.evaluate(z, r=r, mask=mask)
    zm, m = O.apply_mask(z, mask, 1024)
    z_hat = O.forward(sd, dims, O.from_codes(sd, cb, zm))
    target = O.codebook_flatten(z[:, 4:, :])
    flat = O.codebook_flatten(m[:, 4:, :])
    loss = torch.nn.functional.cross_entropy(z_hat, target.masked_fill(~flat.bool(), -100), label_smoothing=0.1)
    assert abs(out["loss"].item() - loss.item()) < 1e-5 * loss.item()

    def accuracy(preds, tgt, top_k, ignore_index=-100):            # train.py:155-183 restated
        preds = rearrange(preds, "b p s -> (b s) p")
        tgt = rearrange(tgt, "b s -> (b s)")
        keep = tgt != ignore_index
        preds, tgt = preds[keep], tgt[keep]
        _, idx = torch.topk(preds, k=top_k, dim=-1)
        return torch.eq(idx, tgt.unsqueeze(1)).sum(1).float().mean()

    for lo, hi in ((0, 0.5), (0.5, 1.0)):
        sel = (r >= lo) & (r < hi)
        for k in (1, 25):
            for name, tg in (("unmasked", target.masked_fill(flat.bool(), -100)), ("masked", target.masked_fill(~flat.bool(), -100))):
                want = accuracy(z_hat[sel], tg[sel], k)
                got = out[f"accuracy-{lo}-{hi}/top{k}/{name}"].cpu()
                assert abs(got.item() - want.item()) < 1e-6, (lo, hi, k, name, got.item(), want.item())


@pytest.mark.parametrize("only_lora", [False, True])
def test_checkpoint_resume_and_interface_load(engine, tmp_path, only_lora):
    """save_checkpoint writes the reference's layout (train.py:380-420); a fresh Trainer resumed from it continues
    bit-identically (same parameters, moments, step counter, dropout stream); torch.optim.AdamW accepts optimizer.pth; the
    Interface twin loads weights.pth (+ lora.pth) for inference."""
    dims = W.TINY_COARSE_DIMS
    base = W.synth_state_dict(dims, 0)
    sd = TO.add_lora(base, dims, seed=1, zero_b=True) if only_lora else base
    cb = W.synth_codebooks()
    kw = dict(max_batch=2, max_T=32, dropout=0.1, seed=3, only_lora=only_lora, use_noam=False, lr=1e-3)
    tr = _trainer(engine, dims, sd, cb, **kw)
    z = W.synth_codes(2, 4, 32, seed=4)
    mask = TO.make_training_mask(z, torch.tensor([0.5, 0.9]), 0, generator=torch.Generator().manual_seed(5))
    for _ in range(3):
        tr.step(z, mask=mask)
    folder = tr.save_checkpoint(str(tmp_path / "run"), tag="latest")
    tr.step(z, mask=mask)
    want_loss, want = tr.loss.clone(), {k: v.clone() for k, v in tr.state_dict().items()}

    tr2 = _trainer(engine, dims, sd, cb, **kw)
    tr2.load_checkpoint(folder)
    assert tr2.steps == 3
    tr2.step(z, mask=mask)
    assert torch.equal(tr2.loss, want_loss)
    got = tr2.state_dict()
    for k, v in want.items():
        assert torch.equal(got[k], v), k

    # the optimizer file is a torch.optim.AdamW state_dict over the same parameter list
    import os
    # (indexed over ALL of the reference model's parameters(), train.py:588-590; state only for the trained subset)
    names = tr._all_param_names()
    ps = [torch.nn.Parameter(torch.zeros(tr._sd_template[k][0])) for k in names]
    opt = torch.optim.AdamW(ps, lr=1e-3)
    opt.load_state_dict(torch.load(os.path.join(folder, "optimizer.pth")))
    trained = set(tr._param_names())
    for k, prm in zip(names, ps):
        assert (prm in opt.state) == (k in trained), k
        if k in trained:
            assert float(opt.state[prm]["step"]) == 3.0
    assert os.path.exists(os.path.join(folder, "lora.pth")) == only_lora

    # inference twin reads the same files
    from vampnet_amd.interface import _load_checkpoint
    sd_ck, kwargs = _load_checkpoint(os.path.join(folder, "vampnet", "weights.pth"))
    assert kwargs["n_layers"] == dims["n_layers"] and kwargs["embedding_dim"] == dims["d_model"]
    if only_lora:
        sd_ck.update(torch.load(os.path.join(folder, "lora.pth")))
    from vampnet_amd.engine import VampNetModel
    from vampnet_amd.synth import model_kwargs
    tr3 = _trainer(engine, dims, sd, cb, **kw)
    tr3.load_checkpoint(folder)
    inf = VampNetModel(engine, sd_ck, cb, **model_kwargs(dims), max_batch=2, max_T=32, precision="f32")
    zm, _ = tr.make_batch(z, mask=mask)
    assert (inf.forward_codes(zm) - tr3.model.forward_codes(zm)).abs().max().item() < 2e-5


def test_full_mode_resume_from_its_own_weights_file(engine, tmp_path):
    """A full-mode Trainer built from an adapter-less state_dict saves a checkpoint; a FRESH Trainer built from that weights.pth
    (not from the original state_dict) must load the optimizer file it was written with: the parameter list is the reference
    model's (adapters included) in both."""
    import os
    from vampnet_amd.interface import _load_checkpoint
    dims = W.TINY_COARSE_DIMS
    sd, cb = W.synth_state_dict(dims, 0), W.synth_codebooks()
    sd = {k: v for k, v in sd.items() if "lora_" not in k}
    kw = dict(max_batch=2, max_T=32, dropout=0.0, seed=5)
    tr = _trainer(engine, dims, sd, cb, **kw)
    z = W.synth_codes(2, 4, 32, seed=3).cuda()
    mask = torch.ones_like(z)
    mask[:, :, ::3] = 0
    for _ in range(2):
        tr.step(z, mask=mask)
    folder = tr.save_checkpoint(str(tmp_path / "run"), tag="latest")
    tr.step(z, mask=mask)
    n_lora = 2 * 5 * dims["n_layers"]
    assert sum("lora_" in k for k in tr._all_param_names()) == n_lora
    osd = torch.load(os.path.join(folder, "optimizer.pth"))
    assert len(osd["param_groups"][0]["params"]) == len(tr._all_param_names())
    sd_ck, _ = _load_checkpoint(os.path.join(folder, "vampnet", "weights.pth"))
    # fresh loralib adapters: lora_B = 0 (the saved function is the merged weights'), lora_A kaiming-uniform — NOT zeros, or a LoRA
    # fine-tune started from this file could never move (tests/test_train_host.py::test_full_mode_save_writes_trainable_adapters)
    assert sum("lora_" in k for k in sd_ck) == n_lora
    assert all(float(v.abs().max()) == 0.0 for k, v in sd_ck.items() if k.endswith(".lora_B"))
    assert all(float(v.abs().max()) > 0.0 for k, v in sd_ck.items() if k.endswith(".lora_A"))
    tr2 = _trainer(engine, dims, sd_ck, cb, **kw)            # template from the SAVED file
    tr2.load_checkpoint(folder)
    assert tr2.steps == 2
    tr2.step(z, mask=mask)
    assert torch.equal(tr2.loss, tr.loss)
    a, b = tr.state_dict(), tr2.state_dict()
    assert list(a) == list(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_zero1_shard_update_equals_full_update(engine):
    """vn_train_update_shard over a partition of the train vector == vn_train_update, bit for bit (same AdamW kernel on the
    same elements, same clip norm), with shard-local moment buffers; vn_train_grad_sumsq of the slices adds up to the norm."""
    import ctypes as C
    dims = W.TINY_COARSE_DIMS
    sd, cb = W.synth_state_dict(dims, 0), W.synth_codebooks()
    kw = dict(max_batch=2, max_T=32, dropout=0.0, seed=5, use_noam=False, lr=1e-3)
    full, sh = _trainer(engine, dims, sd, cb, **kw), _trainer(engine, dims, sd, cb, **kw)
    z = W.synth_codes(2, 4, 32, seed=9)
    mask = TO.make_training_mask(z, torch.tensor([0.4, 0.8]), 0, generator=torch.Generator().manual_seed(2))
    n = full.n_total
    cuts = [0, (n // 3) & ~3, (2 * n // 3 + 8) & ~3, n]                     # three slices, 16-byte multiples
    ms = [torch.zeros(b - a, device="cuda") for a, b in zip(cuts, cuts[1:])]
    vs = [torch.zeros(b - a, device="cuda") for a, b in zip(cuts, cuts[1:])]
    for step in (1, 2):
        zm, tg = full.make_batch(z, mask=mask)
        full.forward_backward(zm, tg, step=step)
        sh.forward_backward(zm, tg, step=step)
        assert torch.equal(full.grads, sh.grads)
        full._apply_update(step, 1e-3)
        st = engine.stream()
        tot = torch.zeros(1, dtype=torch.float64, device="cuda")
        one = torch.zeros(1, dtype=torch.float64, device="cuda")
        for a, b in zip(cuts, cuts[1:]):
            engine.check(engine.lib.vn_train_grad_sumsq(sh.handle, sh.grads[a:b].data_ptr(), b - a, one.data_ptr(), st), "sumsq")
            tot += one
        norm = tot.sqrt().float()
        assert abs(norm.item() - full.grad_norm.item()) <= 2e-6 * full.grad_norm.item()
        tp = sh._tp(step, lr=1e-3)
        for (a, b), m_, v_ in zip(zip(cuts, cuts[1:]), ms, vs):
            g = sh.grads[a:b].clone()                                        # shard-local buffers, indexed from a
            engine.check(engine.lib.vn_train_update_shard(sh.handle, g.data_ptr(), m_.data_ptr(), v_.data_ptr(), C.byref(tp), a, b,
                                                          full.grad_norm.data_ptr(), st), "vn_train_update_shard")
        engine.check(engine.lib.vn_train_sync(sh.handle, st), "vn_train_sync")
        assert torch.equal(sh.params, full.params)
        assert torch.equal(torch.cat(ms), full.adam_m) and torch.equal(torch.cat(vs), full.adam_v)


def test_zero1_trainer_step_on_a_one_rank_rccl_group(engine):
    """Trainer(zero1=True) through reduce_scatter_tensor / all_gather_into_tensor on RCCL (one-rank group on this GPU: the
    collectives are identities, the code path is the multi-GPU one): same losses, parameters and consolidated optimizer state
    as the replicated trainer."""
    import os
    import socket
    import torch.distributed as dist
    dims = W.TINY_COARSE_DIMS
    sd, cb = W.synth_state_dict(dims, 0), W.synth_codebooks()
    own_pg = not dist.is_initialized()
    if own_pg:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        kw = dict(max_batch=2, max_T=32, dropout=0.1, seed=5)
        tr = _trainer(engine, dims, sd, cb, process_group=dist.group.WORLD, zero1=True, **kw)
        ref = _trainer(engine, dims, sd, cb, **kw)
        assert tr.zero1 and not tr.overlap and tr.adam_m.numel() == tr.shard_len
        z = W.synth_codes(2, 4, 32, seed=9)
        mask = TO.make_training_mask(z, torch.tensor([0.4, 0.8]), 0, generator=torch.Generator().manual_seed(2))
        for _ in range(3):
            o1, o2 = tr.step(z, mask=mask), ref.step(z, mask=mask)
            assert o1["loss"].item() == o2["loss"].item()
            assert abs(o1["other/grad_norm"].item() - o2["other/grad_norm"].item()) <= 2e-6 * o2["other/grad_norm"].item()
        for k, v in ref.state_dict().items():
            assert (tr.state_dict()[k] - v).abs().max().item() < 1e-7, k
        with pytest.raises(RuntimeError):
            tr.optimizer_state_dict()                                        # ZeRO-1: needs consolidate() (a collective) first
        tr.consolidate()
        a, b = tr.optimizer_state_dict(), ref.optimizer_state_dict()
        assert a["param_groups"] == b["param_groups"] and set(a["state"]) == set(b["state"])
        for i in b["state"]:
            assert (a["state"][i]["exp_avg"] - b["state"][i]["exp_avg"]).abs().max().item() < 1e-9, i
        tr2 = _trainer(engine, dims, sd, cb, process_group=dist.group.WORLD, zero1=True, **kw)
        tr2.load_state_dict(tr.state_dict())
        tr2.load_optimizer_state_dict(a)                                     # resume into the sharded layout
        assert tr2.steps == 3 and torch.equal(tr2.adam_m, tr.adam_m)
    finally:
        if own_pg:
            dist.destroy_process_group()


def test_staged_backward_with_overlapped_allreduce(engine):
    """The data-parallel step with the gradient exchange overlapped with the backward pass (one-rank RCCL group on this
    GPU: the collectives are identities, everything else is the real code path): same loss and gradients as the
    monolithic call; the bucket slices tile the trainable part of the gradient vector exactly once."""
    import os
    import socket
    import torch.distributed as dist
    dims = W.TINY_COARSE_DIMS
    sd, cb = W.synth_state_dict(dims, 0), W.synth_codebooks()
    own_pg = not dist.is_initialized()
    if own_pg:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        kw = dict(max_batch=2, max_T=32, dropout=0.1, seed=5, use_noam=False, lr=1e-3)
        tr = _trainer(engine, dims, sd, cb, process_group=dist.group.WORLD, layers_per_bucket=1, **kw)
        ref = _trainer(engine, dims, sd, cb, **kw)
        assert tr.overlap and not ref.overlap
        # bucket coverage: every trainable element exactly once, the derived classifier weight never
        cover = torch.zeros(tr.n_total, dtype=torch.int32)
        for _, _, slices in tr._buckets():
            for a, b in slices:
                cover[a:b] += 1
        from vampnet_amd import _lib
        cw = tr._tensor(cover, _lib.W_CLS_W)
        assert int(cw.max()) == 0
        cw += 1
        assert int(cover.min()) == 1 and int(cover.max()) == 1
        z = W.synth_codes(2, 4, 32, seed=9)
        mask = TO.make_training_mask(z, torch.tensor([0.4, 0.8]), 0, generator=torch.Generator().manual_seed(2))
        z_mask, target = tr.make_batch(z, mask=mask)
        a = tr.forward_backward_overlapped(z_mask, target).clone()
        b = ref.forward_backward(z_mask, target).clone()
        assert torch.equal(a, b)
        assert torch.equal(tr.grads, ref.grads)
        for _ in range(2):                                   # whole steps through the overlapped path
            o1 = tr.step(z, mask=mask)
            o2 = ref.step(z, mask=mask)
            assert abs(o1["loss"].item() - o2["loss"].item()) < 1e-6
        for k, v in ref.state_dict().items():
            assert (tr.state_dict()[k] - v).abs().max().item() < 1e-6, k
    finally:
        if own_pg:
            dist.destroy_process_group()


def test_training_trajectory_tracks_oracle(engine):
    """20 consecutive optimiser steps (dropout 0.1, clip, AdamW, Noam) on the engine and on the oracle with the engine's
    keep-masks injected at every step: the two loss curves stay together (no systematic bias accumulates; individual
    parameters may differ by Adam sign flips on ~zero gradients, see _check_update).  Tolerance: 1e-4 relative per step (measured 8e-7)."""
    dims = W.TINY_COARSE_DIMS
    sd, cb = W.synth_state_dict(dims, 0), W.synth_codebooks()
    B, T, p = 2, 32, 0.1
    tr = _trainer(engine, dims, sd, cb, max_batch=B, max_T=T, dropout=p, seed=21, noam_warmup=100)   # lr = 2.5e-3 at step 20
    z = W.synth_codes(B, 4, T, seed=13)
    g = torch.Generator().manual_seed(17)
    cur, state = {k: v.clone() for k, v in sd.items()}, {}
    worst = 0.0
    for step in range(1, 21):
        r = torch.rand(B, generator=g)
        mask = TO.make_training_mask(z, r, 0, generator=g)
        masks = _masks_from_device(tr, dims, B, T, step, p)
        loss_o, grads_o, _ = TO.loss_and_grads(cur, dims, cb, z, mask, masks, p)
        lr = TO.noam_lr(step, dims["d_model"], warmup=100)
        cur, _ = TO.clip_and_adamw(cur, grads_o, state, lr)
        out = tr.step(z, mask=mask)
        assert tr.last_lr == lr
        rel = abs(out["loss"].item() - loss_o.item()) / loss_o.item()
        worst = max(worst, rel)
        assert rel < 1e-4, (step, out["loss"].item(), loss_o.item())
    print(f"20-step trajectory: worst relative loss gap {worst:.2e}, final loss {out['loss'].item():.4f}")
    assert out["loss"].item() < 7.0          # and it learns (starts at ~7.1 = ln(1024) + smoothing)


@pytest.mark.parametrize("B,T", [(2, 32), (3, 29), (1, 575)])
def test_weight_gradients_without_transposes_equal_the_transposed_form(engine, monkeypatch, B, T):
    """Round 6: the dW GEMMs read dY's and X's token-major tiled planes directly (gemm_x3.hip TN operand mode; X's planes are the ones the
    forward GEMM read, kept per layer) — against VN_TRAIN_TN=0, the form that transposes both into fresh planes first.  Same six plane
    products over the same tokens; the two may pick different tile heights / k-splits (the transposed form also has the 96-row tile), so
    the bar is 2e-6 of each tensor's largest gradient, loss bitwise (the forward pass is the same kernels on the same planes).  Ragged
    token counts (B T % 16, % 32 != 0: the zero page and the zero pad rows of the stash), and a SECOND step with fewer tokens on the same
    trainer (the pad rows of the earlier shape must not leak into the contraction)."""
    dims = W.TINY_COARSE_DIMS if T < 100 else W.COARSE_DIMS
    sd, cb = W.synth_state_dict(dims, 0), W.synth_codebooks()
    out = {}
    for tn in ("1", "0"):
        monkeypatch.setenv("VN_TRAIN_TN", tn)
        tr = _trainer(engine, dims, sd, cb, max_batch=B, max_T=T, dropout=0.1 if T < 100 else 0.0, seed=5)
        res = []
        for (b, t_) in ([(B, T), (B, T - 5), (B, T)] if T < 100 else [(B, T)]):
            z = W.synth_codes(b, 4 if T < 100 else dims["n_codebooks"], t_, seed=9 + t_)
            mask = TO.make_training_mask(z, torch.linspace(0.3, 0.9, b), 0, generator=torch.Generator().manual_seed(2))
            z_mask, target = tr.make_batch(z, mask=mask)
            loss = tr.forward_backward(z_mask, target).clone()
            res.append((loss.cpu(), tr.grads.clone().cpu()))
        out[tn] = res
        del tr
    for (la, ga), (lb, gb) in zip(out["1"], out["0"]):
        assert torch.equal(la, lb)
        assert torch.isfinite(ga).all()
        assert (ga - gb).abs().max().item() <= 2e-6 * gb.abs().max().item()
        assert float(((ga - gb).double().norm() / gb.double().norm())) < 1e-6
    if len(out["1"]) > 1:
        assert not torch.equal(out["1"][0][1], out["1"][1][1])          # (the shorter batch is a different problem, not a cached result)


@pytest.mark.parametrize("B,T,dims_name", [(2, 32, "tiny"), (3, 29, "tiny"), (2, 575, "coarse")])
def test_weight_gradient_gemms_on_the_side_stream_change_nothing(engine, B, T, dims_name):
    """Round 6: the layers' dW GEMMs run on the trainer's side stream (a second context, rotating plane buffers, ready / done / join
    events) beside the chain dY -> dX -> ... of the caller's stream.  Same kernels on the same operands: loss and EVERY gradient bitwise
    equal to the single-stream order, five steps in a row on one trainer (buffers rotate 4-deep, so slots are reused and waited for),
    dropout on, a ragged token count; the hook reports the state in effect."""
    dims = W.TINY_COARSE_DIMS if dims_name == "tiny" else W.COARSE_DIMS
    sd, cb = W.synth_state_dict(dims, 0), W.synth_codebooks()
    tr = _trainer(engine, dims, sd, cb, max_batch=B, max_T=T, dropout=0.1, seed=11)
    if not tr.set_overlap(None):
        pytest.skip("the trainer was created without the side stream (VN_TRAIN_OVERLAP=0 / VN_TRAIN_TN=0 / VN_TRAIN_X3=0)")
    z = W.synth_codes(B, 4, T, seed=3)
    mask = TO.make_training_mask(z, torch.linspace(0.3, 0.9, B), 0, generator=torch.Generator().manual_seed(2))
    z_mask, target = tr.make_batch(z, mask=mask)
    res = {}
    for on in (True, False, True):
        assert tr.set_overlap(on) == on
        out = []
        for _ in range(5 if dims_name == "tiny" else 2):
            loss = tr.forward_backward(z_mask, target).clone()
            out.append((loss, tr.grads.clone()))
        res.setdefault(on, []).append(out)
    tr.set_overlap(None)
    ref = res[False][0]
    for run in res[True]:
        for (la, ga), (lb, gb) in zip(run, ref):
            assert torch.equal(la, lb) and torch.equal(ga, gb)
    assert torch.isfinite(ref[0][1]).all() and ref[0][1].abs().max() > 0


@pytest.mark.skipif(os.environ.get("VN_TRAIN_X3") == "0", reason="this IS the child process")
def test_training_step_on_the_fp32_input_mfma():
    """Since round 5 every training GEMM (forward, dX, dW) runs on the split-plane pipe by default (gemm_x3.hip, bf16x3: weights and
    their transposes split once per update, activations split / transposed straight into tiled planes — csrc/train.hip); the tests of
    this file therefore exercise THAT path.  This one repeats the step-vs-oracle / determinism / full-size / trajectory tests with
    VN_TRAIN_X3=0, the fp32-input MFMA kernel — same parity bars.  The switch is read when a trainer is created; a child process keeps
    the two runs apart."""
    import subprocess
    import sys
    env = dict(os.environ, VN_TRAIN_X3="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_train.py", "-x", "-q", "-m", "gpu", "-k",
                        "train_step_vs_oracle or deterministic or full_size_train_step or trajectory"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1200)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
