"""-m gpu: the ragged shapes of the memory-safety harness (round 6).  T in {1, 33, 173, 291, 575, 600} with B T % 32 != 0 puts the end of
an item / of the batch at every position of the kernels' 32-row tiles, first and last head included (H = 2 .. 4): the shapes on which a
last tile's over-read leaves a buffer if it ever does.  Every case checks values (oracle / float64 reference), so the module is an
ordinary part of the suite; tests/test_gpu_guard.py runs it — and the kernel / model / training / codec modules — a second time in child
processes whose EVERY device allocation (torch's and the library's) is a guard block flanked by unmapped pages (csrc/devmem.hip), in
both alignments.  A fault there ends the child; a pass means every access of every launch stayed inside its buffer."""
import numpy as np
import pytest
import torch

from oracle import vampnet_oracle as O, weights as W
from tests.gpu_common import model_kwargs
from tests.test_gpu_kernels import ATTN_FORMS, _attention, _attention_ref, _rand

pytestmark = pytest.mark.gpu

RAGGED = [(1, 1), (3, 33), (3, 173), (3, 291), (1, 575), (2, 600)]          # (B, T): B T = 1, 99, 519, 873, 575, 1200


@pytest.fixture(scope="module")
def eng():
    from vampnet_amd.engine import Engine
    return Engine("cuda:0")


@pytest.fixture(scope="module", params=["bf16x3", "f16x2", "f32"])
def models(eng, request):
    from vampnet_amd.engine import VampNetModel
    cb = W.synth_codebooks()
    csd, fsd = W.synth_state_dict(W.TINY_COARSE_DIMS, 0), W.synth_state_dict(W.TINY_C2F_DIMS, 1)
    kw = dict(precision=request.param, max_batch=3, max_T=600)
    return dict(cb=cb, coarse=(VampNetModel(eng, csd, cb, **kw, **model_kwargs(W.TINY_COARSE_DIMS)), csd, W.TINY_COARSE_DIMS),
                c2f=(VampNetModel(eng, fsd, cb, **kw, **model_kwargs(W.TINY_C2F_DIMS)), fsd, W.TINY_C2F_DIMS))


@pytest.mark.parametrize("B,T", RAGGED)
@pytest.mark.parametrize("which", ["coarse", "c2f"])
def test_ragged_forward_and_generate(models, which, B, T):
    """vn_forward logits vs the oracle (2e-5, the tiny-model bar of tests/test_gpu_model.py) and vn_generate tokens (3 steps, seeded CPU
    noise replayed) bit-equal to the oracle's, at max_batch x max_T = 3 x 600 workspaces that the call does NOT fill"""
    model, sd, dims = models[which]
    codes = W.synth_codes(B, dims["n_codebooks"], T, seed=3 + T)
    masked = codes.clone()
    masked[:, dims["n_cond"]:, ::3] = 1024
    ref = O.forward(sd, dims, O.from_codes(sd, models["cb"], masked))
    got = model.forward_codes(masked).cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=0, atol=2e-5)
    mask = torch.zeros_like(codes)
    mask[:, dims["n_cond"]:, ::2] = 1
    kw = dict(_sampling_steps=3, seed=5)
    want = O.generate(sd, dims, models["cb"], codes.clone(), mask.clone(), **O._gen_kwargs(dict(kw)))
    have = model.generate(return_signal=False, start_tokens=codes.clone(), mask=mask.clone(), typical_filtering=True, **kw).cpu()
    assert torch.equal(have, want)


@pytest.mark.parametrize("form", list(ATTN_FORMS))
@pytest.mark.parametrize("B,H,T", [(3, 2, 33), (3, 2, 173), (3, 2, 291), (2, 3, 600), (3, 1, 575), (5, 2, 7)])
def test_ragged_attention(eng, B, H, T, form):
    """every decomposition of the split-plane attention + the fp32-input kernel on the ragged batch shapes (tests/test_gpu_kernels.py
    holds the single-item tile edges): item boundaries fall inside key tiles, the last tile of the batch is partial"""
    q, k, v = _rand((B, H, T, 64), 20), _rand((B, H, T, 64), 21), _rand((B, H, T, 64), 22)
    table = _rand((32, H), 23)
    ref = _attention_ref(q, k, v, table)
    got = _attention(eng, form, q.cuda(), k.cuda(), v.cuda(), table.cuda()).cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=2e-5, atol=3e-6)


@pytest.mark.parametrize("entry", ["vn_attention_train_bf16x3", "vn_attention_train_f32"])
def test_attention_train_50_calls_in_one_process(eng, entry):
    """the case of profiles/r05_pytest_gpu_one_aborted_run.txt (B = 3, H = 2, T = 291, forward + backward), FIFTY consecutive calls in one
    process: each call allocates and frees its own scratch (bias table, planes, workspace), so the buffers move; results are bitwise the
    first call's (deterministic kernels, fixed-order bias gradient)"""
    B, H, T, p = 3, 2, 291, 0.1
    g = torch.Generator().manual_seed(B * 1000 + T)
    q, k, v = (torch.randn(B, H, T, 64, generator=g).cuda() for _ in range(3))
    dout = torch.randn(B, T, H * 64, generator=g).cuda()
    rel = (torch.randn(32, H, generator=g) * 0.5).cuda()
    first = None
    for it in range(50):
        out = torch.empty(B, T, H * 64, device="cuda")
        lse = torch.empty(B, H, T, device="cuda")
        dqkv = torch.zeros(B * T, 3 * H * 64, device="cuda")
        dbias = torch.zeros(32, H, device="cuda")
        eng.check(getattr(eng.lib, entry)(eng.handle, q.data_ptr(), k.data_ptr(), v.data_ptr(), rel.data_ptr(), out.data_ptr(),
                                          lse.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), dbias.data_ptr(), B, H, T, 32, 128, p, 77,
                                          eng.stream()), entry)
        res = [t.cpu() for t in (out, lse, dqkv, dbias)]
        assert all(bool(torch.isfinite(t).all()) for t in res)
        if first is None:
            first = res
        else:
            assert all(torch.equal(a, b) for a, b in zip(res, first)), f"call {it} differs from call 0"


@pytest.mark.parametrize("B,T", [(3, 33), (2, 173), (3, 291)])
def test_ragged_train_step(eng, B, T):
    """one training step (forward with dropout, CE, backward) of the tiny coarse model on ragged shapes inside a larger workspace: loss
    against the oracle with the engine's keep-masks injected, gradients finite and equal between two runs"""
    from oracle import train_oracle as TO
    from vampnet_amd.train import Trainer
    dims = W.TINY_COARSE_DIMS
    sd, cb = W.synth_state_dict(dims, 0), W.synth_codebooks()
    tr = Trainer(eng, sd, cb, **model_kwargs(dims), max_batch=3, max_T=300, dropout=0.1, seed=11, use_noam=False, lr=1e-3)
    z = W.synth_codes(B, dims["n_codebooks"], T, seed=5)
    mask = TO.make_training_mask(z, torch.linspace(0.3, 0.9, B), dims["n_cond"], generator=torch.Generator().manual_seed(3))
    batch = tr.make_batch(z, mask=mask)
    loss = float(tr.forward_backward(*batch, step=1).item())
    g1 = tr.grads.clone()
    keep = {(l, site): tr.dropout_keep_mask(l, site, B, T, step=1, p=0.1).cpu() for l in range(dims["n_layers"]) for site in TO.DROPOUT_SITES}
    loss_o, _, _ = TO.loss_and_grads(sd, dims, cb, z, mask, keep, 0.1)
    assert abs(loss - float(loss_o)) < 1e-5 * float(loss_o), (loss, float(loss_o))
    loss2 = float(tr.forward_backward(*batch, step=1).item())
    assert loss2 == loss and torch.equal(tr.grads, g1) and bool(torch.isfinite(g1).all())
