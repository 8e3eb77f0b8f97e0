"""CPU tests of the Trainer twin's host logic (batch/target construction, flat-vector pack/export, the data-parallel
gradient exchange) with oracle-backed stand-ins for the two device calls — the HIP kernels are covered by
tests/test_gpu_train.py.  Includes the world_size-2 gloo test of the DDP protocol of BASELINE configs[4]."""
import ctypes as C
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from oracle import train_oracle as TO, vampnet_oracle as O, weights as W
from vampnet_amd import _lib
from vampnet_amd._lib import vn_dims
from vampnet_amd.train import Trainer, noam_lr, IGNORE_INDEX


class _NoEngine:
    device = torch.device("cpu")

    def __init__(self):
        self.lib = _lib.load()          # host-only entry points (layout queries) work without a GPU

    def check(self, rc, what):
        assert rc == 0, what


class OracleBackedTrainer(Trainer):
    """Trainer whose two device calls are replaced by the CPU oracle; everything else (make_batch, all-reduce protocol,
    step counting, Noam schedule, pack/export) is the product's own code."""

    def __init__(self, sd, dims, cb, pg=None, batch_offset=0, **hp):
        self.engine = _NoEngine()
        self.lib = self.engine.lib
        self.odims, self.cb = dims, cb
        self.dims = vn_dims(dims["n_layers"], dims["n_heads"], dims["d_model"], dims["n_codebooks"], dims["n_cond"],
                            dims["vocab"], dims["latent_dim"], 32, 128, 1e-6, 4, 64)
        self.n_codebooks, self.n_cond, self.vocab, self.D = dims["n_codebooks"], dims["n_cond"], dims["vocab"], dims["d_model"]
        self.Cp = self.n_codebooks - self.n_cond
        self.mask_token = self.vocab
        self.hp = dict(lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-2, grad_clip=5.0, label_smoothing=0.1,
                       dropout=0.0)
        self.hp.update(hp)
        self.noam = (2.0, 10000)
        self.seed, self.pg, self.batch_offset, self.steps = 0, pg, batch_offset, 0
        self.overlap, self._reduced, self.only_lora = False, False, False
        n = C.c_int64()
        assert self.lib.vn_train_param_size(C.byref(self.dims), C.byref(n)) == 0
        self.n_total = n.value
        self.params = self.pack(sd, cb)
        self.grads = torch.zeros_like(self.params)
        self.loss, self.grad_norm = torch.zeros(1), torch.zeros(1)
        self.state = {}

    def forward_backward(self, z_mask, target, step=None, dropout=None):
        sd = self.export(self.params)
        lat = O.from_codes(sd, self.cb, z_mask)
        prm = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        logits = TO.forward_train(prm, self.odims, O.from_codes(prm, self.cb, z_mask), None, 0.0)
        loss = torch.nn.functional.cross_entropy(logits, target, label_smoothing=self.hp["label_smoothing"],
                                                 ignore_index=IGNORE_INDEX)
        loss.backward()
        self.grads = self.pack({k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in prm.items()})
        self.loss = loss.detach().reshape(1)
        return self.loss

    def _apply_update(self, step, lr):
        ws = 1
        if self.pg is not None:
            import torch.distributed as dist
            ws = dist.get_world_size(self.pg)
        sd = self.export(self.params)
        grads = {k: v / ws for k, v in self.export(self.grads).items()}
        self.state["step"] = step - 1
        new, norm = TO.clip_and_adamw(sd, grads, self.state, lr, grad_clip=self.hp["grad_clip"])
        self.params = self.pack(new, self.cb)
        self.grad_norm = norm.reshape(1)


def test_pack_export_roundtrip_and_layout():
    dims = W.TINY_C2F_DIMS
    sd, cb = W.synth_state_dict(dims, 1), W.synth_codebooks()
    tr = OracleBackedTrainer(sd, dims, cb)
    back = tr.export(tr.params)
    for k, v in sd.items():
        assert torch.equal(back[k].reshape(v.shape), v), k
    # the prefix of the train vector is the inference blob (same offsets) except for the derived classifier weight
    from vampnet_amd.engine import pack_weights
    blob = pack_weights(tr.lib, tr.dims, sd, cb)
    vec = tr.params[:blob.numel()].clone()
    cw = tr._tensor(vec, _lib.W_CLS_W)
    assert float(cw.abs().max()) == 0.0
    cw.copy_(tr._tensor(blob, _lib.W_CLS_W))
    assert torch.equal(vec, blob)


def test_make_batch_matches_reference_recipe():
    """train.py:250-278: z_mask / target from (z, mask); targets only on predicted codebooks, -100 where not masked."""
    dims = W.TINY_C2F_DIMS
    tr = OracleBackedTrainer(W.synth_state_dict(dims, 1), dims, W.synth_codebooks())
    z = W.synth_codes(2, 14, 20, seed=3)
    g = torch.Generator().manual_seed(5)
    mask = TO.make_training_mask(z, torch.tensor([0.3, 0.9]), 4, generator=g)
    z_mask, target = tr.make_batch(z, mask=mask)
    zm_o, m_o = O.apply_mask(z, mask, 1024)
    assert torch.equal(z_mask, zm_o)
    t_o = O.codebook_flatten(z[:, 4:, :]).masked_fill(~O.codebook_flatten(m_o[:, 4:, :]).bool(), -100)
    assert torch.equal(target, t_o)
    # r-driven masks consume the generator exactly like pmask.random (mask.py:40-54)
    g1, g2 = torch.Generator().manual_seed(9), torch.Generator().manual_seed(9)
    zm1, _ = tr.make_batch(z, r=torch.tensor([0.3, 0.9]), generator=g1)
    m2 = TO.make_training_mask(z, torch.tensor([0.3, 0.9]), 4, generator=g2)
    assert torch.equal(zm1, O.apply_mask(z, m2, 1024)[0])


def test_noam_schedule():
    assert noam_lr(1, 1280) == TO.noam_lr(1, 1280)
    assert noam_lr(10000, 1280) == pytest.approx(2.0 * 1280 ** -0.5 * 10000 ** -0.5)
    assert noam_lr(40000, 1280) < noam_lr(10000, 1280)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data(rank):
    z = W.synth_codes(2, 4, 24, seed=20 + rank)
    mask = TO.make_training_mask(z, torch.tensor([0.4, 0.8]), 0, generator=torch.Generator().manual_seed(rank))
    return z, mask


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    dims = W.TINY_COARSE_DIMS
    tr = OracleBackedTrainer(W.synth_state_dict(dims, 0), dims, W.synth_codebooks(), pg=dist.group.WORLD, batch_offset=2 * rank)
    z, mask = _data(rank)
    outs = []
    for _ in range(2):
        out = tr.step(z, mask=mask)
        outs.append((float(out["loss"]), float(out["other/grad_norm"]), out["other/learning_rate"]))
    q.put((rank, outs, {k: v.numpy() for k, v in tr.state_dict().items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_two_ranks_gloo():
    """DDP semantics of the reference (audiotools Accelerator -> DistributedDataParallel): every rank back-propagates
    the MEAN loss of its own batch, gradients are AVERAGED over ranks, then every rank applies the same clipped AdamW
    step.  The Trainer's exchange (all-reduce SUM of the flat gradient buffer + 1/world in the update) must reproduce
    the single-process computation on the averaged gradients, on both ranks, for two consecutive steps."""
    dims = W.TINY_COARSE_DIMS
    sd, cb = W.synth_state_dict(dims, 0), W.synth_codebooks()
    state, ref = {}, []
    cur = {k: v.clone() for k, v in sd.items()}
    for step in (1, 2):
        per = [TO.loss_and_grads(cur, dims, cb, *_data(r), None, 0.0) for r in range(2)]
        grads = {k: (per[0][1][k] + per[1][1][k]) / 2 for k in per[0][1]}
        lr = TO.noam_lr(step, dims["d_model"])
        cur, norm = TO.clip_and_adamw(cur, grads, state, lr)
        ref.append(((float(per[0][0]) + float(per[1][0])) / 2, float(norm), lr))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, outs, new in got:
        for (l, n, lr), (lo, no, lro) in zip(outs, ref):
            assert l == pytest.approx(lo, rel=1e-6) and n == pytest.approx(no, rel=1e-5) and lr == lro
        for k, v in cur.items():
            assert abs(torch.from_numpy(new[k]).reshape(v.shape) - v).max().item() < 1e-7, (rank, k)


def test_optimizer_and_scheduler_state_roundtrip():
    """optimizer.pth is a torch.optim.AdamW state_dict over the parameters in model.parameters() order; scheduler.pth carries
    the NoamScheduler attributes (vampnet/scheduler.py:30-33) as the reference holds them after N updates: its scheduler has
    stepped N + 1 times (once at construction, train.py:596).  Flat moment buffers -> state_dict -> flat buffers is lossless,
    and torch's own AdamW accepts the file."""
    dims = W.TINY_COARSE_DIMS
    sd, cb = W.synth_state_dict(dims, 0), W.synth_codebooks()
    tr = OracleBackedTrainer(sd, dims, cb)
    tr._sd_template = {k: (tuple(v.shape), v.dtype) for k, v in sd.items()}
    g = torch.Generator().manual_seed(1)
    like = {k: torch.randn(v.shape, generator=g) for k, v in sd.items()}
    tr.adam_m = tr.pack(like)
    tr.adam_v = tr.pack({k: v.abs() for k, v in like.items()})
    tr.steps, tr.last_lr = 7, noam_lr(7, dims["d_model"])
    osd = tr.optimizer_state_dict()
    names = tr._param_names()
    assert osd["param_groups"][0]["params"] == list(range(len(names)))
    assert osd["param_groups"][0]["lr"] == noam_lr(8, dims["d_model"])          # what the reference's group holds after 7 updates
    for i, k in enumerate(names):
        assert torch.equal(osd["state"][i]["exp_avg"], like[k]) and float(osd["state"][i]["step"]) == 7.0
    params = [torch.nn.Parameter(torch.zeros(sd[k].shape)) for k in names]
    opt = torch.optim.AdamW(params, lr=1e-3)
    opt.load_state_dict(osd)
    assert torch.equal(opt.state[params[3]]["exp_avg_sq"], like[names[3]].abs())
    tr2 = OracleBackedTrainer(sd, dims, cb)
    tr2._sd_template = tr._sd_template
    tr2.adam_m, tr2.adam_v = torch.zeros_like(tr.adam_m), torch.zeros_like(tr.adam_v)
    tr2.load_optimizer_state_dict(osd)
    assert tr2.steps == 7 and torch.equal(tr2.adam_m, tr.adam_m) and torch.equal(tr2.adam_v, tr.adam_v)
    assert tr.scheduler_state_dict() == {"warmup": 10000, "factor": 2.0, "d_model": dims["d_model"],
                                         "lr": noam_lr(8, dims["d_model"]), "steps": 8}
    short = dict(osd, param_groups=[dict(osd["param_groups"][0], params=list(range(len(names) - 2)))])
    with pytest.raises(ValueError, match="indexes"):
        tr2.load_optimizer_state_dict(short)


@pytest.mark.reference
def test_resume_is_in_step_with_the_reference_scheduler(tmp_path):
    """A (optimizer.pth, scheduler.pth) pair as the REFERENCE's train loop leaves it — AdamW over model.parameters(),
    `scheduler.step()` once at construction and after every update (train.py:596, :299-301) — resumes in the Trainer at the
    same update count, and the next update uses the rate the reference would use (lr(N + 1)); and the other way round."""
    import importlib
    from oracle import ref_shim
    if not ref_shim.reference_available():
        pytest.skip("needs /root/reference")
    ref_shim.load_reference()
    R = importlib.import_module("vampnet.scheduler")
    dims = W.TINY_COARSE_DIMS
    sd, cb = W.synth_state_dict(dims, 0), W.synth_codebooks()
    tr = OracleBackedTrainer(sd, dims, cb)
    tr._sd_template = {k: (tuple(v.shape), v.dtype) for k, v in sd.items()}
    names = tr._param_names()
    params = [torch.nn.Parameter(sd[k].clone()) for k in names]
    opt = torch.optim.AdamW(params)
    sched = R.NoamScheduler(opt, d_model=dims["d_model"], factor=2.0, warmup=10000)
    sched.step()
    g = torch.Generator().manual_seed(0)
    for _ in range(3):                                    # three reference updates
        for p_ in params:
            p_.grad = torch.randn(p_.shape, generator=g) * 1e-3
        opt.step()
        sched.step()
    torch.save(opt.state_dict(), tmp_path / "optimizer.pth")
    torch.save(sched.state_dict(), tmp_path / "scheduler.pth")
    from vampnet_amd.checkpoint import load_tensor_dict
    tr.adam_m, tr.adam_v = torch.zeros(tr.n_total), torch.zeros(tr.n_total)
    tr.load_optimizer_state_dict(load_tensor_dict(tmp_path / "optimizer.pth"))
    assert tr.steps == 3
    assert torch.equal(tr.export(tr.adam_m)[names[5]].reshape(params[5].shape), opt.state[params[5]]["exp_avg"])
    assert noam_lr(tr.steps + 1, tr.D, *tr.noam) == sched.lr == opt.param_groups[0]["lr"]     # rate of update 4 on both sides
    assert tr.scheduler_state_dict() == sched.state_dict()
    assert tr.optimizer_state_dict()["param_groups"][0]["lr"] == opt.state_dict()["param_groups"][0]["lr"]
    sched2 = R.NoamScheduler(torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))]), d_model=dims["d_model"])
    sched2.load_state_dict(tr.scheduler_state_dict())     # the reference resuming from the Trainer's file
    sched2.step()
    assert sched2.lr == noam_lr(5, dims["d_model"])       # after its 4th update it prepares the 5th


def test_optimizer_indices_follow_the_reference_parameter_order_with_adapters():
    """The reference's AdamW indexes model.parameters(), where every loralib Linear contributes weight, lora_A, lora_B in that
    order (transformer.py:67-68,109-114; w_ks is a plain Linear).  LoRA mode therefore writes / reads its adapter moments at
    THOSE indices (not 0..n-1 of the adapters), full mode skips the adapter slots, and a file for another parameter list is
    refused."""
    from vampnet_amd.train import LORA_KEYS, LORA_R
    dims = W.TINY_COARSE_DIMS
    base, cb = W.synth_state_dict(dims, 0), W.synth_codebooks()
    g = torch.Generator().manual_seed(3)
    sd = {}
    for k, v in base.items():                             # a checkpoint that already holds adapters, in module order
        sd[k] = v
        stem = k[:-len(".weight")]
        if k.endswith(".weight") and any(stem.endswith(key) for key in LORA_KEYS):
            sd[stem + ".lora_A"] = torch.randn(LORA_R, v.shape[1], generator=g) * 0.02
            sd[stem + ".lora_B"] = torch.randn(v.shape[0], LORA_R, generator=g) * 0.02
    allnames = list(sd)
    tr = OracleBackedTrainer(base, dims, cb)
    tr._sd_template = {k: (tuple(v.shape), v.dtype) for k, v in sd.items()}
    assert tr._all_param_names() == allnames
    n_lora = 2 * len(LORA_KEYS) * dims["n_layers"]
    assert len(tr._param_names()) == len(allnames) - n_lora
    like = {k: torch.randn(v.shape, generator=g) for k, v in base.items()}
    tr.adam_m, tr.adam_v, tr.steps = tr.pack(like), tr.pack({k: v.abs() for k, v in like.items()}), 2
    osd = tr.optimizer_state_dict()
    assert osd["param_groups"][0]["params"] == list(range(len(allnames)))
    lora_idx = {i for i, k in enumerate(allnames) if "lora_" in k}
    assert set(osd["state"]) == set(range(len(allnames))) - lora_idx            # full mode: adapters carry no moments
    k = "transformer.layers.1.feed_forward.w_2.weight"
    assert torch.equal(osd["state"][allnames.index(k)]["exp_avg"], like[k])
    # a reference file with state on every parameter: base moments land at their own indices, adapter entries are ignored
    ref_state = {i: {"step": torch.tensor(5.0), "exp_avg": torch.full(tuple(sd[n].shape), float(i)),
                     "exp_avg_sq": torch.full(tuple(sd[n].shape), float(i) + 0.5)} for i, n in enumerate(allnames)}
    tr.load_optimizer_state_dict({"state": ref_state, "param_groups": [{"params": list(range(len(allnames)))}]})
    assert tr.steps == 5
    got = tr.export(tr.adam_m)
    for n in (k, "transformer.layers.0.self_attn.w_ks.weight", "embedding.out_proj.bias"):
        assert torch.all(got[n] == float(allnames.index(n))), n
    # LoRA mode view of the same template: adapter indices are positions in the FULL list
    tr.only_lora = True
    assert [allnames.index(n) for n in tr._param_names()] == sorted(lora_idx)
    with pytest.raises(ValueError, match="indexes"):
        tr.load_optimizer_state_dict({"state": ref_state, "param_groups": [{"params": list(range(len(base)))}]})


def test_full_mode_save_writes_trainable_adapters():
    """A full-mode state_dict() feeds a later LoRA fine-tune (pretrain -> `fine_tune`, train.py:696): its adapters must be loralib's
    FRESH ones — lora_B = 0 (the function is the saved weights') and lora_A kaiming-uniform(a = sqrt 5), not zeros: with A = 0 and
    B = 0 both adapter gradients (dL/dA ~ B^T .., dL/dB ~ .. A^T) vanish and the fine-tune never moves.  Checked on the gradient."""
    import math
    from vampnet_amd.train import LORA_KEYS, LORA_R
    dims = W.TINY_COARSE_DIMS
    base, cb = W.synth_state_dict(dims, 0), W.synth_codebooks()
    tr = OracleBackedTrainer(base, dims, cb)
    tmpl = {}
    for k, v in base.items():                             # the template Trainer.__init__ builds: adapters behind their Linear
        tmpl[k] = (tuple(v.shape), v.dtype)
        stem = k[:-len(".weight")]
        if k.endswith(".weight") and any(stem.endswith(key) for key in LORA_KEYS):
            tmpl[stem + ".lora_A"] = ((LORA_R, v.shape[1]), torch.float32)
            tmpl[stem + ".lora_B"] = ((v.shape[0], LORA_R), torch.float32)
    tr._sd_template = tmpl
    full = tr.state_dict()
    again = tr.state_dict()
    n_ad = 0
    for k, v in full.items():
        if k.endswith(".lora_B"):
            assert not bool(v.any()), k
        elif k.endswith(".lora_A"):
            n_ad += 1
            bound = 1.0 / math.sqrt(v.shape[1])           # kaiming_uniform_(a = sqrt 5) on (r, fan_in): U(+-1 / sqrt(fan_in))
            assert float(v.abs().max()) <= bound and float(v.abs().max()) > 0.5 * bound and float(v.std()) > 0.4 * bound, k
            assert torch.equal(v, again[k]), "repeated saves must write the same adapters"
        else:
            assert torch.equal(v.reshape(base[k].shape), base[k]), k
    assert n_ad == len(LORA_KEYS) * dims["n_layers"]
    # the function is unchanged (B = 0) and dL/dB is NOT zero for y = x W^T + s (x A^T) B^T: dL/dB = s g^T (x A^T)
    k0 = "transformer.layers.0.self_attn.w_qs"
    x = torch.randn(5, full[k0 + ".weight"].shape[1])
    A, B = full[k0 + ".lora_A"], full[k0 + ".lora_B"].clone().requires_grad_(True)
    y = x @ full[k0 + ".weight"].t() + 0.125 * (x @ A.t()) @ B.t()
    assert torch.equal(y.detach(), x @ base[k0 + ".weight"].t())
    y.square().sum().backward()
    assert float(B.grad.abs().max()) > 0.0
    # files of the old form (A = 0 and B = 0) are treated as "no adapters" by the LoRA packer: A is drawn fresh
    tr.only_lora = True
    old = {**full, **{k: torch.zeros_like(v) for k, v in full.items() if "lora_" in k}}
    vec = tr.pack_lora(old)
    assert float(vec.abs().max()) > 0.0
    got = tr.export_lora(vec)
    assert bool(got[k0 + ".lora_A"].any()) and not bool(got[k0 + ".lora_B"].any())


# ---------------------------------------------------------------------------------------- ZeRO-1 (train.py:588-590)
class OracleBackedZeroTrainer(OracleBackedTrainer):
    """ZeRO-1 protocol of the product Trainer (reduce-scatter, norm from the slices' sums of squares, slice update, all-gather,
    consolidate) with torch stand-ins for the three device calls: the flat AdamW below is vn_adamw_kernel's arithmetic."""

    def __init__(self, sd, dims, cb, pg, batch_offset):
        super().__init__(sd, dims, cb, pg=pg, batch_offset=batch_offset)
        import torch.distributed as dist
        self.zero1, self.world, self.rank = True, dist.get_world_size(pg), dist.get_rank(pg)
        self.shard_len = -(-self.n_total // (4 * self.world)) * 4
        pad = torch.zeros(self.shard_len * self.world)
        pad[:self.n_total] = self.params
        self._params_pad, self.params = pad, pad[:self.n_total]
        self._grads_pad = torch.zeros_like(pad)
        self.grads = self._grads_pad[:self.n_total]
        self._gshard = torch.zeros(self.shard_len)
        self.adam_m, self.adam_v = torch.zeros(self.shard_len), torch.zeros(self.shard_len)
        self.grad_norm = torch.zeros(1)
        self._sd_template = {k: (tuple(v.shape), v.dtype) for k, v in sd.items()}
        self.trainable = self.pack({k: torch.ones_like(v) for k, v in sd.items()}) != 0       # pack() leaves every non-parameter slot zero

    def forward_backward(self, z_mask, target, step=None, dropout=None):
        loss = super().forward_backward(z_mask, target, step, dropout)       # rebinds self.grads to a fresh vector
        self._grads_pad.zero_()
        self._grads_pad[:self.n_total] = self.grads
        self.grads = self._grads_pad[:self.n_total]
        return loss

    def _shard_sumsq(self):
        return (self._gshard.double() ** 2).sum().reshape(1)

    def _apply_update_shard(self, step, lr, lo, hi):
        hp, n = self.hp, hi - lo
        t = self.trainable[lo:hi]
        coef = 1.0 / self.world
        if hp["grad_clip"] > 0:
            coef *= min(1.0, hp["grad_clip"] / (float(self.grad_norm) + 1e-6))
        g = self._gshard[:n] * coef
        m = hp["beta1"] * self.adam_m[:n] + (1 - hp["beta1"]) * g
        v = hp["beta2"] * self.adam_v[:n] + (1 - hp["beta2"]) * g * g
        bc1, bc2 = 1 - hp["beta1"] ** step, 1 - hp["beta2"] ** step
        p = self.params[lo:hi]
        new = p * (1 - lr * hp["weight_decay"]) - (lr / bc1) * (m / (v.sqrt() / bc2 ** 0.5 + hp["eps"]))
        self.adam_m[:n] = torch.where(t, m, self.adam_m[:n])
        self.adam_v[:n] = torch.where(t, v, self.adam_v[:n])
        self.params[lo:hi] = torch.where(t, new, p)

    def _sync_derived(self):
        pass


def _zero_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    dims = W.TINY_COARSE_DIMS
    tr = OracleBackedZeroTrainer(W.synth_state_dict(dims, 0), dims, W.synth_codebooks(), dist.group.WORLD, 2 * rank)
    z, mask = _data(rank)
    outs = []
    for _ in range(2):
        out = tr.step(z, mask=mask)
        outs.append((float(out["loss"]), float(out["other/grad_norm"]), out["other/learning_rate"]))
    try:
        tr.optimizer_state_dict()                          # a lone caller must NOT walk into the all_gather
        lone = False
    except RuntimeError:
        lone = True
    assert lone, "optimizer_state_dict() without consolidate() has to refuse, not start a collective on one rank"
    tr.consolidate()                                       # the collective, on every rank
    osd = tr.optimizer_state_dict()                        # ... after which any rank may build the dict alone
    if rank == 0:                                          # ... as often as it likes for this step: the reference writes 'latest',
        again = tr.optimizer_state_dict()                  # 'Nk' and 'best' from ONE consolidate (train.py:376-378, :408-420)
        assert all(torch.equal(again["state"][i]["exp_avg"], e["exp_avg"]) for i, e in osd["state"].items())
    lo, hi = tr.shard_range()
    q.put((rank, outs, {k: v.numpy() for k, v in tr.state_dict().items()}, (lo, hi, tr.n_total, tr.shard_len),
           {i: e["exp_avg"].numpy() for i, e in osd["state"].items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_zero1_sharded_optimizer_two_ranks_gloo():
    """ZeRO-1 (train.py:588-590 ZeroRedundancyOptimizer): Adam moments sharded over two ranks, gradients reduce-scattered, the
    global clip norm rebuilt from the slices' sums of squares, parameters all-gathered — must give the parameters, norms and
    (consolidated) moments of the replicated AdamW on the averaged gradients, on both ranks, for two steps."""
    dims = W.TINY_COARSE_DIMS
    sd, cb = W.synth_state_dict(dims, 0), W.synth_codebooks()
    state, ref = {}, []
    cur = {k: v.clone() for k, v in sd.items()}
    for step in (1, 2):
        per = [TO.loss_and_grads(cur, dims, cb, *_data(r), None, 0.0) for r in range(2)]
        grads = {k: (per[0][1][k] + per[1][1][k]) / 2 for k in per[0][1]}
        lr = TO.noam_lr(step, dims["d_model"])
        cur, norm = TO.clip_and_adamw(cur, grads, state, lr)
        ref.append(((float(per[0][0]) + float(per[1][0])) / 2, float(norm), lr))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_zero_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ranges = sorted(g[3][:2] for g in got)
    n_total, shard_len = got[0][3][2], got[0][3][3]
    assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] == shard_len and ranges[1][1] == n_total and shard_len % 4 == 0
    names = [k for k in sd]
    for rank, outs, new, _, mom in got:
        for (l, n, lr), (lo, no, lro) in zip(outs, ref):
            # the sharded norm is accumulated in float64; the reference side (torch's fp32 vector_norm of per-tensor norms) is
            # itself ~4e-5 off on these sizes
            assert l == pytest.approx(lo, rel=1e-6) and n == pytest.approx(no, rel=1e-4) and lr == lro
        for k, v in cur.items():
            assert abs(torch.from_numpy(new[k]).reshape(v.shape) - v).max().item() < 1e-6, (rank, k)
        for i, k in enumerate(names):                      # consolidated first moments == the replicated optimiser's
            want = state[("m", k)]
            assert abs(torch.from_numpy(mom[i]).reshape(want.shape) - want).max().item() < 1e-7, (rank, k)
