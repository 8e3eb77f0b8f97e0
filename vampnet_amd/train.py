"""Training step of one VampNet on the HIP engine — the twin of `train_loop` in the reference's
scripts/exp/train.py:237-304 (mask -> forward in train() mode -> label-smoothed CE -> backward -> clip -> AdamW ->
NoamScheduler), for the token-level part of the loop (the codec encode in front of it is `Interface.encode`).

All arithmetic runs in libvampnet_hip.so (vn_train_* in include/vampnet_hip.h); this file only builds masks/targets
with torch tensor ops on the device, keeps the four flat state buffers (parameters, gradients, Adam moments) and, in a
data-parallel job, all-reduces the gradient buffer over RCCL between backward and update.
"""
import ctypes as C
import math

import torch

from . import _lib, masks
from ._lib import VnError, vn_dims, vn_train_params
from .engine import Engine, VampNetModel, pack_weights

IGNORE_INDEX = -100           # train.py:68
LORA_R = 8                    # transformer.py:22
LORA_SCALING = 1.0 / LORA_R   # loralib: lora_alpha (1) / r
LORA_KEYS = ("self_attn.w_qs", "self_attn.w_vs", "self_attn.fc", "feed_forward.w_1", "feed_forward.w_2")
SITES = {"attn": 0, "res1": 1, "ffn": 2, "res2": 3}


def noam_lr(step: int, d_model: int, factor: float = 2.0, warmup: int = 10000) -> float:
    """vampnet/scheduler.py:38-46 (conf/vampnet.yml:21-22) at `steps == step`."""
    return factor * (d_model ** (-0.5) * min(step ** (-0.5), step * warmup ** (-1.5)))


class Trainer:
    """Holds model + optimiser + scheduler state of train.py's `State` for one VampNet.

    `state_dict` uses the reference's parameter names; `loralib` adapters, if present, are merged at load
    (every VampNet parameter is then trained, see include/vampnet_hip.h "training step")."""

    def __init__(self, engine: Engine, sd: dict, codebooks: torch.Tensor, *, n_heads, n_layers, n_codebooks,
                 n_conditioning_codebooks=0, latent_dim=8, embedding_dim=1280, vocab_size=1024, max_batch=8, max_T=575,
                 lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_clip=5.0, label_smoothing=0.1,
                 dropout=0.1, noam_factor=2.0, noam_warmup=10000, use_noam=True, seed=0, process_group=None,
                 batch_offset=0, only_lora=False, overlap_allreduce=True, layers_per_bucket=4, zero1=False, **_ignored):
        self.engine, self.lib = engine, engine.lib
        self.dims = vn_dims(n_layers, n_heads, embedding_dim, n_codebooks, n_conditioning_codebooks, vocab_size,
                            latent_dim, 32, 128, 1e-6, max_batch, max_T)
        self.n_codebooks, self.n_cond, self.vocab, self.D = n_codebooks, n_conditioning_codebooks, vocab_size, embedding_dim
        self.Cp = n_codebooks - n_conditioning_codebooks
        self.mask_token = vocab_size
        self.hp = dict(lr=lr, beta1=betas[0], beta2=betas[1], eps=eps, weight_decay=weight_decay, grad_clip=grad_clip,
                       label_smoothing=label_smoothing, dropout=dropout)
        self.noam = (noam_factor, noam_warmup) if use_noam else None
        self.seed, self.pg, self.batch_offset = seed, process_group, batch_offset
        self.steps = 0
        # ZeRO-1 (train.py:588-590: ZeroRedundancyOptimizer when world_size > 1): Adam moments sharded over the ranks, gradients
        # reduce-scattered, parameters all-gathered after the update; the LoRA optimiser (2.9 M parameters) stays replicated
        self.zero1 = bool(zero1) and process_group is not None and not only_lora
        self.overlap = bool(overlap_allreduce) and process_group is not None and not only_lora and not self.zero1
        self.layers_per_bucket = max(1, int(layers_per_bucket))
        self._reduced = False
        n = C.c_int64()
        engine.check(self.lib.vn_train_param_size(C.byref(self.dims), C.byref(n)), "vn_train_param_size")
        self.n_total = n.value
        host = torch.zeros(self.n_total, dtype=torch.float32)
        self.only_lora = only_lora
        # parameter order of the reference model (= model.parameters() = what its AdamW indexes): state_dict order, with the
        # loralib adapters right behind their Linear's weight (lora.Linear registers weight, lora_A, lora_B:
        # transformer.py:67-68,109-114).  The reference model ALWAYS has them, so the slots are inserted whether or not this
        # state_dict holds them and in both modes: optimizer.pth indices then line up with the reference's parameters() and a
        # Trainer rebuilt from its own weights.pth (full mode writes zero adapters, see state_dict) sees the same list
        self._sd_template = {}
        lora_parents = {f"transformer.layers.{l}.{key}.weight" for l in range(n_layers) for key in LORA_KEYS}
        for k, t in sd.items():
            self._sd_template[k] = (tuple(t.shape), t.dtype)
            if k in lora_parents:
                name = k[:-len(".weight")]
                n_out, n_in = t.shape
                if name + ".lora_A" not in sd:
                    self._sd_template[name + ".lora_A"] = ((LORA_R, n_in), torch.float32)
                if name + ".lora_B" not in sd:
                    self._sd_template[name + ".lora_B"] = ((n_out, LORA_R), torch.float32)
        blob = pack_weights(self.lib, self.dims, sd, codebooks, merge_lora=not only_lora)
        self.wsize = blob.numel()
        host[:self.wsize] = blob
        og, ov = self._cls_offsets()
        V, Cp, D = vocab_size, self.Cp, embedding_dim
        if "classifier.layers.0.weight_g" not in sd:
            raise VnError("Trainer needs the weight-normed classifier parameters classifier.layers.0.weight_g / weight_v "
                          "(layers.py:47-48); this state_dict holds a folded weight only")
        g = sd["classifier.layers.0.weight_g"].float().reshape(V, Cp).t().reshape(-1)            # (p c) -> (c p)
        v = sd["classifier.layers.0.weight_v"].float().reshape(V, Cp, D).permute(1, 0, 2).reshape(-1)
        host[og:og + g.numel()] = g
        host[ov:ov + v.numel()] = v
        dev = engine.device
        if self.zero1:
            import torch.distributed as dist
            self.world, self.rank = dist.get_world_size(process_group), dist.get_rank(process_group)
            self.shard_len = -(-self.n_total // (4 * self.world)) * 4           # equal slices, 16-byte multiples
            pad = torch.zeros(self.shard_len * self.world, dtype=torch.float32, device=dev)
            pad[:self.n_total] = host.to(dev)
            self._params_pad = pad
            self.params = pad[:self.n_total]                                    # the engine's view: same storage
        else:
            self.params = host.to(dev)
        if only_lora:
            # train.py:696 lora.mark_only_lora_as_trainable: the optimiser state covers the adapters only
            self._base_sd = {k: v.detach().cpu().clone() for k, v in sd.items() if "lora_" not in k}
            self.lora = self.pack_lora(sd).to(dev)
            self.grads = torch.zeros_like(self.lora)
        elif self.zero1:
            self._grads_pad = torch.zeros_like(self._params_pad)
            self.grads = self._grads_pad[:self.n_total]
            self._gshard = torch.zeros(self.shard_len, dtype=torch.float32, device=dev)
            self._sumsq = torch.zeros(1, dtype=torch.float64, device=dev)
        else:
            self.grads = torch.zeros_like(self.params)
        self.adam_m = torch.zeros_like(self._gshard if self.zero1 else self.grads)
        self.adam_v = torch.zeros_like(self.adam_m)
        self.loss = torch.zeros(1, device=dev)
        self.grad_norm = torch.zeros(1, device=dev)
        # inference view of the same weights (generate / forward share the blob the optimiser updates)
        self.model = VampNetModel(engine, sd, codebooks, n_heads=n_heads, n_layers=n_layers, n_codebooks=n_codebooks,
                                  n_conditioning_codebooks=n_conditioning_codebooks, latent_dim=latent_dim,
                                  embedding_dim=embedding_dim, vocab_size=vocab_size, max_batch=max_batch, max_T=max_T,
                                  precision="f32", _blob=self.params)      # f32: split planes of live parameters would go stale
        h = C.c_void_p()
        engine.check(self.lib.vn_train_create(self.model.handle, self.params.data_ptr(), C.byref(h)), "vn_train_create")
        self.handle = h
        if only_lora:
            engine.check(self.lib.vn_train_enable_lora(self.handle, self.lora.data_ptr(), LORA_SCALING, engine.stream()),
                         "vn_train_enable_lora")
        else:
            engine.check(self.lib.vn_train_sync(self.handle, engine.stream()), "vn_train_sync")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.vn_train_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def _cls_offsets(self):
        off, cnt = C.c_int64(), C.c_int64()
        out = []
        for which in (0, 1):
            self.engine.check(self.lib.vn_train_param_offset(C.byref(self.dims), which, C.byref(off), C.byref(cnt)),
                              "vn_train_param_offset")
            out.append(off.value)
        return out

    def _tp(self, step, lr=None, dropout=None):
        hp = self.hp
        ws = 1
        if self.pg is not None:
            import torch.distributed as dist
            ws = dist.get_world_size(self.pg)
        return vn_train_params(hp["lr"] if lr is None else lr, hp["beta1"], hp["beta2"], hp["eps"], hp["weight_decay"],
                               hp["grad_clip"], hp["label_smoothing"], hp["dropout"] if dropout is None else dropout,
                               self.seed, step, self.batch_offset, ws)

    # ---- pieces of train_loop ------------------------------------------------------------------
    def make_batch(self, z: torch.Tensor, r: torch.Tensor = None, mask: torch.Tensor = None, generator=None):
        """train.py:250-278: (z_mask, target) from clean tokens z (B,C,T) and either mask ratios r (B,) or a mask."""
        z = z.to(self.engine.device, torch.int64)
        if mask is None:
            probs = torch.ones_like(z, dtype=torch.float32) * masks.gamma(r.to(z.device).float())[:, None, None]
            mask = torch.bernoulli(probs, generator=generator).round().long()          # mask.py:40-54
        mask = mask.to(z.device).clone()
        mask[:, :self.n_cond, :] = 0                                                    # codebook_unmask (mask.py:151-158)
        z_mask, mask = masks.apply_mask(z, mask, self.mask_token)
        target = z[:, self.n_cond:, :].permute(0, 2, 1).reshape(z.shape[0], -1)        # "b c t -> b (t c)" (util.py:35-40)
        flat = mask[:, self.n_cond:, :].permute(0, 2, 1).reshape(z.shape[0], -1)
        target = target.masked_fill(~flat.bool(), IGNORE_INDEX)
        return z_mask.contiguous(), target.contiguous()

    def forward_backward(self, z_mask, target, step=None, dropout=None):
        """Fills self.grads / self.loss for this rank's batch (no update)."""
        B, Cn, T = z_mask.shape
        assert Cn == self.n_codebooks and target.shape == (B, T * self.Cp)
        tp = self._tp(self.steps + 1 if step is None else step, dropout=dropout)
        self.engine.check(self.lib.vn_train_forward_backward(
            self.handle, z_mask.data_ptr(), target.data_ptr(), B, T, C.byref(tp), self.grads.data_ptr(),
            self.loss.data_ptr(), self.engine.stream()), "vn_train_forward_backward")
        return self.loss

    # ---- data-parallel exchange overlapped with the backward pass -------------------------------------
    def _buckets(self):
        """[(stage_hi, stage_lo, [(begin, end) gradient slices that are final once those stages ran])]: classifier + final
        norm first, then groups of `layers_per_bucket` layers (one contiguous slice each), last the embedding together with
        the shared relative-position table (accumulated by every layer)."""
        if getattr(self, "_bucket_list", None) is None:
            L = self.dims.n_layers

            def off(tid, layer=0):
                o, c = C.c_int64(), C.c_int64()
                self.engine.check(self.lib.vn_weights_offset(C.byref(self.dims), tid, layer, C.byref(o), C.byref(c)),
                                  "vn_weights_offset")
                return o.value
            layer_start = [off(_lib.W_NORM1, l) for l in range(L)] + [self.wsize]
            og, _ = self._cls_offsets()
            out = [(L, L, [(off(_lib.W_FINAL_NORM), off(_lib.W_CLS_W)), (off(_lib.W_CLS_B), layer_start[0]),
                           (og, self.n_total)])]
            hi = L - 1
            while hi >= 0:
                lo = max(0, hi - self.layers_per_bucket + 1)
                out.append((hi, lo, [(layer_start[lo], layer_start[hi + 1])]))
                hi = lo - 1
            out.append((-1, -1, [(0, off(_lib.W_FINAL_NORM))]))
            self._bucket_list = out
        return self._bucket_list

    def forward_backward_overlapped(self, z_mask, target, step=None):
        """forward + loss, then the backward in stages; as soon as a bucket of gradients is final its all-reduce is queued
        on a side stream (RCCL runs it while the next layers back-propagate).  On return the current stream waits for all
        collectives: self.grads holds the SUM over ranks."""
        import torch.distributed as dist
        B, Cn, T = z_mask.shape
        tp = self._tp(self.steps + 1 if step is None else step)
        st = self.engine.stream()
        self.engine.check(self.lib.vn_train_forward_loss(self.handle, z_mask.data_ptr(), target.data_ptr(), B, T, C.byref(tp),
                                                         self.grads.data_ptr(), self.loss.data_ptr(), st),
                          "vn_train_forward_loss")
        cur = torch.cuda.current_stream(self.engine.device)
        if getattr(self, "_comm_stream", None) is None:
            self._comm_stream = torch.cuda.Stream(self.engine.device)
        works = []
        for hi, lo, slices in self._buckets():
            self.engine.check(self.lib.vn_train_backward(self.handle, C.byref(tp), self.grads.data_ptr(), hi, lo, st),
                              "vn_train_backward")
            ev = torch.cuda.Event()
            ev.record(cur)
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(ev)
                for a, b in slices:
                    works.append(dist.all_reduce(self.grads[a:b], group=self.pg, async_op=True))
        for w in works:
            w.wait()
        cur.wait_stream(self._comm_stream)
        self._reduced = True
        return self.loss

    def forward(self, z_mask, step=None, dropout=None):
        """train()-mode logits [B, V, T*Cp] in the reference layout (transformer.py:634)."""
        B, Cn, T = z_mask.shape
        logits = torch.empty(B, T, self.Cp, self.vocab, device=self.engine.device, dtype=torch.float32)
        tp = self._tp(self.steps + 1 if step is None else step, dropout=dropout)
        self.engine.check(self.lib.vn_train_forward(self.handle, z_mask.data_ptr(), B, T, C.byref(tp), logits.data_ptr(),
                                                    self.engine.stream()), "vn_train_forward")
        return logits.permute(0, 3, 1, 2).reshape(B, self.vocab, T * self.Cp)

    # ---- ZeRO-1 ---------------------------------------------------------------------------------------
    def shard_range(self, rank=None):
        """[lo, hi) of the train vector whose optimiser state rank `rank` owns"""
        r = self.rank if rank is None else rank
        lo = min(r * self.shard_len, self.n_total)
        return lo, min(lo + self.shard_len, self.n_total)

    def _reduce_scatter_grads(self):
        """self._gshard <- SUM over ranks of this rank's slice of the (zero-padded) gradient vector"""
        import torch.distributed as dist
        lo = self.rank * self.shard_len
        if dist.get_backend(self.pg) == "gloo":            # gloo has no reduce-scatter: all-reduce, keep the slice (CPU tests)
            dist.all_reduce(self._grads_pad, group=self.pg)
            self._gshard.copy_(self._grads_pad[lo:lo + self.shard_len])
        else:
            dist.reduce_scatter_tensor(self._gshard, self._grads_pad, group=self.pg)

    def _all_gather_params(self):
        import torch.distributed as dist
        lo = self.rank * self.shard_len
        mine = self._params_pad[lo:lo + self.shard_len]
        if dist.get_backend(self.pg) == "gloo":
            parts = [self._params_pad[r * self.shard_len:(r + 1) * self.shard_len] for r in range(self.world)]
            tmp = [torch.empty_like(mine) for _ in range(self.world)]
            dist.all_gather(tmp, mine.clone(), group=self.pg)
            for dst, src in zip(parts, tmp):
                dst.copy_(src)
        else:
            dist.all_gather_into_tensor(self._params_pad, mine, group=self.pg)      # in place: `mine` is its own slot of the output

    def _shard_sumsq(self):
        """sum of squares (float64, 1 element) of the reduced gradient slice"""
        self.engine.check(self.lib.vn_train_grad_sumsq(self.handle, self._gshard.data_ptr(), self.shard_len, self._sumsq.data_ptr(),
                                                       self.engine.stream()), "vn_train_grad_sumsq")
        return self._sumsq

    def _apply_update_shard(self, step, lr, lo, hi):
        tp = self._tp(step, lr=lr)
        self.engine.check(self.lib.vn_train_update_shard(self.handle, self._gshard.data_ptr(), self.adam_m.data_ptr(),
                                                         self.adam_v.data_ptr(), C.byref(tp), lo, hi, self.grad_norm.data_ptr(),
                                                         self.engine.stream()), "vn_train_update_shard")

    def _sync_derived(self):
        self.engine.check(self.lib.vn_train_sync(self.handle, self.engine.stream()), "vn_train_sync")

    def set_overlap(self, on):
        """run the layers' weight-gradient GEMMs on the trainer's side stream (True; the default, VN_TRAIN_OVERLAP) or in the caller's
        stream (False); None = as created.  Between steps only.  Returns the state in effect."""
        return bool(self.lib.vn_debug_train_overlap(self.handle, -1 if on is None else int(bool(on))))

    def _update_zero1(self, step, lr):
        """reduce-scatter -> global norm from the slices' sums of squares -> clip + AdamW on the own slice -> all-gather."""
        import torch.distributed as dist
        self._reduce_scatter_grads()
        sumsq = self._shard_sumsq()
        dist.all_reduce(sumsq, group=self.pg)
        self.grad_norm.copy_((sumsq.sqrt() / self.world).to(torch.float32))     # || sum / world ||, as vn_train_update computes it
        lo, hi = self.shard_range()
        self._apply_update_shard(step, lr, lo, hi)
        self._all_gather_params()
        self._sync_derived()

    def consolidate_state(self):
        """ZeroRedundancyOptimizer.consolidate_state_dict (train.py:376-378): every rank receives the full moment vectors
        (returned as (m, v) of length n_total; the sharded buffers stay as they are)."""
        import torch.distributed as dist
        out = []
        for buf in (self.adam_m, self.adam_v):
            parts = [torch.empty_like(buf) for _ in range(self.world)]
            dist.all_gather(parts, buf, group=self.pg)
            out.append(torch.cat(parts)[:self.n_total])
        return tuple(out)

    def consolidate(self):
        """ZeRO-1: gather the sharded Adam moments on every rank and keep them for the next optimizer_state_dict() / save_checkpoint()
        of THIS step count.  A COLLECTIVE: every rank of the group calls it (the reference does the same before its rank-0 save,
        train.py:376-378 `consolidate_state_dict`).  No-op without ZeRO-1."""
        if getattr(self, "zero1", False):
            # kept on the HOST (2 x n_total fp32 = 2.7 GB for the coarse model would otherwise sit on every rank's GPU and undo the
            # ZeRO-1 saving), usable by every optimizer_state_dict() / save_checkpoint() of this step count, dropped by the next update()
            self._consolidated = (self.steps, tuple(t.cpu() for t in self.consolidate_state()))

    def update(self):
        """(all-reduce) -> clip -> AdamW -> scheduler.step(); advances self.steps."""
        if getattr(self, "zero1", False):
            import torch.distributed as dist
            dist.all_reduce(self.loss, group=self.pg)
            self.loss /= self.world
            step = self.steps + 1
            lr = noam_lr(step, self.D, *self.noam) if self.noam else self.hp["lr"]
            self._update_zero1(step, lr)
            self.steps = step
            self._consolidated = None            # the gathered moments of the previous step are stale now
            self.last_lr = lr
            return self.grad_norm
        if self.pg is not None:
            import torch.distributed as dist
            if not self._reduced:
                dist.all_reduce(self.grads, group=self.pg)             # SUM; the update kernel divides by world_size
            self._reduced = False
            dist.all_reduce(self.loss, group=self.pg)
            self.loss /= dist.get_world_size(self.pg)
        step = self.steps + 1
        lr = noam_lr(step, self.D, *self.noam) if self.noam else self.hp["lr"]
        self._apply_update(step, lr)
        self.steps = step
        self.last_lr = lr
        return self.grad_norm

    def _apply_update(self, step, lr):
        """grad norm (of grads / world_size) -> clip -> AdamW on the flat buffers -> re-derive folded/transposed weights."""
        tp = self._tp(step, lr=lr)
        self.engine.check(self.lib.vn_train_update(self.handle, self.grads.data_ptr(), self.adam_m.data_ptr(),
                                                   self.adam_v.data_ptr(), C.byref(tp), self.grad_norm.data_ptr(),
                                                   self.engine.stream()), "vn_train_update")

    def step(self, z, r=None, mask=None, generator=None):
        """One train_loop iteration; returns device scalars (no host sync): loss, grad_norm, and the lr used."""
        z_mask, target = self.make_batch(z, r, mask, generator)
        if self.overlap:
            self.forward_backward_overlapped(z_mask, target)
        else:
            self.forward_backward(z_mask, target)
        self.update()
        return {"loss": self.loss, "other/grad_norm": self.grad_norm, "other/learning_rate": self.last_lr,
                "other/batch_size": z.shape[0]}

    # ---- validation (train.py:327-377 val_loop, :155-215 accuracy / _metrics) -----------------------
    def evaluate(self, z, r=None, mask=None, generator=None):
        """One val_loop iteration: eval()-mode forward, CE on the masked targets and the reference's top-1 / top-25
        accuracies split by masked / unmasked tokens and by mask ratio r in [0, 0.5) / [0.5, 1).  Per-row loss and the rank
        of the true token come from the engine (vn_train_eval); only the bookkeeping over those B*T*Cp numbers is torch."""
        z = z.to(self.engine.device, torch.int64)
        if mask is None:
            probs = torch.ones_like(z, dtype=torch.float32) * masks.gamma(r.to(z.device).float())[:, None, None]
            mask = torch.bernoulli(probs, generator=generator).round().long()
        mask = mask.to(z.device).clone()
        mask[:, :self.n_cond, :] = 0
        z_mask, mask = masks.apply_mask(z, mask, self.mask_token)
        z_mask = z_mask.contiguous()
        B, _, T = z.shape
        target = z[:, self.n_cond:, :].permute(0, 2, 1).reshape(B, -1).contiguous()
        flat = mask[:, self.n_cond:, :].permute(0, 2, 1).reshape(B, -1).bool()
        row_loss = torch.empty(B, T * self.Cp, device=z.device, dtype=torch.float32)
        rank = torch.empty(B, T * self.Cp, device=z.device, dtype=torch.int32)
        self.engine.check(self.lib.vn_train_eval(self.handle, z_mask.data_ptr(), target.data_ptr(), B, T,
                                                 self.hp["label_smoothing"], row_loss.data_ptr(), rank.data_ptr(),
                                                 self.engine.stream()), "vn_train_eval")
        out = {"loss": row_loss[flat].mean() if bool(flat.any()) else row_loss.sum() * 0.0}
        if r is not None:
            r = r.to(z.device).float()
            for lo, hi in ((0, 0.5), (0.5, 1.0)):
                sel = ((r >= lo) & (r < hi))[:, None].expand_as(flat)
                for k in (1, 25):
                    hit = (rank < k).float()
                    for name, m in (("unmasked", sel & ~flat), ("masked", sel & flat)):
                        out[f"accuracy-{lo}-{hi}/top{k}/{name}"] = hit[m].mean()      # nan when the group is empty, like the reference
        return out

    # ---- checkpoint / resume (train.py:380-420 checkpoint(), :560-600 load) -------------------------
    def _all_param_names(self):
        """every parameter of the reference model in model.parameters() order (adapters included)"""
        return [k for k in self._sd_template if not k.endswith("num_batches_tracked")]

    def _param_names(self):
        """the parameters THIS trainer updates: the adapters in LoRA mode, everything else in full mode (adapters a
        checkpoint holds are merged into their weights at load — DESIGN.md section 7 — and have no moments here)"""
        names = self._all_param_names()
        return [k for k in names if "lora_" in k] if self.only_lora else [k for k in names if "lora_" not in k]

    def optimizer_state_dict(self) -> dict:
        """torch.optim.AdamW.state_dict() layout.  Parameter index = position in the reference model's parameters()
        (train.py:588-590 builds AdamW over ALL of them, adapters and frozen weights included), so indices line up with a
        reference optimizer.pth whatever subset carries state: LoRA mode writes state for the adapters only (what the
        reference has after mark_only_lora_as_trainable), full mode for every non-adapter parameter.  `lr` is what the
        reference's param group holds after N steps: NoamScheduler has already stepped to N + 1 (train.py:596)."""
        exp = self.export_lora if self.only_lora else self.export
        if getattr(self, "zero1", False):
            # sharded moments: the gathered copy of consolidate() for this step count, so that a rank-0-only caller does not enter a
            # collective alone (the usual `if rank == 0: save` pattern would hang in all_gather)
            cons = getattr(self, "_consolidated", None)
            if cons is None or cons[0] != self.steps:
                raise RuntimeError("ZeRO-1: call Trainer.consolidate() on EVERY rank (a collective) before optimizer_state_dict() / "
                                   "save_checkpoint() of this step; save_checkpoint(all_ranks=True) does it when every rank calls it")
            # NOT dropped here: the reference consolidates once and then writes every tag of the step ('latest', 'Nk', 'best' —
            # train.py:376-378, :408-420) from rank 0; the host copy lives until the next update() makes it stale
            m, v = (exp(t) for t in cons[1])
        else:
            m, v = exp(self.adam_m), exp(self.adam_v)
        index = {k: i for i, k in enumerate(self._all_param_names())}
        state = {}
        if self.steps > 0:
            for k in self._param_names():
                shp = self._sd_template[k][0]
                state[index[k]] = {"step": torch.tensor(float(self.steps)), "exp_avg": m[k].reshape(shp).clone(),
                                   "exp_avg_sq": v[k].reshape(shp).clone()}
        hp = self.hp
        lr = noam_lr(self.steps + 1, self.D, *self.noam) if self.noam else hp["lr"]
        group = {"lr": lr, "betas": (hp["beta1"], hp["beta2"]), "eps": hp["eps"],
                 "weight_decay": hp["weight_decay"], "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
                 "differentiable": False, "fused": None, "params": list(range(len(index)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, osd: dict):
        """Moments + step count from an AdamW state_dict indexed like optimizer_state_dict() (ours or the reference's).
        Raises when the file was written for a different parameter list; entries for parameters this trainer does not
        update (adapters in full mode: merged at load) are ignored, missing ones start from zero moments."""
        allnames = self._all_param_names()
        index = {k: i for i, k in enumerate(allnames)}
        groups = osd.get("param_groups", [])
        n_file = sum(len(g["params"]) for g in groups)
        if groups and n_file != len(allnames):
            raise ValueError(f"optimizer state indexes {n_file} parameters, this model has {len(allnames)} "
                             f"({sum('lora_' in k for k in allnames)} of them loralib adapters): it was written for another "
                             "architecture / adapter set")
        st = osd.get("state", {})
        if not st:
            self.adam_m.zero_(); self.adam_v.zero_()
            return
        names = self._param_names()
        have = [k for k in names if index[k] in st]
        if not have:
            raise ValueError("optimizer state holds no entry for any parameter this trainer updates "
                             f"({'LoRA adapters' if self.only_lora else 'base weights'})")
        for k in have:
            if tuple(st[index[k]]["exp_avg"].shape) != tuple(self._sd_template[k][0]):
                raise ValueError(f"optimizer state entry {index[k]} has shape {tuple(st[index[k]]['exp_avg'].shape)}, "
                                 f"parameter {k} is {tuple(self._sd_template[k][0])}")
        zeros = lambda k: torch.zeros(self._sd_template[k][0])
        m = {k: (st[index[k]]["exp_avg"] if index[k] in st else zeros(k)) for k in names}
        v = {k: (st[index[k]]["exp_avg_sq"] if index[k] in st else zeros(k)) for k in names}
        if self.only_lora:
            self.adam_m.copy_(self.pack_lora(m, init_missing=False))
            self.adam_v.copy_(self.pack_lora(v, init_missing=False))
        elif getattr(self, "zero1", False):                # keep this rank's slice of the full moment vectors
            lo = self.rank * self.shard_len
            for dst, full in ((self.adam_m, self.pack(m)), (self.adam_v, self.pack(v))):
                dst.zero_()
                n = max(0, min(self.shard_len, self.n_total - lo))
                dst[:n].copy_(full[lo:lo + n])
        else:
            self.adam_m.copy_(self.pack(m)); self.adam_v.copy_(self.pack(v))
        self.steps = int(float(st[index[have[0]]]["step"]))

    def scheduler_state_dict(self) -> dict:
        """vampnet/scheduler.py:30-33: every attribute of NoamScheduler except the optimizer.  The reference steps its
        scheduler once at construction (train.py:596) and once after every optimizer step, so after N updates it holds
        steps = N + 1 and lr = lr(N + 1) — the rate the NEXT update will use."""
        f, w = self.noam if self.noam else (None, None)
        nxt = self.steps + 1
        return {"warmup": w, "factor": f, "d_model": self.D, "lr": noam_lr(nxt, self.D, f, w) if self.noam else None,
                "steps": nxt}

    def save_checkpoint(self, save_path: str, tag: str = "latest", fine_tune: bool = None, extra: dict = None,
                        all_ranks: bool = False) -> str:
        """Writes what train.py:380-420 writes for one tag: <save_path>/<tag>/vampnet/weights.pth in the audiotools
        BaseModel layout ({"state_dict", "metadata": {"kwargs"}}), optimizer.pth, scheduler.pth and, when
        fine-tuning, lora.pth (= loralib.lora_state_dict).  Readable by vampnet_amd.Interface(coarse_ckpt=..., coarse_lora_ckpt=...).
        Multi-rank jobs: either call consolidate() on every rank and then this on rank 0 (the reference's pattern), or call this with
        all_ranks=True on EVERY rank — it consolidates (ZeRO-1) and only rank 0 writes."""
        import os
        fine_tune = self.only_lora if fine_tune is None else fine_tune
        folder = os.path.join(save_path, tag)
        if all_ranks:
            self.consolidate()
            if self.pg is not None:
                import torch.distributed as dist
                if dist.get_rank(self.pg) != 0:
                    return folder
        os.makedirs(os.path.join(folder, "vampnet"), exist_ok=True)
        d = self.dims
        kwargs = dict(n_heads=d.n_heads, n_layers=d.n_layers, n_codebooks=d.n_codebooks, n_conditioning_codebooks=d.n_cond,
                      latent_dim=d.latent_dim, embedding_dim=d.d_model, vocab_size=d.vocab, dropout=self.hp["dropout"])
        sd = self.state_dict()
        if fine_tune:
            torch.save(self.lora_state_dict(), os.path.join(folder, "lora.pth"))
        torch.save({"state_dict": sd, "metadata": {"kwargs": kwargs, **(extra or {})}}, os.path.join(folder, "vampnet", "weights.pth"))
        torch.save(self.optimizer_state_dict(), os.path.join(folder, "optimizer.pth"))
        torch.save(self.scheduler_state_dict(), os.path.join(folder, "scheduler.pth"))
        return folder

    def load_checkpoint(self, folder: str):
        """Resume: parameters (+ adapters), Adam moments and the step counter from a folder written by save_checkpoint (or by
        the reference's checkpoint() for the same architecture)."""
        import os
        from .checkpoint import load_model_checkpoint, load_tensor_dict
        sd, _ = load_model_checkpoint(os.path.join(folder, "vampnet", "weights.pth"))
        lp = os.path.join(folder, "lora.pth")
        if os.path.exists(lp):
            sd = {**sd, **load_tensor_dict(lp)}
        self.load_state_dict(sd)
        op = os.path.join(folder, "optimizer.pth")
        if os.path.exists(op):
            self.load_optimizer_state_dict(load_tensor_dict(op))
        sp = os.path.join(folder, "scheduler.pth")
        if os.path.exists(sp) and not os.path.exists(op):      # the optimizer state is authoritative; scheduler steps = N + 1
            self.steps = max(int(load_tensor_dict(sp).get("steps", self.steps + 1)) - 1, 0)

    def load_state_dict(self, sd: dict):
        """Replace the parameters (full mode: everything; LoRA mode: the adapters — the frozen base is what the Trainer was
        built with) and re-derive the engine's folded / transposed / merged copies."""
        if self.only_lora:
            self.lora.copy_(self.pack_lora({**self._base_sd, **{k: v for k, v in sd.items() if "lora_" in k}}))
            self._remerge()
        else:
            cb = self._tensor(self.params, _lib.W_EMB_TABLES).view(self.n_codebooks, self.vocab + 1, -1)[:, :self.vocab].cpu()
            if any("lora_" in k for k in sd):              # adapters of a checkpoint are merged at load, as in the constructor
                from .checkpoint import merge_lora_state_dict
                sd = merge_lora_state_dict(sd)
            self.params.copy_(self.pack(sd, cb))
            self.engine.check(self.lib.vn_train_sync(self.handle, self.engine.stream()), "vn_train_sync")

    def _remerge(self):
        """LoRA mode: W_eff = W + B A / 8 after an external change of the adapters (a zero-lr update would also decay)."""
        self.engine.check(self.lib.vn_train_lora_merge(self.handle, self.engine.stream()), "vn_train_lora_merge")

    # ---- debugging / parity helpers ---------------------------------------------------------------
    def dropout_keep_mask(self, layer, site, B, T, step=None, p=None):
        """The keep-mask of one dropout site in the REFERENCE tensor layout: attn (H,B,T,T); res1/res2 (B,T,D); ffn (B,T,2D)."""
        H = self.dims.n_heads
        p = self.hp["dropout"] if p is None else p
        step = self.steps + 1 if step is None else step
        if site == "attn":
            rows, cols, row0 = B * H * T, T, self.batch_offset * H * T
        else:
            rows, cols, row0 = B * T, (2 * self.D if site == "ffn" else self.D), self.batch_offset * T
        out = torch.empty(rows, cols, dtype=torch.uint8, device=self.engine.device)
        self.engine.check(self.lib.vn_dropout_keep_mask(self.engine.handle, self.seed, step, layer, SITES[site], p, row0,
                                                        rows, cols, out.data_ptr(), self.engine.stream()),
                          "vn_dropout_keep_mask")
        if site == "attn":
            return out.view(B, H, T, T).permute(1, 0, 2, 3).float()
        return out.view(B, T, cols).float()

    def _tensor(self, buf, tid, layer=0):
        off, cnt = C.c_int64(), C.c_int64()
        self.engine.check(self.lib.vn_weights_offset(C.byref(self.dims), tid, layer, C.byref(off), C.byref(cnt)),
                          "vn_weights_offset")
        return buf[off.value:off.value + cnt.value]

    def export(self, buf=None) -> dict:
        """Unpacks a train-vector buffer (default: the parameters) into the reference's state_dict naming — the inverse of
        engine.pack_weights (+ classifier weight_g / weight_v).  Used for checkpoints (`weights.pth` payload) and tests."""
        buf = (self.params if buf is None else buf).detach().cpu()
        d, D, V, Cp, Cn, L = self.dims, self.D, self.vocab, self.Cp, self.n_codebooks, self.dims.n_layers
        ld = d.latent_dim
        out = {}
        tables = self._tensor(buf, _lib.W_EMB_TABLES).view(Cn, V + 1, ld)
        out["embedding.special.MASK"] = tables[:, V, :].clone()
        out["embedding.out_proj.weight"] = self._tensor(buf, _lib.W_EMB_WT).view(Cn * ld, D).t().unsqueeze(-1).clone()
        out["embedding.out_proj.bias"] = self._tensor(buf, _lib.W_EMB_B).clone()
        out["transformer.layers.0.self_attn.relative_attention_bias.weight"] = self._tensor(buf, _lib.W_REL_BIAS).view(32, d.n_heads).clone()
        out["transformer.norm.weight"] = self._tensor(buf, _lib.W_FINAL_NORM).clone()
        og, ov = self._cls_offsets()
        out["classifier.layers.0.weight_g"] = buf[og:og + Cp * V].view(Cp, V).t().reshape(V * Cp, 1, 1).clone()
        out["classifier.layers.0.weight_v"] = buf[ov:ov + Cp * V * D].view(Cp, V, D).permute(1, 0, 2).reshape(V * Cp, D, 1).clone()
        out["classifier.layers.0.bias"] = self._tensor(buf, _lib.W_CLS_B).view(Cp, V).t().reshape(-1).clone()
        for l in range(L):
            p = f"transformer.layers.{l}."
            out[p + "norm_1.weight"] = self._tensor(buf, _lib.W_NORM1, l).clone()
            qkv = self._tensor(buf, _lib.W_QKV, l).view(3, D, D)
            out[p + "self_attn.w_qs.weight"] = qkv[0].clone()
            out[p + "self_attn.w_ks.weight"] = qkv[1].clone()
            out[p + "self_attn.w_vs.weight"] = qkv[2].clone()
            out[p + "self_attn.fc.weight"] = self._tensor(buf, _lib.W_WO, l).view(D, D).clone()
            out[p + "norm_3.weight"] = self._tensor(buf, _lib.W_NORM3, l).clone()
            w1 = self._tensor(buf, _lib.W_W1, l).view(2 * D // 32, 2, 32, D)
            out[p + "feed_forward.w_1.weight"] = torch.cat([w1[:, 0].reshape(2 * D, D), w1[:, 1].reshape(2 * D, D)], 0)
            out[p + "feed_forward.w_2.weight"] = self._tensor(buf, _lib.W_W2, l).view(D, 2 * D).clone()
        return out

    def state_dict(self) -> dict:
        if self.only_lora:          # frozen base (as loaded) + the current adapters, like the reference model's state_dict()
            return {**self._base_sd, **self.lora_state_dict()}
        # full mode: adapters a checkpoint held were merged into the weights at load, so the reference model that loads this
        # file (it always owns lora_A / lora_B) gets the merged weights with FRESH adapters — lora_B = 0 (the function is the merged
        # weights'), lora_A as loralib.Linear.reset_parameters leaves it (kaiming-uniform, a = sqrt 5).  Not zeros for A: with A = 0
        # AND B = 0 both adapter gradients vanish (dL/dA ~ B^T, dL/dB ~ A^T) and a LoRA fine-tune started from this file — here or
        # in the reference — would never move.  The draw is seeded by the parameter name, so repeated saves write the same file.
        out = self.export(self.params)
        full = {}
        for k, (shape, dtype) in getattr(self, "_sd_template", {}).items():
            if k in out:
                full[k] = out[k]
            elif k.endswith(".lora_A"):
                import zlib
                a = torch.empty(shape, dtype=torch.float32)
                g = torch.Generator().manual_seed(zlib.crc32(k.encode()))
                bound = math.sqrt(6.0 / ((1.0 + 5.0) * shape[1]))                 # kaiming_uniform_(a = sqrt 5): gain^2 = 2 / (1 + a^2)
                full[k] = a.uniform_(-bound, bound, generator=g)
            elif "lora_" in k:
                full[k] = torch.zeros(shape, dtype=torch.float32)
        full.update({k: v for k, v in out.items() if k not in full})
        return full

    # ---- LoRA vector <-> loralib naming -------------------------------------------------------------
    def _lora_slot(self, layer, which, ab):
        off, cnt = C.c_int64(), C.c_int64()
        self.engine.check(self.lib.vn_lora_param_offset(C.byref(self.dims), layer, which, ab, C.byref(off), C.byref(cnt)),
                          "vn_lora_param_offset")
        return off.value, cnt.value

    def _w1_perm(self):
        """packed row order of VN_W_W1: 64*g + i <- 32*g + i (value), 64*g + 32 + i <- 2D + 32*g + i (gate)"""
        D = self.D
        val = torch.arange(2 * D).view(2 * D // 32, 32)
        return torch.stack([val, val + 2 * D], 1).reshape(-1)

    def pack_lora(self, sd: dict, init_missing: bool = True) -> torch.Tensor:
        """loralib tensors (lora_A (r,in), lora_B (out,r)) -> engine LoRA vector; missing adapters are initialised like
        loralib.Linear.reset_parameters (A kaiming-uniform(a=sqrt 5), B zeros), or left zero (`init_missing=False`)."""
        n = C.c_int64()
        self.engine.check(self.lib.vn_lora_param_size(C.byref(self.dims), C.byref(n)), "vn_lora_param_size")
        out = torch.zeros(n.value, dtype=torch.float32)
        for l in range(self.dims.n_layers):
            for w, key in enumerate(LORA_KEYS):
                name = f"transformer.layers.{l}.{key}"
                n_out, n_in = self._sd_template[name + ".weight"][0]
                a = sd.get(name + ".lora_A")
                b = sd.get(name + ".lora_B")
                if a is not None and b is not None and init_missing and not bool(a.any()) and not bool(b.any()):
                    a = None                  # an all-zero pair (files of an older full-mode save) can never train: treat as missing
                if a is None:
                    a = torch.zeros(LORA_R, n_in)
                    if init_missing:
                        torch.nn.init.kaiming_uniform_(a, a=math.sqrt(5))
                    b = torch.zeros(n_out, LORA_R)
                a, b = a.float(), b.float()
                if key == "feed_forward.w_1":
                    b = b[self._w1_perm()]
                oa, ca = self._lora_slot(l, w, 0)
                ob, cb = self._lora_slot(l, w, 1)
                out[oa:oa + ca] = a.t().contiguous().reshape(-1)            # stored transposed: At [in][r]
                out[ob:ob + cb] = b.contiguous().reshape(-1)
        return out

    def export_lora(self, buf=None) -> dict:
        """engine LoRA vector (parameters by default, or gradients / moments) -> {name.lora_A, name.lora_B}"""
        buf = (self.lora if buf is None else buf).detach().cpu()
        out = {}
        inv = torch.argsort(self._w1_perm())
        for l in range(self.dims.n_layers):
            for w, key in enumerate(LORA_KEYS):
                name = f"transformer.layers.{l}.{key}"
                oa, ca = self._lora_slot(l, w, 0)
                ob, cb = self._lora_slot(l, w, 1)
                out[name + ".lora_A"] = buf[oa:oa + ca].view(-1, LORA_R).t().clone()
                b = buf[ob:ob + cb].view(-1, LORA_R)
                out[name + ".lora_B"] = (b[inv] if key == "feed_forward.w_1" else b).clone()
        return out

    def lora_state_dict(self) -> dict:
        """What train.py:401-405 saves as lora.pth (loralib.lora_state_dict: the names containing 'lora_')."""
        return self.export_lora(self.lora)

    def pack(self, sd_like: dict, codebooks: torch.Tensor = None) -> torch.Tensor:
        """Inverse of `export`: a state_dict-shaped set of tensors (parameters, gradients or Adam moments in the
        reference's naming) -> flat train vector on the host.  The derived VN_W_CLS_W region and, unless `codebooks` is
        given, the codec rows of VN_W_EMB_TABLES are left zero."""
        d, D, V, Cp, Cn, L = self.dims, self.D, self.vocab, self.Cp, self.n_codebooks, self.dims.n_layers
        ld = d.latent_dim
        out = torch.zeros(self.n_total, dtype=torch.float32)

        def put(tid, layer, t):
            dst = self._tensor(out, tid, layer)
            dst.copy_(t.contiguous().float().reshape(-1))

        tables = torch.zeros(Cn, V + 1, ld)
        tables[:, V] = sd_like["embedding.special.MASK"].float()
        if codebooks is not None:
            tables[:, :V] = codebooks[:Cn].float()
        put(_lib.W_EMB_TABLES, 0, tables)
        put(_lib.W_EMB_WT, 0, sd_like["embedding.out_proj.weight"].float().squeeze(-1).t())
        put(_lib.W_EMB_B, 0, sd_like["embedding.out_proj.bias"])
        put(_lib.W_REL_BIAS, 0, sd_like["transformer.layers.0.self_attn.relative_attention_bias.weight"])
        put(_lib.W_FINAL_NORM, 0, sd_like["transformer.norm.weight"])
        put(_lib.W_CLS_B, 0, sd_like["classifier.layers.0.bias"].float().reshape(V, Cp).t())
        og, ov = self._cls_offsets()
        out[og:og + Cp * V] = sd_like["classifier.layers.0.weight_g"].float().reshape(V, Cp).t().reshape(-1)
        out[ov:ov + Cp * V * D] = sd_like["classifier.layers.0.weight_v"].float().reshape(V, Cp, D).permute(1, 0, 2).reshape(-1)
        for l in range(L):
            p = f"transformer.layers.{l}."
            put(_lib.W_NORM1, l, sd_like[p + "norm_1.weight"])
            put(_lib.W_QKV, l, torch.cat([sd_like[p + "self_attn.w_qs.weight"], sd_like[p + "self_attn.w_ks.weight"],
                                          sd_like[p + "self_attn.w_vs.weight"]], 0))
            put(_lib.W_WO, l, sd_like[p + "self_attn.fc.weight"])
            put(_lib.W_NORM3, l, sd_like[p + "norm_3.weight"])
            w1 = sd_like[p + "feed_forward.w_1.weight"].float()
            put(_lib.W_W1, l, torch.stack([w1[:2 * D].reshape(2 * D // 32, 32, D), w1[2 * D:].reshape(2 * D // 32, 32, D)], 1))
            put(_lib.W_W2, l, sd_like[p + "feed_forward.w_2.weight"])
        return out
