#!/usr/bin/env python
"""bench.py — VampNet `vamp()` hot path on N MI355X (contract: see the task statement / DESIGN.md §Measurement).

A "step" = one full Interface.vamp() (coarse 12 sampling steps + coarse-to-fine 4 chunks x 2 steps) over a batch of
synthetic 10 s clips (T = 575 tokens, 14 codebooks) that is already resident in HBM.  Workload = BASELINE.json
configs[2] ("coarse + c2f full vamp(), batch=8, 10 s clips, typical_filtering=True, 1xMI355X"); with --gpus N each
GPU keeps 8 clips (weak scaling; N = 8 is configs[3], batch 64 sharded 8-way with one all-gather of the tokens).
Random-init weights of the real architecture (no checkpoints in this image), device Philox RNG.
Precision of the headline (`value`, `dtype`, `roofline`): "bf16x3" — every GEMM / attention operand as three bf16 planes whose sum IS
the fp32 value, six bf16-MFMA products, fp32 accumulation: arithmetic not narrower than the reference's fp32 (--dtype f32 = the
fp32-input MFMA).  The opt-in fast mode "f16x2" (two fp16 planes per operand: 22 significand bits, fp16's range — NARROWER than fp32,
guarded by the saturation ledger) is timed AFTER the primary region on the same inputs and reported as the labelled secondary block
`"alt"` of the same JSON line; it never is `value`.
--config 1 = BASELINE configs[1] (coarse model only, batch 1, 12 steps); --config 2 (default) = configs[2].

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TOKENS_PER_CLIP = 14 * 575
PEAK_F32_MFMA_TF = 157.3           # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 chip peak
PEAK_BF16_MFMA_TF = 2500.0         # dense bf16 MFMA (32x32x16)
MEASURED_PIPE_LIMIT_TF = 1790.0    # what the matrix pipe sustains with NO data movement on the split planes of Gaussian operands in the
                                   # kernel's six-product order (profiles/r05_mfma_power_probe.txt, mode 4; 1.69-1.74 PF on uniformly random words)


def _host_facts():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"os_cpu_count": os.cpu_count(), "cpu_model": model}


def _gemm_source_sha():
    """sha256 of csrc/gemm_x3.hip: the committed PMC traffic figure is only quoted for the kernel revision it was captured on"""
    import hashlib
    try:
        return hashlib.sha256(open(os.path.join(ROOT, "vampnet_amd", "csrc", "gemm_x3.hip"), "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def _numa_bind(dev_index):
    """Pin this rank's host threads to the CPUs of its GPU's NUMA node (PCI bus id -> /sys/bus/pci/devices/<bdf>/numa_node ->
    /sys/devices/system/node/nodeN/cpulist): the launch thread of rank r then does not wander over the other socket while it feeds
    GPU r.  Returns what was done (goes into devices.numa_binding of the JSON line); never raises."""
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read().strip())
        if node < 0:
            return {"device": dev_index, "pci": bdf, "numa_node": None, "bound": False, "why": "the platform reports no NUMA node for the device"}
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = cpus & os.sched_getaffinity(0)
        if not allowed:
            return {"device": dev_index, "pci": bdf, "numa_node": node, "bound": False, "why": "no CPU of that node is in this process's affinity mask"}
        os.sched_setaffinity(0, allowed)
        return {"device": dev_index, "pci": bdf, "numa_node": node, "bound": True, "cpus": len(allowed)}
    except Exception as e:                                   # no sysfs, no attribute, no permission: run unbound
        return {"device": dev_index, "numa_node": None, "bound": False, "why": f"{type(e).__name__}: {e}"}


def _with_deadline(what, seconds, fn, rank=0):
    """run fn(); if it has not returned after `seconds`, say what hung and end the process (a collective that cannot complete —
    one rank missing, a wrong interface, a dead link — otherwise blocks inside RCCL for ever and the driver learns nothing)"""
    import threading
    done = threading.Event()

    def dog():
        if not done.wait(seconds):
            sys.stderr.write(f"bench.py rank {rank}: {what} did not finish within {seconds:.0f} s — check that all ranks started "
                             f"(WORLD_SIZE={os.environ.get('WORLD_SIZE')}), MASTER_ADDR/PORT={os.environ.get('MASTER_ADDR')}:"
                             f"{os.environ.get('MASTER_PORT')}, HSA_ENABLE_IPC_MODE_LEGACY=0, and NCCL_DEBUG=INFO for RCCL's own account\n")
            sys.stderr.flush()
            os._exit(3)
    threading.Thread(target=dog, daemon=True).start()
    try:
        return fn()
    finally:
        done.set()


def _rccl_preflight(itf, device, world, rank):
    """What RCCL itself says before anything is timed: ncclCommCount / ncclCommUserRank of a communicator of the library (the one the
    "c_abi" exchange uses, or a throw-away one when the exchange goes through torch.distributed — torch does not expose its own) and
    one all-gather of the rank ids through the exchange that will be used, counted.  All under a deadline."""
    import ctypes as C
    import torch.distributed as dist
    out = {}

    def count():
        comm, own = getattr(itf, "_comm", None), False
        if comm is None:
            comm, own = itf._make_comm(), True
        n, r = C.c_int(-1), C.c_int(-1)
        itf.engine.check(itf.engine.lib.vn_comm_count(comm, C.byref(n), C.byref(r)), "vn_comm_count")
        probe = torch.full((4,), rank, dtype=torch.int64, device=device)
        got = torch.empty(4 * world, dtype=torch.int64, device=device)
        itf.engine.check(itf.engine.lib.vn_allgather_tokens(comm, probe.data_ptr(), got.data_ptr(), 4, itf.engine.stream()), "vn_allgather_tokens")
        torch.cuda.synchronize()
        if own:
            itf.engine.lib.vn_comm_destroy(comm)
        return n.value, r.value, sorted(set(got.cpu().tolist()))
    n, r, seen = _with_deadline("creating the library's RCCL communicator (vn_comm_create) + first all-gather", 180, count, rank)
    out["rccl_nranks"] = n                                    # ncclCommCount
    out["rccl_user_rank"] = r
    out["c_abi_allgather_ranks_seen"] = len(seen)

    def torch_gather():
        got = torch.empty(world, dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(got, torch.tensor([rank], dtype=torch.int64, device=device))
        torch.cuda.synchronize()
        return len(set(got.cpu().tolist()))
    out["torch_allgather_ranks_seen"] = _with_deadline("torch.distributed all_gather_into_tensor on the nccl (= RCCL) group", 180, torch_gather, rank)
    out["ok"] = n == world and out["c_abi_allgather_ranks_seen"] == world and out["torch_allgather_ranks_seen"] == world
    if not out["ok"]:
        sys.stderr.write(f"bench.py rank {rank}: RCCL preflight mismatch: {out} for WORLD_SIZE={world}\n")
    return out


def _ref_models():
    """The reference's OWN modules (vampnet.modules.transformer.VampNet + vampnet.interface.Interface through the import shim),
    when /root/reference is mounted (this container; never on the GPU box) -> callable(z, mask) running Interface.vamp."""
    try:
        from oracle import ref_shim                      # CPU baseline leg only
        if not ref_shim.reference_available():
            return None
        from vampnet_amd import synth as W
        ns = ref_shim.load_reference()
        cb = W.synth_codebooks()
        coarse = ref_shim.build_reference_model(ns, W.COARSE_DIMS, W.synth_state_dict(W.COARSE_DIMS, 0))
        c2f = ref_shim.build_reference_model(ns, W.C2F_DIMS, W.synth_state_dict(W.C2F_DIMS, 1))
        itf = ref_shim.build_reference_interface(ns, coarse, c2f, ref_shim.FakeCodec(cb))
        return lambda z, mask, **kw: itf.vamp(z, mask, **kw)
    except Exception:
        return None


def cpu_baseline(threads=None, coarse_only=False, b8=True):
    """The CPU path timed on the host cores over ONE WHOLE 10 s clip (12 coarse steps + 4 x 2 c2f steps, B = 1; coarse_only:
    the 12 coarse steps of configs[1]) after a one-step warm-up of each model.  kind = "reference": the reference's own
    modules through oracle/ref_shim.py (only where /root/reference is mounted); kind = "port": the oracle, a port of the
    reference's torch-CPU path that the tests pin bitwise to it (tests/test_oracle_vs_reference.py, tests/golden/)."""
    from oracle import vampnet_oracle as O            # the ONLY place bench.py touches oracle/: the CPU baseline leg
    from vampnet_amd import synth as W
    cb = W.synth_codebooks()
    csd, fsd = W.synth_state_dict(W.COARSE_DIMS, 0), W.synth_state_dict(W.C2F_DIMS, 1)
    z = W.synth_codes(1, 14, 575, seed=2)
    torch.manual_seed(0)
    mask = O.build_mask(z)                             # periodic_prompt=7, upper_codebook_mask=3
    m1 = torch.ones(1, 14, 173, dtype=torch.long)
    m1[:, :4] = 0

    def warm():
        O.generate(csd, W.COARSE_DIMS, cb, z[:, :4], mask[:, :4], sampling_steps=1, seed=0)
        O.generate(fsd, W.C2F_DIMS, cb, z[:, :, :173], m1, sampling_steps=1, seed=0)

    probe = []
    if threads:
        torch.set_num_threads(threads)
    else:
        # torch's default (one thread per logical CPU) over-subscribes big hosts.  Probe ONE COARSE sampling step (T = 575: 86 % of
        # the clip's FLOPs are coarse steps) at a few thread counts and keep the fastest; the c2f steps run at the same count
        best = (1e30, torch.get_num_threads())
        for n in sorted({min(os.cpu_count() or 8, c) for c in (16, 32, 64, 128)}):
            torch.set_num_threads(n)
            O.generate(csd, W.COARSE_DIMS, cb, z[:, :4], mask[:, :4], sampling_steps=1, seed=0)
            t0 = time.perf_counter()
            O.generate(csd, W.COARSE_DIMS, cb, z[:, :4], mask[:, :4], sampling_steps=1, seed=0)
            dt = time.perf_counter() - t0
            probe.append((n, round(dt, 3)))
            best = min(best, (dt, n))
        torch.set_num_threads(best[1])
    cores = torch.get_num_threads()
    # the coarse-only line is timed through the port in every environment; VN_BENCH_CPU_KIND=port forces the port where the reference is mounted
    ref = None if coarse_only or os.environ.get("VN_BENCH_CPU_KIND") == "port" else _ref_models()
    kind = "reference" if ref is not None else "port"
    models = O.OracleModels(csd, W.COARSE_DIMS, fsd, W.C2F_DIMS, cb)
    warm()
    t0 = time.perf_counter()
    if coarse_only:
        O.coarse_vamp(models, z, mask, _sampling_steps=12, seed=0)
        tokens = 4 * 575
    else:
        if ref is not None:
            ref(z, mask, batch_size=1, _sampling_steps=12, seed=0, typical_filtering=True)
        else:
            O.vamp(models, z, mask, batch_size=1, _sampling_steps=12, seed=0, typical_filtering=True)
        tokens = TOKENS_PER_CLIP
    clip_s = time.perf_counter() - t0
    what = "12 coarse steps (B=1, T=575)" if coarse_only else "12 coarse steps (T=575) + 4 chunks x 2 c2f steps (T=173), B=1"
    res = {"value": tokens / clip_s, "unit": "codec-tokens/s", "cores": cores, "kind": kind,
           "sample": f"one whole clip, not extrapolated: {what} = {clip_s:.2f} s after a 1-step warm-up of each model; "
                     f"torch {torch.__version__} CPU fp32, {cores} threads (coarse-step probe, s per step: {probe})",
           # like-for-like notes: the GPU line is B = 8, so `batch8` below times the same batch on the host; and the port
           # does not EXECUTE typical_filter (the reference computes the filter and discards its result, transformer.py:989-993: ~6 % of
           # the reference's own wall on 8 threads) — so kind "port" flatters the CPU; the calibration against the reference's own
           # Python, run in the build container (8 threads, no GPU box ever holds /root/reference), is profiles/r06_cpu_reference_in_container.json
           "port_skips": None if kind == "reference" else "typical_filter (executed and discarded by the reference): the port is the FASTER of the two",
           "calibration": "profiles/r06_cpu_reference_in_container.json (kind = reference, this repository's build container)",
           **_host_facts()}
    if not coarse_only and b8:
        # B = 8 on the host, bounded: one coarse sampling step (T = 575) and one c2f sampling step over the 4 x 8 chunks of the batch
        # (T = 173, B = 32 — how the reference's loop would batch them at best), at the thread counts of the probe; a clip costs 12 of
        # the first and 2 of the second: tokens/s = 8 clips x 8050 tokens / (12 t_coarse + 2 t_c2f)
        z8 = W.synth_codes(8, 14, 575, seed=2)
        mask8 = O.build_mask(z8)
        zc = torch.cat([z8[:, :, i * 173:(i + 1) * 173] for i in range(3)] + [torch.nn.functional.pad(z8[:, :, 519:], (0, 117))])
        mc = torch.ones(32, 14, 173, dtype=torch.long)
        mc[:, :4] = 0
        sweep = []
        for n in sorted({min(os.cpu_count() or 8, c) for c in (16, 32)}):
            torch.set_num_threads(n)
            t0 = time.perf_counter()
            O.generate(csd, W.COARSE_DIMS, cb, z8[:, :4], mask8[:, :4], sampling_steps=1, seed=0)
            tc = time.perf_counter() - t0
            t0 = time.perf_counter()
            O.generate(fsd, W.C2F_DIMS, cb, zc, mc, sampling_steps=1, seed=0)
            tf = time.perf_counter() - t0
            sweep.append({"threads": n, "coarse_step_s": round(tc, 3), "c2f_step_s": round(tf, 3),
                          "tokens_per_s": round(8 * TOKENS_PER_CLIP / (12 * tc + 2 * tf), 1)})
        best8 = max(sweep, key=lambda r: r["tokens_per_s"])
        res["batch8"] = {"value": best8["tokens_per_s"], "unit": "codec-tokens/s", "cores": best8["threads"], "kind": "port",
                         "sample": "bounded: ONE coarse sampling step (B=8, T=575) and ONE c2f sampling step (the batch's 32 chunks, T=173) "
                                   "per thread count, composed as 12 coarse + 2 c2f steps per 8 clips (no warm-up run: the B=1 pass above "
                                   "warmed both models)", "thread_sweep": sweep}
        torch.set_num_threads(cores)
    return res


def cpu_baseline_train(threads=16):
    """BASELINE configs[4] on the host: one train_loop iteration of the coarse model (B=1, T=575) through the oracle
    (torch-CPU fp32 autograd + clip + AdamW), dropout 0.1."""
    from oracle import train_oracle as TO          # CPU baseline leg only
    from vampnet_amd import synth as W
    torch.set_num_threads(min(threads, os.cpu_count() or threads))
    dims = W.COARSE_DIMS
    sd, cb = W.synth_state_dict(dims, 0), W.synth_codebooks()
    z = W.synth_codes(1, 4, 575, seed=2)
    g = torch.Generator().manual_seed(0)
    mask = TO.make_training_mask(z, torch.tensor([0.6]), 0, generator=g)
    masks = TO.draw_dropout_masks(dims, 1, 575, 0.1, g)
    state, times = {}, []
    for it in range(2):                               # first iteration = warm-up
        t0 = time.perf_counter()
        _, grads, _ = TO.loss_and_grads(sd, dims, cb, z, mask, masks, 0.1)
        sd, _ = TO.clip_and_adamw(sd, grads, state, TO.noam_lr(it + 1, dims["d_model"]))
        times.append(time.perf_counter() - t0)
    return {"value": 4 * 575 / times[-1], "unit": "codec-tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle train step (forward+CE+autograd backward+clip+AdamW), coarse model, B=1, T=575: {times[-1]:.2f} s"}


def bench_train(args, rank, world, device, pg, barrier, result_out=sys.stdout):
    """BASELINE configs[4]: conf/vampnet.yml training step of the coarse model on synthetic DAC tokens, batch 8 per GPU,
    gradient all-reduce on RCCL.  A "step" = mask -> forward (dropout 0.1) -> CE(label_smoothing 0.1) -> backward ->
    all-reduce -> clip 5.0 -> AdamW -> Noam.  value = masked-LM tokens consumed per second over all ranks."""
    from vampnet_amd import synth as W
    from vampnet_amd.engine import Engine
    from vampnet_amd.synth import model_kwargs
    from vampnet_amd.train import Trainer
    dims = W.COARSE_DIMS
    Bg, T = args.batch_per_gpu, 575
    eng = Engine(device)
    tr = Trainer(eng, W.synth_state_dict(dims, 0), W.synth_codebooks(), **model_kwargs(dims), max_batch=Bg, max_T=T,
                 dropout=0.1, label_smoothing=0.1, grad_clip=5.0, lr=1e-3, process_group=pg, batch_offset=rank * Bg, seed=0,
                 only_lora=args.lora_only)
    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    z = torch.randint(0, 1024, (Bg, 4, T), device=device, generator=gen)
    rs = torch.rand(args.warmup + args.steps, Bg, device=device, generator=gen)

    for i in range(args.warmup):
        tr.step(z, r=rs[i], generator=gen)
    barrier()
    if not args.no_kernel_events:
        eng.profile_begin(1200 * max(args.steps, 1), stride=args.event_stride)
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = tr.step(z, r=rs[args.warmup + i], generator=gen)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = eng.profile_end() if not args.no_kernel_events else None
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # The layers' weight-gradient GEMMs run on a side stream beside the caller's (csrc/train.hip, VN_TRAIN_OVERLAP): a bracket of the timed
    # region then contains the time its kernel SHARED the chip — a lower bound of the kernel's own rate.  One extra, untimed step with the
    # side stream off gives the brackets of the kernels alone (`roofline` is taken from it; the timed region's own brackets are reported
    # next to it as `roofline.overlapped`).
    prof_serial = None
    if prof is not None and tr.set_overlap(None):
        tr.set_overlap(False)
        tr.step(z, r=rs[args.warmup], generator=gen)
        barrier()
        eng.profile_begin(1200, stride=args.event_stride)
        t1 = time.perf_counter()
        tr.step(z, r=rs[args.warmup], generator=gen)
        barrier()
        serial_s = time.perf_counter() - t1
        prof_serial = eng.profile_end()
        tr.set_overlap(None)
    eng.health_check()
    loss = float(out["loss"].item())
    assert loss == loss and loss < 20.0, loss
    if rank != 0:
        return
    tokens = world * Bg * 4 * T * args.steps
    # the step's GEMMs (forward, dX, dW) run on the split-plane pipe unless VN_TRAIN_X3=0 (csrc/train.hip): bf16x3 = three exact bf16
    # planes per operand, six bf16-MFMA products, fp32 accumulation — not narrower than the reference's fp32 (amp: false)
    train_dtype = "f32" if os.environ.get("VN_TRAIN_X3") == "0" else "bf16x3"
    fwd_gflop = 416.76                                     # coarse forward per item
    # forward + dX + dW of every product; LoRA-only skips the dW products (the rank-8 gradients are HBM-bound passes)
    step_tflop = (2.0 if args.lora_only else 3.0) * fwd_gflop * Bg / 1e3
    res = {"metric": "codec-tokens/s, " + ("LoRA-only fine-tuning step" if args.lora_only else "conf/vampnet.yml training step")
                     + " (coarse model)", "value": tokens / elapsed,
           "unit": "codec-tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": train_dtype, "data": "synthetic (random-init weights of the real architecture, random DAC tokens, r ~ U(0,1))",
           "config": {"workload": "BASELINE configs[4]: conf/vampnet.yml training step, coarse VampNet (20 layers, d=1280), "
                                  f"batch {Bg}/GPU x T=575 x 4 codebooks, dropout 0.1, label smoothing 0.1, clip 5.0, "
                                  "AdamW + Noam, fp32-grade arithmetic (amp: false): " +
                                  (("GEMMs" + ("" if os.environ.get("VN_TRAIN_ATTN_X3") == "0" else " and attention (forward + backward)") +
                                    " as six bf16-MFMA products of exact three-way operand splits") if train_dtype == "bf16x3"
                                   else "fp32-input MFMA"),
                      "global_batch": world * Bg, "parallelism": f"dp{world}" if world > 1 else "single GPU",
                      "step_tflop_per_gpu": step_tflop, "achieved_tflops_per_gpu": step_tflop / (elapsed / args.steps),
                      "final_loss": loss}}
    if prof is not None:
        over = None
        if prof_serial is not None:
            on_, oms, ofl, _ = prof["gemm"]
            over = {"note": "the timed region's own brackets: kernels of the caller's stream while the weight-gradient GEMMs run beside them on "
                            "the side stream (a bracket includes the time its kernel shared the chip); `roofline` itself = one extra untimed "
                            "step with the side stream off", "launches": int(on_), "avg_launch_us": 1e3 * oms / on_ if on_ else None,
                    "achieved": ofl / (oms * 1e-3) / 1e12 if oms else None}
            prof = prof_serial
        n, ms, fl, by = prof["gemm"]
        an, ams, afl, _ = prof["attention"]
        peak = PEAK_BF16_MFMA_TF / 6.0 if train_dtype == "bf16x3" else PEAK_F32_MFMA_TF
        el_b = serial_s if over is not None else elapsed           # wall time of the steps the brackets cover
        if over is not None:
            over["serial_step_ms"] = 1e3 * serial_s
        res["roofline"] = {"bound": "mfma", "kernel": "vn_gemm_x3_kernel" if train_dtype == "bf16x3" else "vn_gemm_f32[_sk]_kernel",
                           "achieved": fl / (ms * 1e-3) / 1e12 if ms else None,
                           "peak": peak, "unit": "TFLOP/s",
                           "peak_basis": "2500 TF dense bf16 MFMA / 6 plane products per fp32-grade product" if train_dtype == "bf16x3"
                                         else "fp32-input MFMA",
                           "frac": fl / (ms * 1e-3) / 1e12 / peak if ms else None, "traffic": None,
                           "algorithmic_bytes_per_launch": by / n if n else None, "launches": int(n),
                           "avg_launch_us": 1e3 * ms / n if n else None, "event_stride": args.event_stride,
                           "gemm_time_frac": args.event_stride * ms / (1e3 * el_b) if el_b else None,
                           "attention": {"launches": int(an), "avg_launch_us": 1e3 * ams / an if an else None,
                                         "achieved": afl / (ams * 1e-3) / 1e12 if ams else None,
                                         "time_frac": args.event_stride * ams / (1e3 * el_b)},
                           **({"overlapped": over} if over is not None else {})}
    if args.lora_only:
        res["config"]["workload"] += "; LoRA-only (r=8 adapters on w_qs, w_vs, fc, w_1, w_2; everything else frozen)"
    if not args.no_cpu_baseline and not args.lora_only:
        res["cpu_baseline"] = cpu_baseline_train()
    result_out.write(json.dumps(res) + "\n")
    result_out.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-per-gpu", type=int, default=8)
    ap.add_argument("--coarse-steps", type=int, default=12)
    ap.add_argument("--coarse-only", action="store_true", help="coarse_vamp only (4 codebooks)")
    ap.add_argument("--config", type=int, choices=[1, 2], default=2,
                    help="BASELINE.json configs index: 1 = coarse model only, batch 1, 12 sampling steps (= --coarse-only "
                         "--batch-per-gpu 1); 2 (default) = coarse + c2f vamp(), batch 8")
    ap.add_argument("--dtype", choices=["f32", "bf16x3", "f16x2", "bf16"], default="bf16x3",
                    help="bf16x3 (default) = fp32-grade GEMMs as six bf16-MFMA products of exact 3-way operand splits (8+8+8 significand "
                         "bits, fp32's exponent range), fp32 accumulate: held to the same parity bars as f32 (tests/test_gpu_bf16x3.py: "
                         "tokens bit-identical to the oracle and the reference's golden tokens); f32 = exact-fp32 MFMA; "
                         "f16x2 = OPT-IN fast mode, three fp16-MFMA products of two-plane operand splits (22 significand bits, fp16's "
                         "range: narrower than fp32; saturation ledger + bf16x3 fallback) — by default it is timed as the secondary "
                         "`alt` block; bf16 = fast mode, not bit-exact")
    ap.add_argument("--no-alt", action="store_true", help="skip the secondary f16x2 block (`alt`) that follows the primary timed region")
    ap.add_argument("--workload", choices=["vamp", "train"], default="vamp",
                    help="vamp = the headline inference path (default); train = BASELINE configs[4] training step")
    ap.add_argument("--lora-only", action="store_true",
                    help="train workload: LoRA-only fine-tuning step (train.py:696) instead of full training")
    ap.add_argument("--rng", choices=["device", "torch_device"], default="device",
                    help="device (default) = in-kernel Philox noise; torch_device = seed-exact parity mode: torch's CPU mt19937 stream "
                         "continued on the GPU (tokens identical to the reference's seeded run), timed at device speed")
    ap.add_argument("--e2e", action="store_true",
                    help="time the whole request encode -> build_mask -> vamp -> decode (DAC codec with seeded synthetic weights; "
                         "codec parity is UNPINNED) instead of vamp() alone, with per-stage ms")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true",
                    help="print the cpu_baseline object alone (no GPU needed): how profiles/r06_cpu_reference_in_container.json was made")
    ap.add_argument("--no-sharded-check", action="store_true",
                    help="N = 1 only: skip the extra timed region that runs the same steps through the sharded code path (a one-rank RCCL group)")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--event-stride", type=int, default=8,
                    help="bracket ~1 of every N MFMA launches with hipEvents (1 = all: +2.3 %% step time at B=8)")
    args = ap.parse_args()
    if args.config == 1:
        args.coarse_only, args.batch_per_gpu = True, 1
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(coarse_only=args.coarse_only)), flush=True)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, rendezvous on 127.0.0.1) and hand
        # the same arguments on; rank 0 of the children prints the JSON line
        import socket
        import subprocess
        ndev = torch.cuda.device_count()
        if ndev < args.gpus and os.environ.get("VN_BENCH_ONE_GPU") != "1":
            raise SystemExit(f"--gpus {args.gpus}: only {ndev} device(s) visible (VN_BENCH_ONE_GPU=1 runs the {args.gpus} ranks on "
                             "device 0 with a gloo exchange: exercises the sharded path, measures nothing about scaling)")
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    # ONE JSON line on stdout, whatever the libraries underneath print: RCCL writes a version banner to fd 1 when its first communicator
    # is made (rank 0 of every multi-GPU run, and the one-rank group of the sharded-path check).  From here on fd 1 IS stderr; the result
    # line goes to the saved descriptor.
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # debugging aid (single-GPU boxes): VN_BENCH_ONE_GPU=1 maps every rank to cuda:0 and uses gloo for the exchange
    one_gpu = os.environ.get("VN_BENCH_ONE_GPU") == "1"
    dev_index = 0 if one_gpu else local_rank
    torch.cuda.set_device(dev_index)
    device = f"cuda:{dev_index}"
    numa = _numa_bind(dev_index) if world > 1 and not one_gpu else None      # one process per GPU, each on its GPU's socket
    pg = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")

        def _init():
            if one_gpu:
                dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=300))
            else:
                dist.init_process_group("nccl", device_id=torch.device(device), timeout=datetime.timedelta(seconds=300))     # "nccl" is RCCL on ROCm
            # build the communicator (rings over xGMI) now, outside any timed region, whatever --warmup is
            _t = torch.ones(1, device=device) if not one_gpu else torch.ones(1)
            dist.all_reduce(_t)
            torch.cuda.synchronize()
            return int(_t.item())
        n_seen = _with_deadline(f"torch.distributed init + first all-reduce over {world} ranks", 240, _init, rank)
        pg = dist.group.WORLD
        assert dist.get_world_size() == args.gpus and n_seen == world, (dist.get_world_size(), n_seen, args.gpus)

    if args.workload == "train":
        def _barrier():
            torch.cuda.synchronize()
            if world > 1:
                import torch.distributed as dist
                dist.barrier()
            torch.cuda.synchronize()
        bench_train(args, rank, world, device, pg, _barrier, result_out)
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()
        return

    from vampnet_amd import synth as W         # synthetic weights / inputs (data generators only)
    from vampnet_amd.interface import Interface
    from vampnet_amd.synth import SynthCodec, model_kwargs

    cb = W.synth_codebooks()
    if args.e2e:
        if world > 1 or args.coarse_only:
            raise SystemExit("--e2e times the whole single-GPU request (encode -> build_mask -> vamp -> decode)")
        from vampnet_amd.codec import DacCodec
        codec = DacCodec(W.synth_dac_state_dict(W.DAC_DEFAULT_CFG, 0), W.DAC_DEFAULT_CFG, device=device)
    else:
        codec = SynthCodec(cb)
    csd_host, fsd_host = W.synth_state_dict(W.COARSE_DIMS, 0), W.synth_state_dict(W.C2F_DIMS, 1)
    torch.cuda.synchronize()
    t_setup = time.perf_counter()
    itf = Interface.from_state_dicts(codec, csd_host, model_kwargs(W.COARSE_DIMS), fsd_host, model_kwargs(W.C2F_DIMS), device=device,
                                     max_batch=args.batch_per_gpu, rng=args.rng, process_group=pg, precision=args.dtype)
    torch.cuda.synchronize()
    # what a hot-swap costs (the reference app reloads weights per request, app.py:181): pack + upload + plane build of both models
    setup_s = {args.dtype: round(time.perf_counter() - t_setup, 3)}
    itf.exchange_log = [] if world > 1 else None
    preflight = None
    if world > 1 and not one_gpu:
        try:                                 # what RCCL reports about itself; a failure here is recorded, the measurement goes on
            preflight = _rccl_preflight(itf, device, world, rank)
        except SystemExit:
            raise
        except Exception as e:
            preflight = {"rccl_nranks": None, "preflight_error": f"{type(e).__name__}: {e}"}
    B = args.batch_per_gpu * world
    codes = W.synth_codes(B, 14, 575, seed=2).to(device)
    torch.manual_seed(0)
    mask = itf.build_mask(codes)                # periodic_prompt=7, upper_codebook_mask=3 (hello.py / BASELINE cfg)
    kw = dict(batch_size=B, _sampling_steps=args.coarse_steps, typical_filtering=True)
    # noise: "device" = one Philox stream per call keyed by device_seed; "torch_device" = the reference's own seeding
    # (seed= -> torch.manual_seed, transformer.py:718-719), the mt19937 stream continued on the GPU
    seed_kw = (lambda sd: {"device_seed": sd}) if args.rng == "device" else (lambda sd: {"seed": sd})

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    if one_gpu and world > 1:       # gloo has no device all_gather_into_tensor: stage the token exchange through the host
        import torch.distributed as dist

        def _gather_via_host(z, _itf=itf):
            B_ = z.shape[0]
            per = -(-B_ // _itf.world)
            b0, b1 = _itf._shard(B_)
            local = torch.zeros((per,) + tuple(z.shape[1:]), dtype=z.dtype)
            local[:b1 - b0] = z[b0:b1].cpu()
            full = [torch.empty_like(local) for _ in range(_itf.world)]
            dist.all_gather(full, local)
            return torch.cat(full)[:B_].to(z.device)
        itf._allgather_batch = _gather_via_host

    stages = None
    if args.e2e:
        # the whole request of hello.py on resident inputs: 10 s of 44.1 kHz audio per item -> Interface._preprocess (BS.1770 loudness,
        # gain, peak limit, pad: csrc/preprocess.hip; only a resample of input at another rate would still run on the host) -> DAC
        # encode + RVQ -> build_mask -> vamp -> DAC decode
        from vampnet_amd.codec import AudioSignal
        audio = 0.1 * torch.randn(B, 1, 575 * codec.hop_length, device=device, generator=torch.Generator(device=device).manual_seed(7))

        def run(seed, sync=None):
            st = []

            def stage(name, fn):
                if sync is None:
                    return fn()
                torch.cuda.synchronize()
                t = time.perf_counter()
                o = fn()
                torch.cuda.synchronize()
                st.append((name, round(1e3 * (time.perf_counter() - t), 3)))
                return o
            sig = stage("preprocess", lambda: itf._preprocess(AudioSignal(audio, codec.sample_rate)))     # loudness -> gain -> peak limit -> pad, on the device
            c = stage("encode", lambda: codec.encode(sig.samples)["codes"])
            m = stage("build_mask", lambda: itf.build_mask(c))
            z = stage("vamp", lambda: itf.vamp(c, m, **seed_kw(seed), **kw))
            stage("decode", lambda: itf.decode(z))
            if sync is not None:
                sync.extend(st)
            return z
    elif args.coarse_only:
        kw.pop("batch_size")
        run = lambda seed: itf.coarse_vamp(codes, mask, **seed_kw(seed), **kw)
    else:
        run = lambda seed: itf.vamp(codes, mask, **seed_kw(seed), **kw)
    def timed_region():
        """W warm-up steps, then EXACTLY K steps between two barriers (+ device synchronisation); max over ranks"""
        for i in range(args.warmup):
            run(100 + i)
        barrier()
        if itf.exchange_log is not None:
            del itf.exchange_log[:]
        codec_eng = getattr(codec, "engine", None) if args.e2e else None
        if not args.no_kernel_events:
            itf.engine.profile_begin(4000 * max(args.steps, 1), stride=args.event_stride)
            if codec_eng is not None and codec_eng is not itf.engine:       # the codec's convolutions: every launch bracketed (few, long)
                codec_eng.profile_begin(600 * max(args.steps, 1), stride=1)
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = run(i)
        barrier()
        el = time.perf_counter() - t0
        pr = itf.engine.profile_end() if not args.no_kernel_events else None
        if pr is not None and codec_eng is not None and codec_eng is not itf.engine:
            pr["codec"] = codec_eng.profile_end()
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([el], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        assert tuple(out.shape) == (B, 14, 575), out.shape   # coarse_vamp also returns all 14 codebooks
        itf.engine.health_check()
        ex_ms = None
        if itf.exchange_log:                                  # the one RCCL all-gather of every vamp() call, event-timed
            ex_ms = sum(a.elapsed_time(b) for a, b in itf.exchange_log) / len(itf.exchange_log)
        return el, pr, ex_ms

    elapsed, prof, exchange_ms = timed_region()
    # ---- N = 1: the SAME steps once more through the sharded code path (a one-rank RCCL group: _shard, the padded local block, the device
    # all-gather through the exchange in use) — the code every rank of an N-GPU run executes; its rate must be the plain line's (1 %)
    sharded_n1 = None
    if world == 1 and not args.e2e and not args.no_sharded_check and os.environ.get("VN_BENCH_SHARDED_CHECK", "1") != "0":
        try:
            import datetime
            import socket
            import torch.distributed as dist
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port1 = sk.getsockname()[1]; sk.close()

            def _init1():
                dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port1}", rank=0, world_size=1,
                                        device_id=torch.device(device), timeout=datetime.timedelta(seconds=120))
                dist.all_reduce(torch.ones(1, device=device))
                torch.cuda.synchronize()
            _with_deadline("one-rank RCCL group for the sharded-path check", 150, _init1)
            itf.pg, itf.rank, itf.world = dist.group.WORLD, 0, 1
            pf = _rccl_preflight(itf, device, 1, 0)
            save_ev, args.no_kernel_events = args.no_kernel_events, True
            itf.exchange_log = []
            s_el, _, s_ex = timed_region()
            args.no_kernel_events = save_ev
            itf.pg, itf.exchange_log = None, None
            dist.destroy_process_group()
            ratio = s_el / elapsed
            sharded_n1 = {"ms_per_step": 1e3 * s_el / args.steps, "ratio_vs_plain": ratio, "within_1pct": abs(ratio - 1.0) <= 0.01,
                          "exchange_ms": s_ex, **pf,
                          "note": "the timed region again with a one-rank RCCL process group attached (kernel event brackets off: they are "
                                  "on in the plain region and cost it ~0.3 %)"}
            if not sharded_n1["within_1pct"]:
                sys.stderr.write(f"bench.py: the sharded code path at N = 1 runs at {ratio:.4f} x the plain step time (> 1 % apart)\n")
        except Exception as e:                                # a box without a usable RCCL must still produce the N = 1 line
            sharded_n1 = {"error": f"{type(e).__name__}: {e}"}
    # ---- secondary block: the opt-in fast precision on the SAME inputs, timed after (never inside) the primary region
    alt = None
    if args.dtype in ("bf16x3", "f32") and not args.no_alt and not args.e2e:
        import warnings
        from vampnet_amd.engine import PrecisionFallbackWarning
        with warnings.catch_warnings(record=True) as wl:
            warnings.simplefilter("always")
            torch.cuda.synchronize()
            t_sw = time.perf_counter()
            for m_ in (itf.coarse, itf.c2f):
                m_.set_precision("f16x2")
            torch.cuda.synchronize()
            setup_s["f16x2 (switch of both resident models: plane build + probe forward)"] = round(time.perf_counter() - t_sw, 3)
            a_el, a_prof, _ = timed_region()
        alt = {"elapsed": a_el, "prof": a_prof, "effective": itf.effective_precision,
               "fallbacks": sum(1 for w_ in wl if issubclass(w_.category, PrecisionFallbackWarning))}
    if args.e2e:                                          # one more, untimed pass with a device sync around every stage
        stages = []
        run(args.steps, sync=stages)
    if rank == 0 and args.dtype in ("bf16x3", "f32", "f16x2") and not args.e2e:
        # what the other kinds of hot swap cost (the reference app may swap per request, app.py:181), after everything that is timed:
        # (1) adapters only, on the resident models: device merge of rank-8 adapters on the five LoRA'd linears of every layer + planes;
        # (2) back to a model that is still resident: a dictionary lookup
        from vampnet_amd.engine import LORA_KEYS
        gl = torch.Generator().manual_seed(5)

        def adapters(dims):
            sd, D = {}, dims["d_model"]
            shp = {"self_attn.w_qs": (D, D), "self_attn.w_vs": (D, D), "self_attn.fc": (D, D), "feed_forward.w_1": (4 * D, D),
                   "feed_forward.w_2": (D, 2 * D)}
            for l in range(dims["n_layers"]):
                for k in LORA_KEYS:
                    o, i = shp[k]
                    sd[f"transformer.layers.{l}.{k}.lora_A"] = torch.randn(8, i, generator=gl) * 0.01
                    sd[f"transformer.layers.{l}.{k}.lora_B"] = torch.randn(o, 8, generator=gl) * 0.01
            return sd
        la, lb = adapters(W.COARSE_DIMS), adapters(W.C2F_DIMS)
        for m_ in (itf.coarse, itf.c2f):                           # (the `alt` block left the models in f16x2: time the swap at the headline precision)
            if m_.precision != args.dtype:
                m_.set_precision(args.dtype)
        itf.coarse.apply_lora(la)                                  # first use: snapshots the un-merged blob (one-off, 1.1-1.3 GB clone each)
        itf.c2f.apply_lora(lb)
        torch.cuda.synchronize()
        t_l = time.perf_counter()
        itf.coarse.apply_lora(la)
        itf.c2f.apply_lora(lb)
        torch.cuda.synchronize()
        setup_s["adapter swap on the resident models (both: pack 8-rank adapters, upload, device merge, planes)"] = round(time.perf_counter() - t_l, 4)
        itf.coarse.apply_lora(None)
        itf.c2f.apply_lora(None)
        itf._resident_put("coarse", "resident://a", itf.coarse)
        t_c = time.perf_counter()
        hit = itf._resident_get("coarse", "resident://a")
        setup_s["swap to a model that is still resident (LRU lookup)"] = round(time.perf_counter() - t_c, 6)
        assert hit is itf.coarse

    if rank == 0:
        tokens = B * (4 * 575 if args.coarse_only else TOKENS_PER_CLIP) * args.steps
        rng_txt = {"device": "device RNG (Philox)",
                   "torch_device": "seed-exact parity mode: torch's mt19937 stream continued on the GPU (rng=torch_device)"}[args.rng]
        res = {
            "metric": "codec-tokens/s, coarse vamp (4 codebooks), 10 s clips" if args.coarse_only else
                      ("codec-tokens/s, whole request: _preprocess + DAC encode + build_mask + coarse+c2f vamp() + DAC decode, 10 s clips" if args.e2e
                       else "codec-tokens/s, coarse+c2f vamp(), 10 s clips"), "value": tokens / elapsed,
            "unit": "codec-tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic (random-init weights of the real architecture, "
            "random codes, periodic-7 prompt mask)",
            "config": {"workload": (f"BASELINE configs[1]: Interface.coarse_vamp(), coarse model only ({args.coarse_steps} sampling "
                                    f"steps), batch {args.batch_per_gpu}/GPU x 10 s @ 44.1 kHz (T=575, 4 codebooks), device RNG"
                                    if args.coarse_only else
                                    f"BASELINE configs[2]: Interface.vamp() coarse({args.coarse_steps} steps)+c2f(4x2 steps), "
                                    f"batch {args.batch_per_gpu}/GPU x 10 s @ 44.1 kHz (T=575, 14 codebooks), "
                                    "typical_filtering=True, " + rng_txt),
                       "codec": ("INSIDE the timed region (--e2e): DAC encode + RVQ and DAC decode on seeded synthetic weights of the "
                                 "published 44.1 kHz configuration; codec_parity = unpinned (lac sources and weights absent upstream, "
                                 "HIP == oracle/dac_oracle.py only)") if args.e2e else
                                "outside the timed region (tokens in, tokens out); DAC encode/decode parity is UNPINNED "
                                "(lac sources and weights absent) and no codec figure is part of this line",
                       **({"codec_parity": "unpinned", "stages_ms": dict(stages), "stages_note": "one extra pass with a device "
                           "synchronise around every stage (the timed passes have none); Interface._preprocess runs on the device "
                           "inside the timed region (its parity with audiotools is unpinned like the codec's)"} if args.e2e else {}),
                       "rng": args.rng,
                       "global_batch": B, "coarse_steps": args.coarse_steps,
                       "parallelism": f"batch-shard x{world}" if world > 1 else "single GPU",
                       **({"ranks": world, "devices": 1 if one_gpu else world,
                           "exchange": "gloo all_gather staged through the host, every rank on device 0 (VN_BENCH_ONE_GPU=1: exercises "
                                       "the sharded path on a one-GPU box, says nothing about scaling)" if one_gpu else
                                       "RCCL all_gather_into_tensor of the (B,14,T) tokens, one rank per GPU"} if world > 1 else {}),
                       "s_per_clip": elapsed / args.steps / B * world,
                       "precision": {"f32": "fp32 operands on the fp32-input MFMA, fp32 accumulate",
                                     "bf16x3": "fp32-grade: each GEMM operand = 3 exact bf16 split planes (sum == fp32 value), 6 bf16-MFMA "
                                               "products per k-step, fp32 accumulate — the GEMMs (gemm_x3.hip) AND both attention products "
                                               "(attention_x3.hip; P split in registers); norms / softmax / sampling fp32; "
                                               "same parity bars as f32 (tests/test_gpu_bf16x3.py)",
                                     "f16x2": "opt-in fast mode, operands NARROWER than fp32: each GEMM operand = 2 fp16 planes (h0 = fp16(x), h1 = "
                                              "fp16((x - h0) 2^11): x to 2^-22), 3 fp16-MFMA products per k-step into two fp32 accumulators "
                                              "(gemm_x3.hip); attention on fp16 two-plane operands as well (q / 8, k, 16 v, second plane unscaled; "
                                              "attention_x3.hip NP = 2); norms / softmax / sampling fp32; values outside fp16's range are recorded on "
                                              "the saturation ledger and the call is repeated on bf16x3 (tests/test_gpu_saturation.py)",
                                     "bf16": "bf16 GEMM/attention operands (fast mode, not bit-exact)"}[args.dtype]},
        }
        def roofline_of(dtype, prof, elapsed):
            n, ms, fl, gbytes = prof["gemm_bf16"] if dtype == "bf16" else prof["gemm"]
            an, ams, afl, _ = prof["attention"]
            # fabric bytes per launch: NOT measured by this run — read from the committed rocprofv3 --pmc passes of this same command
            traffic = traffic_source = None
            tname = next((n for n in {"f32": ["history/r01_traffic.json"], "bf16x3": ["r06_traffic_x3.json", "r05_traffic_x3.json"],
                                      "f16x2": ["r06_traffic_h2.json", "r05_traffic_h2.json"]}.get(dtype, [])
                          if os.path.exists(os.path.join(ROOT, "profiles", n))), "-")
            tpath = os.path.join(ROOT, "profiles", tname)
            if os.path.exists(tpath) and world == 1 and args.batch_per_gpu == 8 and not args.coarse_only and not args.e2e:
                tj = json.load(open(tpath))
                sha_now, sha_then = _gemm_source_sha(), tj.get("gemm_x3_sha256_16")
                if sha_then is not None and sha_then == sha_now:
                    traffic = tj["bytes_per_launch"]
                    traffic_source = (f"profiles/{tname} (separate --pmc FETCH_SIZE / WRITE_SIZE passes of this command on gemm_x3.hip "
                                      f"sha256 {sha_then}: the kernel source of THIS build; a committed capture, not this run)")
                else:       # the capture belongs to another revision of the kernel: do not quote it for this one
                    traffic_source = (f"null: profiles/{tname} was captured on gemm_x3.hip sha256 {sha_then}, this build is {sha_now} "
                                      "(re-run scripts/gpu_pmc_x3.sh)")
            # bf16x3: `achieved` stays ALGORITHMIC (2MNK per GEMM, fp32-equivalent); every such flop costs six bf16-MFMA flops,
            # so the ceiling of this algorithm is the dense bf16 MFMA peak / 6 (the kernel executes 6 x achieved on the pipe)
            # f16x2: three fp16-MFMA flops per algorithmic flop: ceiling = dense fp16 MFMA peak (= the bf16 one) / 3
            nprod = {"bf16x3": 6.0, "f16x2": 3.0}.get(dtype)
            peak = {"f32": PEAK_F32_MFMA_TF, "bf16": PEAK_BF16_MFMA_TF, "bf16x3": PEAK_BF16_MFMA_TF / 6.0, "f16x2": PEAK_BF16_MFMA_TF / 3.0}[dtype]
            kname = {"f32": "vn_gemm_f32[_sk]_kernel", "bf16": "vn_gemm_f32_kernel<128,128,BF16>",
                     "bf16x3": "vn_gemm_x3_kernel", "f16x2": "vn_gemm_x3_kernel<.., FMT = 1>"}[dtype]
            return {"bound": "mfma", "kernel": kname, "achieved": fl / (ms * 1e-3) / 1e12 if ms else None,
                    "peak": peak, "unit": "TFLOP/s",
                    "frac": (fl / (ms * 1e-3) / 1e12) / peak if ms else None,
                    "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes_per_launch": gbytes / n if n else None, "launches": int(n), "avg_launch_us": 1e3 * ms / n if n else None,
                    **({"peak_basis": f"2500 TF dense bf16 / fp16 MFMA / {int(nprod)} plane products per fp32-grade product",
                        "executed_mfma_tflops": nprod * fl / (ms * 1e-3) / 1e12 if ms else None,
                        # context: the same algorithmic rate against the bf16x3 formulation's ceiling (2500 / 6)
                        "achieved_over_bf16x3_ceiling": fl / (ms * 1e-3) / 1e12 / (PEAK_BF16_MFMA_TF / 6.0) if ms else None,
                        # context: the same algorithmic rate against the fp32-input MFMA peak (157.3 TF), the
                        # ceiling of the exact-fp32 kernel this precision replaces
                        "achieved_over_f32_mfma_peak": fl / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TF if ms else None,
                        # context, NOT `frac`: a pure matrix-pipe loop with no data movement sustains 1.79 PF on the real split planes of
                        # Gaussian operands (1.69-1.74 PF on uniformly random words, 2.42 PF on a smooth ramp: the firmware holds the socket
                        # at ~1.3 kW by lowering the clock) — a committed measurement (profiles/r05_mfma_power_probe.txt), not of this run
                        "executed_over_measured_pipe_limit": (nprod * fl / (ms * 1e-3) / 1e12) / MEASURED_PIPE_LIMIT_TF if ms else None,
                        "measured_pipe_limit": {"tflops": MEASURED_PIPE_LIMIT_TF, "source": "profiles/r05_mfma_power_probe.txt mode 4: v_mfma_f32_32x32x16_bf16 on "
                                                "register-resident split planes of Gaussian activations x weights, six-product order, no global / LDS "
                                                "traffic, socket at ~1.26 kW"}}
                       if nprod else {}),
                    "event_stride": args.event_stride,        # launches / times above: the bracketed sample
                    "gemm_time_frac": args.event_stride * ms / (1e3 * elapsed) if elapsed else None,
                    "attention": {"launches": int(an), "avg_launch_us": 1e3 * ams / an if an else None,
                                  "achieved": afl / (ams * 1e-3) / 1e12 if ams else None}}

        if prof is not None:
            res["roofline"] = roofline_of(args.dtype, prof, elapsed)
            if prof.get("codec") and prof["codec"]["conv1d"][0]:
                # the codec's convolutions (both directions; hipEvents around EVERY launch of the timed region), booked by the roofline that
                # BOUNDS each layer (csrc/vn_common.h vn_conv_class: arithmetic intensity of the layer's own operand bytes against the ridge
                # of the pipe it runs on): the matrix-pipe-bound layers on the split-plane pipe (ceiling 2500 / 6 TF-eq), those on the
                # fp32-input MFMA (157.3 TF), and the byte-bound layers (audio-rate 64-channel layers, 2-tap phases, k = 1 tails; ceiling 8 TB/s).
                # Algorithmic flops = 2 x MACs of each convolution as launched; algorithmic bytes = every operand read once, every result
                # written once.  Parity of every one of these kernels is unpinned (lac absent).
                def grp(key, bound, peak, unit):
                    n_, ms_, fl_, by_ = prof["codec"][key]
                    if not n_:
                        return None
                    ach = (fl_ / (ms_ * 1e-3) / 1e12) if unit == "TFLOP/s" else (by_ / (ms_ * 1e-3) / 1e9)
                    return {"bound": bound, "launches": int(n_), "ms_per_step": ms_ / args.steps, "achieved": ach, "peak": peak, "unit": unit,
                            "frac": ach / peak, "algorithmic_tflop_per_step": fl_ / args.steps / 1e12,
                            "algorithmic_gbytes_per_step": by_ / args.steps / 1e9}
                cn, cms, cfl, cby = prof["codec"]["conv1d"]
                res["codec_roofline"] = {
                    "groups": {"mfma_split_plane_pipe": grp("conv_x3", "mfma", PEAK_BF16_MFMA_TF / 6.0, "TFLOP/s"),
                               "mfma_fp32_input": grp("conv_f32", "mfma", PEAK_F32_MFMA_TF, "TFLOP/s"),
                               "hbm": grp("conv_hbm", "hbm", 8000.0, "GB/s")},
                    "kernel": "vn_gemm_x3_kernel<CONV / CONVT> + vn_conv1d_f32_kernel", "launches": int(cn), "ms_per_step": cms / args.steps,
                    "algorithmic_tflop_per_step": cfl / args.steps / 1e12, "algorithmic_gbytes_per_step": cby / args.steps / 1e9,
                    "codec_parity": "unpinned"}
        res["setup_s"] = setup_s
        res["devices"] = {"world_size": world, "device_count": torch.cuda.device_count(),
                          "exchange_ms": exchange_ms,            # event-timed Interface._allgather_batch, mean per vamp() (None at N = 1)
                          "exchange": (("the library's RCCL communicator (vn_allgather_tokens)" if itf.exchange == "c_abi" else
                                        "RCCL all_gather_into_tensor (torch.distributed 'nccl')") + " of the (B,14,T) int64 tokens")
                                      if world > 1 and not one_gpu else None,
                          # what RCCL ITSELF reports (ncclCommCount) + how many distinct ranks one all-gather saw through each exchange:
                          # a record with rccl_nranks == n_gpus shows the collective library ran with N ranks, not just N processes
                          **(preflight or {}), "numa_binding": numa, "sharded_path_at_n1": sharded_n1}
        if alt is not None:
            a_el = alt["elapsed"]
            res["alt"] = {"dtype": "f16x2", "value": tokens / a_el, "unit": "codec-tokens/s", "ms_per_step": 1e3 * a_el / args.steps,
                          "steps": args.steps, "warmup": args.warmup,
                          "note": "OPT-IN fast mode, timed after the primary region on the same inputs; operands are two fp16 planes "
                                  "(22 significand bits, fp16's exponent range): NARROWER than the reference's fp32, so this is never "
                                  "the headline.  Guard: every fp16 plane writer records a clamped value on the saturation ledger, read "
                                  "after each generate(); a call that saturated is repeated on bf16x3 (tests/test_gpu_saturation.py)",
                          "effective_precision": alt["effective"], "fallbacks": alt["fallbacks"],
                          **({"roofline": roofline_of("f16x2", alt["prof"], a_el)} if alt["prof"] is not None else {})}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(coarse_only=args.coarse_only)
        result_out.write(json.dumps(res) + "\n")
        result_out.flush()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
