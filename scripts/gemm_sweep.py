"""GPU micro-benchmark of vn_gemm_f32 over the model's GEMM shapes x tile configs x tile orders (tuning aid).
usage (gpurun): python scripts/gemm_sweep.py > gpurun_out/gemm_sweep.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampnet_amd import _lib
from vampnet_amd.engine import Engine

eng = Engine("cuda:0")
lib = eng.lib
D = 1280
SHAPES = [  # name, M, N, K, epilogue
    ("qkv   B8", 4600, 3 * D, D, _lib.EPI_STORE), ("wo    B8", 4600, D, D, _lib.EPI_RESIDUAL),
    ("w1geg B8", 4600, 4 * D, D, _lib.EPI_GEGLU), ("w2    B8", 4600, D, 2 * D, _lib.EPI_RESIDUAL),
    ("cls   B8", 4600, 4096, D, _lib.EPI_BIAS),
    ("qkv  c2f", 1384, 3 * D, D, _lib.EPI_STORE), ("wo   c2f", 1384, D, D, _lib.EPI_RESIDUAL),
    ("w1g  c2f", 1384, 4 * D, D, _lib.EPI_GEGLU), ("w2   c2f", 1384, D, 2 * D, _lib.EPI_RESIDUAL),
    ("cls  c2f", 1384, 10240, D, _lib.EPI_BIAS),
    ("qkv   B1", 575, 3 * D, D, _lib.EPI_STORE), ("wo    B1", 575, D, D, _lib.EPI_RESIDUAL),
    ("w1geg B1", 575, 4 * D, D, _lib.EPI_GEGLU), ("w2    B1", 575, D, 2 * D, _lib.EPI_RESIDUAL),
    ("cls   B1", 575, 4096, D, _lib.EPI_BIAS),
    ("sq 4096 ", 4096, 4096, 4096, _lib.EPI_STORE),
]
TILES = [(0, 0), (128, 128), (64, 128), (64, 64)]


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us



def cfg(bm, bn, walk, sched):        # sched: -1 auto, 0 data-parallel, 1 stream-K
    lib.vn_debug_gemm_config(eng.handle, bm, bn, walk | ((sched + 1) << 1))


# long warm-up so clocks settle before the first measurement
_w = torch.randn(4096, 4096, device="cuda")
for _ in range(30):
    eng.gemm(_w, _w)
torch.cuda.synchronize()

print(f"{'shape':10s} {'tile':>8s} {'DP us':>9s} {'TF':>6s} {'SK us':>9s} {'TF':>6s}  {'maxerr DP/SK vs f64':>22s} sk-deterministic")
for name, M, N, K, epi in SHAPES:
    A = torch.randn(M, K, device="cuda")
    Wt = torch.randn(N, K, device="cuda") / K ** 0.5
    bias = torch.randn(N, device="cuda")
    Nout = N // 2 if epi == _lib.EPI_GEGLU else N
    flops = 2.0 * M * N * K
    ref = None
    if M * N <= 4600 * 5120 and epi in (_lib.EPI_STORE, _lib.EPI_BIAS, _lib.EPI_RESIDUAL):
        ref = A.double() @ Wt.double().t()
        if epi == _lib.EPI_BIAS:
            ref = ref + bias.double()
    for bm, bn in TILES:
        if epi == _lib.EPI_GEGLU and bn == 64:
            continue
        cells, errs, outs = [], [], []
        for sched in (0, 1):
            if bm == 0 and sched == 1:
                sched = -1          # row "0x0": DP-auto vs full auto (may pick stream-K)
            cfg(bm, bn, 1, sched)
            out = torch.zeros(M, Nout, device="cuda")
            run = lambda: eng.gemm(A, Wt, bias=bias if epi == _lib.EPI_BIAS else None, epilogue=epi, out=out)
            if ref is not None:
                out.zero_()
                run()
                errs.append(f"{(out.double() - ref).abs().max().item():.2e}")
                o1 = out.clone()
                out.zero_()
                run()
                outs.append(bool(torch.equal(o1, out)))
            us = bench(run)
            cells.append(f"{us:9.1f} {flops / us / 1e6:6.1f}")
        print(f"{name:10s} {bm:>4d}x{bn:<3d} " + " ".join(cells) + "  " + "/".join(errs) + f"  {outs[-1] if outs else ''}", flush=True)
cfg(0, 0, 1, -1)
