"""A/B for the claim "a launch that leaves CUs idle runs its k-tiles faster (higher clock)": the SAME bf16x3 kernel (128-row tiles, no
k-split, K = 1280 = 40 k-tiles, store epilogue) on launches of 30 .. 300 tiles.  Up to 256 tiles every tile has a CU of its own, so
the launch lasts one tile's k-loop whatever the count — unless the clock (power) depends on how many CUs work.  Also the time of
back-to-back launches vs single launches (launch overhead inside the number).  Output -> profiles/r05_gemm_underfill_ab.txt"""
import sys

import torch

sys.path.insert(0, ".")
from vampnet_amd import _lib
from vampnet_amd.engine import Engine

eng = Engine("cuda:0")


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


w = torch.randn(4096, 4096, device="cuda")
for _ in range(20):
    eng.gemm(w, w)
K = 1280
print("bf16x3 GEMM, 128 x 128 tiles, K = 1280 (40 k-tiles), store epilogue, tiled operands; one block per tile")
print(f"{'N':>5s} {'M':>5s} {'tiles':>5s} {'us/launch':>10s} {'us/k-tile':>10s} {'rel. to 240 tiles':>18s}")
for N in (3840, 5120):
    g = torch.Generator(device="cuda").manual_seed(0)
    w32 = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    w3 = eng.tile3(eng.split3(w32))
    res = []
    for r in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10):
        M = 128 * r
        tiles = r * (N // 128)
        if tiles > 320:
            continue
        a3 = eng.tile3(eng.split3(torch.randn(M, K, device="cuda", generator=g)))
        out = torch.zeros(M, N, device="cuda")
        eng.lib.vn_debug_x3_config(eng.handle, 128, 1, -1)
        us = timeit(lambda: eng.gemm_bf16x3(a3, w3, epilogue=_lib.EPI_STORE, out=out, tiled_shape=(M, N, K)))
        res.append((M, tiles, us))
    eng.lib.vn_debug_x3_config(eng.handle, 0, -1, -1)
    ref = min((abs(t - 240), us) for _, t, us in res)[1]
    for M, tiles, us in res:
        print(f"{N:5d} {M:5d} {tiles:5d} {us:10.1f} {us / 40:10.3f} {us / ref:18.3f}", flush=True)
