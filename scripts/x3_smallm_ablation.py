"""Where does the k-loop time of a ONE-ROUND bf16x3 launch go?  128-row tile (CFG 1) at M = 575 (and smaller: fewer busy CUs), K = 1280, with the
kernel's ablation switches: 1 = no LDS-DMA inside the k-loop, 2 = no fragment reads, 3 = both (results invalid, timing only)."""
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
for N in (3840, 5120):
    g = torch.Generator(device="cuda").manual_seed(0)
    w3 = eng.tile3(eng.split3(torch.randn(N, K, device="cuda", generator=g) / K ** 0.5))
    for M in (128, 575, 1024):
        a3 = eng.tile3(eng.split3(torch.randn(M, K, device="cuda", generator=g)))
        out = torch.zeros(M, N, device="cuda")
        fn = lambda: eng.gemm_bf16x3(a3, w3, epilogue=_lib.EPI_STORE, out=out, tiled_shape=(M, N, K))
        res = []
        for abl in (0, 1, 2, 3):
            eng.lib.vn_debug_x3_config(eng.handle, 128, 1, abl)
            res.append(timeit(fn))
        eng.lib.vn_debug_x3_config(eng.handle, 0, -1, -1)
        tiles = -(-M // 128) * (N // 128)
        print(f"N={N} M={M:5d} ({tiles:3d} tiles): full {res[0]:6.1f} us | no DMA {res[1]:6.1f} | no fragment reads {res[2]:6.1f} | neither {res[3]:6.1f}"
              f"   (40 k-tiles; MFMA-only bound at 2.4 GHz: 25.6 us)", flush=True)
