"""f16x2 GEMM (gemm_x3.hip, FMT = 1), store epilogue, tiled operands: what a k-tile's time is made of — ablations of the shipped
kernel (results invalid): 1 = no DMA inside the k-loop, 2 = no fragment reads inside the k-loop, 3 = MFMAs + barriers only, 4 (f16x2) = DMA issued but never waited for."""
import sys

import torch

sys.path.insert(0, ".")
from vampnet_amd.engine import Engine

eng = Engine("cuda:0")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3


for name, M, N, K in [("qkv B8", 4600, 3840, 1280), ("sq 4096", 4096, 4096, 4096)]:
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    a2, w2 = eng.split2h(a, tiled=True), eng.split2h(w, tiled=True)
    a3, w3 = eng.tile3(eng.split3(a)), eng.tile3(eng.split3(w))
    out = torch.zeros(M, N, device="cuda")
    fl = 2.0 * M * N * K
    for bm in (128, 192, 256):
        line = f"{name:8s} bm {bm}:"
        for fmt, fn in (("h2", lambda: eng.gemm_f16x2(a2, w2, out=out, tiled_shape=(M, N, K))),
                        ("x3", lambda: eng.gemm_bf16x3(a3, w3, out=out, tiled_shape=(M, N, K)))):
            for abl in (0, 1, 2, 3, 4):
                if abl == 4 and fmt == "x3":
                    continue
                eng.check(eng.lib.vn_debug_x3_config(eng.handle, bm, 1, abl), "cfg")
                t = timeit(fn)
                line += f"  {fmt} abl{abl} {t * 1e6:6.1f} us ({fl / t / 1e12:5.1f})"
            line += " |"
        print(line, flush=True)
eng.check(eng.lib.vn_debug_x3_config(eng.handle, 0, -1, -1), "cfg")
