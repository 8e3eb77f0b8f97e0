"""bf16x3 GEMM ablations at every tile height on the B = 8 shapes (store epilogue, no split): abl 0 = shipped, 1 = no DMA in the
k-loop, 2 = no fragment reads, 3 = MFMA + barriers only, 4 = DMA of whole 128-byte lines (same volume).  Results of abl != 0 are invalid."""
import sys
import torch
sys.path.insert(0, ".")
from vampnet_amd import _lib
from vampnet_amd.engine import Engine
eng = Engine("cuda:0")
def timeit(fn, n=15):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
w = torch.randn(4096, 4096, device="cuda")
for _ in range(20): eng.gemm(w, w)
for name, M, N, K in [("qkv B8", 4600, 3840, 1280), ("wo B8", 4600, 1280, 1280), ("w2 B8", 4600, 1280, 2560), ("w1 B8 (store)", 4600, 5120, 1280), ("sq 4096", 4096, 4096, 4096)]:
    g = torch.Generator(device="cuda").manual_seed(0)
    a3 = eng.split3(torch.randn(M, K, device="cuda", generator=g)); w3 = eng.split3(torch.randn(N, K, device="cuda", generator=g) / K ** 0.5)
    out = torch.zeros(M, N, device="cuda")
    for bm in (128, 192, 256):
        row = []
        for abl in (0, 1, 2, 3, 4):
            eng.lib.vn_debug_x3_config(eng.handle, bm, 1, abl if abl else -1)
            row.append(timeit(lambda: eng.gemm_bf16x3(a3, w3, out=out)))
        eng.lib.vn_debug_x3_config(eng.handle, 0, -1, -1)
        fl = 2.0 * M * N * K
        print(f"{name:14s} bm {bm}: " + "  ".join(f"abl{i} {us:6.1f} us ({fl / us / 1e6:5.1f} TF-eq)" for i, us in enumerate(row)) + f"   full-line gain {100 * (row[0] / row[4] - 1):+.1f} %", flush=True)
