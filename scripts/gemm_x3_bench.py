"""Time the exact-fp32 MFMA GEMM against the bf16x3 GEMM on the model's shapes (fp32-equivalent TFLOP/s = 2MNK / t)."""
import sys

import torch

sys.path.insert(0, ".")
from vampnet_amd import _lib
from vampnet_amd.engine import Engine

eng = Engine("cuda:0")
SHAPES = [("qkv B8", 4600, 3840, 1280, _lib.EPI_STORE), ("wo  B8", 4600, 1280, 1280, _lib.EPI_RESIDUAL),
          ("w1g B8", 4600, 5120, 1280, _lib.EPI_GEGLU), ("w2  B8", 4600, 1280, 2560, _lib.EPI_RESIDUAL),
          ("cls B8", 4600, 4096, 1280, _lib.EPI_STORE), ("qkv c2f", 1384, 3840, 1280, _lib.EPI_STORE),
          ("w2  c2f", 1384, 1280, 2560, _lib.EPI_RESIDUAL), ("qkv B1", 575, 3840, 1280, _lib.EPI_STORE),
          ("w1g B1", 575, 5120, 1280, _lib.EPI_GEGLU), ("sq 4096", 4096, 4096, 4096, _lib.EPI_STORE)]


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


print(f"{'shape':10s} {'f32 us':>9s} {'TF':>7s} {'x3 us':>9s} {'TF':>7s} {'speedup':>8s}")
for name, M, N, K, epi in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    a3, w3 = eng.split3(a), eng.split3(w)
    out = torch.zeros(M, N // 2 if epi == _lib.EPI_GEGLU else N, device="cuda")
    t32 = timeit(lambda: eng.gemm(a, w, epilogue=epi, out=out))
    tx3 = timeit(lambda: eng.gemm_bf16x3(a3, w3, epilogue=epi, out=out))
    fl = 2.0 * M * N * K
    print(f"{name:10s} {t32 * 1e6:9.1f} {fl / t32 / 1e12:7.1f} {tx3 * 1e6:9.1f} {fl / tx3 / 1e12:7.1f} {t32 / tx3:8.2f}x")
