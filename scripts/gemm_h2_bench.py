"""STAGED: time the f16x2 GEMM (tile 128 / 256, auto split) against bf16x3 and the fp32-MFMA GEMM on the model's shapes, and
report its error against float64 next to theirs.  fp32-equivalent TFLOP/s = 2MNK / t."""
import sys

import torch

sys.path.insert(0, ".")
from vampnet_amd import _lib
from vampnet_amd.engine import Engine

eng = Engine("cuda:0")
SHAPES = [("qkv B8", 4600, 3840, 1280, 1), ("wo  B8", 4600, 1280, 1280, 2), ("w1  B8", 4600, 5120, 1280, 1),
          ("w2  B8", 4600, 1280, 2560, 2), ("cls B8", 4600, 4096, 1280, 1), ("qkv c2f", 1384, 3840, 1280, 1),
          ("w2  c2f", 1384, 1280, 2560, 4), ("qkv B1", 575, 3840, 1280, 1), ("sq 4096", 4096, 4096, 4096, 1)]


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


print(f"{'shape':9s} {'f32 TF':>7s} {'x3 TF':>7s} {'h2/128':>7s} {'h2/256':>7s} | max err vs f64: f32 / x3 / h2")
for name, M, N, K, ns in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    a3, w3 = eng.split3(a), eng.split3(w)
    a2, w2 = eng.split2h(a), eng.split2h(w)
    out = torch.zeros(M, N, device="cuda")
    fl = 2.0 * M * N * K
    t32 = timeit(lambda: eng.gemm(a, w, out=out))
    e32 = out.clone()
    tx3 = timeit(lambda: eng.gemm_bf16x3(a3, w3, out=out))
    ex3 = out.clone()
    th1 = timeit(lambda: eng.gemm_f16x2(a2, w2, out=out, tile_m=128, nsplit=ns))
    eh = out.clone()
    th2 = timeit(lambda: eng.gemm_f16x2(a2, w2, out=out, tile_m=256, nsplit=ns))
    errs = ""
    if M * N <= 20_000_000:
        ref = a.double() @ w.double().t()
        errs = " / ".join(f"{(x.double() - ref).abs().max().item():.2e}" for x in (e32, ex3, eh))
    print(f"{name:9s} {fl / t32 / 1e12:7.1f} {fl / tx3 / 1e12:7.1f} {fl / th1 / 1e12:7.1f} {fl / th2 / 1e12:7.1f} | {errs}")
