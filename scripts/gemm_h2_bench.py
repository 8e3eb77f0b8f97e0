"""bf16x3 (six products) vs f16x2 (three products) of gemm_x3.hip on the model's shapes, tiled operands, every tile height:
kernel time over 20 back-to-back launches and fp32-equivalent TFLOP/s = 2MNK / t."""
import sys

import torch

sys.path.insert(0, ".")
from vampnet_amd import _lib
from vampnet_amd.engine import Engine

eng = Engine("cuda:0")
SHAPES = [("qkv B8", 4600, 3840, 1280, _lib.EPI_STORE), ("wo  B8", 4600, 1280, 1280, _lib.EPI_RESIDUAL),
          ("w1g B8", 4600, 5120, 1280, _lib.EPI_GEGLU), ("w2  B8", 4600, 1280, 2560, _lib.EPI_RESIDUAL),
          ("cls B8", 4600, 4096, 1280, _lib.EPI_BIAS), ("qkv c2f", 5536, 3840, 1280, _lib.EPI_STORE),
          ("qkv B1", 575, 3840, 1280, _lib.EPI_STORE), ("w1g B1", 575, 5120, 1280, _lib.EPI_GEGLU),
          ("w2  B1", 575, 1280, 2560, _lib.EPI_RESIDUAL), ("sq 4096", 4096, 4096, 4096, _lib.EPI_STORE)]


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


for name, M, N, K, epi in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    bias = torch.randn(N, device="cuda", generator=g)
    a3, w3 = eng.tile3(eng.split3(a)), eng.tile3(eng.split3(w))
    a2, w2 = eng.split2h(a, tiled=True), eng.split2h(w, tiled=True)
    out = torch.zeros(M, N // 2 if epi == _lib.EPI_GEGLU else N, device="cuda")
    fl = 2.0 * M * N * K
    line = f"{name:8s} M={M:5d} N={N:5d} K={K:5d}:"
    for bm in (128, 192, 256, 0):
        if bm == 192 and epi == _lib.EPI_GEGLU:
            continue
        eng.check(eng.lib.vn_debug_x3_config(eng.handle, bm, -1, -1), "cfg")
        kw = dict(bias=bias if epi == _lib.EPI_BIAS else None, epilogue=epi, out=out, tiled_shape=(M, N, K))
        t3 = timeit(lambda: eng.gemm_bf16x3(a3, w3, **kw))
        t2 = timeit(lambda: eng.gemm_f16x2(a2, w2, **kw))
        line += f"  {bm or 'auto'}: x3 {t3 * 1e6:6.1f} us ({fl / t3 / 1e12:5.1f} TF-eq)  h2 {t2 * 1e6:6.1f} us ({fl / t2 / 1e12:5.1f} TF-eq) {t3 / t2:4.2f}x |"
    eng.check(eng.lib.vn_debug_x3_config(eng.handle, 0, -1, -1), "cfg")
    print(line, flush=True)
