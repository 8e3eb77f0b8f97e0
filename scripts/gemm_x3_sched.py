"""bf16x3 GEMM: tile height (128 / 256 / by shape) on the model's shapes,
interleaved rounds in ONE process (median of ROUNDS), real split planes of N(0,1) operands.  TF-eq = 2MNK / t
(fp32-equivalent); the matrix pipe executes 6x that.  Second table: ablations of the kernels (results invalid)."""
import statistics
import sys

import torch

sys.path.insert(0, ".")
from vampnet_amd import _lib
from vampnet_amd.engine import Engine

eng = Engine("cuda:0")
SHAPES = [("qkv B8", 4600, 3840, 1280, _lib.EPI_STORE), ("wo  B8", 4600, 1280, 1280, _lib.EPI_RESIDUAL),
          ("w1g B8", 4600, 5120, 1280, _lib.EPI_GEGLU), ("w2  B8", 4600, 1280, 2560, _lib.EPI_RESIDUAL),
          ("cls B8", 4600, 4096, 1280, _lib.EPI_BIAS), ("qkv c2f", 1384, 3840, 1280, _lib.EPI_STORE),
          ("w2  c2f", 1384, 1280, 2560, _lib.EPI_RESIDUAL), ("qkv B1", 575, 3840, 1280, _lib.EPI_STORE),
          ("sq 4096", 4096, 4096, 4096, _lib.EPI_STORE), ("sq 8192", 8192, 8192, 8192, _lib.EPI_STORE)]
PIPES = [128, 256, 0]
ROUNDS = 5


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3


def cfg(p=0, split=-1, abl=-1):
    eng.check(eng.lib.vn_debug_x3_config(eng.handle, p, split, abl), "vn_debug_x3_config")


def name_of(p):
    return "bm" + (str(p) if p else "auto")


print(f"{'shape':10s} " + " ".join(f"{name_of(p) + ' us':>10s} {'TF-eq':>6s} {'%pipe':>6s}" for p in PIPES))
for name, M, N, K, epi in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    a3, w3 = eng.split3(a), eng.split3(w)
    bias = torch.zeros(N, device="cuda")
    out = torch.zeros(M, N // 2 if epi == _lib.EPI_GEGLU else N, device="cuda")
    ts = {p: [] for p in PIPES}
    for r in range(ROUNDS):
        for p in PIPES:
            cfg(p)
            ts[p].append(timeit(lambda: eng.gemm_bf16x3(a3, w3, bias=bias if epi == _lib.EPI_BIAS else None, epilogue=epi, out=out)))
    fl = 2.0 * M * N * K
    row = f"{name:10s} "
    for p in PIPES:
        t = statistics.median(ts[p])
        row += f"{t * 1e6:10.1f} {fl / t / 1e12:6.1f} {6 * fl / t / 2.5e15:6.1%} "
    print(row, flush=True)
    del a, w, a3, w3, out

print("\nablations (store epilogue, data-parallel form): abl 0 = shipped, 1 = no DMA in the k-loop, 2 = no fragment reads, 3 = MFMA + barriers only, "
      "4 = DMA of whole 128-B lines (8 rows per instruction, same volume)")
for name, M, N, K in [("qkv B8", 4600, 3840, 1280), ("sq 4096", 4096, 4096, 4096)]:
    g = torch.Generator(device="cuda").manual_seed(0)
    a3 = eng.split3(torch.randn(M, K, device="cuda", generator=g))
    w3 = eng.split3(torch.randn(N, K, device="cuda", generator=g) / K ** 0.5)
    out = torch.zeros(M, N, device="cuda")
    for p in (128, 256):
        row = f"{name:8s} {name_of(p)}: "
        for abl in (0, 1, 2, 3, 4):
            cfg(p, 1, abl)
            t = statistics.median(timeit(lambda: eng.gemm_bf16x3(a3, w3, out=out)) for _ in range(3))
            row += f"abl{abl} {2.0 * M * N * K / t / 1e12:6.1f} TF-eq ({6 * 2.0 * M * N * K / t / 2.5e15:5.1%})  "
        print(row, flush=True)
cfg()
