"""gemm_x3.hip, bf16x3 (default) or f16x2 operands (argv[1] = "f16x2"): every (tile height, k-split) option on the model's shapes,
against the by-shape choice of the kernel's cost model (x3_choose).  Output -> profiles/r0N_gemm_x3_plan_sweep*.txt."""
import sys

import torch

sys.path.insert(0, ".")
from vampnet_amd import _lib
from vampnet_amd.engine import Engine

eng = Engine("cuda:0")
H2 = len(sys.argv) > 1 and sys.argv[1] == "f16x2"
S, R, G, Bi = _lib.EPI_STORE, _lib.EPI_RESIDUAL, _lib.EPI_GEGLU, _lib.EPI_BIAS
SHAPES = [("qkv  B8 (store proxy)", 4600, 3840, 1280, S), ("wo   B8", 4600, 1280, 1280, R), ("w1g  B8", 4600, 5120, 1280, G),
          ("w2   B8", 4600, 1280, 2560, R), ("cls  B8", 4600, 4096, 1280, Bi),
          ("qkv  c2f B8 (32 x 173)", 5536, 3840, 1280, S), ("wo   c2f B8", 5536, 1280, 1280, R), ("w1g  c2f B8", 5536, 5120, 1280, G),
          ("w2   c2f B8", 5536, 1280, 2560, R), ("cls  c2f B8", 5536, 10240, 1280, Bi),
          ("qkv  B4", 2300, 3840, 1280, S), ("wo   B4", 2300, 1280, 1280, R), ("w1g  B4", 2300, 5120, 1280, G), ("w2   B4", 2300, 1280, 2560, R),
          ("qkv  B1", 575, 3840, 1280, S), ("wo   B1", 575, 1280, 1280, R), ("w1g  B1", 575, 5120, 1280, G), ("w2   B1", 575, 1280, 2560, R),
          ("qkv  c2f B1 (4 x 173)", 692, 3840, 1280, S), ("w2   c2f B1", 692, 1280, 2560, R)]


def timeit(fn, n=15):
    for _ in range(3):
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
for name, M, N, K, epi in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(0)
    a32, w32 = torch.randn(M, K, device="cuda", generator=g), torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    if H2:
        a3, w3 = eng.split2h(a32, tiled=True), eng.split2h(w32, tiled=True)
    else:
        a3, w3 = eng.tile3(eng.split3(a32)), eng.tile3(eng.split3(w32))          # the model path's tiled operand layout
    bias = torch.randn(N, device="cuda", generator=g)
    out = torch.zeros(M, N // 2 if epi == G else N, device="cuda")
    fn = lambda: (eng.gemm_f16x2 if H2 else eng.gemm_bf16x3)(a3, w3, bias=bias if epi == Bi else None, epilogue=epi, out=out, tiled_shape=(M, N, K))
    res = []
    for bm in (128, 192, 256):
        if epi == G and bm == 192:
            continue
        for ns in ((1, 2, 4) if epi in (S, R) else (1,)):
            if ns > 1 and (K // 32) // ns < 8:
                continue
            eng.lib.vn_debug_x3_config(eng.handle, bm, ns, -1)
            res.append((timeit(fn), bm, ns))
    eng.lib.vn_debug_x3_config(eng.handle, 0, -1, -1)
    auto = timeit(fn)
    best = min(res)
    fl = 2.0 * M * N * K
    print(f"{name:24s} M={M:5d} N={N:5d} K={K:4d}: " + "  ".join(f"{bm}/{ns}:{us:6.1f}" for us, bm, ns in res) +
          f"  | auto {auto:6.1f} us ({fl / auto / 1e6:5.1f} TF-eq)  best {best[1]}/{best[2]} {best[0]:6.1f}" +
          ("" if auto <= 1.03 * best[0] else "   <-- model misses by %.0f %%" % (100 * (auto / best[0] - 1))), flush=True)
