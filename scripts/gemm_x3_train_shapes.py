"""gemm_x3.hip on the TRAINING step's shapes (B = 8, M = 4600; dW GEMMs contract over Mp = 4608 tokens): every (tile height, k-split) against
the planner's choice.  Output -> profiles/r05_gemm_train_shapes_sweep.txt"""
import sys
import torch
sys.path.insert(0, ".")
from vampnet_amd import _lib
from vampnet_amd.engine import Engine
eng = Engine("cuda:0")
S = _lib.EPI_STORE
SHAPES = [("dX qkv   ", 4600, 1280, 3840), ("dX wo    ", 4600, 1280, 1280), ("dX w1    ", 4600, 1280, 5120), ("dX w2    ", 4600, 2560, 1280),
          ("dX cls   ", 4600, 1280, 4096), ("dW qkv   ", 3840, 1280, 4608), ("dW wo    ", 1280, 1280, 4608), ("dW w1    ", 5120, 1280, 4608),
          ("dW w2    ", 1280, 2560, 4608), ("dW cls   ", 4096, 1280, 4608), ("fwd w1   ", 4600, 5120, 1280), ("fwd w2   ", 4600, 1280, 2560)]

def timeit(fn, n=12):
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
for name, M, N, K in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(0)
    a3 = eng.tile3(eng.split3(torch.randn(M, K, device="cuda", generator=g)))
    w3 = eng.tile3(eng.split3(torch.randn(N, K, device="cuda", generator=g) / K ** 0.5))
    out = torch.zeros(M, N, device="cuda")
    fn = lambda: eng.gemm_bf16x3(a3, w3, epilogue=S, out=out, tiled_shape=(M, N, K))
    res = []
    for bm in (96, 128, 192, 256):
        for ns in (1, 2, 4):
            if ns > 1 and (K // 32) // ns < 8:
                continue
            eng.lib.vn_debug_x3_config(eng.handle, bm, ns, -1)
            res.append((timeit(fn), bm, ns))
    eng.lib.vn_debug_x3_config(eng.handle, 0, -1, -1)
    auto = timeit(fn)
    best = min(res)
    fl = 2.0 * M * N * K
    print(f"{name} M={M:5d} N={N:5d} K={K:4d}: " + "  ".join(f"{bm}/{ns}:{us:6.1f}" for us, bm, ns in res) +
          f"  | auto {auto:6.1f} us ({fl / auto / 1e6:5.1f} TF-eq)  best {best[1]}/{best[2]} {best[0]:6.1f}" +
          ("" if auto <= 1.03 * best[0] else "   <-- model misses by %.0f %%" % (100 * (auto / best[0] - 1))), flush=True)
