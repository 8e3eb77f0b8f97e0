"""gemm_x3.hip, the 96-row k-split tile (CFG 4): every epilogue against the 128-row tile on the model's one-sequence shapes (values to
fp32 re-association noise, run-to-run bitwise), then kernel times 96 vs 128 / 192 / 256 rows vs the planner's choice."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from vampnet_amd import _lib
from vampnet_amd.engine import Engine

eng = Engine("cuda:0")
S, R, G, Bi = _lib.EPI_STORE, _lib.EPI_RESIDUAL, _lib.EPI_GEGLU, _lib.EPI_BIAS


def cfg(bm=0, ns=-1):
    eng.check(eng.lib.vn_debug_x3_config(eng.handle, bm, ns, -1), "cfg")


def timeit(fn, n=30):
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


ok = True
for (M, N, K) in [(575, 3840, 1280), (575, 1280, 2560), (96, 128, 32), (97, 256, 64), (692, 5120, 1280), (1, 128, 96), (200, 768, 256)]:
    g = torch.Generator(device="cuda").manual_seed(M + N)
    a32 = torch.randn(M, K, device="cuda", generator=g)
    w32 = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    bias = torch.randn(N, device="cuda", generator=g)
    c0 = torch.randn(M, N, device="cuda", generator=g)
    ref64 = (a32.double() @ w32.double().t())
    tol = 2e-6 * (a32.abs().double() @ w32.abs().double().t()) + 1e-6
    for tiled in (False, True):
        a3, w3 = eng.split3(a32), eng.split3(w32)
        kw = {}
        if tiled:
            a3, w3, kw = eng.tile3(a3), eng.tile3(w3), {"tiled_shape": (M, N, K)}
        for ns in ((1, 2, 4) if K // 32 >= 16 else (1,)):
            res = {}
            for bm in (128, 96):
                cfg(bm, ns)
                st = eng.gemm_bf16x3(a3, w3, **kw).clone()
                bi = eng.gemm_bf16x3(a3, w3, bias=bias, epilogue=Bi, **kw).clone()
                rr = c0.clone()
                eng.gemm_bf16x3(a3, w3, epilogue=R, out=rr, **kw)
                st2 = eng.gemm_bf16x3(a3, w3, **kw).clone()
                res[bm] = (st, bi, rr)
                if not torch.equal(st, st2):
                    print(f"  !! run-to-run mismatch bm={bm} ns={ns} {M}x{N}x{K}")
                    ok = False
            cfg()
            e_ref = (res[96][0].double() - ref64).abs()
            bad = int((e_ref > tol).sum().item())
            d = [float((res[96][i] - res[128][i]).abs().max()) for i in range(3)]
            eb = float((res[96][1].double() - (ref64 + bias.double())).abs().sub(tol).max())
            er = float((res[96][2].double() - (ref64 + c0.double())).abs().sub(tol).max())
            flag = bad == 0 and eb <= 0 and er <= 0
            ok &= flag
            print(f"{M}x{N}x{K} tiled={int(tiled)} ns={ns}: 96 vs 128 max|d| store {d[0]:.2e} bias {d[1]:.2e} resid {d[2]:.2e}; vs float64: "
                  f"max err {float(e_ref.max()):.2e}, over tol {bad}  {'ok' if flag else 'FAIL'}", flush=True)
    if N % 128 == 0 and M > 1:
        # GEGLU (packed value / gate columns): fp32 output and planes
        w1 = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
        a3, w3 = eng.split3(a32), eng.split3(w1)
        outs = {}
        for bm in (128, 96):
            cfg(bm, 1)
            outs[bm] = eng.gemm_bf16x3(a3, w3, epilogue=G).clone()
        cfg()
        d = float((outs[96] - outs[128]).abs().max())
        print(f"{M}x{N}x{K} GEGLU (fp32 out, direct epilogue of the 128-row tile): 96 vs 128 max|d| {d:.2e} {'ok' if d < 2e-5 else 'FAIL'}")
        ok &= d < 2e-5
print("VALUES", "OK" if ok else "FAILED")

w = torch.randn(4096, 4096, device="cuda")
for _ in range(20):
    eng.gemm(w, w)
SHAPES = [("qkv  B1", 575, 3840, 1280, S), ("wo   B1", 575, 1280, 1280, R), ("w1g  B1", 575, 5120, 1280, G), ("w2   B1", 575, 1280, 2560, R),
          ("cls  B1", 575, 4096, 1280, Bi), ("qkv  c2f B1", 692, 3840, 1280, S), ("w1g  c2f B1", 692, 5120, 1280, G), ("w2   c2f B1", 692, 1280, 2560, R),
          ("cls  c2f B1", 692, 10240, 1280, Bi),
          ("qkv  B2", 1150, 3840, 1280, S), ("wo   B2", 1150, 1280, 1280, R), ("w1g  B2", 1150, 5120, 1280, G), ("w2   B2", 1150, 1280, 2560, R),
          ("qkv  B4", 2300, 3840, 1280, S), ("wo   B4", 2300, 1280, 1280, R), ("w1g  B4", 2300, 5120, 1280, G), ("w2   B4", 2300, 1280, 2560, R),
          ("qkv  B8", 4600, 3840, 1280, S), ("w1g  B8", 4600, 5120, 1280, G), ("w2   B8", 4600, 1280, 2560, R)]
for name, M, N, K, epi in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(0)
    a32, w32 = torch.randn(M, K, device="cuda", generator=g), torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    a3, w3 = eng.tile3(eng.split3(a32)), eng.tile3(eng.split3(w32))
    bias = torch.randn(N, device="cuda", generator=g)
    out = torch.zeros(M, N // 2 if epi == G else N, device="cuda")
    out16 = None
    fn = lambda: eng.gemm_bf16x3(a3, w3, bias=bias if epi == Bi else None, epilogue=epi, out=out, tiled_shape=(M, N, K))
    res = []
    for bm in (96, 128, 192, 256):
        if epi == G and bm == 192:
            continue
        for ns in ((1, 2, 4) if epi in (S, R) else (1,)):
            if ns > 1 and (K // 32) // ns < 8:
                continue
            cfg(bm, ns)
            res.append((timeit(fn), bm, ns))
    cfg()
    auto = timeit(fn)
    best = min(res)
    fl = 2.0 * M * N * K
    print(f"{name:14s} M={M:5d} N={N:5d} K={K:4d}: " + "  ".join(f"{bm}/{ns}:{us:6.1f}" for us, bm, ns in res) +
          f"  | auto {auto:6.1f} us ({fl / auto / 1e6:5.1f} TF-eq)  best {best[1]}/{best[2]} {best[0]:6.1f}" +
          ("" if auto <= 1.03 * best[0] else "   <-- model misses by %.0f %%" % (100 * (auto / best[0] - 1))), flush=True)

# ---- model level: QKV3 / GEGLU-plane / folded-norm producer epilogues of the 96-row tile inside a forward (tiny and full size)
from oracle import vampnet_oracle as O            # checker (this is a test script)
from vampnet_amd import synth as W
from vampnet_amd.engine import VampNetModel
from vampnet_amd.synth import model_kwargs

cb = W.synth_codebooks()
for dims, B, T, name in [(W.TINY_COARSE_DIMS, 3, 200, "tiny coarse"), (W.TINY_C2F_DIMS, 2, 173, "tiny c2f"), (W.COARSE_DIMS, 1, 575, "coarse"),
                         (W.C2F_DIMS, 4, 173, "c2f")]:
    sd = W.synth_state_dict(dims, 4)
    m = VampNetModel(eng, sd, cb, max_batch=4, max_T=575, precision="bf16x3", **model_kwargs(dims))
    codes = W.synth_codes(B, dims["n_codebooks"], T, seed=6)
    codes[:, dims["n_cond"]:, ::2] = 1024
    outs = {}
    eng.lib.vn_debug_attention_x3_force(eng.handle, 1)
    for bm in (128, 96, 0):
        cfg(bm, -1)
        outs[bm] = m.forward_codes(codes).clone()
        again = m.forward_codes(codes)
        if not torch.equal(outs[bm], again):
            print(f"  !! forward not reproducible at bm={bm}")
            ok = False
    cfg()
    eng.lib.vn_debug_attention_x3_force(eng.handle, -1)
    d = float((outs[96] - outs[128]).abs().max())
    d0 = float((outs[0] - outs[128]).abs().max())
    line = f"{name} B={B} T={T}: logits 96 vs 128 rows max|d| {d:.3e}; auto vs 128 {d0:.3e}"
    if dims in (W.TINY_COARSE_DIMS, W.TINY_C2F_DIMS):
        ref = O.forward(sd, dims, O.from_codes(sd, cb, codes))
        e = float((outs[96].cpu() - ref).abs().max())
        line += f"; 96 vs oracle {e:.3e}"
        ok &= e <= 2e-5
    ok &= d <= 2e-5
    print(line, flush=True)
    del m
print("ALL", "OK" if ok else "FAILED")
