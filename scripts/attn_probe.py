"""bf16x3 attention, shared-tile kernel probes (tuning): blocks per CU (dynamic-LDS override), start stagger and the per-phase cycle
trace — through vn_debug_attention_x3_config / vn_debug_attention_x3_time (include/vampnet_hip_debug.h)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampnet_amd.engine import Engine

eng = Engine("cuda:0")
lib = eng.lib
w = torch.randn(4096, 4096, device="cuda")
for _ in range(20):
    eng.gemm(w, w)


def run(B, H, T, lds=0, stagger=0, trace=None, iters=30):
    q, k, v = (torch.randn(B, H, T, 64, device="cuda") for _ in range(3))
    table = torch.randn(32, H, device="cuda")
    out = torch.empty(B, T, H * 64, device="cuda")
    us = C.c_float()
    lib.vn_debug_attention_x3_config(eng.handle, 0, lds, stagger, trace.data_ptr() if trace is not None else None)
    try:
        eng.check(lib.vn_debug_attention_x3_time(eng.handle, q.data_ptr(), k.data_ptr(), v.data_ptr(), table.data_ptr(),
                                                 out.data_ptr(), B, H, T, iters, C.byref(us), eng.stream()), "attention_x3_time")
    finally:
        lib.vn_debug_attention_x3_config(eng.handle, -1, 0, -1, None)
    torch.cuda.synchronize()
    return us.value


H = 20
QUICK = os.environ.get("ATTN_PROBE_QUICK") == "1"
for (B, T) in ([(8, 575), (32, 173), (2, 575)] if QUICK else [(8, 575), (4, 575), (2, 575), (32, 173)]):
    fl = 4.0 * T * T * 64 * H * B
    variants = [("3 blocks/CU (shipped)", {}), ("2 blocks/CU", dict(lds=60 * 1024)), ("1 block/CU", dict(lds=90 * 1024))]
    if not QUICK:
        variants += [(f"stagger {s} x 64 cyc", dict(stagger=s)) for s in (4, 8, 16, 24, 32, 48, 64)] + \
                    [(f"stagger {s} x 64 cyc, 2/CU", dict(stagger=s, lds=60 * 1024)) for s in (16, 32, 48)]
    for name, kw in variants:
        us = run(B, H, T, **kw)
        print(f"B={B:2d} T={T}: {name:28s} {us:8.1f} us  {fl / us / 1e6:6.1f} TF-eq", flush=True)

# phase trace: cycles summed over the tiles of wave 0 of every 16th block
names = ["wait dma", "barrier", "dma issue", "tile math", "-", "-"]
for (B, T) in [(8, 575), (2, 575)]:
    for name, kw in [("3/CU", {}), ("1/CU", dict(lds=90 * 1024))]:
        nblk = ((T + 127) // 128) * H * B
        tr = torch.zeros((nblk + 15) // 16, 8, dtype=torch.int32, device="cuda")
        us = run(B, H, T, trace=tr, iters=3, **kw)
        t = tr.cpu().to(torch.int64) & 0xFFFFFFFF
        t = t[t[:, 6] > 0]
        ntile = (T + 31) // 32 + 1
        mean = t[:, :7].double().mean(0)
        slots = sorted(set((t[:, 7] & 15).tolist()))
        print(f"trace B={B} T={T} {name:16s} kernel {us:7.1f} us; traced waves {len(t)}; per tile (cycles, ~{ntile} tiles): " +
              ", ".join(f"{n} {mean[i].item() / ntile:7.0f}" for i, n in enumerate(names)) +
              f"; block total {mean[6].item():9.0f} cyc (min {t[:, 6].min().item()}, max {t[:, 6].max().item()}); wave slots seen {slots}", flush=True)
