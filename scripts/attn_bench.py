"""bf16x3 attention (attention_x3.hip), kernel-only timing of every work decomposition at the model's shapes: shared 128-query tiles
vs key-split 32-query blocks (1 / 2 / 4 waves), next to the launcher's own choice and to the fp32-input MFMA kernel
(attention_f32.hip, timed with events around the single-op entry, which also allocates: an upper bound)."""
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


def x3(q, k, v, table, out, split, iters=30, np=3):
    B, H, T, _ = q.shape
    us = C.c_float()
    eng.check(lib.vn_debug_attention_x3_config(eng.handle, split, 0, -1, None), "cfg")
    try:
        eng.check(lib.vn_debug_attention_x3_time(eng.handle, q.data_ptr(), k.data_ptr(), v.data_ptr(), table.data_ptr(),
                                                 out.data_ptr(), B, H, T, iters if np == 3 else -iters, C.byref(us), eng.stream()), "attention_x3_time")
    finally:
        eng.check(lib.vn_debug_attention_x3_config(eng.handle, -1, 0, -1, None), "cfg")
    return us.value


H = 20
for (B, T) in [(1, 575), (2, 575), (3, 575), (4, 575), (4, 173), (8, 173)]:
    q, k, v = (torch.randn(B, H, T, 64, device="cuda") for _ in range(3))
    table = torch.randn(32, H, device="cuda")
    out = torch.empty(B, T, H * 64, device="cuda")
    fl = 4.0 * T * T * 64 * H * B
    res = {name: x3(q, k, v, table, out, split) for name, split in (("auto", -1), ("shared", 0), ("ks1", 1), ("ks2", 2), ("ks4", 4), ("pair8", 8))}
    res2 = {name: x3(q, k, v, table, out, split, np=2) for name, split in (("auto", -1), ("shared", 0), ("ks2", 2), ("pair8", 8))}
    eng.attention(q, k, v, table)                  # fp32-input kernel through its single-op entry (allocates + frees per call)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        eng.attention(q, k, v, table)
    e1.record()
    torch.cuda.synchronize()
    f32 = e0.elapsed_time(e1) * 100.0
    print(f"B={B:2d} T={T}: f16x2 operands: " + "  ".join(f"{n} {u:7.1f} us" for n, u in res2.items()) +
          f"  ({fl / res2['auto'] / 1e6:6.1f} TF-eq = {3 * fl / res2['auto'] / 1e6 / 2500:5.1%} of the fp16 pipe)", flush=True)
    print(f"B={B:2d} T={T}: bf16x3 operands: " + "  ".join(f"{n} {u:7.1f} us" for n, u in res.items()) +
          f"  | auto = {fl / res['auto'] / 1e6:6.1f} TF-eq ({6 * fl / res['auto'] / 1e6 / 2500:5.1%} of the bf16 pipe); "
          f"fp32-input kernel (single-op entry, upper bound) {f32:7.1f} us", flush=True)
