"""Micro-benchmark of vn_transpose_f32 on the shapes of the training step (GB/s of read+write traffic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vampnet_amd.engine import Engine

eng = Engine("cuda:0")
lib = eng.lib
for R, C in [(4600, 1280), (4600, 2560), (4600, 3840), (4600, 5120), (4600, 4096), (5120, 1280), (1280, 2560), (3840, 1280)]:
    ldd = (R + 31) // 32 * 32
    src = torch.randn(R, C, device="cuda")
    dst = torch.empty(C, ldd, device="cuda")
    for _ in range(3):
        eng.check(lib.vn_transpose_f32(eng.handle, src.data_ptr(), dst.data_ptr(), R, C, ldd, eng.stream()), "t")
    torch.cuda.synchronize()
    assert torch.equal(dst[:, :R], src.t()) and (dst[:, R:] == 0).all()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(20):
        lib.vn_transpose_f32(eng.handle, src.data_ptr(), dst.data_ptr(), R, C, ldd, eng.stream())
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"[{R}x{C}] -> [{C}x{ldd}]: {us:8.1f} us  {(R * C + C * ldd) * 4 / us / 1e3:8.1f} GB/s")
