"""Attention kernels alone: fp32-input MFMA (attention_f32.hip, timed through the single-op entry's launch only is not possible
— it allocates per call — so it is bracketed generously) vs bf16x3 (attention_x3.hip, kernel-only timing hook).  VN_ATTN_X3_WAVES
selects the block size of the x3 kernel (0 = cost model)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampnet_amd.engine import Engine

eng = Engine("cuda:0")
w = torch.randn(4096, 4096, device="cuda")
for _ in range(20):
    eng.gemm(w, w)
for (B, H, T) in [(8, 20, 575), (32, 20, 173), (1, 20, 575), (2, 20, 575), (4, 20, 173)]:
    q, k, v = (torch.randn(B, H, T, 64, device="cuda") for _ in range(3))
    table = torch.randn(32, H, device="cuda")
    out = torch.empty(B, T, H * 64, device="cuda")
    us = C.c_float()
    eng.check(eng.lib.vn_debug_attention_x3_time(eng.handle, q.data_ptr(), k.data_ptr(), v.data_ptr(), table.data_ptr(),
                                                 out.data_ptr(), B, H, T, 30, C.byref(us), eng.stream()), "attention_x3_time")
    fl = 4.0 * T * T * 64 * H * B
    print(f"attn_x3 waves={os.environ.get('VN_ATTN_X3_WAVES', 'auto'):4s} B={B:2d} H={H} T={T}: {us.value:8.1f} us  {fl / us.value / 1e6:6.1f} TF-eq "
          f"({6 * fl / us.value / 1e6 / 2500:5.1%} of the bf16 pipe)", flush=True)
