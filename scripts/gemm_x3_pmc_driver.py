"""Kernel-only driver for rocprofv3 --pmc runs of the split-plane GEMMs: VN_PMC_KERNEL = x3 (default) | h2 (staged f16x2)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampnet_amd.engine import Engine

eng = Engine("cuda:0")
which = os.environ.get("VN_PMC_KERNEL", "x3")
for (M, N, K) in [(4096, 4096, 4096), (4600, 3840, 1280), (4600, 5120, 1280)]:
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    out = torch.zeros(M, N, device="cuda")
    if which == "h2":
        a2, w2 = eng.split2h(a), eng.split2h(w)
        tile = int(os.environ.get("VN_H2_TILE", "128"))
        fn = lambda: eng.gemm_f16x2(a2, w2, out=out, tile_m=tile)
    else:
        a3, w3 = eng.split3(a), eng.split3(w)
        fn = lambda: eng.gemm_bf16x3(a3, w3, out=out)
    for _ in range(3):
        fn()
torch.cuda.synchronize()
