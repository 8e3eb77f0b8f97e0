"""Kernel-only driver for rocprofv3 --pmc runs of the split-plane GEMM (VN_PMC_FMT = bf16x3 | f16x2): the model's B = 8 shapes + 4096^3, schedule from
VN_X3_BM (128 / 256) and VN_X3_SK (1 stream-K, 0 data-parallel).  Three launches per shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampnet_amd import _lib
from vampnet_amd.engine import Engine

eng = Engine("cuda:0")
for (M, N, K, epi) in [(4096, 4096, 4096, _lib.EPI_STORE), (4600, 3840, 1280, _lib.EPI_STORE), (4600, 5120, 1280, _lib.EPI_GEGLU),
                       (4600, 1280, 1280, _lib.EPI_RESIDUAL), (4600, 1280, 2560, _lib.EPI_RESIDUAL), (4600, 4096, 1280, _lib.EPI_STORE)]:
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    out = torch.zeros(M, N // 2 if epi == _lib.EPI_GEGLU else N, device="cuda")
    if os.environ.get("VN_PMC_FMT", "bf16x3") == "f16x2":
        a3, w3, gemm = eng.split2h(a, tiled=True), eng.split2h(w, tiled=True), eng.gemm_f16x2
    else:
        a3, w3, gemm = eng.tile3(eng.split3(a)), eng.tile3(eng.split3(w)), eng.gemm_bf16x3          # the model path's tiled operand layout
    for _ in range(3):
        gemm(a3, w3, epilogue=epi, out=out, tiled_shape=(M, N, K))
torch.cuda.synchronize()
