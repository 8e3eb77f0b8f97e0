"""bf16x3 GEMM timing on two shapes (run once per VN_X3_ABL / VN_X3_PIPE setting; ablated results are not valid products)."""
import os
import sys

import torch

sys.path.insert(0, ".")
from vampnet_amd.engine import Engine

eng = Engine("cuda:0")
tag = f"PIPE={os.environ.get('VN_X3_PIPE', '1')} ABL={os.environ.get('VN_X3_ABL', '0')} {os.environ.get('VN_X3_TAG', '')}"
for name, M, N, K in [("qkv B8", 4600, 3840, 1280), ("sq 4096", 4096, 4096, 4096), ("wo B8", 4600, 1280, 1280)]:
    a3 = torch.randn(3, M, K, device="cuda").to(torch.bfloat16)
    w3 = (torch.randn(3, N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    out = torch.zeros(M, N, device="cuda")
    fn = lambda: eng.gemm_bf16x3(a3, w3, out=out)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    if os.environ.get("VN_X3_ABL", "0") == "0":          # gross-error check of the schedule variants (not a parity test)
        A, Wp = a3.float(), w3.float()             # the six products the kernel keeps (planes here are independent randoms)
        ref = sum(A[i] @ Wp[j].t() for i, j in ((0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)))
        rel = ((out - ref).abs().max() / ref.abs().max()).item()
        assert rel < 2e-3, (name, rel)
    print(f"{tag:24s} {name:8s} {t * 1e6:8.1f} us  {2.0 * M * N * K / t / 1e12:6.1f} TF-eq  (matrix pipe {6 * 2.0 * M * N * K / t / 1e12 / 2500:5.1%} of 2.5 PF)")
