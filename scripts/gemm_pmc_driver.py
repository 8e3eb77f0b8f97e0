import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampnet_amd import _lib
from vampnet_amd.engine import Engine
eng = Engine("cuda:0")
for (M, N, K) in [(4096, 4096, 4096), (4600, 3840, 1280)]:
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5
    for sched in (0, 1):
        eng.lib.vn_debug_gemm_config(eng.handle, 128, 128, 1 | ((sched + 1) << 1))
        for _ in range(3):
            eng.gemm(A, W)
torch.cuda.synchronize()
