import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampnet_amd import _lib
from vampnet_amd.engine import Engine
eng = Engine("cuda:0")
def bench(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
w = torch.randn(4096, 4096, device="cuda")
for _ in range(30): eng.gemm(w, w)
D = 1280
for name, M, N, K in [("sq4096", 4096, 4096, 4096), ("sq8192", 8192, 8192, 8192), ("qkv B8", 4600, 3 * D, D), ("w1 B8", 4600, 4 * D, D), ("w2 B8", 4600, D, 2 * D), ("wo B8", 4600, D, D), ("cls B8", 4600, 4096, D), ("qkv c2f4", 5536, 3 * D, D)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16); Wt = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    A32, W32 = A.float(), Wt.float()
    row = []
    for sched in (0, 1):
        eng.lib.vn_debug_gemm_config(eng.handle, 0, 0, 1 | ((sched + 1) << 1))
        us = bench(lambda: eng.gemm_bf16(A, Wt))
        row.append(f"{'SK' if sched else 'DP'} {us:7.1f}us {2.0*M*N*K/us/1e6:7.1f}TF")
    eng.lib.vn_debug_gemm_config(eng.handle, 0, 0, 1)
    us32 = bench(lambda: eng.gemm(A32, W32), 10)
    print(f"{name:9s} bf16: " + " | ".join(row) + f" || f32 auto {us32:7.1f}us {2.0*M*N*K/us32/1e6:6.1f}TF", flush=True)
