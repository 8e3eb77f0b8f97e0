import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampnet_amd.engine import Engine
eng = Engine("cuda:0")
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
w = torch.randn(4096, 4096, device="cuda")
for _ in range(20): eng.gemm(w, w)
table = torch.randn(32, 20, device="cuda")
for (B, H, T) in [(8, 20, 575), (32, 20, 173), (1, 20, 575), (4, 20, 173), (2, 20, 575)]:
    q, k, v = (torch.randn(B, H, T, 64, device="cuda") for _ in range(3))
    us = bench(lambda: eng.attention(q, k, v, table))
    fl = 4.0 * T * T * 64 * H * B
    print(f"attn B={B} H={H} T={T}: {us:8.1f} us  {fl/us/1e6:6.1f} TF", flush=True)
