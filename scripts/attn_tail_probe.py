import ctypes as C, os, sys, torch
sys.path.insert(0, os.getcwd())
from vampnet_amd.engine import Engine
eng = Engine("cuda:0"); lib = eng.lib
w = torch.randn(4096, 4096, device="cuda")
for _ in range(20): eng.gemm(w, w)
def run(B, H, T, iters=30):
    q, k, v = (torch.randn(B, H, T, 64, device="cuda") for _ in range(3))
    table = torch.randn(32, H, device="cuda"); out = torch.empty(B, T, H * 64, device="cuda"); us = C.c_float()
    eng.check(lib.vn_debug_attention_x3_time(eng.handle, q.data_ptr(), k.data_ptr(), v.data_ptr(), table.data_ptr(), out.data_ptr(), B, H, T, iters, C.byref(us), eng.stream()), "t")
    return us.value
for (B, H, T) in [(8, 19, 575), (8, 20, 575), (8, 21, 575), (8, 24, 575), (8, 38, 575), (8, 39, 575), (8, 20, 512), (8, 24, 512), (8, 25, 512), (6, 20, 575), (7, 20, 575), (32, 20, 173), (32, 12, 173), (32, 13, 173)]:
    nblk = ((T + 127) // 128) * H * B
    us = run(B, H, T)
    print(f"B={B} H={H} T={T}: blocks {nblk:5d} = {nblk/768:.2f} x 768 slots: {us:7.1f} us  ({us/ (B*H) * 160:7.1f} us per 160 heads)", flush=True)
