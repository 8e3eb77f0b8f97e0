"""Is VMM-backed memory (hipMemCreate / hipMemMap guard blocks behind torch's pluggable allocator) trustworthy for plain copies and torch ops?"""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampnet_amd import _lib
alloc = torch.cuda.memory.CUDAPluggableAllocator(_lib.LIB_PATH, "vn_guard_torch_alloc", "vn_guard_torch_free")
torch.cuda.memory.change_current_allocator(alloc)
bad = 0
g = torch.Generator().manual_seed(0)
for rnd in range(3):
    for n in (1, 3, 255, 1000, 4097, 65537, 1 << 20, 3_000_001):
        x = torch.randn(n, generator=g)
        y = x.cuda()
        if not torch.equal(y.cpu(), x): bad += 1; print("H2D/D2H mismatch", n)
        xp = x.pin_memory()
        z = xp.to("cuda", non_blocking=True)
        torch.cuda.synchronize()
        if not torch.equal(z.cpu(), x): bad += 1; print("pinned async mismatch", n)
        w = (y * 2 + z).cpu()
        if not torch.equal(w, x * 2 + x): bad += 1; print("op mismatch", n)
        i = torch.randint(0, 1000, (n,), generator=g)
        if not torch.equal(i.cuda().to(torch.int32).cpu(), i.to(torch.int32)): bad += 1; print("int mismatch", n)
        del y, z, w
lib = _lib.load()
n_, live, nb = C.c_int64(), C.c_int64(), C.c_int64()
mode = lib.vn_guard_stats(C.byref(n_), C.byref(live), C.byref(nb))
print("VMM probe: mismatches", bad, "mode", mode, "blocks", n_.value, "live", live.value)
