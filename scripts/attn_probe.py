"""bf16x3 attention, shared-tile kernel probes (tuning): blocks per CU (dynamic-LDS override), start stagger and the per-phase cycle
trace — through vn_debug_attention_x3_config / vn_debug_attention_x3_time (include/vampnet_hip_debug.h)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampnet_amd.engine import Engine

eng = Engine("cuda:0")
lib = eng.lib
w = torch.randn(4096, 4096, device="cuda")
for _ in range(20):
    eng.gemm(w, w)


def run(B, H, T, lds=0, stagger=0, trace=None, iters=30, split=0):
    q, k, v = (torch.randn(B, H, T, 64, device="cuda") for _ in range(3))
    table = torch.randn(32, H, device="cuda")
    out = torch.empty(B, T, H * 64, device="cuda")
    us = C.c_float()
    lib.vn_debug_attention_x3_config(eng.handle, split, lds, stagger, trace.data_ptr() if trace is not None else None)
    try:
        eng.check(lib.vn_debug_attention_x3_time(eng.handle, q.data_ptr(), k.data_ptr(), v.data_ptr(), table.data_ptr(),
                                                 out.data_ptr(), B, H, T, iters, C.byref(us), eng.stream()), "attention_x3_time")
    finally:
        lib.vn_debug_attention_x3_config(eng.handle, -1, 0, -1, None)
    torch.cuda.synchronize()
    return us.value


H = 20
QUICK = os.environ.get("ATTN_PROBE_QUICK") == "1"
for (B, T) in ([(8, 575), (32, 173), (2, 575)] if QUICK else [(8, 575), (4, 575), (2, 575), (32, 173)]):
    fl = 4.0 * T * T * 64 * H * B
    variants = [("3 blocks/CU (shipped)", {}), ("2 blocks/CU", dict(lds=60 * 1024)), ("1 block/CU", dict(lds=90 * 1024)),
                ("tail blocks at the default wave priority", dict(stagger=0x10000))]
    if not QUICK:
        variants += [(f"stagger {s} x 64 cyc", dict(stagger=s)) for s in (4, 8, 16, 24, 32, 48, 64)] + \
                    [(f"stagger {s} x 64 cyc, 2/CU", dict(stagger=s, lds=60 * 1024)) for s in (16, 32, 48)]
    for name, kw in variants:
        us = run(B, H, T, **kw)
        print(f"B={B:2d} T={T}: {name:28s} {us:8.1f} us  {fl / us / 1e6:6.1f} TF-eq", flush=True)

# timeline: every block's phase cycles (wave 0, full-role blocks), entry / exit time and CU — what a launch is made of.
# s_memtime is per CU (its offset differs from CU to CU), so the timeline is read CU by CU.
names = ["wait dma", "barrier", "dma issue", "tile math"]
for (B, T) in [(8, 575), (32, 173)]:
    for name, kw in [("3/CU", {}), ("3/CU, tail blocks at the default wave priority", dict(stagger=0x10000)), ("1/CU", dict(lds=90 * 1024))]:
        nqbf, rq = T // 128, T % 128
        tail = 0 < rq <= 64
        nfull = (nqbf + (1 if rq and not tail else 0)) * H * B
        nblk = nfull + (H * B if tail else 0)
        tr = torch.zeros(nblk, 8, dtype=torch.int32, device="cuda")
        us = run(B, H, T, trace=tr, iters=1, **kw)
        t = tr.cpu().to(torch.int64) & 0xFFFFFFFF
        dur = (t[:, 5] - t[:, 4]) & 0xFFFFFFFF
        ntile = (T + 31) // 32 + 1
        full, tl = slice(0, nfull), slice(nfull, nblk)
        mean = t[full, :4].double().mean(0)
        print(f"trace B={B} T={T} {name}: kernel {us:7.1f} us, {nfull} full + {nblk - nfull} tail blocks; full-role wave 0 per tile (cycles, ~{ntile} tiles): " +
              ", ".join(f"{n} {mean[i].item() / ntile:6.0f}" for i, n in enumerate(names)) +
              f"; full block {dur[full].double().mean().item():8.0f} cyc (min {dur[full].min().item()}, max {dur[full].max().item()})" +
              (f"; tail block {dur[tl].double().mean().item():8.0f} (min {dur[tl].min().item()}, max {dur[tl].max().item()})" if tail else ""), flush=True)
        cu = ((t[:, 6] & 0xF) << 8) | ((t[:, 7] >> 8) & 0xFF)          # XCC id, SE / SH / CU id
        done, nb, firsts, late_s, late_e = [], [], [], [], []
        for c in sorted(set(cu.tolist())):
            sel = (cu == c).nonzero().flatten()
            base = t[sel, 4].min()
            rs, re = (t[sel, 4] - base) & 0xFFFFFFFF, (t[sel, 5] - base) & 0xFFFFFFFF
            done.append(re.max().item()); nb.append(len(sel))
            first = rs < 20000
            firsts.append(sorted(re[first].tolist()))
            late_s += rs[~first].tolist(); late_e += re[~first].tolist()
        done = torch.tensor(done, dtype=torch.float64)
        q = lambda v, f: sorted(v)[int(f * (len(v) - 1))]
        ends = [[f[i] for f in firsts if len(f) > i] for i in range(3)]
        print(f"    {len(done)} CUs, blocks per CU {min(nb)}..{max(nb)}; a CU is busy for {done.mean().item():8.0f} cycles (min {done.min().item():.0f}, "
              f"median {done.median().item():.0f}, max {done.max().item():.0f}; the launch = {us * 1e3:.0f} ns => {done.max().item() / us / 1e3:4.2f} GHz if the slowest CU is the launch)", flush=True)
        print("    first-wave blocks of a CU end at (median over CUs): " + ", ".join(f"#{i + 1} {q(e, 0.5)}" for i, e in enumerate(ends) if e) +
              (f"; {len(late_s)} late blocks start at {min(late_s)}..{max(late_s)} (median {q(late_s, 0.5)}) and end at {min(late_e)}..{max(late_e)} (median {q(late_e, 0.5)})" if late_s else ""), flush=True)
