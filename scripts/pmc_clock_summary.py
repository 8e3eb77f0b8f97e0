"""Per-kernel shader clock and matrix-pipe occupancy INSIDE the model, from one rocprofv3 pass
    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -- python bench.py ...
clock [GHz]   = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration          (MI355X_MICROARCH.md, "DVFS give-back")
MFMA busy     = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs) / (GRBM_GUI_ACTIVE / 8)
executed TF   = 32768 flop x (SQ_VALU_MFMA_BUSY_CYCLES / 32) / duration       (a 32x32x16 bf16 MFMA holds its pipe 32 cycles)
usage: pmc_clock_summary.py <rocprof output dir> [name filter ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
filters = sys.argv[2:] or ["vn_gemm_x3", "vn_attention_x3"]
dur = {}
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
per = defaultdict(lambda: defaultdict(float))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if not any(x in name for x in filters):
            continue
        per[(r["Dispatch_Id"], name.split("(")[0][:60], r.get("Grid_Size", ""))][r["Counter_Name"]] += float(r["Counter_Value"])
agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0])
for (did, name, grid), c in per.items():
    if did not in dur or "GRBM_GUI_ACTIVE" not in c:
        continue
    a = agg[(name, grid)]
    a[0] += 1
    a[1] += dur[did]
    a[2] += c["GRBM_GUI_ACTIVE"]
    a[3] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    a[4] += c.get("SQ_BUSY_CYCLES", 0.0)
tot = [0, 0.0, 0.0, 0.0]
print(f"{'kernel':60s} {'grid':>9s} {'n':>5s} {'avg us':>8s} {'clock GHz':>9s} {'MFMA busy':>9s} {'executed TF':>11s}")
for (name, grid), (n, ns, gui, mf, sq) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    cyc = gui / 8.0
    print(f"{name:60s} {grid:>9s} {n:5d} {ns / n / 1e3:8.1f} {cyc / ns:9.3f} {mf / 1024.0 / cyc if cyc else 0:9.3f} "
          f"{mf / 32.0 * 32768 / ns / 1e3:11.1f}")
    if "gemm_x3" in name:
        tot[0] += n; tot[1] += ns; tot[2] += cyc; tot[3] += mf
if tot[1]:
    print(f"ALL vn_gemm_x3 launches: n={tot[0]} time={tot[1] / 1e6:.2f} ms  clock={tot[2] / tot[1]:.3f} GHz  "
          f"MFMA busy={tot[3] / 1024.0 / tot[2]:.3f} of the cycles  executed={tot[3] / 32.0 * 32768 / tot[1] / 1e3:.1f} TF "
          f"(= {tot[3] / 32.0 * 32768 / tot[1] / 1e3 / 2500:.3f} of 2.5 PF; the pipe saturated at this clock would give "
          f"{2500.0 * (tot[2] / tot[1]) / 2.4:.0f} TF)")
