"""Per-kernel statistics of the LAST step in a rocprofv3 --kernel-trace CSV (steady state: excludes warm-up, first-touch
and library initialisation).  A step starts at the last launch of the marker kernel (default vn_embed_kernel)."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "vn_embed_kernel"
nth = int(sys.argv[4]) if len(sys.argv) > 4 else 1          # the step starts at the nth-from-last marker launch
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if not starts:
    raise SystemExit(f"marker {marker} not found")
# the grads memset of the step precedes the marker by a few dispatches: back up to the previous fillBuffer if adjacent
i0 = starts[-nth]
while i0 > 0 and ("fillBuffer" in rows[i0 - 1]["Kernel_Name"] or "i64_to_i32" in rows[i0 - 1]["Kernel_Name"]):
    i0 -= 1
step = rows[i0:]
agg = defaultdict(lambda: [0, 0])
for r in step:
    a = agg[r["Kernel_Name"]]
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(a[1] for a in agg.values())
span = int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])
print(f"last step: {len(step)} dispatches, kernel time {tot / 1e6:.2f} ms, wall span {span / 1e6:.2f} ms, GPU busy {100.0 * tot / span:.1f} %")
for name, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print("%-70s calls=%5d total_ms=%8.2f avg_us=%8.1f pct=%5.1f" % (name[:70], n, ns / 1e6, ns / 1e3 / n, 100.0 * ns / tot))
