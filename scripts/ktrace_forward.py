"""Timeline of ONE model forward out of a rocprofv3 --kernel-trace CSV: every dispatch between the last-but-`skip` launch of the
marker kernel (vn_embed_kernel = first kernel of a forward) and the next one, with its duration and the idle gap in front of it;
then the per-kernel-name totals, the GPU-busy fraction of the span and a per-layer summary (mean over the layers of that forward).
usage: ktrace_forward.py kernel_trace.csv [marker] [skip-from-last (default 2)] [--full]"""
import csv
import re
import sys
from collections import defaultdict

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "vn_embed_kernel"
skip = int(sys.argv[3]) if len(sys.argv) > 3 and not sys.argv[3].startswith("--") else 2
full = "--full" in sys.argv
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if len(starts) < skip + 1:
    raise SystemExit(f"only {len(starts)} launches of {marker}")
i0, i1 = starts[-skip - 1], starts[-skip]
fwd = rows[i0:i1]


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "")
    m = re.match(r"(\w+)<(.*)>$", name)
    return name if not m else f"{m.group(1)}<{m.group(2)[:28]}>"


t0 = int(fwd[0]["Start_Timestamp"])
span = int(fwd[-1]["End_Timestamp"]) - t0
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in fwd)
print(f"one forward step (sampling kernels of the step included): {len(fwd)} dispatches, span {span / 1e3:.1f} us, kernel time "
      f"{busy / 1e3:.1f} us, GPU busy {100.0 * busy / span:.1f} %, idle {(span - busy) / 1e3:.1f} us")
agg = defaultdict(lambda: [0, 0, 0])
prev_end = t0
gaps = []
for r in fwd:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    a = agg[short(r["Kernel_Name"])]
    a[0] += 1
    a[1] += e - s
    a[2] += max(0, s - prev_end)
    gaps.append(max(0, s - prev_end))
    if full:
        print(f"  +{(s - t0) / 1e3:9.1f} us  gap {max(0, s - prev_end) / 1e3:6.2f}  dur {(e - s) / 1e3:7.2f}  wg {r.get('Workgroup_Size', '?'):>5} "
              f"grid {r.get('Grid_Size', '?'):>8}  {short(r['Kernel_Name'])}")
    prev_end = max(prev_end, e)
gaps.sort()
print(f"gaps between consecutive dispatches: median {gaps[len(gaps) // 2] / 1e3:.2f} us, p90 {gaps[int(0.9 * len(gaps))] / 1e3:.2f} us, "
      f"sum {sum(gaps) / 1e3:.1f} us")
print(f"{'kernel':64s} {'calls':>5s} {'total us':>9s} {'avg us':>8s} {'% busy':>7s} {'gap us':>7s}")
for name, (n, ns, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{name[:64]:64s} {n:5d} {ns / 1e3:9.1f} {ns / 1e3 / n:8.2f} {100.0 * ns / busy:7.1f} {g / 1e3:7.1f}")
