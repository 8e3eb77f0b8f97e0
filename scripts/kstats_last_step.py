"""Per-kernel statistics of the LAST step in a rocprofv3 --kernel-trace CSV (steady state: excludes warm-up, first-touch
and library initialisation).  A step starts at the last launch of the marker kernel (default vn_embed_kernel)."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "vn_embed_kernel"
nth = sys.argv[4] if len(sys.argv) > 4 else "1"             # the step starts at the nth-from-last marker launch; "vamp" = one vamp()
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if not starts:
    raise SystemExit(f"marker {marker} not found")
if nth == "vamp":
    # ONE Interface.vamp(): every model forward starts with the marker; a forward of the coarse model holds more layers (GEGLU GEMM
    # launches: vn_gemm_x3_kernel<3, ..> / the f32 kernel's GEGLU instance) than one of the c2f model.  Walking back from the end:
    # the trailing run of SHORT forwards (c2f steps) and the run of LONG forwards in front of it (coarse steps) are the last call
    # (round 4's fixed count of 20 markers took 12 + 8 forwards for a call that has 12 + 2: the four c2f chunks run as one batch)
    bounds = starts + [len(rows)]
    depth = [sum(1 for r in rows[bounds[j]:bounds[j + 1]] if "vn_attention" in r["Kernel_Name"]) for j in range(len(starts))]
    deep = max(depth)
    j = len(starts) - 1
    while j >= 0 and depth[j] < deep:
        j -= 1
    n_short = len(starts) - 1 - j
    k = j
    while k >= 0 and depth[k] == deep:
        k -= 1
    n_long = j - k
    if n_short == 0:          # a coarse-only run (configs[1]): the caller gives the step count instead
        raise SystemExit("no c2f forwards at the end of the trace: pass the number of forwards of a call instead of 'vamp'")
    # several calls back to back look like ...LLLLSS LLLLSS: cut the long run at one call's worth = the shortest long run in the trace
    runs, cur = [], 0
    for dpt in depth:
        if dpt == deep:
            cur += 1
        elif cur:
            runs.append(cur)
            cur = 0
    n_long = min([r for r in runs if r > 0] + [n_long])
    nth = n_long + n_short
    print(f"last vamp(): {n_long} coarse + {n_short} c2f forwards")
nth = int(nth)
# the grads memset of the step precedes the marker by a few dispatches: back up to the previous fillBuffer if adjacent
i0 = starts[-nth]
while i0 > 0 and ("fillBuffer" in rows[i0 - 1]["Kernel_Name"] or "i64_to_i32" in rows[i0 - 1]["Kernel_Name"]):
    i0 -= 1
step = rows[i0:]
# what bench.py runs AFTER its timed region (the adapter-swap cost measurement: merges, plane builds) is not part of the step
for j, r in enumerate(step):
    if "vn_lora_merge_kernel" in r["Kernel_Name"]:
        step = step[:j]
        break
# ... and neither is anything that follows a long host-side pause behind the last forward (the process going on to something else)
last_marker = max((j for j, r in enumerate(step) if marker in r["Kernel_Name"]), default=0)
for j in range(last_marker + 1, len(step)):
    if int(step[j]["Start_Timestamp"]) - int(step[j - 1]["End_Timestamp"]) > 10_000_000:        # 10 ms
        step = step[:j]
        break
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
