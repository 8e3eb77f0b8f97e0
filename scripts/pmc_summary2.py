"""per (kernel, grid) mean of every counter in a rocprofv3 --pmc csv run"""
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        key = (row["Kernel_Name"].split("(")[0][:48], row.get("Grid_Size", ""))
        a = acc[key][row["Counter_Name"]]
        a[0] += 1; a[1] += float(row["Counter_Value"])
for key, cs in acc.items():
    if "gemm" not in key[0]:
        continue
    print(key[0], "grid", key[1], " ".join(f"{c}={v[1] / v[0]:.4g}" for c, v in sorted(cs.items())))
