"""Summarise a rocprofv3 --pmc run: per kernel name, launches and mean counter value per launch."""
import csv
import glob
import os
import sys
from collections import defaultdict

root, counter = sys.argv[1], sys.argv[2]
files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(lambda: [0, 0.0])
for f in files:
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"].split("(")[0][:70]
            a = acc[name]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
print(f"# {counter} per launch (raw counter units as reported by rocprofv3), files={len(files)}")
for name, (n, tot) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:70s} launches={n:6d} mean={tot / n:14.1f} total={tot:16.1f}")
