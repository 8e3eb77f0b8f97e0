import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# measured pass = after the second-to-last elementwise "marker" kernel
idx = [i for i, r in enumerate(rows) if "elementwise" in r["Kernel_Name"] and "vn_" not in r["Kernel_Name"]]
start = idx[-2] if len(idx) >= 2 else 0
tot = 0
for r in rows[start:]:
    n = r["Kernel_Name"]
    if not n.startswith("void vn_") and not n.startswith("vn_"):
        print("---", n[:60]); continue
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += us
    print("%-46s grid=%8s wg=%4s  %9.1f us" % (n[:46], r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")), us))
print("total vn kernel us", tot)
