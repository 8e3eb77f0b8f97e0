import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 18]:
    print("%-64s calls=%6s total_ms=%9.2f avg_us=%9.1f pct=%6.2f" % (r["Name"][:64], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                                     float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
print("total ms", tot / 1e6)
