"""Per-step kernel table from a rocprofv3 kernel_stats.csv: python tools/prof_table.py <csv> <steps-in-run>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = sum(float(r["TotalDurationNs"]) for r in rows) / n / 1e6
for r in rows[:top]:
    name = r["Name"].split("(")[0][:58]
    ms = float(r["TotalDurationNs"]) / n / 1e6
    print(f"{name:60s} {int(r['Calls'])/n:6.1f}/step {ms:7.3f} ms  avg {float(r['AverageNs'])/1e3:8.1f} us")
print(f"total {tot:.3f} ms/step")
