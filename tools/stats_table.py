"""Per-iteration table from a rocprofv3 kernel_stats.csv. Usage: python tools/stats_table.py <csv> <iterations>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows) / n / 1e3
print(f"all kernels: {tot:.1f} us per iteration ({int(n)} iterations profiled)")
for r in rows[:32]:
    name = r["Name"].split("(")[0][:100]
    print(f"{name:100s} {int(r['Calls']):5d} {float(r['TotalDurationNs']) / n / 1e3:9.1f} us/iter {float(r['AverageNs']) / 1e3:9.1f} us avg")
