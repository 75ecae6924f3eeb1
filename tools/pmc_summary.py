"""Summarises rocprofv3 --pmc CSV output (counter_collection.csv) per kernel: mean counter value per
dispatch. Usage: python tools/pmc_summary.py <dir> [name-filter]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "mvi::"
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if flt not in k:
            continue
        k = k.split("(")[0]
        c = acc[k][row["Counter_Name"]]
        c[0] += float(row["Counter_Value"])
        c[1] += 1
for k in sorted(acc):
    print(k)
    for c, (s, n) in sorted(acc[k].items()):
        print(f"    {c:28s} {s / n:16.1f}  (n={n})")
