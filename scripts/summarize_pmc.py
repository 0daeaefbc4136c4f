"""Aggregate a rocprofv3 counter_collection CSV per (kernel, counter): mean per dispatch."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0.0, 0])
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        k = row.get("Kernel_Name", "?")
        k = k.split("(")[0][-60:]
        key = (k, row.get("Counter_Name", "?"))
        acc[key][0] += float(row.get("Counter_Value", 0) or 0)
        acc[key][1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    print(f"{k:60s} {c:28s} dispatches={n:5d} mean={s / n:.6g}")
