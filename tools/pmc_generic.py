#!/usr/bin/env python
"""Per (kernel, grid) averages of arbitrary PMC counters from one rocprofv3 counter_collection.csv."""
import csv, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int); order = []
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"].split("(")[0][-58:], r["Grid_Size"])
    if k not in agg: order.append(k)
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == sys.argv[2]: cnt[k] += 1
names = sys.argv[2:]
for k in order:
    n = max(cnt[k], 1)
    print(f"{k[0]:60s} {k[1]:>9s} x{n:3d} " + " ".join(f"{c}={agg[k][c]/n:12.0f}" for c in names))
