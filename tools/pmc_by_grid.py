#!/usr/bin/env python
"""Per (kernel, grid) HBM traffic from two rocprofv3 --pmc counter_collection.csv files (FETCH_SIZE x2 gfx950 correction)."""
import csv, sys
from collections import defaultdict
def load(path, counter):
    agg = defaultdict(lambda: [0, 0.0, 0.0]); order = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter: continue
        k = (r["Kernel_Name"].split("(")[0][-60:], r["Grid_Size"])
        if k not in agg: order.append(k)
        a = agg[k]; a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return agg, order
f, order = load(sys.argv[1], "FETCH_SIZE"); w, _ = load(sys.argv[2], "WRITE_SIZE")
for k in order:
    n, fk, us = f[k]; wk = w.get(k, [1, 0, 0])
    print(f"{k[0]:62s} grid {k[1]:>9s} x{n:3d} {us/n:7.1f} us  read {2*fk*1024/n/1e6:8.2f} MB  write {wk[1]*1024/max(wk[0],1)/1e6:8.2f} MB")
