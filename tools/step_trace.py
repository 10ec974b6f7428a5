#!/usr/bin/env python
"""One whole training step from a rocprofv3 --kernel-trace csv as a per-queue listing: start / end relative to the previous
optimizer launch's end, duration, queue, kernel.  usage: step_trace.py trace.csv [k-th step from the end]"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0][-56:]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r[3] or "adamw_rest_kernel" in r[3]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lo, hi = idx[-k - 1], idx[-k]
t0 = rows[lo][1]
qs = sorted({r[2] for r in rows[lo + 1:hi + 1]})
for s, e, q, n in rows[lo + 1:hi + 1]:
    print(f"{(s - t0) * 1e-3:8.1f} {(e - t0) * 1e-3:8.1f} ({(e - s) * 1e-3:6.1f}) {'    ' * qs.index(q)}q{q} {n}")
