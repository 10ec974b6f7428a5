#!/usr/bin/env python
"""The kernels around a step boundary from a rocprofv3 --kernel-trace csv: everything that starts within [-pre, +post] us of the
end of an adamw_kernel launch, with queue, start and end relative to that moment.  usage: step_head.py trace.csv [k-th adamw from the end] [pre] [post]"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0][-60:]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r[3] or "adamw_rest_kernel" in r[3]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
pre, post = (float(sys.argv[3]) if len(sys.argv) > 3 else 50.0), (float(sys.argv[4]) if len(sys.argv) > 4 else 700.0)
t0 = rows[idx[-k]][1]
for s, e, q, n in rows:
    if -pre * 1e3 <= s - t0 <= post * 1e3:
        print(f"{(s - t0) * 1e-3:8.1f} -> {(e - t0) * 1e-3:8.1f} us  ({(e - s) * 1e-3:6.1f})  q{q}  {n}")
