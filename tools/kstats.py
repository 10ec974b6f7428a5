#!/usr/bin/env python
"""Per-step table from a rocprofv3 `*kernel_stats.csv`: kstats.py <csv> <steps in the trace> [rows]."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
print(f"{sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / n:.3f} ms of kernels per step")
for r in rows[:top]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']) / n:7.1f} {float(r['AverageNs']) / 1e3:8.1f} us {float(r['TotalDurationNs']) / 1e6 / n:7.3f} ms")
