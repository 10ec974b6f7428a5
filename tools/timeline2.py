#!/usr/bin/env python
"""Per-queue activity segments of one step from a rocprofv3 kernel-trace csv (kernels overlap across queues here)."""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0][-40:]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r[3]]
lo, hi = idx[-2] + 1, idx[-1] + 1
step = rows[lo:hi]; t0 = step[0][0]
gap_us = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
for q in sorted({r[2] for r in step}):
    ks = [r for r in step if r[2] == q]
    print(f"queue {q}: {len(ks)} kernels, busy {sum(e-s for s,e,_,_ in ks)*1e-3:.0f} us, span {1e-3*(ks[0][0]-t0):.0f}..{1e-3*(ks[-1][1]-t0):.0f} us")
    seg_s, seg_e, n, first = ks[0][0], ks[0][1], 1, ks[0][3]
    small = 0.0
    for s, e, _, nm in ks[1:]:
        if (s - seg_e) * 1e-3 > gap_us:
            print(f"    {1e-3*(seg_s-t0):8.0f} .. {1e-3*(seg_e-t0):8.0f}  {n:3d} kernels from {first:40s} then idle {1e-3*(s-seg_e):6.0f} us")
            seg_s, n, first = s, 0, nm
        else:
            small += max(0, s - seg_e) * 1e-3
        seg_e = max(seg_e, e); n += 1
    print(f"    {1e-3*(seg_s-t0):8.0f} .. {1e-3*(seg_e-t0):8.0f}  {n:3d} kernels from {first}")
    print(f"    sum of small gaps (<= {gap_us} us): {small:.0f} us")
