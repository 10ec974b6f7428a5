#!/usr/bin/env python
"""Timeline summary of one train step from a rocprofv3 --kernel-trace csv: busy/idle/overlap time, per-queue sums, the
largest idle gaps.  usage: timeline.py kernel_trace.csv [step_index_from_end]"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0][-50:]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# steps are delimited by adamw_kernel launches
idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r[3] or "adamw_rest_kernel" in r[3]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lo, hi = idx[-k - 1] + 1, idx[-k] + 1
step = rows[lo:hi]
t0, t1 = step[0][0], max(r[1] for r in step)
print(f"step wall {1e-3*(t1-t0):.1f} us, {len(step)} kernels")
ev = []
for s, e, q, n in step: ev += [(s, 1), (e, -1)]
ev.sort()
depth = 0; last = t0; hist = {}
for t, d in ev:
    hist[depth] = hist.get(depth, 0) + (t - last); last = t; depth += d
for d in sorted(hist): print(f"  {d} kernels in flight: {hist[d]*1e-3:8.1f} us")
byq = {}
for s, e, q, n in step: byq.setdefault(q, [0, 0]); byq[q][0] += e - s; byq[q][1] += 1
for q, (t, c) in byq.items(): print(f"  queue {q}: {c} kernels, {t*1e-3:.1f} us kernel time")
# idle gaps
gaps = []; cur_end = step[0][1]
for s, e, q, n in step[1:]:
    if s > cur_end: gaps.append((s - cur_end, n))
    cur_end = max(cur_end, e)
gaps.sort(reverse=True)
print("  total idle", sum(g for g, _ in gaps) * 1e-3, "us in", len(gaps), "gaps; largest:")
for g, n in gaps[:12]: print(f"    {g*1e-3:6.1f} us before {n}")
# per kernel name totals
agg = {}
for s, e, q, n in step: a = agg.setdefault(n, [0, 0]); a[0] += e - s; a[1] += 1
for n, (t, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:30]: print(f"  {t*1e-3:8.1f} us {c:4d}x {n}")
if len(sys.argv) > 3:
    print("kernels longer than", sys.argv[3], "us:")
    import csv as _c
    full = {(int(r["Start_Timestamp"])): r for r in _c.DictReader(open(sys.argv[1]))}
    for s, e, q, n in step:
        if (e - s) * 1e-3 > float(sys.argv[3]):
            r = full[s]
            g = "x".join(str(r.get(k, "?")) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")) if "Grid_Size_X" in r else str(r.get("Grid_Size", "?"))
            print(f"   {(e-s)*1e-3:7.1f} us grid {g:>14s} q{q} {n}")
