#!/usr/bin/env python
"""Per-kernel summary (calls, total/avg/min/max duration) from a rocprofv3 rocpd sqlite database -- the same table
`rocprofv3 --stats` prints, for runs whose output format was the default .db.  Usage: rocpd_stats.py results.db [top]"""
import re
import sqlite3
import sys


def main(path, top=40):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    scols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
    q = (f"select s.{name_col}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
         f"from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.{name_col} order by 3 desc")
    rows = cur.execute(q).fetchall()
    total = sum(r[2] for r in rows)
    print(f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for name, n, tot, mn, mx in rows[:top]:
        short = re.sub(r"\(.*", "", name)[:90]
        print(f"{short:90s} {n:7d} {tot / 1e6:10.3f} {tot / n / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * tot / total:6.2f}")
    print(f"{'TOTAL':90s} {sum(r[1] for r in rows):7d} {total / 1e6:10.3f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
