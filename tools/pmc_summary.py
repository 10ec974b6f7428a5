#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counter unit = KiB).
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE under-reports wide coalesced reads by exactly 2x."""
import csv
import sys
from collections import defaultdict


def load(path, counter):
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return agg


def main(fetch_csv, write_csv, top=14):
    f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    rows = []
    for k in f:
        n, fk, us = f[k]
        wk = w.get(k, [0, 0.0, 0.0])[1]
        rows.append((us, k, n, 2 * fk * 1024 / n, wk * 1024 / max(w.get(k, [1])[0], 1), us / n))
    rows.sort(reverse=True)
    print(f"{'kernel':80s} {'calls':>6s} {'avg_us':>8s} {'read_MB':>9s} {'write_MB':>9s} {'HBM GB/s':>9s}")
    for us, k, n, rd, wr, avg in rows[:top]:
        print(f"{k[:80]:80s} {n:6d} {avg:8.1f} {rd / 1e6:9.2f} {wr / 1e6:9.2f} {(rd + wr) / avg / 1e3:9.0f}")
    if len(sys.argv) > 3:
        import json
        fam = [r for r in rows if any(t in r[1] for t in ("gemm_glds_kernel", "gemm_glds4_kernel", "gemm_dw256_kernel", "gemm_dw_grouped_kernel", "simnce_kernel", "simnce_res_kernel", "mlp_panel_kernel", "attnblk_fwd_kernel", "attnblk_bwd_kernel"))]
        calls = sum(r[2] for r in fam)
        out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB units, FETCH_SIZE x2 (gfx950 correction)",
               "mfma_gemm_family": {"launches": calls,
                                    "hbm_read_bytes_per_launch": sum(r[3] * r[2] for r in fam) / calls,
                                    "hbm_write_bytes_per_launch": sum(r[4] * r[2] for r in fam) / calls},
               "kernels": [{"kernel": r[1], "calls": r[2], "avg_us": r[5], "read_bytes": r[3], "write_bytes": r[4]} for r in rows[:top]]}
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
