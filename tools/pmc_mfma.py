#!/usr/bin/env python
"""Per-kernel MFMA utilisation from one rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass (with --kernel-trace).
MfmaUtil = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max GRBM_GUI_ACTIVE x 1024 SIMDs) -- the derived_counters.xml formula (the file has
no gfx950 section; this is the gfx94x expression, stated in MI355X_MICROARCH.md).  SQ_VALU_MFMA_BUSY_CYCLES counts cycles
(32 per v_mfma_f32_32x32x16_bf16)."""
import csv
import json
import sys
from collections import defaultdict

SIMDS = 256 * 4


def derived(path):
    """kernel -> (calls, mean MfmaUtil) from a --pmc MfmaUtil pass (rocprofv3's own derived counter)"""
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == "MfmaUtil":
            a = agg[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return {k: (a[0], a[1] / a[0]) for k, a in agg.items()}


def main(path, out_json=None, derived_csv=None, top=16):
    disp = defaultdict(dict)          # dispatch id -> {counter: value, 'k': name, 'us': duration}
    for r in csv.DictReader(open(path)):
        d = disp[r["Dispatch_Id"]]
        d["k"] = r["Kernel_Name"]
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for d in disp.values():
        a = agg[d["k"]]
        a[0] += 1
        a[1] += d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        a[2] += d.get("GRBM_GUI_ACTIVE", 0.0)
        a[3] += d["us"]
    rows = sorted(((a[3], k, a[0], 100.0 * a[1] / (a[2] * SIMDS) if a[2] else 0.0, a[3] / a[0]) for k, a in agg.items()), reverse=True)
    dv = derived(derived_csv) if derived_csv else {}
    print(f"{'kernel':80s} {'calls':>6s} {'avg_us':>8s} {'busy/(active*1024) %':>21s} {'rocprofv3 MfmaUtil %':>21s}")
    for us, k, n, util, avg in rows[:top]:
        print(f"{k[:80]:80s} {n:6d} {avg:8.1f} {util:21.2f} {dv.get(k, (0, float('nan')))[1]:21.2f}")
    tot_busy = sum(a[1] for a in agg.values())
    tot_act = sum(a[2] for a in agg.values())
    print(f"all kernels, time-weighted: MFMA busy {100.0 * tot_busy / (tot_act * SIMDS):.1f} % of the SIMD-cycles the GPU was active")
    if out_json:
        json.dump({"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; util = busy / (active x 1024 SIMDs)",
                   "overall_mfma_busy_pct": 100.0 * tot_busy / (tot_act * SIMDS),
                   "kernels": [{"kernel": k, "calls": n, "avg_us": avg, "busy_over_active_x1024_pct": util,
                                "rocprofv3_MfmaUtil_pct": dv.get(k, (0, None))[1]} for us, k, n, util, avg in rows[:top]]},
                  open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, sys.argv[3] if len(sys.argv) > 3 else None)
