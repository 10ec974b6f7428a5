# per-kernel duration of simnce_dl_dvn_kernel under the TAN_DVN_DBG ablations (rocprofv3 --stats)
export PYTHONPATH=$PWD; R=$PWD; cd /tmp; export TMPDIR=/tmp
for d in "$@"; do
  rm -rf /tmp/prof_$d
  TAN_DVN_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$d -- python $R/tools/lab/dvn_time.py 1 > /dev/null 2>&1
  f=$(find /tmp/prof_$d -name "*kernel_stats.csv" | head -1)
  python - "$f" "$d" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "dl_dvn" in r["Name"] or "dl_kept" in r["Name"] or "pack_textT" in r["Name"]:
        print(f"dbg={sys.argv[2]} {r['Name'][:40]:40s} calls {r['Calls']} avg {float(r['AverageNs'])/1e3:.1f} us min {float(r['MinNs'])/1e3:.1f}")
PY
done
