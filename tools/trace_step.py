"""Print one training step of a `rocprofv3 --kernel-trace --output-format csv` run (kernel_trace.csv): start offset, duration, HW queue,
workgroups, kernel -- from one adamw launch covering the whole buffer (grid 8192) to the next.  usage: trace_step.py <csv> [step] [all]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
want = sys.argv[3] if len(sys.argv) > 3 else 'simnce_res_kernel'
idx = [i for i, r in enumerate(rows) if want in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
per = [idx[i] for i in range(0, len(idx), max(1, len(idx) // max(1, len(set(idx)))))]
a, b = idx[k], idx[k + 2] if want == 'simnce_res_kernel' else idx[k + 1]
t0 = int(rows[a]['Start_Timestamp'])
for i in range(a, b + 1):
    r = rows[i]
    n = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '').replace('at::native::', '')[:64]
    s = (int(r['Start_Timestamp']) - t0) / 1e3
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    print(f"{s:8.1f} {d:7.1f} q{r['Queue_Id']:>2s} g{int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):6d} {n}")
