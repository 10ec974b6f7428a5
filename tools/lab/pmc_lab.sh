#!/bin/bash
# PMC counters of one lab command, one counter group per pass (never together with hip/hsa traces).  usage: pmc_lab.sh <out> <cmd...>
out=$1; shift
cd /tmp; export TMPDIR=/tmp
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES"; do
  d=/tmp/pmc_$(echo $grp | tr ' ' '_')
  rm -rf $d
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -- "$@" > /dev/null 2> /tmp/pmc_err.txt || tail -3 /tmp/pmc_err.txt
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY' >> $out
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    if "dw" in k or "gemm" in k:
        print(k, {c: round(v / n[(k, c)]) for c, v in d.items()})
PY
done
