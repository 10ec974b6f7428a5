#!/bin/bash
# kernel time of the similarity sweep for each ablation build (tools/lab/build_variant.sh labN tan_simnce.hip -DTAN_SIM_LAB=N), by rocprofv3
R=$PWD; cd /tmp; export TMPDIR=/tmp
for n in "$@"; do
  rm -rf /tmp/prof_sl
  PYTHONPATH=$R TAN_HIP_LIB=$R/ab_libs/lab$n.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sl -- python $R/tools/lab/simres_time.py 0 > /dev/null 2>&1
  f=$(find /tmp/prof_sl -name "*kernel_stats.csv" | head -1)
  echo "lab $n: $(grep 'simnce_res_kernel<0>' $f | awk -F, '{printf "calls %s avg %.1f us min %.1f", $2, $4/1000, $6/1000}')"
done
