#!/bin/bash
# ABBA of two BUILDS of the library inside the training step (60-step runs):  tools/ab_lib2.sh ab_libs/a.so ab_libs/b.so [rounds]
A=$1; B=$2; ROUNDS=${3:-2}
for r in $(seq 1 $ROUNDS); do
  for v in $A $B $B $A; do
    TAN_HIP_LIB=$PWD/$v python bench.py --warmup 5 --steps 60 --no-cpu-baseline --no-extra --no-kernel-timer 2>/dev/null | python -c '
import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print("'$v'", d["ms_per_step"], "ms/step  p50", d["step_ms_p50"])'
  done
done
