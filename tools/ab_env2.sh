#!/bin/bash
# like ab_env.sh with longer runs and the per-step median:  tools/ab_env2.sh VAR A B [rounds] [steps]
VAR=$1; A=$2; B=$3; ROUNDS=${4:-2}; STEPS=${5:-80}
for r in $(seq 1 $ROUNDS); do
  for v in $A $B $B $A; do
    env $VAR=$v python bench.py --warmup 5 --steps $STEPS --no-cpu-baseline --no-extra --no-kernel-timer 2>/dev/null | python -c '
import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print("'$VAR=$v'", d["ms_per_step"], "ms/step  p50", d["step_ms_p50"], " min", d["step_ms_min"])'
  done
done
