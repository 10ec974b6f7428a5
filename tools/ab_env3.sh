#!/bin/bash
# ab_env2.sh with extra bench arguments (e.g. --stage 2, --batch 16):  tools/ab_env3.sh VAR A B rounds steps [bench args...]
VAR=$1; A=$2; B=$3; ROUNDS=$4; STEPS=$5; shift 5
for r in $(seq 1 $ROUNDS); do
  for v in $A $B $B $A; do
    env $VAR=$v python bench.py --warmup 5 --steps $STEPS --no-cpu-baseline --no-extra --no-kernel-timer "$@" 2>/dev/null | python -c '
import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print("'$VAR=$v'", d["ms_per_step"], "ms/step  p50", d["step_ms_p50"], " min", d["step_ms_min"])'
  done
done
