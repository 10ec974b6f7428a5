#!/bin/bash
# A/B of an environment switch inside the training step, interleaved on ONE box (boxes differ by +-3 %):
#   tools/ab_env.sh VAR A B [rounds] [bench args...]      e.g. tools/ab_env.sh TAN_ATTN_PANEL 0 1 2 --steps 30
VAR=$1; A=$2; B=$3; ROUNDS=${4:-2}; shift 4
for r in $(seq 1 $ROUNDS); do
  for v in $A $B $B $A; do
    ms=$(env $VAR=$v python bench.py --warmup 5 --steps 30 --no-cpu-baseline --no-extra "$@" 2>/dev/null | python -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"): print(json.loads(l)["ms_per_step"])')
    echo "$VAR=$v $ms ms/step"
  done
done
