#!/bin/bash
# ABBA of two builds of the library inside the training step on ONE box:  tools/ab_lib.sh ab_libs/old.so ab_libs/new.so [rounds] [bench args]
A=$1; B=$2; ROUNDS=${3:-2}; shift 3
for r in $(seq 1 $ROUNDS); do
  for v in $A $B $B $A; do
    ms=$(env TAN_HIP_LIB=$PWD/$v python bench.py --warmup 5 --steps 30 --no-cpu-baseline --no-extra "$@" 2>/dev/null | python -c 'import sys,json
for l in sys.stdin:
    if l.startswith("{"): print(json.loads(l)["ms_per_step"])')
    echo "$v $ms ms/step"
  done
done
