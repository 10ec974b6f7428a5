#!/bin/bash
# step time against the batch size: ms per step and the same scaled to 128 videos (CU-slot effects of the panel kernels)
for b in "$@"; do
  python bench.py --batch $b --no-cpu-baseline --no-kernel-timer --no-extra --steps 30 --warmup 8 --settle-s 1 2>/dev/null | B=$b python -c "
import sys,json,os
b=int(os.environ['B']); d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b, d['ms_per_step'], round(d['ms_per_step']/b*128,3))"
done
