#!/bin/bash
# batch sweep incl. the small-batch points of SURVEY 8(d) config 3 (stage 1 and stage 2), and host issue time per step at B = 16
echo "# stage 1: B  ms/step  ms/step scaled to 128 videos"
bash tools/batch_sweep.sh "$@"
echo "# stage 2 (co-training): B  ms/step"
for b in 16 32 128; do
  python bench.py --stage 2 --batch $b --no-cpu-baseline --no-kernel-timer --no-extra --steps 30 --warmup 8 --settle-s 1 2>/dev/null | B=$b python -c "
import sys,json,os
b=int(os.environ['B']); d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b, d['ms_per_step'])"
done
echo "# host issue time per step (tools/host_ahead.py)"
B=16 python tools/host_ahead.py | tail -1
B=16 KIND=cotrain python tools/host_ahead.py | tail -1
B=128 python tools/host_ahead.py | tail -1
