for p in 1 0 1 0; do for b in 16 32; do
TAN_PANEL=$p python bench.py --batch $b --no-cpu-baseline --no-kernel-timer --no-extra --steps 40 --warmup 10 --settle-s 1 2>/dev/null | P=$p B=$b python -c "
import sys,json,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('panel', os.environ['P'], 'B', os.environ['B'], d['ms_per_step'])"
done; done
