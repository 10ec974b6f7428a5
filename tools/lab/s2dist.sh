export MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
run() { python bench.py --stage 2 --steps 10 --warmup 5 --no-cpu-baseline --no-extra "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], (d.get('comm') or {}).get('exposed_comm_ms_per_step'))"; }
echo nodist timer5; run --timer-every 5
echo dist timer5; TAN_FORCE_DIST=1 run --timer-every 5
echo dist timer20; TAN_FORCE_DIST=1 run --timer-every 20
echo dist notimer; TAN_FORCE_DIST=1 run --no-kernel-timer
