#!/bin/bash
# round-5 baseline: gpu tests, a fresh bench line, a kernel trace of one step (start offsets per kernel)
R=$PWD; O=$R/gpurun_out/r5base; rm -rf $O; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr -- python $R/bench.py --steps 10 --warmup 3 --settle-s 0 --no-cpu-baseline --no-extra --no-kernel-timer > $O/under_rocprof.json 2> $O/under_rocprof.err
cp $(find /tmp/prof_tr -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
T=$(find /tmp/prof_tr -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_step.py $T > $O/trace_step.txt 2>&1
cd $R
python tools/stack_timeline.py > $O/stack_timeline.txt 2>/dev/null
tail -1 $O/bench.json | cut -c1-600
tail -3 $O/pytest.log
