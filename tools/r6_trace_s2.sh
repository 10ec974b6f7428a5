#!/bin/bash
# kernel trace of one stage-2 step (start offsets per kernel) + kernel stats, into gpurun_out/$1  (bench args: $2...)
R=$PWD; O=$R/gpurun_out/${1:-r6trace_s2}; shift; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr2 -- python $R/bench.py --stage 2 --steps 10 --warmup 3 --settle-s 0 --no-cpu-baseline --no-extra --no-kernel-timer "$@" > $O/under_rocprof.json 2> $O/under_rocprof.err
cp $(find /tmp/prof_tr2 -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
T=$(find /tmp/prof_tr2 -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_step.py $T 6 adamw_rest > $O/trace_step.txt 2>&1
tail -1 $O/under_rocprof.json | cut -c1-200
