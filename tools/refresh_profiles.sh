#!/bin/bash
# Regenerates profiles/ on an MI355X box: run as  gpurun -- 'bash tools/refresh_profiles.sh'  (writes under gpurun_out/refresh/);
# then `bash tools/install_profiles.sh r06` copies gpurun_out/refresh/* over profiles/r06_*.  Counters are collected in their own passes (--pmc with --kernel-trace only).
set -x
R=$PWD; O=$R/gpurun_out/refresh; rm -rf $O; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $R/bench.py --steps 10 --warmup 3 --settle-s 0 --no-cpu-baseline --no-extra > $O/under_rocprof.json 2> $O/under_rocprof.err
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python $R/bench.py --steps 10 --warmup 5 --settle-s 0 --no-cpu-baseline --no-kernel-timer --no-extra > /dev/null 2> $O/pmc_$c.err
done
python $R/tools/pmc_summary.py $(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json > $O/pmc_traffic.txt
# MFMA utilisation (north_star: "rocprof-reported ... MFMA utilisation")
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_mfma -- python $R/bench.py --steps 10 --warmup 5 --settle-s 0 --no-cpu-baseline --no-kernel-timer --no-extra > /dev/null 2> $O/pmc_mfma.err
rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d /tmp/pmc_mfmau -- python $R/bench.py --steps 10 --warmup 5 --settle-s 0 --no-cpu-baseline --no-kernel-timer --no-extra > /dev/null 2> $O/pmc_mfmau.err
head -40 $(find /tmp/pmc_mfma -name "*counter_collection.csv" | head -1) > $O/pmc_mfma_raw_head.csv
python $R/tools/pmc_mfma.py $(find /tmp/pmc_mfma -name "*counter_collection.csv" | head -1) $O/pmc_mfma.json $(find /tmp/pmc_mfmau -name "*counter_collection.csv" | head -1) > $O/pmc_mfma.txt
# ---- BASELINE configs[3]: len=256 (joint L = 272), B=32 -- streamed attention kernels, HBM GB/s per kernel
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats4 -- python $R/bench.py --seq-len 256 --batch 32 --steps 10 --warmup 3 --settle-s 0 --no-cpu-baseline --no-kernel-timer > $O/cfg4_bench_under_rocprof.json 2> $O/cfg4_under_rocprof.err
cp $(find /tmp/prof_stats4 -name "*kernel_stats.csv" | head -1) $O/cfg4_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc4_$c -- python $R/bench.py --seq-len 256 --batch 32 --steps 10 --warmup 5 --settle-s 0 --no-cpu-baseline --no-kernel-timer > /dev/null 2> $O/cfg4_pmc_$c.err
done
python $R/tools/pmc_summary.py $(find /tmp/pmc4_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc4_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/cfg4_pmc_traffic.json > $O/cfg4_pmc_traffic.txt
# ---- stage 2 (configs[2] on one GPU)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats2 -- python $R/bench.py --stage 2 --steps 10 --warmup 3 --settle-s 0 --no-cpu-baseline --no-kernel-timer > $O/stage2_bench_under_rocprof.json 2> $O/stage2_under_rocprof.err
cp $(find /tmp/prof_stats2 -name "*kernel_stats.csv" | head -1) $O/stage2_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc2_$c -- python $R/bench.py --stage 2 --steps 10 --warmup 5 --settle-s 0 --no-cpu-baseline --no-kernel-timer > /dev/null 2> $O/stage2_pmc_$c.err
done
python $R/tools/pmc_summary.py $(find /tmp/pmc2_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc2_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/stage2_b128_pmc_traffic.json > $O/stage2_b128_pmc_traffic.txt
# ---- SURVEY 8(d) config 3's small-batch point (B_local = 16, both stages): the split-hidden / head-pair-split launches of round 6
for st in 1 2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b16_$st -- python $R/bench.py --stage $st --batch 16 --steps 20 --warmup 5 --settle-s 0 --no-cpu-baseline --no-kernel-timer --no-extra > $O/b16_stage${st}_bench_under_rocprof.json 2> $O/b16_stage${st}.err
  cp $(find /tmp/prof_b16_$st -name "*kernel_stats.csv" | head -1) $O/b16_stage${st}_kernel_stats.csv
done
# ---- where the step's time is (events per stream, host running ahead) and the step against the batch size
cd $R
python tools/stack_timeline.py > $O/stack_timeline.txt 2>/dev/null
python tools/stack_timeline.py --stage 2 >> $O/stack_timeline.txt 2>/dev/null
bash tools/batch_sweep2.sh 16 32 64 96 112 128 144 160 > $O/batch_sweep.txt 2>/dev/null
tail -1 $O/bench.json | cut -c1-400
