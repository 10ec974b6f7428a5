#!/bin/bash
# Copy what tools/refresh_profiles.sh wrote under gpurun_out/refresh/ over profiles/<round>_*:  tools/install_profiles.sh r03
set -e
R=${1:-r06}; S=gpurun_out/refresh; P=profiles
cp $S/bench.json                        $P/${R}_bench_e6d6_b128_bf16.json
cp $S/kernel_stats.csv                  $P/${R}_bench_e6d6_b128_bf16_kernel_stats.csv
cp $S/under_rocprof.json                $P/${R}_bench_e6d6_b128_bf16_under_rocprof.json
cp $S/pmc_traffic.json $S/pmc_traffic.txt $S/pmc_mfma.json $S/pmc_mfma.txt $P/ 2>/dev/null || true
for f in pmc_traffic.json pmc_traffic.txt pmc_mfma.json pmc_mfma.txt; do mv $P/$f $P/${R}_$f; done
cp $S/cfg4_bench_under_rocprof.json     $P/${R}_cfg4_len256_b32_bench_under_rocprof.json
cp $S/cfg4_kernel_stats.csv             $P/${R}_cfg4_len256_b32_kernel_stats.csv
cp $S/cfg4_pmc_traffic.json             $P/${R}_cfg4_len256_b32_pmc_traffic.json
cp $S/cfg4_pmc_traffic.txt              $P/${R}_cfg4_len256_b32_pmc_traffic.txt
cp $S/stage2_bench_under_rocprof.json   $P/${R}_stage2_b128_bench_under_rocprof.json
cp $S/stage2_kernel_stats.csv           $P/${R}_stage2_b128_kernel_stats.csv
cp $S/stage2_b128_pmc_traffic.json      $P/${R}_stage2_b128_pmc_traffic.json
cp $S/stage2_b128_pmc_traffic.txt       $P/${R}_stage2_b128_pmc_traffic.txt
for st in 1 2; do
  cp $S/b16_stage${st}_kernel_stats.csv $P/${R}_b16_stage${st}_kernel_stats.csv 2>/dev/null || true
  cp $S/b16_stage${st}_bench_under_rocprof.json $P/${R}_b16_stage${st}_bench_under_rocprof.json 2>/dev/null || true
done
cp $S/stack_timeline.txt $P/${R}_stack_timeline.txt
cp $S/batch_sweep.txt $P/${R}_batch_sweep.txt
ls -la $P | grep ${R}_
