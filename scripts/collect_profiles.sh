#!/bin/bash
# After `gpurun -- 'bash scripts/gpu_r05.sh profile'` (and the other stages): copies the summaries out of gpurun_out/ (scratch) into
# profiles/ (tracked) under the names bench.py and the README tables use.   bash scripts/collect_profiles.sh r05
TAG=${1:-r05}
G=gpurun_out
cp -v $G/prof_$TAG/summary/${TAG}_kernel_stats_timed.csv profiles/${TAG}_kernel_stats_fused_b1024.csv
cp -v $G/prof_$TAG/summary/${TAG}_kernel_stats_all.csv profiles/${TAG}_kernel_stats_fused_b1024_all_launches.csv
cp -v $G/prof_$TAG/summary/${TAG}_pmc_rti_kernel.json profiles/${TAG}_pmc_rti_kernel.json
cp -v $G/prof_rows_$TAG/summary/${TAG}_kernel_stats_rows.csv $G/prof_rows_$TAG/summary/${TAG}_pmc_rows.json profiles/
[ -f $G/${TAG}_bench/bench_driver.json ] && cp -v $G/${TAG}_bench/bench_driver.json profiles/${TAG}_bench_b1024_fused.json
[ -f $G/${TAG}_bench/bench300.json ] && cp -v $G/${TAG}_bench/bench300.json profiles/${TAG}_bench_b1024_fused_300steps.json
[ -f $G/${TAG}_tick/tick_rate.txt ] && cp -v $G/${TAG}_tick/tick_rate.txt profiles/${TAG}_tick_rate.txt
[ -f $G/${TAG}_tests/tests.txt ] && cp -v $G/${TAG}_tests/tests.txt profiles/${TAG}_gpu_tests.txt
for f in hbm_ceiling tick_stamps; do [ -f $G/${TAG}_profile/$f.txt ] && cp -v $G/${TAG}_profile/$f.txt profiles/${TAG}_$f.txt; done
for f in cadence_50hz host_latency stress_parity; do [ -f $G/${TAG}_side/$f.txt ] && cp -v $G/${TAG}_side/$f.txt profiles/${TAG}_$f.txt; done
true
