#!/bin/bash
# final-binary refresh of the side evidence: stress parity, widened rows
O=gpurun_out/r04r; rm -rf $O; mkdir -p $O
timeout 900 python scripts/stress_parity.py > $O/r04_stress_parity.txt 2>/dev/null; tail -n 5 $O/r04_stress_parity.txt
timeout 300 python scripts/bench_rows.py --row ref_window --batch 1048576 2>/dev/null | tail -n 1 > $O/r04_bench_row_f1_ref_window_b1M.json
timeout 300 python scripts/bench_rows.py --row ref_list --batch 262144 2>/dev/null | tail -n 1 > $O/r04_bench_row_f1_ref_list_b256k.json
timeout 300 python scripts/bench_rows.py --row throttle 2>/dev/null | tail -n 1 > $O/r04_bench_row_f3_throttle.json
timeout 300 python scripts/bench_rows.py --row rollout --batch 1024 2>/dev/null | tail -n 1 > $O/r04_bench_row_rollout.json
for f in $O/r04_bench_row*.json; do echo $f; cut -c1-400 $f; done
