#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -k "ref_window or flat or list or rollout or plant or control_error" 2>&1 | tail -4 | tee $O/s6_tests.txt
for row in "ref_window --batch 1048576" "ref_window --batch 1024" "ref_list --batch 262144" "rollout --batch 1024 --steps 500"; do
  timeout 600 python scripts/bench_rows.py --row $row 2>/dev/null | tail -1
done | tee $O/s6_rows.txt
for i in 1 2 3; do timeout 400 python bench.py --only-timed --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-shaped: value %.4g ms/step %.5f kernel_us %.2f'%(d['value'],d['ms_per_step'],d['roofline']['kernel_us']))"; done | tee $O/s6_driver.txt
