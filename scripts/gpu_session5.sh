#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -s 2>&1 | grep -E "passed|failed|Error|error|scale 1|FAILED" | tail -30 | tee $O/s5_tests.txt
for row in "ref_window --batch 1048576" "ref_window --batch 1024" "ref_list --batch 262144" "ref_list --batch 1024" "throttle --batch 4194304" "throttle --batch 1024" "rollout --batch 1024 --steps 500"; do
  timeout 600 python scripts/bench_rows.py --row $row 2>/dev/null | tail -1
done | tee $O/s5_rows.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $O/s5_bench_driver.json 2> $O/s5_bench_driver.err
python - <<PY
import json
d=json.loads(open("$O/s5_bench_driver.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("driver-shaped: value %.4g ms/step %.5f kernel_us %.2f rocprof %s frac %.3f mismatch %s"%(d["value"],d["ms_per_step"],r["kernel_us"],r["kernel_us_rocprof"],r["frac"],r["profile_mismatch"]))
for k in ("value_host_inclusive","mixed","ipm_always","cpu_baseline","config1_single_vehicle"):
    print("  ",k,json.dumps(d.get(k))[:1200])
PY
