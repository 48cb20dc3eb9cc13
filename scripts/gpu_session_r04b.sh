#!/bin/bash
# round 4, session b: after the register diet (no rti_kernel instantiation uses scratch): full GPU suite, driver-shaped bench, config 5 numbers
O=gpurun_out/r04b; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_b1024_20.json 2> $O/bench_b1024_20.err; echo "bench rc=$?"
timeout 600 python bench.py --steps 300 --warmup 30 --no-cpu-baseline > $O/bench_b1024_300.json 2> $O/bench_b1024_300.err; echo "bench300 rc=$?"
timeout 600 python scripts/config5_precision.py > $O/config5_precision.json 2> $O/config5_precision.err; echo "config5 rc=$?"
python - <<'PY'
import json
for f in ("bench_b1024_20","bench_b1024_300"):
    try:
        d=json.loads(open("gpurun_out/r04b/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel_us"], d["roofline"]["kernel_us_dispatch_events"], "ipm", d.get("ipm_always",{}).get("value"), "mixed", {k:v.get("value") for k,v in d.get("mixed",{}).items() if isinstance(v,dict) and "value" in v})
    except Exception as e: print(f, "unreadable", e)
try:
    d=json.loads(open("gpurun_out/r04b/config5_precision.json").read().strip().splitlines()[-1])
    for lab in ("nominal","perturbed"):
        print(lab, {k:(round(v.get("solves_per_s",0)/1e6,2), v.get("max_rel_err_vs_oracle")) for k,v in d[lab].items() if isinstance(v,dict)})
except Exception as e: print("config5 unreadable", e)
PY
