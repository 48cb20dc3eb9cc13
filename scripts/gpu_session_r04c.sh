#!/bin/bash
# round 4, session c: refinement / robust paths + configs block
O=gpurun_out/r04c; mkdir -p $O
export TMPDIR=/tmp
python scripts/refine_probe.py > $O/refine_probe.txt 2>&1; cat $O/refine_probe.txt | grep -v amdgpu.ids
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_parity.py::test_active_state_bounds_at_a_tight_tolerance_on_the_device > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
SECONDS=0
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_b1024_20.json 2> $O/bench_b1024_20.err; echo "bench rc=$? seconds=$SECONDS"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04c/bench_b1024_20.json").read().strip().splitlines()[-1])
print("value", d["value"], d["ms_per_step"], "kernel_us", d["roofline"]["kernel_us"], d["roofline"]["kernel_us_dispatch_events"], "ipm", d.get("ipm_always",{}).get("value"))
print("mixed", {k:round(v.get("value")/1e6,2) for k,v in d.get("mixed",{}).items() if isinstance(v,dict) and "value" in v})
print("scaling_baseline", {k:(round(v.get("value",0)/1e6,2), v.get("ms_per_step")) for k,v in d["scaling_baseline"]["forms"].items()})
c=d.get("configs",{})
print("configs seconds", c.get("seconds"), c.get("error"))
print("config2", c.get("config2"))
for k,v in (c.get("config4_one_gpu") or {}).items(): print("config4", k, v if not isinstance(v,dict) else {kk:v[kk] for kk in v if kk in ("value","ms_per_step","parity_max_rel_vs_oracle","launch","error")})
for lab in ("nominal","perturbed"):
    for k,v in (c.get("config5",{}).get(lab) or {}).items(): print("config5", lab, k, {kk:v[kk] for kk in ("value","ms_per_step","max_rel_err_vs_oracle","status_nonzero","frac_interior_point")})
print("config1", d.get("config1_single_vehicle"))
PY
