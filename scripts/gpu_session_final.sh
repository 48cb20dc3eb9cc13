#!/bin/bash
# what the driver runs at round end, in one call: GPU tests, smoke, the default bench line, the self-launched 2-rank mechanics check
O=gpurun_out/final; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
SECONDS=0
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $? in $SECONDS s"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/final/bench_default.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("%.2f M solves/s  %.2f us/step  frac %.4f (rocprof %.4f)  parity %.2e  not converged %d  mismatch %s"%(d["value"]/1e6, d["ms_per_step"]*1e3, r["frac"], r["frac_rocprof"] or 0, d["parity_max_rel_vs_oracle"], d["instances_not_converged"], r["profile_mismatch"]))
print("keys:", sorted(d.keys()))
PY
NDP_BENCH_SAME_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 40 --warmup 5 --no-cpu-baseline > $O/self2.json 2> $O/self2.err; echo "self-launch rc $?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/final/self2.json").read().strip().splitlines()[-1])
print(d["n_gpus"], "%.2f M"%(d["value"]/1e6), {m:(v.get("ok"), v.get("parity_max_rel_vs_oracle")) for m,v in d["exchange"].items() if isinstance(v,dict)})
PY
