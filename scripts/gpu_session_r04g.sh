#!/bin/bash
O=gpurun_out/r04g; mkdir -p $O
export TMPDIR=/tmp
python scripts/refine_probe.py 2>&1 | grep -v amdgpu.ids > $O/refine_probe.txt; cat $O/refine_probe.txt
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04g/bench20.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["ipm_always"]["value"], d["configs"].get("seconds"), {k:round(v["value"]/1e6,2) for k,v in d["configs"]["config4_one_gpu"].items() if isinstance(v,dict) and "value" in v})
PY
