#!/bin/bash
# quick A/B: shipped library against libndp_nmpc_hip_old.so, headline + NMPC-only + config 4 shard + interior point always
O=gpurun_out/r04k; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do
for v in new old; do
  if [ $v = old ]; then export NDP_NMPC_LIB=$PWD/ndp_nmpc_qd_amd/libndp_nmpc_hip_old.so; else unset NDP_NMPC_LIB; fi
  timeout 600 python bench.py --steps 300 --warmup 30 --only-timed > $O/b300_${v}_$rep.json 2>/dev/null
  timeout 600 python bench.py --steps 300 --warmup 30 --only-timed --workload nmpc > $O/nmpc_${v}_$rep.json 2>/dev/null
  timeout 600 python bench.py --steps 200 --warmup 30 --only-timed --qp-mode 1 > $O/ipm_${v}_$rep.json 2>/dev/null
  timeout 600 python bench.py --steps 200 --warmup 30 --only-timed --config 4 --formations 512 --placement formation > $O/c4_1536_${v}_$rep.json 2>/dev/null
done; done
unset NDP_NMPC_LIB
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04k/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round((d["value"] or d.get("value_unchecked") or 0)/1e6,2), round(d["ms_per_step"]*1e3,2))
PY
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
