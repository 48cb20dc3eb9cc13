#!/bin/bash
# A/B of library variants ndp_nmpc_qd_amd/libndp_v*.so against the shipped one: headline, NMPC-only, interior point always
O=gpurun_out/r04l; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do
for v in new $(ls ndp_nmpc_qd_amd/libndp_v*.so | sed 's/.*libndp_\(v.*\)\.so/\1/'); do
  if [ $v = new ]; then unset NDP_NMPC_LIB; else export NDP_NMPC_LIB=$PWD/ndp_nmpc_qd_amd/libndp_$v.so; fi
  timeout 600 python bench.py --steps 300 --warmup 30 --only-timed > $O/b300_${v}_$rep.json 2>/dev/null
  timeout 600 python bench.py --steps 300 --warmup 30 --only-timed --workload nmpc > $O/nmpc_${v}_$rep.json 2>/dev/null
  timeout 600 python bench.py --steps 200 --warmup 30 --only-timed --qp-mode 1 > $O/ipm_${v}_$rep.json 2>/dev/null
done; done
unset NDP_NMPC_LIB
python - <<'PY'
import json,glob,collections
R=collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r04l/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception: continue
    k=f.split('/')[-1].rsplit('_',1)[0]; R[k].append(round(d["ms_per_step"]*1e3,2))
for k,v in sorted(R.items()): print(k, v)
PY
