#!/bin/bash
# A/B on ONE box: the shipped library against the same source with -DNDP_DEV_NO_STIFF (no robust / refinement code in the loop)
O=gpurun_out/r04d; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
for v in stiff nostiff; do
  if [ $v = nostiff ]; then export NDP_NMPC_LIB=$PWD/ndp_nmpc_qd_amd/libndp_nmpc_hip_nostiff.so; else unset NDP_NMPC_LIB; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline --downwash-form fused --exchange peer > $O/b20_${v}_$rep.json 2> $O/b20_${v}_$rep.err
  timeout 600 python bench.py --steps 300 --warmup 30 --only-timed > $O/b300_${v}_$rep.json 2> $O/b300_${v}_$rep.err
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04d/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d["value"]/1e6,2), round(d["ms_per_step"]*1e3,2), "kernel_us", round(d["roofline"]["kernel_us"],2), round(d["roofline"]["kernel_us_dispatch_events"],2), "ipm", round(d.get("ipm_always",{}).get("value",0)/1e6,2))
    except Exception as e: print(f, e)
PY
