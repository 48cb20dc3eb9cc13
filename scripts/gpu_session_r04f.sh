#!/bin/bash
O=gpurun_out/r04f; mkdir -p $O
export TMPDIR=/tmp
python scripts/batch_stamps.py 1024 2>&1 | grep -v amdgpu.ids > $O/stamps_new.txt
NDP_NMPC_LIB=$PWD/ndp_nmpc_qd_amd/libndp_nmpc_hip_nostiff.so python scripts/batch_stamps.py 1024 2>&1 | grep -v amdgpu.ids > $O/stamps_old_minors.txt
paste <(grep -E "backward|forward|whole|cost|linearize|end " $O/stamps_new.txt) <(grep -E "backward|forward|whole|cost|linearize|end " $O/stamps_old_minors.txt | cut -c28-)
python scripts/refine_probe.py 2>&1 | grep -v amdgpu.ids > $O/refine_probe.txt; cat $O/refine_probe.txt
for rep in 1 2; do
  timeout 600 python bench.py --steps 300 --warmup 30 --only-timed > $O/b300_new_$rep.json 2>/dev/null
  NDP_NMPC_LIB=$PWD/ndp_nmpc_qd_amd/libndp_nmpc_hip_nostiff.so timeout 600 python bench.py --steps 300 --warmup 30 --only-timed > $O/b300_old_$rep.json 2>/dev/null
  timeout 600 python bench.py --steps 200 --warmup 30 --only-timed --qp-mode 1 > $O/ipm_new_$rep.json 2>/dev/null
  NDP_NMPC_LIB=$PWD/ndp_nmpc_qd_amd/libndp_nmpc_hip_nostiff.so timeout 600 python bench.py --steps 200 --warmup 30 --only-timed --qp-mode 1 > $O/ipm_old_$rep.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04f/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d["value"]/1e6,2), round(d["ms_per_step"]*1e3,2))
PY
