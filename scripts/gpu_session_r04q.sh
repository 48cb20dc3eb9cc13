#!/bin/bash
# the whole-run deadline of bench.py: forced to fire (1 s), then a normal default run
O=gpurun_out/r04q; rm -rf $O; mkdir -p $O
timeout 300 python bench.py --steps 20 --warmup 5 --deadline-s 1 > $O/deadline.out 2> $O/deadline.err; echo "exit code $?" | tee $O/deadline.rc
grep "bench\]" $O/deadline.err
SECONDS=0
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/default.json 2> $O/default.err; echo "default run rc $? in $SECONDS s"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04q/default.json").read().strip().splitlines()[-1])
print("%.2f M"%(d["value"]/1e6), "%.2f us"%(d["ms_per_step"]*1e3), d["roofline"]["frac"], d["roofline"]["profile_mismatch"], list(d.keys()))
PY
