#!/bin/bash
# soak: 20 000 control ticks of the headline and of every exchange form on one GPU (protocol counters must stay clean)
O=gpurun_out/soak; rm -rf $O; mkdir -p $O
SECONDS=0
timeout 900 python bench.py --steps 20000 --warmup 20 --no-cpu-baseline --no-configs > $O/soak_default.json 2> $O/soak_default.err; echo "default rc $? at $SECONDS s"
timeout 900 python bench.py --config 4 --formations 512 --steps 20000 --warmup 20 --no-cpu-baseline > $O/soak_c4.json 2> $O/soak_c4.err; echo "config4 rc $? at $SECONDS s"
python - <<'PY'
import json
for f in ("soak_default","soak_c4"):
    d=json.loads(open("gpurun_out/soak/%s.json"%f).read().strip().splitlines()[-1])
    print(f, "%.2f M solves/s, %.2f us/step, parity %.2e, not converged %d"%(d["value"]/1e6, d["ms_per_step"]*1e3, d["parity_max_rel_vs_oracle"], d["instances_not_converged"]), d.get("watchdog"))
    for blk in ("downwash_forms","scaling_baseline","exchange"):
        b=d.get(blk)
        if not b: continue
        forms=b.get("forms", b)
        for m,v in forms.items():
            if isinstance(v,dict) and "value" in v: print("   ",blk,m,"%.2f M %.2f us"%(v["value"]/1e6, v["ms_per_step"]*1e3), "ok=%s"%v.get("ok"), v.get("peer_stats"), v.get("prefetch_stats"))
PY
