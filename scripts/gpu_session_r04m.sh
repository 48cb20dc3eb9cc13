#!/bin/bash
# the peer_ahead exchange form: N = 1 baseline forms, config 4 shard on one GPU, two ranks time-slicing one GPU
O=gpurun_out/r04m; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-configs > $O/b_default.json 2> $O/b_default.err
timeout 600 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-configs > $O/b_default_300.json 2>> $O/b_default.err
timeout 600 python bench.py --config 4 --formations 512 --steps 200 --warmup 20 --no-cpu-baseline > $O/c4_1536.json 2> $O/c4_1536.err
timeout 900 python -m pytest tests -m gpu -q -x -k "peer or prefetch or ahead" 2>&1 | tail -2
NDP_BENCH_SAME_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 100 --warmup 10 --no-cpu-baseline > $O/self2.json 2> $O/self2.err
python - <<'PY'
import json
for f in ("b_default","b_default_300","c4_1536","self2"):
    try: d=json.loads(open("gpurun_out/r04m/%s.json"%f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    print(f, "%.2f M %.2f us"%(d["value"]/1e6, d["ms_per_step"]*1e3), d.get("watchdog"))
    for blk in ("scaling_baseline","exchange"):
        b=d.get(blk)
        if not b: continue
        forms=b.get("forms", b)
        for m,v in forms.items():
            if isinstance(v,dict): print("   ",blk,m, ("%.2f M %.2f us ok=%s par=%s"%(v["value"]/1e6, v["ms_per_step"]*1e3, v.get("ok"), v.get("parity_max_rel_vs_oracle")) if "value" in v else v), v.get("peer_stats"), v.get("prefetch_stats"), str(v.get("launch"))[:90])
PY
for f in $O/*.err; do tail -n 3 $f; done
