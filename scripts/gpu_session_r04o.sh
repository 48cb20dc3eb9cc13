#!/bin/bash
O=gpurun_out/r04o; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do for nb in 2 3; do
NDP_BENCH_GATHER_BUFS=$nb timeout 600 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-configs --exchange rccl --graph-exchange off > $O/b1024_nb${nb}_$rep.json 2> $O/b.err
NDP_BENCH_GATHER_BUFS=$nb timeout 600 python bench.py --config 4 --formations 512 --steps 200 --warmup 20 --no-cpu-baseline --exchange rccl --graph-exchange off > $O/c4_1536_nb${nb}_$rep.json 2> $O/c4.err
NDP_BENCH_GATHER_BUFS=$nb timeout 600 python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline --exchange rccl --graph-exchange off > $O/c4_12288_nb${nb}_$rep.json 2>> $O/c4.err
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04o/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,"ERR",e); continue
    b=d.get("scaling_baseline",{}).get("forms") or d.get("exchange")
    print(f.split('/')[-1], {m:("%.2f us"%(v["ms_per_step"]*1e3), "%.2f M"%(v["value"]/1e6), v.get("ok"), v.get("parity_max_rel_vs_oracle")) for m,v in b.items() if isinstance(v,dict) and "ms_per_step" in v})
PY
tail -n 3 $O/c4.err
timeout 600 python -m pytest tests -m gpu -q -x -k "rccl or exchange or gather or tracked" 2>&1 | tail -2
