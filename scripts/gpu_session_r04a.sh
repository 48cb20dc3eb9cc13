#!/bin/bash
# round 4, session a: full GPU suite, the driver-shaped bench line (new: scaling_baseline legs), config 4 shards on one GPU
O=gpurun_out/r04a; mkdir -p $O
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_b1024_20.json 2> $O/bench_b1024_20.err; echo "bench rc=$?"
for ord in interleaved leaders_first; do
  timeout 300 python bench.py --config 4 --formations 512 --instance-order $ord --steps 200 --warmup 20 > $O/cfg4_f512_$ord.json 2> $O/cfg4_f512_$ord.err; echo "cfg4 512 $ord rc=$?"
  timeout 300 python bench.py --config 4 --formations 512 --placement formation --instance-order $ord --steps 200 --warmup 20 > $O/cfg4_f512_formation_$ord.json 2> $O/cfg4_f512_formation_$ord.err; echo "cfg4 512 formation $ord rc=$?"
  timeout 300 python bench.py --config 4 --instance-order $ord --steps 100 --warmup 10 > $O/cfg4_f4096_$ord.json 2> $O/cfg4_f4096_$ord.err; echo "cfg4 4096 $ord rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04a/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f.split('/')[-1], d.get("value"), d.get("ms_per_step"), {k:(v.get("value"),v.get("ms_per_step")) if isinstance(v,dict) and "value" in v else None for k,v in (d.get("exchange") or {}).items()})
    if "scaling_baseline" in d: print("   scaling_baseline", json.dumps(d["scaling_baseline"])[:600])
PY
