# the automatic work-list rule: nominal and mixed workloads at large batches, against the forced forms
O=gpurun_out/e1; rm -rf $O; mkdir -p $O
for rep in 1 2; do for wq in 0 1 2; do
for b in 2048 4096 16384; do
timeout 600 python bench.py --steps 200 --warmup 40 --batch $b --work-queue $wq --no-cpu-baseline --no-configs > $O/nom${b}_wq${wq}_$rep.json 2>/dev/null
done
timeout 600 python bench.py --steps 100 --warmup 40 --config 4 --placement formation --work-queue $wq --no-cpu-baseline > $O/c4f_wq${wq}_$rep.json 2>/dev/null
timeout 600 python bench.py --steps 100 --warmup 40 --batch 4096 --perturb mixed --work-queue $wq --no-cpu-baseline --no-configs > $O/mixed4096_wq${wq}_$rep.json 2>/dev/null
timeout 600 python bench.py --steps 100 --warmup 40 --batch 2048 --perturb mixed --work-queue $wq --no-cpu-baseline --no-configs > $O/mixed2048_wq${wq}_$rep.json 2>/dev/null
done; done
python - <<'PY'
import json,glob,collections
R=collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/e1/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: R[f.split('/')[-1].rsplit('_',1)[0]].append("ERR"); continue
    R[f.split('/')[-1].rsplit('_',1)[0]].append((round(d["ms_per_step"]*1e3,2), "%.1e"%(d.get("parity_max_rel_vs_oracle") or -1), d.get("instances_not_converged"), d["config"].get("work_queue")))
for k,v in sorted(R.items()): print(k,v)
PY
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
