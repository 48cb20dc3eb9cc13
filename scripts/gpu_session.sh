#!/bin/bash
# One gpurun call of a development session: GPU tests, the driver-shaped bench run, and the one-GPU mechanics checks of the
# N > 1 paths.  Usage: gpurun --timeout 2400 -- 'bash scripts/gpu_session.sh <tag> [pytest -k expression]'
TAG=${1:-s}
KEXPR=${2:-}
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
if [ -n "$KEXPR" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -k "$KEXPR" 2>&1 | tail -40 > $O/${TAG}_tests.txt
else
  timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -40 > $O/${TAG}_tests.txt
fi
tail -5 $O/${TAG}_tests.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_driver.json 2> $O/${TAG}_bench_driver.err
tail -c 600 $O/${TAG}_bench_driver.err
timeout 400 python bench.py --only-timed --steps 300 --warmup 30 > $O/${TAG}_bench_300.json 2>> $O/${TAG}_bench_driver.err
timeout 400 python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline > $O/${TAG}_bench_cfg4.json 2> $O/${TAG}_bench_cfg4.err
tail -c 400 $O/${TAG}_bench_cfg4.err
NDP_BENCH_SAME_DEVICE=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 2 --steps 100 --warmup 10 --no-cpu-baseline > $O/${TAG}_bench_2proc_peer.json 2> $O/${TAG}_bench_2proc_peer.err
tail -c 400 $O/${TAG}_bench_2proc_peer.err
python - <<PY
import json
for f in ("bench_driver","bench_300","bench_cfg4","bench_2proc_peer"):
    try:
        d=json.loads(open("$O/${TAG}_%s.json"%f).read().strip().splitlines()[-1])
        r=d.get("roofline",{})
        print(f, "value %.4g ms/step %.5f kernel_us %.2f parity %s bad %s err %s"%(d["value"] or -1, d["ms_per_step"], r.get("kernel_us",0), d.get("parity_max_rel_vs_oracle"), d.get("instances_not_converged"), d.get("error")))
        for k in ("exchange","value_host_inclusive","mixed","ipm_always"):
            if k in d: print("   ",k, json.dumps(d[k])[:900])
    except Exception as e:
        print(f, "unreadable:", e)
PY
