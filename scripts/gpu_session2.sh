#!/bin/bash
# host-path variants + graph-launch overhead + config 4 peer re-measure
export TMPDIR=/tmp
O=gpurun_out
python - <<'PY' > $O/s2_cpuinfo.txt 2>&1
import os
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try: print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e: print(e)
PY
cat $O/s2_cpuinfo.txt
( for path in zero dma dma1; do for th in 0 3 7 15; do
    NDP_HOST_PATH=$path NDP_PACK_THREADS=$th timeout 120 python scripts/host_path_rate.py 1024 2>/dev/null | grep "^B="
done; done ) | tee $O/s2_host_path.txt
NDP_HOST_PATH=zero NDP_PACK_THREADS=7 timeout 120 python scripts/host_path_rate.py 4096 2>/dev/null | grep "^B=" | tee -a $O/s2_host_path.txt
timeout 300 python scripts/graph_overhead.py 2>/dev/null | tee $O/s2_graph_overhead.txt
timeout 400 python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(d['exchange'])[:1500])" | tee $O/s2_cfg4.txt
