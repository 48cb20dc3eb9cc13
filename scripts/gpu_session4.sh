#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
( for p in zero dma1; do NDP_HOST_PATH=$p NDP_PACK_THREADS=7 timeout 120 python scripts/host_path_rate.py 1024 2>/dev/null | grep -E "^B=|all loops"; done ) | tee $O/s4_host_path.txt
timeout 400 python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(d['exchange'])[:1500])" | tee $O/s4_cfg4.txt
for w in 4 2; do
  NDP_DEV_WAVES=$w timeout 300 python bench.py --only-timed --steps 300 --warmup 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('waves $w: value %.4g ms/step %.5f kernel_us %.2f'%(d['value'],d['ms_per_step'],d['roofline']['kernel_us']))"
  NDP_DEV_WAVES=$w python scripts/batch_stamps.py 1024 2>&1 | grep -v amdgpu.ids
done | tee $O/s4_waves.txt
timeout 600 python -m pytest tests -m gpu -q --timeout 600 -k "peer" 2>&1 | tail -5 | tee $O/s4_tests.txt
