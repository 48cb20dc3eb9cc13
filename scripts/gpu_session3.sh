#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
./scripts/ubench/publish_copy.bin | tee $O/s3_publish_copy.txt
( for th in 0 7; do NDP_HOST_PATH=zero NDP_PACK_THREADS=$th timeout 120 python scripts/host_path_rate.py 1024 2>/dev/null | grep "^B="; done
  NDP_HOST_PATH=dma1 NDP_PACK_THREADS=7 timeout 120 python scripts/host_path_rate.py 1024 2>/dev/null | grep "^B=" ) | tee $O/s3_host_path.txt
for cw in 0 30; do
  timeout 300 python bench.py --only-timed --steps 20 --warmup 5 --clock-warm-ms $cw 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('clock-warm $cw: value %.4g ms/step %.5f kernel_us %.2f extra %s'%(d['value'],d['ms_per_step'],d['roofline']['kernel_us'],d['config'].get('warmup_untimed_extra_steps')))"
done | tee $O/s3_clock_warm.txt
python scripts/batch_stamps.py 1024 2>&1 | grep -v amdgpu.ids | tee $O/s3_stamps.txt
python scripts/launch_ramp.py 2>&1 | grep -v amdgpu.ids | tee $O/s3_ramp.txt
