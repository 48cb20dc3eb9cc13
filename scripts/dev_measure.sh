#!/bin/bash
# Kernel-development measurement on the GPU box (through gpurun): phase stamps + timed steps of the headline workload with the
# development library (scripts/dev_kernel.sh).  Usage: gpurun -- 'bash scripts/dev_measure.sh [tag]'
export NDP_NMPC_LIB=$PWD/ndp_nmpc_qd_amd/libndp_nmpc_hip_dev.so
TAG=${1:-dev}
python3 scripts/batch_stamps.py 1024 2>&1 | grep -v amdgpu.ids > gpurun_out/stamps_$TAG.txt
cat gpurun_out/stamps_$TAG.txt
python3 bench.py --only-timed --steps 1000 --warmup 50 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step %.5f value %.3e kernel_us %.2f' % (d['ms_per_step'], d['value'], d['roofline']['kernel_us']))" | tee gpurun_out/bench_$TAG.txt
python3 bench.py --only-timed --steps 200 --warmup 20 --qp-mode 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ipm_always: ms_per_step %.5f value %.3e sweeps %.2f' % (d['ms_per_step'], d['value'], d['roofline']['riccati_sweeps_per_solve']))" | tee -a gpurun_out/bench_$TAG.txt
python3 bench.py --only-timed --steps 200 --warmup 20 --perturb mixed 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('mixed b1024: ms_per_step %.5f value %.3e ipm frac %.3f' % (d['ms_per_step'], d['value'], d['roofline']['frac_interior_point']))" | tee -a gpurun_out/bench_$TAG.txt
