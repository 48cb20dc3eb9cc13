#!/bin/bash
# A/B of BASELINE config 2 (batch 1024, no downwash: the unfused in-place kernel) on one box; staging as in scripts/ab_headline.sh
R=$PWD
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s value %.4g ms/step %.5f kernel_us %.3f' % (sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['kernel_us']))" "$1"; }
for i in 1 2; do
  (cd $R && NDP_NMPC_LIB=$R/ndp_nmpc_qd_amd/libndp_nmpc_hip_dev.so python3 bench.py --only-timed --steps 300 --warmup 30 --workload nmpc 2>/dev/null | val new)
  (cd $R && NDP_NMPC_LIB=$R/ndp_nmpc_qd_amd/libndp_nmpc_hip_dev.so python3 bench.py --only-timed --steps 300 --warmup 30 --workload nmpc --as-iter-max 0 2>/dev/null | val new_as0)
  (cd $R/.ab_base && python3 bench.py --only-timed --steps 300 --warmup 30 --workload nmpc 2>/dev/null | val base)
done
NDP_NMPC_LIB=$R/ndp_nmpc_qd_amd/libndp_nmpc_hip_dev.so python3 scripts/batch_stamps.py 1024 nmpc 2>/dev/null
(cd .ab_base && python3 ../scripts/batch_stamps.py 1024 nmpc 2>/dev/null)
