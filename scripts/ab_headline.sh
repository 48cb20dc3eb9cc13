#!/bin/bash
# A/B of the headline step on ONE box: the tree's development library (scripts/dev_kernel.sh) against a snapshot of another commit staged
# under .ab_base/ (git archive <commit> ndp_nmpc_qd_amd bench.py oracle | tar -x -C .ab_base, its own dev library as
# .ab_base/ndp_nmpc_qd_amd/libndp_nmpc_hip.so).  Alternating runs, 300 steps each.  gpurun -- 'bash scripts/ab_headline.sh [rounds]'
R=${1:-3}
ROOT=$PWD
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s value %.4g ms/step %.5f kernel_us %.3f' % (sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['kernel_us']))" "$1"; }
for i in $(seq $R); do
  (cd $ROOT && NDP_NMPC_LIB=$ROOT/ndp_nmpc_qd_amd/libndp_nmpc_hip_dev.so python3 bench.py --only-timed --steps 300 --warmup 30 --downwash-form fused 2>/dev/null | val new)
  (cd $ROOT/.ab_base && python3 bench.py --only-timed --steps 300 --warmup 30 --downwash-form fused 2>/dev/null | val base)
done
for i in $(seq $R); do
  (cd $ROOT && NDP_NMPC_LIB=$ROOT/ndp_nmpc_qd_amd/libndp_nmpc_hip_dev.so python3 bench.py --only-timed --steps 20 --warmup 5 2>/dev/null | val new20)
  (cd $ROOT/.ab_base && python3 bench.py --only-timed --steps 20 --warmup 5 2>/dev/null | val base20)
done
