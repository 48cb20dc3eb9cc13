R=$PWD
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s value %.4g ms/step %.5f kernel_us %.3f' % (sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['kernel_us']))" "$1"; }
export NDP_NMPC_LIB=$R/ndp_nmpc_qd_amd/libndp_nmpc_hip_dev.so
for i in 1 2; do
python3 bench.py --only-timed --steps 300 --warmup 30 --downwash-form fused 2>/dev/null | val new_default
python3 bench.py --only-timed --steps 300 --warmup 30 --downwash-form fused --as-iter-max 0 2>/dev/null | val new_as0
(cd .ab_base && NDP_NMPC_LIB= python3 bench.py --only-timed --steps 300 --warmup 30 --downwash-form fused 2>/dev/null | val base)
done
