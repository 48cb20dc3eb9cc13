#!/bin/bash
# A/B of development libraries on ONE box: scripts/ab_libs.sh "<bench.py flags>" lib1.so lib2.so ...  (three alternating rounds)
FL=$1; shift
ROOT=$PWD
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-12s value %.4g ms/step %.5f kernel_us %.3f' % (sys.argv[1], d['value'] or d.get('value_unchecked'), d['ms_per_step'], d['roofline']['kernel_us']))" "$1"; }
for i in 1 2 3; do
  for L in "$@"; do
    NDP_NMPC_LIB=$ROOT/ndp_nmpc_qd_amd/$L python3 bench.py --only-timed $FL 2>/dev/null | val $L
  done
done
