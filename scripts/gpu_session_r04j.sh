#!/bin/bash
# per-phase shader-clock stamps of the fused step: shipped library against libndp_nmpc_hip_old.so
O=gpurun_out/r04j; mkdir -p $O
for rep in 1 2; do
for v in new old; do
  if [ $v = old ]; then export NDP_NMPC_LIB=$PWD/ndp_nmpc_qd_amd/libndp_nmpc_hip_old.so; else unset NDP_NMPC_LIB; fi
  python scripts/batch_stamps.py 1024 2>&1 | grep -v amdgpu.ids > $O/stamps_${v}_$rep.txt
done; done
paste -d'|' <(cut -c1-75 $O/stamps_new_1.txt) <(cut -c29-75 $O/stamps_old_1.txt)
paste -d'|' <(cut -c1-75 $O/stamps_new_2.txt) <(cut -c29-75 $O/stamps_old_2.txt)
