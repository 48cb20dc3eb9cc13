#!/bin/bash
# Kernel-development build: only the reference configuration's two rti_kernel instantiations (N = 20, 1 RTI iteration, fused /
# unfused), ~25 s instead of ~3 min.  Output ndp_nmpc_qd_amd/libndp_nmpc_hip_dev.so (git-ignored, travels with gpurun); use it
# with NDP_NMPC_LIB=$PWD/ndp_nmpc_qd_amd/libndp_nmpc_hip_dev.so.  Extra hipcc flags: "$@".
set -e
cd "$(dirname "$0")/../ndp_nmpc_qd_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-schedule-relaxed-occupancy=true -DNDP_DEV_HEADLINE_ONLY -fPIC -shared "$@" \
    -o ../libndp_nmpc_hip_dev.so ndp_hip.hip
