#!/bin/bash
# Register study of ONE rti_kernel shape without building the library: device-only compile to assembly, then the kernel's
# resource lines.  Default: config 5's in-place kernel rti_kernel<5,2,false,40,0,2,0> (-DNDP_DEV_QMODE=2: the work-list consumer).
#   scripts/dev_regs.sh [-DNDP_DEV_QMODE=2] [other hipcc flags]
set -e
cd "$(dirname "$0")/../ndp_nmpc_qd_amd/csrc"
OUT=${TMPDIR:-/tmp}/ndp_dev_regs.s
hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-schedule-relaxed-occupancy=true -DNDP_DEV_N40_ONLY --cuda-device-only -S "$@" -o $OUT ndp_hip.hip 2>/dev/null
python3 - "$OUT" <<'PY'
import re, sys
t = open(sys.argv[1]).read()
for blk in re.split(r"\n  - \.agpr_count:", t)[1:]:
    blk = ".agpr_count:" + blk
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    if "rti_kernel" in g("name"):
        print(g("name")[19:52], "vgpr", g("vgpr_count"), "agpr", g("agpr_count"), "vgpr_spill", g("vgpr_spill_count"), "sgpr_spill", g("sgpr_spill_count"),
              "scratch", g("private_segment_fixed_size"))
n = t.count("scratch_load"), t.count("scratch_store"), t.count("v_accvgpr_read"), t.count("v_accvgpr_write")
print("scratch loads %d stores %d; accvgpr reads %d writes %d" % n)
PY
