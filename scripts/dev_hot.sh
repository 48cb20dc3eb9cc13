#!/bin/bash
# Static A/B of the headline kernel's HOT path (everything up to the end of the first Riccati sweep) after a change elsewhere in the
# wave program: development build (scripts/dev_kernel.sh), then registers + the accumulation-register / lane-spill traffic the
# allocator put into the hot region.  scripts/dev_hot.sh [hipcc flags]
set -e
bash "$(dirname "$0")/dev_kernel.sh" "$@" 2>/dev/null
cd "$(dirname "$0")/.."
python3 - <<'PY'
from ndp_nmpc_qd_amd import isa_inspect as I
import collections
co = I.CodeObject("ndp_nmpc_qd_amd/libndp_nmpc_hip_dev.so"); k = co.kernels()
for fused in (True, False):
    n = I.rti_kernel_name(3, 4, fused, 20); d = co.disassemble(n)
    idx = [i for i, l in enumerate(d) if "mfma" in l]
    cl = []; s = p = idx[0]
    for i in idx[1:]:
        if i - p > 150: cl.append((s, p)); s = i
        p = i
    cl.append((s, p))
    end = cl[1 if fused else 0][1] + 150        # fused: cluster 0 is the network tile, cluster 1 the first sweep
    ops = collections.Counter(l.split()[0] for l in d[:end] if l.strip())
    print("fused" if fused else "plain", k[n], "instr", len(d), "hot", end, "acc_read", ops["v_accvgpr_read_b32"], "acc_write", ops["v_accvgpr_write_b32"],
          "readlane", ops["v_readlane_b32"], "writelane", ops["v_writelane_b32"], "s_nop", ops["s_nop"])
PY
