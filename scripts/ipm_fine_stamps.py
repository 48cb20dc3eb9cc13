"""Fine stamps of the FIRST interior-point iteration (B = 1, debug path, library built with -DNDP_FINE_STAMPS:
bash scripts/dev_kernel.sh -DNDP_FINE_STAMPS; NDP_NMPC_LIB=.../libndp_nmpc_hip_dev.so python scripts/ipm_fine_stamps.py)."""
import sys; sys.path.insert(0, '.')
import numpy as np
import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import synth, _lib
b = synth.make_batch(1, seed=3, downwash=False)
KT = _lib.lds_layout(20)["stamps"]
eng = ndp.BatchedNMPC(1, qp_mode=1)
for rep in range(3):
    eng.reset(b['xr'], b['ur'])
    u0, d = eng.update_debug(b['x0'], b['xr'], b['ur'])
    ft = d[KT + 16:KT + 32]
    names = ["factorisation sweep (riccati_sweep)", "slack / multiplier directions + step length (pass 0)", "centring + corrector right-hand side",
             "second solve (delta_sweep)", "directions + step length (pass 1)", "iterate / slack update, mu"]
    print("rep", rep, " | ".join(f"{n}: {int(ft[7 + i] - ft[6 + i])}" for i, n in enumerate(names)), "| iteration", int(ft[12] - ft[6]))
