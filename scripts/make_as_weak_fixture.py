"""tests/golden/as_weak_multiplier_cases.npz from gpurun_out/stress_as_dump.npz (scripts/stress_as_dump.py on a GPU box): QPs out
of closed-loop recoveries on which round 6's FIRST active-set form (multiplier read off du - d at as_gamma = 1e12: right to ~1e-3)
kept a pin with a negative multiplier or cycled into the interior-point loop -- 1e-4 .. 1.6e-2 off the QP's solution with status 0.
Inputs of one control step each (x0, xr, ur, the iterate X, U and the kept set in front of it) + the step's exact solution (dense KKT
active-set iteration, tests/ref_numpy.py).  No reference code involved: the inputs are this repo's synthetic workload."""
import sys

sys.path.insert(0, ".")
import numpy as np

from oracle import oracle as O
from tests import ref_numpy as R

O.build()
d = np.load("gpurun_out/stress_as_dump.npz", allow_pickle=True)
cfg = O.default_cfg()
n = len(d["inst"])
Xex, Uex, nact = [], [], []
for k in range(n):
    qp = O.linearize(cfg, d["x0"][k], d["xr"][k], d["ur"][k], None, d["Xp"][k], d["Up"][k])
    dxa, dua, active = R.active_set_solve(qp)
    Xex.append(d["Xp"][k] + dxa)
    Uex.append(d["Up"][k] + dua)
    nact.append(len(active))
np.savez_compressed("tests/golden/as_weak_multiplier_cases.npz",
                    x0=d["x0"].astype(np.float64), xr=d["xr"].astype(np.float64), ur=d["ur"].astype(np.float64),
                    X=d["Xp"].astype(np.float64), U=d["Up"].astype(np.float64), act=d["actp"].astype(np.int8),
                    X_exact=np.array(Xex), U_exact=np.array(Uex), n_active=np.array(nact, dtype=np.int32),
                    origin=np.array([f"seed {s} {w} tick {t} inst {i}" for s, w, t, i in zip(d["seed"], d["work"], d["tick"], d["inst"])]))
print(n, "cases", nact)
