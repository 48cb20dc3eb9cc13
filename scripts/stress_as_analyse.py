"""CPU analysis of gpurun_out/stress_as_dump.npz: every dumped instance's QP solved exactly (dense KKT active-set iteration,
tests/ref_numpy.py) and compared with what the device, the oracle twin and the tol-1e-11 interior point returned."""
import sys

sys.path.insert(0, ".")
import numpy as np

from oracle import oracle as O
from tests import ref_numpy as R

O.build()
d = np.load("gpurun_out/stress_as_dump.npz", allow_pickle=True)
cfg = O.default_cfg()
n = len(d["inst"])
for k in range(n):
    qp = O.linearize(cfg, d["x0"][k], d["xr"][k], d["ur"][k], None, d["Xp"][k], d["Up"][k])
    try:
        dxa, dua, active = R.active_set_solve(qp)
        how = "as"
    except Exception as e:
        dxa, dua, active = R.pdas_solve(qp)
        how = "pdas"
    Uex = d["Up"][k] + dua
    e_dev = np.abs(d["U"][k] - Uex).max()
    e_twin = np.abs(d["Uo"][k] - Uex).max()
    e_ipm = np.abs(d["Ui"][k] - Uex).max()
    print(f"{k:2d} seed {d['seed'][k]} {d['work'][k]:5s} tick {d['tick'][k]:2d} inst {d['inst'][k]:4d} | dev st {d['st'][k]} it {d['it'][k]} sw {d['sw'][k]} pins {(d['act'][k] != 0).sum():2d} err {e_dev:.1e}"
          f" | twin st {d['sto'][k]} it {d['ito'][k]} sw {d['swo'][k]} pins {(d['acto'][k] != 0).sum():2d} err {e_twin:.1e} | ipm st {d['sti'][k]} it {d['iti'][k]} err {e_ipm:.1e} | exact({how}) active {len(active)} kept-in {(d['actp'][k] != 0).sum()}")
