"""scripts/stress_active_set.py's closed loop (downwash off) with the inputs of every instance on which device, oracle twin and the
tol-1e-11 interior point disagree dumped to gpurun_out/stress_as_dump.npz (analysis on the CPU: scripts/stress_as_analyse.py)."""
import sys

sys.path.insert(0, ".")
import os

import numpy as np

import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import synth
from oracle import oracle as O

O.build()
n_seed, ticks, B = int(sys.argv[1]), int(sys.argv[2]), 1024
WORK = {"mixed": dict(pos_sigma=0.5, vel_sigma=1.0, quat_sigma=0.15), "hard": dict(pos_sigma=1.5, vel_sigma=3.0, quat_sigma=0.2),
        "fast": dict(pos_sigma=0.8, vel_sigma=1.5, quat_sigma=0.25, omega_range=(1.5, 2.5))}
rec = []
# AS_DUMP_LIST=seed:work:tick:inst,... -- instances dumped whatever the comparison says (regression fixtures)
want = {tuple(w.split(':')) for w in os.environ.get('AS_DUMP_LIST', '').split(',') if w}
for seed in range(n_seed):
    for name, kw in WORK.items():
        b = synth.make_batch(B, seed=1000 + seed, **kw)
        eng = ndp.BatchedNMPC(B)
        eng.reset(b["xr"], b["ur"])
        twin = O.default_cfg()
        twin.qp_mode = 0
        tight = O.default_cfg()
        tight.tol = 1e-11
        Xo, Uo = b["xr"].copy(), b["ur"].copy()
        acto = np.zeros((B, 20, 4), dtype=np.int8)
        x = b["x0"].copy()
        for t in range(ticks):
            bt = synth.make_batch(B, seed=1000 + seed, t0=0.02 * t, **kw)
            Xp, Up, actp = Xo.copy(), Uo.copy(), acto.copy()
            u0, X, U, st, it = eng.update(x, bt["xr"], bt["ur"], raise_on_status=False, full=True)
            sw, act = eng.active_set()
            uo, sto, ito, swo = O.step_batch_as(twin, x, bt["xr"], bt["ur"], None, Xo, Uo, acto)
            Xi, Ui = Xp.copy(), Up.copy()
            ui, sti, iti = O.step_batch(tight, x, bt["xr"], bt["ur"], None, Xi, Ui)
            d_twin = np.max(np.abs(U - Uo), axis=(1, 2))
            d_tight = np.max(np.abs(u0 - ui) / np.maximum(1.0, np.abs(ui)), axis=1)
            pick = (st != sto) | (sw != swo) | (it != ito) | (act != acto).any(axis=(1, 2)) | (d_twin > 1e-7) | ((sti == 0) & (st == 0) & (d_tight > 1e-6))
            for (ws, ww, wt, wi) in want:
                if int(ws) == seed and ww == name and int(wt) == t:
                    pick[int(wi)] = True
            for i in np.flatnonzero(pick):
                rec.append(dict(seed=seed, work=name, tick=t, inst=i, x0=x[i].copy(), xr=bt["xr"][i].copy(), ur=bt["ur"][i].copy(), Xp=Xp[i], Up=Up[i], actp=actp[i],
                                U=U[i].copy(), X=X[i].copy(), st=st[i], it=it[i], sw=sw[i], act=act[i].copy(),
                                Uo=Uo[i].copy(), sto=sto[i], ito=ito[i], swo=swo[i], acto=acto[i].copy(), Ui=Ui[i].copy(), sti=sti[i], iti=iti[i]))
            Xo[:], Uo[:] = X, U
            acto[:] = act
            x = O.plant_step(twin, x.copy(), u0, np.zeros((B, 3)), 0.02)
        del eng
print(len(rec), "records")
os.makedirs("gpurun_out", exist_ok=True)
keys = rec[0].keys() if rec else []
np.savez("gpurun_out/stress_as_dump.npz", **{k: np.array([r[k] for r in rec]) for k in keys})
