"""Stress of QP_AUTO's active-set iterations on the device against the oracle's restatement of the same rule (qp_mode 0, kept sets
carried along) and against the oracle's interior-point loop at tol 1e-11 -- closed loop: the plant (oracle RK4) is driven by the
DEVICE's u0, recovering from large initial errors over `ticks` control periods, so the kept sets grow, shrink and empty again.
Seeds x workloads x {downwash off, on}.  gpurun -- 'python3 scripts/stress_active_set.py [seeds] [ticks] [B]'  -> one line per
case, then the worst figures."""
import sys

sys.path.insert(0, ".")
import numpy as np

import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import synth
from oracle import oracle as O

O.build()
n_seed = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 12
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
N = int(sys.argv[4]) if len(sys.argv) > 4 else 20          # 40: the five-slot kernels (config 5's shape), the kept set parked in LDS
NRTI = int(sys.argv[5]) if len(sys.argv) > 5 else 1
WQ = {"auto": None, "on": 1, "off": 2}[sys.argv[6] if len(sys.argv) > 6 else "auto"]      # the work list (producer / consumer launches) forced on / off
WORK = {"mixed": dict(pos_sigma=0.5, vel_sigma=1.0, quat_sigma=0.15), "hard": dict(pos_sigma=1.5, vel_sigma=3.0, quat_sigma=0.2),
        "fast": dict(pos_sigma=0.8, vel_sigma=1.5, quat_sigma=0.25, omega_range=(1.5, 2.5))}
blob = np.fromfile("ndp_nmpc_qd_amd/weights/downwash_sn4.bin", dtype="<f4")
worst = dict(twin=0.0, tight=0.0, mism=0, bad_dev=0, bad_tight=0, ipm=0, sweeps=0, n=0, con=0)
for seed in range(n_seed):
    for name, kw in WORK.items():
        for dw in (False, True):
            b = synth.make_batch(B, N=N, seed=1000 + seed, downwash=dw, **kw)
            eng = ndp.BatchedNMPC(B, N=N, n_rti=NRTI, disturbance=dw, **({} if WQ is None else {"work_queue": WQ}))
            eng.reset(b["xr"], b["ur"])
            twin = O.default_cfg(N=N, n_rti=NRTI, use_fd=dw)
            twin.qp_mode = 0
            tight = O.default_cfg(N=N, n_rti=NRTI, use_fd=dw)
            tight.tol = 1e-11
            Xo, Uo = b["xr"].copy(), b["ur"].copy()
            acto = np.zeros((B, N, 4), dtype=np.int8)
            x = b["x0"].copy()
            e_twin = e_tight = 0.0
            mism = bad_dev = bad_tight = n_ipm = sw_max = con = 0
            for t in range(ticks):
                bt = synth.make_batch(B, N=N, seed=1000 + seed, downwash=dw, t0=0.02 * t, **kw)
                f = None
                kwu = {}
                if dw:
                    kwu = dict(other=bt["other"], ego_xy=x[:, 0:2].copy())
                    f = O.downwash_batch(blob, bt["other"], bt["xr"], x[:, 0:2].copy())
                Xp, Up = Xo.copy(), Uo.copy()
                u0, X, U, st, it = eng.update(x, bt["xr"], bt["ur"], raise_on_status=False, full=True, **kwu)
                sw, act = eng.active_set()
                uo, sto, ito, swo = O.step_batch_as(twin, x, bt["xr"], bt["ur"], f, Xo, Uo, acto)
                ui, sti, _ = O.step_batch(tight, x, bt["xr"], bt["ur"], f, Xp, Up)
                ok = (st == 0) & (sto == 0)
                mism += int((st != sto).sum() + (sw != swo).sum() + (it != ito).sum() + (act != acto).any(axis=(1, 2)).sum())
                if ok.any():
                    e_twin = max(e_twin, float(np.max(np.abs(u0[ok] - uo[ok]) / np.maximum(1.0, np.abs(uo[ok])))),
                                 float(np.max(np.abs(U[ok] - Uo[ok]))))
                okt = ok & (sti == 0)
                if okt.any():
                    e_tight = max(e_tight, float(np.max(np.abs(u0[okt] - ui[okt]) / np.maximum(1.0, np.abs(ui[okt])))))
                bad_dev += int((st != 0).sum())
                bad_tight += int(((sti != 0) & (st == 0)).sum())
                n_ipm += int((it > 0).sum())
                sw_max = max(sw_max, int(sw.max()))
                con += int(act.any(axis=(1, 2)).sum())
                # the oracle continues from the DEVICE's iterate (a failed instance would otherwise drift apart and hide later ticks)
                Xo[:], Uo[:] = X, U
                acto[:] = act
                ff = np.zeros((B, 3)) if f is None else f[:, 0, :].astype(np.float64)
                x = O.plant_step(twin, x.copy(), u0, ff, 0.02)
            print(f"seed {seed} {name:5s} downwash {int(dw)}: twin {e_twin:.1e} tight-ipm {e_tight:.1e} mismatches {mism} "
                  f"not solved dev {bad_dev} (tight ipm failed where dev solved: {bad_tight}) ipm fallbacks {n_ipm} max sweeps {sw_max} "
                  f"constrained {con / (ticks * B):.3f}", flush=True)
            worst["twin"] = max(worst["twin"], e_twin); worst["tight"] = max(worst["tight"], e_tight)
            worst["mism"] += mism; worst["bad_dev"] += bad_dev; worst["bad_tight"] += bad_tight; worst["ipm"] += n_ipm
            worst["sweeps"] = max(worst["sweeps"], sw_max); worst["n"] += ticks * B; worst["con"] += con
            del eng
print("TOTAL", worst)
