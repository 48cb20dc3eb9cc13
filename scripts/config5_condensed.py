"""BASELINE config 5 as worded: "N = 40 horizon, 2 RTI iterations, fp32 vs bf16 MFMA on the CONDENSED QP, batch 4096" -- the study
mode qp_precision 5 / 6 (csrc/cond_qp.hpp) next to the fp64 Riccati product path and the round-1..5 sweep forms (qp_precision 3 / 4):
u0 error against the fp64 oracle, solves/s, how many QPs kept the condensed result.  python scripts/config5_condensed.py [B] [N]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import synth
from oracle import oracle as O

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
n_rti = 2
dev = torch.device("cuda", 0)
for name, kw in (("nominal", {}), ("perturbed", dict(pos_sigma=0.5, vel_sigma=1.0, quat_sigma=0.15))):
    b = synth.make_batch(B, N=N, seed=synth.SEED0 + 5, **kw)
    NS = min(B, 256)
    cfgo = O.default_cfg(N=N, n_rti=n_rti)
    cfgo.tol = 1e-11
    Xo, Uo = b["xr"][:NS].copy(), b["ur"][:NS].copy()
    uo, sto, _ = O.step_batch(cfgo, b["x0"][:NS], b["xr"][:NS], b["ur"][:NS], None, Xo, Uo)
    d = {k: torch.from_numpy(b[k]).to(dev) for k in ("x0", "xr", "ur")}
    for prec, label in ((0, "fp64 Riccati (product)"), (3, "fp32 sweeps"), (4, "bf16 sweeps"), (5, "condensed fp32"), (6, "condensed bf16")):
        eng = ndp.BatchedNMPC(B, N=N, n_rti=n_rti, qp_precision=prec, work_queue=2 if prec == 0 else 0)
        eng.reset(b["xr"], b["ur"])
        u0 = eng.update(b["x0"], b["xr"], b["ur"], raise_on_status=False)
        st, it = eng.status()
        ok = (st[:NS] == 0) & (sto == 0)
        err = float(np.max(np.abs(u0[:NS][ok] - uo[ok]) / np.maximum(1.0, np.abs(uo[ok]))))
        med = float(np.median(np.max(np.abs(u0[:NS][ok] - uo[ok]) / np.maximum(1.0, np.abs(uo[ok])), axis=1)))
        kept = eng.condensed_kept() if prec >= 5 else None
        u = torch.empty(B, 4, dtype=torch.float64, device=dev)
        eng.reset_device(d["xr"], d["ur"])
        for _ in range(3):
            eng.update_device(d["x0"], d["xr"], d["ur"], u)
        torch.cuda.synchronize()
        ta = time.perf_counter()
        n = 10
        for _ in range(n):
            eng.update_device(d["x0"], d["xr"], d["ur"], u)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - ta) / n
        print(f"{name:9s} N={N} B={B} {label:24s}: u0 err max {err:.2e} median {med:.2e} | bad {int((st != 0).sum())} ipm {float((it > 0).mean()):.3f}"
              + (f" | QPs kept condensed {float(kept.mean()):.2f} of {n_rti}" if kept is not None else "") + f" | {B / dt / 1e6:.3f} M solves/s ({dt * 1e3:.3f} ms/step)")
        eng.close()
