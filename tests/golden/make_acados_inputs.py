#!/usr/bin/env python3
"""Inputs of the acados cross-check (scripts/acados_crosscheck.py), committed so that the machine that HAS acados needs nothing but
this repository and a checkout of the reference: tests/golden/acados_inputs.npz.  Four cases x 3 control ticks (the iterate persists
across the ticks, as in the node):
   nmpc_nominal    192 instances  synth.make_batch, SURVEY 8d's figure-eights and noise          (NMPCBodyRateController)
   nmpc_perturbed  192 instances  0.5 m / 1 m/s / 0.15 initial errors: ~20 % hit an input bound   (NMPCBodyRateController)
   ndp_nominal     128 instances  + the downwash force of the reference's own network (fp32 values, mlp_golden-style: DownwashNN
                                  semantics through the CPU restatement of the net, gate on)       (NDPNMPCBodyRateController)
   ndp_fast        64 instances   omega in [1.5, 2.5] rad/s: large tilt, thrust near its bounds    (NDPNMPCBodyRateController)
Per case: <case>_x0 [3, B, 10], <case>_xr [3, B, 21, 10], <case>_ur [3, B, 20, 4] (, <case>_f [3, B, 21, 3] float32 values as float64).
Tick t's reference is the window at t0 = 0.02 t; x0 is re-drawn per tick (odometry noise)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ndp_nmpc_qd_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

O.build()
blob = np.fromfile(os.path.join(ROOT, "ndp_nmpc_qd_amd", "weights", "downwash_sn4.bin"), dtype="<f4")
out = {}
for case, B, seed, kw, dw in (("nmpc_nominal", 192, 11, {}, False),
                              ("nmpc_perturbed", 192, 12, dict(pos_sigma=0.5, vel_sigma=1.0, quat_sigma=0.15), False),
                              ("ndp_nominal", 128, 13, {}, True),
                              ("ndp_fast", 64, 14, dict(omega_range=(1.5, 2.5)), True)):
    x0, xr, ur, f = [], [], [], []
    for t in range(3):
        b = synth.make_batch(B, seed=synth.SEED0 + seed, downwash=dw, t0=0.02 * t, **kw)
        rng = np.random.default_rng(1000 * seed + t)                      # fresh odometry noise per tick, same trajectories
        x = b["xr"][:, 0, :].copy()
        x[:, 0:3] += rng.normal(0, kw.get("pos_sigma", 0.1), (B, 3))
        x[:, 3:6] += rng.normal(0, kw.get("vel_sigma", 0.2), (B, 3))
        x[:, 6:10] += rng.normal(0, kw.get("quat_sigma", 0.03), (B, 4))
        x[:, 6:10] /= np.linalg.norm(x[:, 6:10], axis=1, keepdims=True)
        x0.append(x); xr.append(b["xr"]); ur.append(b["ur"])
        if dw:
            f.append(O.downwash_batch(blob, b["other"], b["xr"], x[:, 0:2].copy()).astype(np.float64))
    out[case + "_x0"], out[case + "_xr"], out[case + "_ur"] = np.array(x0), np.array(xr), np.array(ur)
    if dw:
        out[case + "_f"] = np.array(f)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "acados_inputs.npz"), **out)
print({k: v.shape for k, v in out.items()})
