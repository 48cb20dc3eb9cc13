"""Cycle budget of the interior-point path: whole-program shader-clock span per instance against its iteration count
(B = 1024, every instance iterating).  Run on the GPU box: python scripts/ipm_stamps.py [nmpc]"""
import sys; sys.path.insert(0, '.')
import numpy as np
import torch
import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import dist as ndist, synth

B = 1024
downwash = (sys.argv[1] if len(sys.argv) > 1 else "ndp_downwash") == "ndp_downwash"
dev = torch.device("cuda", 0)
for name, b in (("nominal", ndist.make_formation_shard(B, 0, 1, N=20, t0=0.0)),
                ("perturbed", synth.make_batch(B, N=20, seed=synth.SEED0 + 40, downwash=True, pos_sigma=0.5, vel_sigma=1.0, quat_sigma=0.15))):
    d = {k: torch.from_numpy(b[k]).to(dev) for k in ("x0", "xr", "ur", "other", "ego_xy")}
    eng = ndp.BatchedNMPC(B, N=20, disturbance=downwash, device=0, qp_mode=1)
    u0 = torch.empty(B, 4, dtype=torch.float64, device=dev)
    kw = dict(other=d["other"], ego_xy=d["ego_xy"]) if downwash else {}
    eng.reset_device(d["xr"], d["ur"])
    eng.debug_stamps(True)
    eng.update_device(d["x0"], d["xr"], d["ur"], u0, **kw)
    eng.synchronize()
    t = eng.debug_stamps(False, read=True)
    st, it = eng.status()
    span = t[:, 8] - t[:, 4]          # linearised -> end: the QP solve + step
    print(f"{name}: iterations min {it.min()} median {np.median(it)} max {it.max()}; status != 0: {(st != 0).sum()}")
    for k in sorted(set(it.tolist())):
        m = it == k
        print(f"  {k:2d} iterations: {m.sum():5d} instances, QP solve + step median {np.median(span[m]):9.0f} cycles")
    A = np.stack([it, np.ones_like(it)], 1).astype(float)
    coef = np.linalg.lstsq(A, span, rcond=None)[0]
    print(f"  least squares: {coef[0]:.0f} cycles per iteration + {coef[1]:.0f} fixed")
