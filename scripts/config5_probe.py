"""N = 40, 2 RTI iterations, batch 4096 (BASELINE config 5 shape), fp64: step time with the work list on / off, nominal and
perturbed starts."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import synth

B, N = 4096, 40
dev = torch.device("cuda", 0)
for label, kw in (("nominal", {}), ("perturbed", dict(pos_sigma=0.5, vel_sigma=1.0, quat_sigma=0.15))):
    b = synth.make_batch(B, N=N, seed=20231213 + 5, **kw)
    t = {k: torch.from_numpy(b[k]).to(dev) for k in ("x0", "xr", "ur")}
    for wq in (2, 1):
        eng = ndp.BatchedNMPC(B, N=N, n_rti=2, work_queue=wq)
        u0d = torch.empty(B, 4, dtype=torch.float64, device=dev)
        eng.reset_device(t["xr"], t["ur"])
        for _ in range(10):
            eng.update_device(t["x0"], t["xr"], t["ur"], u0d)
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            eng.update_device(t["x0"], t["xr"], t["ur"], u0d)
        eng.synchronize()
        el = (time.perf_counter() - t0) / 40
        st, it = eng.status()
        print(f"{label:10s} work list {'on ' if wq == 1 else 'off'}: {el * 1e3:.4f} ms/step, {B / el / 1e6:.2f} M solves/s, "
              f"interior-point frac {(it > 0).mean():.3f}, status != 0: {(st != 0).sum()}", flush=True)
