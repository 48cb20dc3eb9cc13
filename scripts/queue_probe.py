"""Mixed workload (about 20 % of the instances need the interior-point loop) with and without the in-kernel work queue."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import synth

dev = torch.device("cuda:0")
out = {}
for B in (1024, 2048, 4096, 8192, 16384):
    tk = []
    for t in range(4):
        m = synth.make_batch(B, seed=synth.SEED0 + 40, downwash=True, t0=0.02 * t, pos_sigma=0.5, vel_sigma=1.0, quat_sigma=0.15)
        tk.append({k: torch.from_numpy(m[k]).to(dev) for k in ("x0", "xr", "ur", "other", "ego_xy")})
    for wq in (2, 1):
        e = ndp.BatchedNMPC(B, disturbance=True, work_queue=wq)
        u = torch.empty(B, 4, dtype=torch.float64, device=dev)
        e.reset_device(tk[0]["xr"], tk[0]["ur"])
        for i in range(10):
            d = tk[i % 4]
            e.update_device(d["x0"], d["xr"], d["ur"], u, other=d["other"], ego_xy=d["ego_xy"])
        e.synchronize()
        n = 40
        t0 = time.perf_counter()
        for i in range(n):
            d = tk[i % 4]
            e.update_device(d["x0"], d["xr"], d["ur"], u, other=d["other"], ego_xy=d["ego_xy"])
        e.synchronize()
        dt = (time.perf_counter() - t0) / n
        st, it = e.status()
        out[f"B{B}_queue_{'on' if wq == 1 else 'off'}"] = dict(ms_per_step=dt * 1e3, Msolves_s=B / dt / 1e6, frac_ipm=float((it > 0).mean()),
                                                          bad=int((st != 0).sum()))
        print(B, wq, out[f"B{B}_queue_{'on' if wq == 1 else 'off'}"], flush=True)
print(json.dumps(out))
