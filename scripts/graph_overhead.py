"""Wall time of ONE replay of a hipGraph holding n control steps (batch 1024, fused downwash), bracketed by synchronisations
like bench.py's timed region: a + b n.  The driver's short run (--steps 20) pays the fixed part a once per 20 steps."""
import sys
import time

sys.path.insert(0, '.')
import numpy as np
import torch

import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import synth

B, T = 1024, 8
dev = torch.device("cuda", 0)
ticks = []
for t in range(T):
    b = synth.make_batch(B, seed=1, downwash=True, t0=0.02 * t)
    ticks.append({k: torch.from_numpy(b[k]).to(dev) for k in ("x0", "xr", "ur", "other", "ego_xy")})
eng = ndp.BatchedNMPC(B, disturbance=True)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
u0 = torch.empty(B, 4, dtype=torch.float64, device=dev)
eng.reset_device(ticks[0]["xr"], ticks[0]["ur"], stream=stream)


def step(i):
    d = ticks[i % T]
    eng.update_device(d["x0"], d["xr"], d["ur"], u0, other=d["other"], ego_xy=d["ego_xy"], stream=stream)


for i in range(16):
    step(i)
torch.cuda.synchronize()
res = []
for n in (1, 2, 5, 10, 20, 40, 100, 300):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream, capture_error_mode="relaxed"):
        for i in range(n):
            step(i)
    torch.cuda.set_stream(stream)
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for rep in range(15):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    # the same n steps launched from the host
    hs = []
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            step(i)
        torch.cuda.synchronize()
        hs.append(time.perf_counter() - t0)
    res.append((n, np.median(ts), np.min(ts), np.median(hs)))
    print("n=%4d graph replay median %8.1f us (min %8.1f) = %6.2f us/step | host launches %8.1f us = %6.2f us/step"
          % (n, np.median(ts) * 1e6, np.min(ts) * 1e6, np.median(ts) / n * 1e6, np.median(hs) * 1e6, np.median(hs) / n * 1e6))
ns = np.array([r[0] for r in res], float)
tm = np.array([r[1] for r in res]) * 1e6
A = np.vstack([np.ones_like(ns), ns]).T
a, bb = np.linalg.lstsq(A, tm, rcond=None)[0]
print("fit: %.1f us + %.2f us per step" % (a, bb))
