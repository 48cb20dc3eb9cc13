"""What one launch + completion costs on this box, seen from the host (the fixed part of bench.py's short timed region):
an empty-ish torch kernel, the control step alone (host launch / graph of 1), with the fence bench.py uses (event query poll)
and with a plain synchronize."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch

import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import synth

dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
x = torch.zeros(64, device=dev)


def timeit(fn, fence, n=300):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        fence()
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts[n // 3:]) * 1e6
    return np.median(ts), ts.min()


ev = torch.cuda.Event()


def fence_poll():
    ev.record(stream)
    while not ev.query():
        pass
    torch.cuda.synchronize()


def fence_sync():
    torch.cuda.synchronize()


def fence_stream():
    stream.synchronize()


def fence_stream_dev():
    stream.synchronize()
    torch.cuda.synchronize()


FENCES = (("event-query poll + synchronize", fence_poll), ("synchronize", fence_sync), ("stream.synchronize", fence_stream),
          ("stream.synchronize + synchronize", fence_stream_dev))
t_idle = []
for _ in range(200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    torch.cuda.synchronize()
    t_idle.append(time.perf_counter() - t0)
print("torch.cuda.synchronize() on an idle device: median %.2f us" % (np.median(t_idle) * 1e6))
for name, fence in FENCES:
    print("tiny torch kernel (x.add_(1), 64 floats), %-32s: median %.1f us, min %.1f us" % ((name,) + timeit(lambda: x.add_(1.0), fence)))

for B in (1024,):
    b = synth.make_batch(B, seed=1, downwash=True)
    t = {k: torch.from_numpy(b[k]).to(dev) for k in ("x0", "xr", "ur", "other", "ego_xy")}
    eng = ndp.BatchedNMPC(B, disturbance=True)
    u0 = torch.empty(B, 4, dtype=torch.float64, device=dev)
    eng.reset_device(t["xr"], t["ur"], stream=stream)
    step = lambda: eng.update_device(t["x0"], t["xr"], t["ur"], u0, other=t["other"], ego_xy=t["ego_xy"], stream=stream)  # noqa: E731
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(200):
        step()
    e1.record(stream)
    torch.cuda.synchronize()
    print("batch %d: control step back to back (HIP events / 200): %.2f us" % (B, e0.elapsed_time(e1) * 1e3 / 200))
    for name, fence in FENCES:
        print("batch %d: ONE host-launched control step, %-32s: median %.1f us, min %.1f us" % ((B, name) + timeit(step, fence)))
    for n in (1, 20):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream, capture_error_mode="relaxed"):
            for _ in range(n):
                step()
        torch.cuda.set_stream(stream)
        g.replay()
        torch.cuda.synchronize()
        for name, fence in FENCES:
            print("batch %d: graph of %2d steps, one replay,   %-32s: median %.1f us, min %.1f us" % ((B, n, name) + timeit(g.replay, fence)))
