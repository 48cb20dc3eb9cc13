"""The prefetch protocol's two chains (ndp_step_device_prefetched / ndp_downwash_prefetch_device) as two hipGraphs side by side."""
import sys, time, os
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np, torch
import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import synth
B, T, n = 1024, 8, 200
dev = torch.device("cuda", 0)
ticks = []
for t in range(T):
    b = synth.make_batch(B, seed=1, downwash=True, t0=0.02 * t)
    ticks.append({k: torch.from_numpy(b[k]).to(dev) for k in ("x0", "xr", "ur", "other", "ego_xy")})
for prio in (0, -1):
    sa = torch.cuda.Stream(device=dev)
    sb = torch.cuda.Stream(device=dev, priority=prio)
    eng = ndp.BatchedNMPC(B, disturbance=True)
    u0 = torch.empty(B, 4, dtype=torch.float64, device=dev)
    eng.reset_device(ticks[0]["xr"], ticks[0]["ur"], stream=sa)
    d = ticks[0]
    eng.downwash_prefetch_device(d["other"], d["xr"], ego_xy=d["ego_xy"], on_stream=sb)
    eng.update_device_prefetched(d["x0"], d["xr"], d["ur"], u0, stream=sa)
    torch.cuda.synchronize()
    ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(ga, stream=sa, capture_error_mode="relaxed"):
        for i in range(n):
            d = ticks[i % T]
            eng.update_device_prefetched(d["x0"], d["xr"], d["ur"], u0, stream=sa)
    with torch.cuda.graph(gb, stream=sb, capture_error_mode="relaxed"):
        for i in range(n):
            d = ticks[i % T]
            eng.downwash_prefetch_device(d["other"], d["xr"], ego_xy=d["ego_xy"], on_stream=sb)
    for order in ("b_first",):
        best = 1e9
        for rep in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if order == "b_first":
                with torch.cuda.stream(sb): gb.replay()
                with torch.cuda.stream(sa): ga.replay()
            else:
                with torch.cuda.stream(sa): ga.replay()
                with torch.cuda.stream(sb): gb.replay()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print("priority %d %s: %.2f us per tick; stats %s" % (prio, order, best / n * 1e6, eng.prefetch_stats()))
    # the downwash chain alone (its waits for free slots never end: bounded) is not timed; the control chain alone needs its forces
    eng.close()
