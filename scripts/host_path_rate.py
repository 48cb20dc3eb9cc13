import sys, time; sys.path.insert(0,'.')
import numpy as np
import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import synth
B=1024
b=synth.make_batch(B, seed=1, downwash=True)
eng=ndp.BatchedNMPC(B, disturbance=True)
eng.reset(b["xr"], b["ur"])
for _ in range(5): eng.update(b["x0"], b["xr"], b["ur"], other=b["other"], ego_xy=b["ego_xy"])
t0=time.perf_counter()
n=100
for _ in range(n): eng.update(b["x0"], b["xr"], b["ur"], other=b["other"], ego_xy=b["ego_xy"])
el=(time.perf_counter()-t0)/n
print("host-pointer ndp_step (pageable numpy buffers, H2D + kernel + D2H + status): %.1f us per step, %.2f M solves/s" % (el*1e6, B/el/1e6))
