"""Host-array step rate (SURVEY 8d's wording of the metric: H2D of the inputs and D2H of u0 inside the time), pageable numpy in /
numpy out, batch 1024 (or argv[1]), N = 20, downwash on.  One tick at a time (ndp_step) and two ticks in flight
(ndp_step_begin / ndp_step_end).  Environment: NDP_PACK_THREADS (pack threads beside the caller), NDP_HOST_PATH (measurement
switch of csrc/ndp_hip.hip: zero | dma | dma1).  Also a plain multi-threaded memcpy of the same bytes for reference.
Run on the GPU box: python scripts/host_path_rate.py [B]"""
import os
import sys
import time

sys.path.insert(0, '.')
import numpy as np

import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
b = synth.make_batch(B, seed=1, downwash=True)
kw = dict(other=b["other"], ego_xy=b["ego_xy"])
nbytes = sum(b[k].nbytes for k in ("x0", "xr", "ur", "ego_xy")) + b["other"].nbytes * 6 // 10   # as they cross PCIe: 6 of the windows' 10 columns
eng = ndp.BatchedNMPC(B, disturbance=True)
eng.reset(b["xr"], b["ur"])
for _ in range(10):
    eng.update(b["x0"], b["xr"], b["ur"], **kw)
n = 200
u = np.empty((B, 4))


def run_sync():
    t0 = time.perf_counter()
    for _ in range(n):
        eng.update(b["x0"], b["xr"], b["ur"], **kw)
    return (time.perf_counter() - t0) / n


def run_pipe():
    eng.update_begin(b["x0"], b["xr"], b["ur"], **kw)
    t0 = time.perf_counter()
    for _ in range(n):
        eng.update_begin(b["x0"], b["xr"], b["ur"], **kw)
        eng.update_end(out=u)
    dt = (time.perf_counter() - t0) / n
    eng.update_end(out=u)
    return dt


res = []
for rep in range(3):                      # alternate: the first loop after start-up runs at ramping clocks / link state
    res.append(("sync", run_sync()))
    ht = np.zeros(4)
    eng._lib.ndp_debug_host_timing(eng._h, ht.ctypes.data)
    res.append(("pipe", run_pipe()))
sync = min(v for k, v in res if k == "sync")
pipe = min(v for k, v in res if k == "pipe")
tag = "threads=%s path=%s" % (os.environ.get("NDP_PACK_THREADS", "auto"), os.environ.get("NDP_HOST_PATH", "default"))
print("B=%d %-62s one tick at a time %7.1f us/step %6.2f M solves/s %5.1f GB/s | two in flight %7.1f us/step %6.2f M solves/s %5.1f GB/s"
      % (B, tag + " [pack %.0f enq %.0f wait %.0f out %.0f]" % tuple(ht), sync * 1e6, B / sync / 1e6, nbytes / sync / 1e9, pipe * 1e6, B / pipe / 1e6, nbytes / pipe / 1e9))
print("   all loops (us/step): " + " ".join("%s %.0f" % (k, v * 1e6) for k, v in res))
