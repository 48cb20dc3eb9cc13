"""Latency of the host-array entry points at small batch (numpy in, numpy out, one synchronisation per call), back to back.
Every case is warmed for a fixed TIME (0.3 s), not a fixed count: the first ~100 ms of launches after an idle period run at low
clocks and through cold paths of the runtime -- rounds 3 / 4 warmed 50 calls and their FIRST row (the first case measured in the
process) read 120 / 195 us for a 36 us call.  The reference's real cadence (one call per 20 ms) is scripts/cadence_50hz.py.
Run on the GPU box: python scripts/host_latency.py"""
import sys, time; sys.path.insert(0, '.')
import numpy as np
import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import synth
from ndp_nmpc_qd_amd.nmpc_ctl.nmpc_body_rate_ctl import NMPCBodyRateController

for B in (1, 16, 64):
    b = synth.make_batch(B, seed=5, downwash=True)
    for dw in (False, True):
        e = ndp.BatchedNMPC(B, disturbance=dw)
        e.reset(b["xr"], b["ur"])
        kw = dict(other=b["other"], ego_xy=b["ego_xy"]) if dw else {}
        for full in (False, True):
            tw = time.perf_counter()
            while time.perf_counter() - tw < 0.3:
                e.update(b["x0"], b["xr"], b["ur"], full=full, **kw)
            t = time.perf_counter()
            n = 400
            for _ in range(n):
                e.update(b["x0"], b["xr"], b["ur"], full=full, **kw)
            print(f"B={B:3d} downwash={int(dw)} {'ndp_step_ex (u0 + iterate + status)' if full else 'ndp_step (u0)':38s} {(time.perf_counter() - t) / n * 1e6:7.1f} us")
        e.close()
b = synth.make_batch(1, seed=5)
ctl = NMPCBodyRateController()
ctl.reset(b["xr"][0], b["ur"][0])
tw = time.perf_counter()
while time.perf_counter() - tw < 0.3:
    ctl.update(b["x0"][0], b["xr"][0], b["ur"][0])
t = time.perf_counter()
for _ in range(400):
    ctl.update(b["x0"][0], b["xr"][0], b["ur"][0])
print(f"NMPCBodyRateController.update                                  {(time.perf_counter() - t) / 400 * 1e6:7.1f} us")
