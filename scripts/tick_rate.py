#!/usr/bin/env python3
"""ndp_tick (odometry in, actuator command out, references resident) against ndp_step (x0 + xr + ur + neighbour columns across
PCIe): solves/s with two ticks in flight and one at a time, B = 1 latency.  -> profiles/r05_tick_rate.txt

    python3 scripts/tick_rate.py [--batch 1024] [--ticks 400]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ndp_nmpc_qd_amd as ndp                      # noqa: E402
from ndp_nmpc_qd_amd import synth                  # noqa: E402


def setup(B, pairs=True, seed=synth.SEED0 + 3):
    """B vehicles on SURVEY 8d's figure-eights (synth.figure_eight_traj: 20 s of trajectory = 1000 control ticks), the reference list
    built, vehicle i's neighbour = vehicle i ^ 1 (gate on the odometry), the controller reset to the first window."""
    tr = synth.figure_eight_traj(B, seed=seed, n_seg=80, t_seg=0.25, pairs=pairs and B % 2 == 0)
    eng = ndp.BatchedNMPC(B, disturbance=True)
    eng.ref_set_trajectory(tr["coeff_x"], tr["coeff_y"], tr["coeff_z"], tr["coeff_yaw"], tr["time_cum"], tr["time_seg"], tr["final_pt"])
    eng.ref_list_reset()
    if pairs and B > 1 and B % 2 == 0:
        eng.tick_config(np.arange(B, dtype=np.int32) ^ 1, gate=True)
    eng.tick_reset()
    return eng


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--ticks", type=int, default=300)
    ap.add_argument("--uniform-t", action="store_true", help="one clock for every vehicle: t as a scalar (NDP_TICK_T_UNIFORM)")
    ap.add_argument("--no-estimator", action="store_true", help="the reference stops the estimator's timer while a trajectory is tracked")
    args = ap.parse_args()
    B, n = args.batch, args.ticks
    est = not args.no_estimator
    eng = setup(B)
    rng = np.random.default_rng(0)
    # tick i happens at trajectory time 0.02 i; the odometry follows the reference (node 0 of that tick's window) with SURVEY 8d's noise
    n1 = min(n, 150)                                  # ticks of the one-at-a-time loop
    nt = 32 + n1 + n + 1
    assert nt <= 890, "the trajectories last 1000 ticks; the list looks 100 ticks ahead"
    ts = [np.full(B, 0.02 * i) for i in range(nt)]
    tk = [float(t[0]) for t in ts] if args.uniform_t else ts          # what the tick is handed
    xs = []
    for i in range(nt):
        x = eng.ref_window(ts[i])[0][:, 0, :].copy()
        x[:, 0:3] += rng.normal(0, 0.1, (B, 3))
        x[:, 3:6] += rng.normal(0, 0.2, (B, 3))
        xs.append(x)
    x0 = xs[0]
    cmd = np.empty((B, 4))
    it = iter(range(nt))
    for _ in range(32):                               # warm: link, clocks, mirrors
        i = next(it)
        eng.tick(xs[i], t=tk[i], estimate=est)
    t0 = time.perf_counter()
    for _ in range(n1):
        i = next(it)
        eng.tick(xs[i], t=tk[i], estimate=est)
    one = (time.perf_counter() - t0) / n1
    i = next(it)
    eng.tick_begin(xs[i], t=tk[i], estimate=est)
    t0 = time.perf_counter()
    for _ in range(n):
        i = next(it)
        eng.tick_begin(xs[i], t=tk[i], estimate=est)
        eng.tick_end(out=cmd)
    two = (time.perf_counter() - t0) / n
    _, _, st, itn = eng.tick_end(full=False), None, *eng.status()
    print(f"  last tick: {int((st != 0).sum())} instances not converged, {float((itn > 0).mean()):.3f} in the interior-point loop, "
          f"{float(np.any(eng.device_force().cpu().numpy() != 0, axis=(1, 2)).mean()):.2f} of the gates open")
    in_b = 80 + 8
    print(f"ndp_tick batch {B} (estimator {'on' if est else 'off'}, t {'scalar' if args.uniform_t else 'per vehicle'}): two ticks in flight {two * 1e6:8.2f} us/tick = {B / two / 1e6:7.3f} M solves/s "
          f"(PCIe {B * (in_b + 36) / two / 1e9:.2f} GB/s implied); one at a time {one * 1e6:8.2f} us = {B / one / 1e6:7.3f} M solves/s")
    print("  host us of the last tick (pack, enqueue, wait, copy-out):", [round(v, 1) for v in host_timing(eng)])
    # ---- the same workload through ndp_step_begin / _end: everything across PCIe
    xr, ur = eng.ref_list_window(None)
    other = xr[np.arange(B) ^ 1] if B % 2 == 0 else xr
    st = ndp.BatchedNMPC(B, disturbance=True)
    st.reset(xr, ur)
    u0 = np.empty((B, 4))
    kw = dict(other=other, ego_xy=x0[:, 0:2].copy())
    st.update_begin(x0, xr, ur, **kw)
    for _ in range(100):
        st.update_begin(x0, xr, ur, **kw)
        st.update_end(out=u0)
    t0 = time.perf_counter()
    for _ in range(n // 2):
        st.update_begin(x0, xr, ur, **kw)
        st.update_end(out=u0)
    full = (time.perf_counter() - t0) / (n // 2)
    st.update_end(out=u0)
    print(f"ndp_step_begin/_end batch {B} (x0 + xr + ur + neighbour columns + ego xy across PCIe): {full * 1e6:8.2f} us/tick = "
          f"{B / full / 1e6:7.3f} M solves/s")
    # ---- one vehicle, one tick at a time (BASELINE config 1's shape through the tick)
    e1 = setup(1, pairs=False)
    x1s = [e1.ref_window(ts[i][:1])[0][:, 0, :].copy() for i in range(350)]
    for i in range(50):
        e1.tick(x1s[i], t=ts[i][:1])
    lat = []
    for i in range(300):
        a = time.perf_counter()
        e1.tick(x1s[50 + i], t=ts[50 + i][:1], estimate=est)
        lat.append(time.perf_counter() - a)
    lat = np.array(lat) * 1e6
    print(f"ndp_tick batch 1: p50 {np.median(lat):.1f} us, p99 {np.percentile(lat, 99):.1f} us, max {lat.max():.1f} us per tick (back to back)")


def host_timing(eng):
    import ctypes as C
    out = (C.c_double * 4)()
    eng._lib.ndp_debug_host_timing(eng._h, out)
    return list(out)


if __name__ == "__main__":
    main()
