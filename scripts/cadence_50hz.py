#!/usr/bin/env python3
"""BASELINE config 1 at the reference's cadence: one vehicle, one control tick every 20 ms (rospy.Timer(ts_nmpc), nmpc_node.py:94;
params/nmpc_params.py:11), the GPU idle in between -- not a back-to-back loop.  p50 / p99 / max latency of
NMPCBodyRateController.update, ndp_step_ex and ndp_tick, beside the back-to-back figure and with a keep-warm launch every
millisecond from another thread.  bench.py's config1.hz50 uses paced() / single_vehicle_tick().   -> profiles/r05_cadence_50hz.txt
    python3 scripts/cadence_50hz.py [--ticks 500] [--wake]
"""
import argparse
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def paced(fn, n, period=0.02, pre=None, lead=1e-3):
    """fn() once per period on an absolute grid (sleep, then spin the last 200 us); returns the call latencies in us.
    pre: called `lead` seconds before every fn() (a wake-up call on the same handle)."""
    lat = np.empty(n)
    t0 = time.perf_counter() + period
    for k in range(n):
        due = t0 + k * period
        if pre is not None:
            while due - lead - time.perf_counter() > 3e-4:
                time.sleep(due - lead - time.perf_counter() - 2e-4)
            while time.perf_counter() < due - lead:
                pass
            pre()
        while True:
            left = due - time.perf_counter()
            if left <= 0:
                break
            if left > 3e-4:
                time.sleep(left - 2e-4)
        a = time.perf_counter()
        fn()
        lat[k] = time.perf_counter() - a
    return lat * 1e6


def back_to_back(fn, n, warm_s=0.3):
    t = time.perf_counter()
    while time.perf_counter() - t < warm_s:           # warm for a fixed TIME: the first ~100 ms after an idle period run at low clocks
        fn()
    lat = np.empty(n)
    for k in range(n):
        a = time.perf_counter()
        fn()
        lat[k] = time.perf_counter() - a
    return lat * 1e6


def stats(lat):
    return "p50 %6.1f  p99 %6.1f  max %6.1f us" % (np.median(lat), np.percentile(lat, 99), lat.max())


def single_vehicle_tick(ndp, synth, device=0, engine_out=None):
    """One vehicle on a figure-eight through ndp_tick; returns tick(): each call is the next 20 ms control tick (odometry = node 0 of
    the tick's window; after 800 ticks the trajectory starts over: list and controller reset, as at a new goal)."""
    tr = synth.figure_eight_traj(1, seed=synth.SEED0 + 1, n_seg=80, t_seg=0.25)
    et = ndp.BatchedNMPC(1, load_mlp=False, device=device)
    et.ref_set_trajectory(tr["coeff_x"], tr["coeff_y"], tr["coeff_z"], tr["coeff_yaw"], tr["time_cum"], tr["time_seg"], tr["final_pt"])
    et.ref_list_reset()
    et.tick_reset()
    xs = [et.ref_window(np.array([0.02 * i]))[0][:, 0, :].copy() for i in range(800)]
    no = [0]
    if engine_out is not None:
        engine_out.append(et)

    def tick():
        i = no[0] % 800
        if i == 0 and no[0]:
            et.ref_list_reset()
            et.tick_reset()
        no[0] += 1
        et.tick(xs[i], t=0.02 * i)
    return tick


class KeepWarm:
    """A launch every `period` seconds on its own engine and stream from a background thread."""

    def __init__(self, ndp, period=1e-3):
        self.eng = ndp.BatchedNMPC(1, load_mlp=False)
        self.period, self.stop = period, False
        self.th = threading.Thread(target=self.run, daemon=True)

    def run(self):
        while not self.stop:
            self.eng.throttle_reset()                 # one tiny launch + synchronise
            time.sleep(self.period)

    def __enter__(self):
        self.th.start()
        return self

    def __exit__(self, *a):
        self.stop = True
        self.th.join()


def main():
    import ndp_nmpc_qd_amd as ndp
    from ndp_nmpc_qd_amd import synth
    from ndp_nmpc_qd_amd.nmpc_ctl import NMPCBodyRateController
    ap = argparse.ArgumentParser()
    ap.add_argument("--ticks", type=int, default=500)
    ap.add_argument("--wake", action="store_true", help="also: the same handle woken shortly before every paced tick (three ways)")
    args = ap.parse_args()
    n = args.ticks
    b = synth.make_batch(1, seed=synth.SEED0 + 1)
    x0, xr, ur = b["x0"][0], b["xr"][0], b["ur"][0]
    ctl = NMPCBodyRateController()
    ctl.reset(xr, ur)
    e1 = ndp.BatchedNMPC(1, load_mlp=False)
    e1.reset(xr[None], ur[None])
    engs = []
    tick = single_vehicle_tick(ndp, synth, engine_out=engs)
    et = engs[0]
    cases = (("NMPCBodyRateController.update", lambda: ctl.update(x0, xr, ur)),
             ("ndp_step_ex (BatchedNMPC(1).update, full)", lambda: e1.update(x0[None], xr[None], ur[None], full=True)),
             ("ndp_tick (odometry in, command out)", tick))
    for name, fn in cases:
        print(f"{name}\n   back to back                                   {stats(back_to_back(fn, n))}")
        print(f"   one per 20 ms                                  {stats(paced(fn, n))}")
        with KeepWarm(ndp):
            print(f"   one per 20 ms, a keep-warm launch every 1 ms   {stats(paced(fn, n))}")
    import ctypes as C
    parts = []

    def tick_timed():
        tick()
        out = (C.c_double * 4)()
        et._lib.ndp_debug_host_timing(et._h, out)
        parts.append(list(out))
    for title, run in (("one per 20 ms", lambda: paced(tick_timed, n)), ("back to back ", lambda: back_to_back(tick_timed, n))):
        parts.clear()
        lat = run()
        print(f"ndp_tick {title}: {stats(lat)}; inside the library, median us: pack %.1f  enqueue %.1f  wait %.1f  copy-out %.1f"
              % tuple(np.median(np.array(parts[-n:]), axis=0)))
    if args.wake:
        for lead in (1e-3, 2e-4):
            print(f"ndp_tick one per 20 ms, the same handle woken {lead * 1e6:.0f} us earlier by ...")
            print(f"   status()  (a 4-byte device-to-host copy)             {stats(paced(tick, n, pre=et.status, lead=lead))}")
            print(f"   ref_list_window(None) (a launch that reads the list) {stats(paced(tick, n, pre=lambda: et.ref_list_window(None), lead=lead))}")
            print(f"   synchronize()                                        {stats(paced(tick, n, pre=et.synchronize, lead=lead))}")
    print("deadline: 20 000 us (nmpc_node.py:216-220)")


if __name__ == "__main__":
    main()
