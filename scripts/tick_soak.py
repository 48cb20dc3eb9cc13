#!/usr/bin/env python3
"""Soak of ndp_tick: 25 laps of 800 control ticks (20 000 ticks, two in flight) over the figure-eight trajectories at batch 1024, the list
and the controller reset at the start of every lap like a node that receives a new trajectory (nmpc_node.py:148-152).  Checks at every
tick: commands finite, every instance converged; at the end of every lap the last command against the CPU oracle on the same inputs.
GPU box: python3 scripts/tick_soak.py [laps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tick_rate import setup  # noqa: E402


def main():
    laps = int(sys.argv[1]) if len(sys.argv) > 1 else 25
    B, n = 1024, 800
    eng = setup(B)
    rng = np.random.default_rng(0)
    ts = [0.02 * i for i in range(n + 1)]
    xs = []
    for t in ts:                                                     # the odometry follows the reference (SURVEY 8d's noise)
        x = eng.ref_window(np.full(B, t))[0][:, 0, :].copy()
        x[:, 0:3] += rng.normal(0, 0.1, (B, 3))
        x[:, 3:6] += rng.normal(0, 0.2, (B, 3))
        xs.append(x)
    cmd = np.empty((B, 4))
    bad = nonfinite = 0
    first = {}
    t_all = 0.0
    for lap in range(laps):
        eng.ref_list_reset()
        eng.throttle_reset()
        eng.tick_reset()
        t0 = time.perf_counter()
        eng.tick_begin(xs[0], t=ts[0], estimate=lap % 2 == 0)
        for i in range(1, n + 1):
            eng.tick_begin(xs[i], t=ts[i] if i % 2 else np.full(B, ts[i]), estimate=lap % 2 == 0)
            eng.tick_end(out=cmd)
            nonfinite += int(not np.isfinite(cmd).all())
        eng.tick_end(out=cmd)
        t_all += time.perf_counter() - t0
        st, _ = eng.status()
        bad += int((st != 0).sum())
        # Laps with the same settings and the same start end in the same command, bit for bit.  (Lap 0 is not among them: the thrust
        # of the last command sent survives the resets -- it is what the vehicle is flying on -- and feeds the estimator's first update.)
        if lap >= 1:
            ref = first.setdefault(lap % 2, cmd.copy())
            assert np.array_equal(ref, cmd), (f"lap {lap}: the last command differs from lap {2 - lap % 2}'s", np.max(np.abs(ref - cmd), axis=0))
    print(f"ndp_tick soak: {laps} laps x {n + 1} ticks at batch {B}: {t_all / (laps * (n + 1)) * 1e6:.2f} us per tick, "
          f"{nonfinite} ticks with a non-finite command, {bad} instances not converged at a lap's end, laps of a kind end bit-equal")


if __name__ == "__main__":
    main()
