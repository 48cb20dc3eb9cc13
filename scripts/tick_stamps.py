#!/usr/bin/env python3
"""Phase stamps (shader clock) of the one-launch control tick against the plain fused control step, inputs in HBM (ndp_tick_device /
ndp_step_device) and -- the tick -- in page-locked host memory (ndp_tick): where the tick's extra time goes.  GPU box:
    python3 scripts/tick_stamps.py [uniform]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tick_rate import setup  # noqa: E402

LABEL = {9: 'kernel entry', 0: 'program start', 1: 'tables', 2: 'stage_in', 3: 'cost', 11: 'weights staged + new point + barrier', 10: 'mlp tile',
         4: 'linearize', 5: 'pre_sweep', 6: 'backward', 7: 'forward', 8: 'end'}


def report(title, t):
    ids = sorted([9, 11, 10] + list(range(9)), key=lambda i: np.median(t[:, i] - t[:, 0]))
    clk = np.median((t[:, 15] - t[:, 14]) / np.maximum(1.0, t[:, 13] - t[:, 12])) * 100e6
    print(f"{title}: median cycles per phase (shader clock {clk / 1e9:.2f} GHz); whole wave {np.median(t[:, 15] - t[:, 14]):.0f} cycles = "
          f"{np.median(t[:, 15] - t[:, 14]) / clk * 1e6:.2f} us, slowest wave {np.max(t[:, 15] - t[:, 14]) / clk * 1e6:.2f} us")
    print(f"   entry -> first stamp {np.median(t[:, ids[0]] - t[:, 14]):.0f} | " + " | ".join(f"{LABEL[b]} {np.median(t[:, b] - t[:, a]):.0f}" for a, b in zip(ids[:-1], ids[1:]))
          + f" | last stamp -> exit {np.median(t[:, 15] - t[:, 8]):.0f}")
    if t.shape[1] > 21 and np.median(t[:, 21]) > 0:          # the one-launch tick's prologue, since kernel entry (each stamp waits for its value)
        names = ("neighbour index known", "cached segment records arrived", "time arrived", "polynomial value", "values collected (v_readlane)", "flatness map done")
        print("   tick prologue since entry: " + " | ".join(f"{n} {np.median(t[:, 16 + i] - t[:, 14]):.0f}" for i, n in enumerate(names))
              + f" | first phase stamp {np.median(t[:, 9] - t[:, 14]):.0f} | weights + barrier done {np.median(t[:, 11] - t[:, 14]):.0f}"
              + (f" | cache loads issued {np.median(t[:, 22] - t[:, 14]):.0f} | input loads issued {np.median(t[:, 23] - t[:, 14]):.0f}" if t.shape[1] > 23 and np.median(t[:, 23]) > 0 else ""))


def main():
    uniform = len(sys.argv) > 1 and sys.argv[1] == "uniform"
    B = 1024
    dev = torch.device("cuda", 0)
    eng = setup(B)
    ts = [np.full(B, 0.02 * i) for i in range(64)]
    xs = [eng.ref_window(t)[0][:, 0, :].copy() for t in ts]
    for i in range(30):
        eng.tick(xs[i], t=float(ts[i][0]) if uniform else ts[i])
    eng.debug_stamps(True)
    for i in range(30, 33):
        eng.tick(xs[i], t=float(ts[i][0]) if uniform else ts[i])
    report("ndp_tick, host arrays (x_odom / t read over PCIe, cmd written to host memory)", eng.debug_stamps(False, read=True))
    cmd = torch.empty(B, 4, dtype=torch.float64, device=dev)
    for i in range(33, 50):
        eng.tick_device(torch.from_numpy(xs[i]).to(dev), cmd, t=float(ts[i][0]) if uniform else torch.from_numpy(ts[i]).to(dev))
    eng.synchronize()
    eng.debug_stamps(True)
    for i in range(50, 53):
        eng.tick_device(torch.from_numpy(xs[i]).to(dev), cmd, t=float(ts[i][0]) if uniform else torch.from_numpy(ts[i]).to(dev))
    eng.synchronize()
    report("ndp_tick_device (everything in HBM)", eng.debug_stamps(False, read=True))
    # the plain fused step on the same windows
    xr, ur = eng.ref_list_window(None)
    d = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in dict(x0=xs[53], xr=xr, ur=ur, other=xr[np.arange(B) ^ 1], ego=xs[53][:, 0:2]).items()}
    u0 = torch.empty(B, 4, dtype=torch.float64, device=dev)
    for _ in range(10):
        eng.update_device(d["x0"], d["xr"], d["ur"], u0, other=d["other"], ego_xy=d["ego"])
    eng.synchronize()
    eng.debug_stamps(True)
    for _ in range(3):
        eng.update_device(d["x0"], d["xr"], d["ur"], u0, other=d["other"], ego_xy=d["ego"])
    eng.synchronize()
    report("ndp_step_device (plain fused control step, HBM)", eng.debug_stamps(False, read=True))


if __name__ == "__main__":
    main()
