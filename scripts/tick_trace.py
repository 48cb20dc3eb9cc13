#!/usr/bin/env python3
"""Kernel timeline of ndp_tick under rocprofv3 --kernel-trace: per kernel the average duration and the average gap to the previous
kernel's end, over the last ticks of a two-in-flight loop.
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 <repo>/scripts/tick_trace.py run
    python3 scripts/tick_trace.py summarise <dir>
"""
import csv
import glob
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    from tick_rate import setup
    B, n = 1024, 300
    eng = setup(B)
    rng = np.random.default_rng(0)
    ts = [np.full(B, 0.02 * i) for i in range(n + 1)]
    xs = []
    for i in range(n + 1):
        x = eng.ref_window(ts[i])[0][:, 0, :].copy()
        x[:, 0:3] += rng.normal(0, 0.1, (B, 3))
        x[:, 3:6] += rng.normal(0, 0.2, (B, 3))
        xs.append(x)
    cmd = np.empty((B, 4))
    est = len(sys.argv) < 3 or "noest" not in sys.argv[2]
    if len(sys.argv) >= 3 and "uni" in sys.argv[2]:
        ts = [float(t[0]) for t in ts]
    eng.tick_begin(xs[0], t=ts[0], estimate=est)
    for i in range(1, n + 1):
        eng.tick_begin(xs[i], t=ts[i], estimate=est)
        eng.tick_end(out=cmd)
    eng.tick_end(out=cmd)


def summarise(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-600:]                                  # the last 200 ticks
    stat = {}
    prev_end = None
    for r in rows:
        nm = r["Kernel_Name"].split("(")[0].replace("void ndp::", "")[:40]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        st = stat.setdefault(nm, [[], []])
        st[0].append(e - s)
        if prev_end is not None:
            st[1].append(s - prev_end)
        prev_end = e
    span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
    print(f"{len(rows)} launches over {span:.1f} us")
    for nm, (du, gap) in stat.items():
        print(f"  {nm:42s} n {len(du):4d}  duration avg {np.mean(du) / 1e3:7.2f} us  gap in front avg {np.mean(gap) / 1e3:7.2f} us (median {np.median(gap) / 1e3:.2f})")


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    if sys.argv[1] == "run":
        run()
    else:
        summarise(sys.argv[2])
