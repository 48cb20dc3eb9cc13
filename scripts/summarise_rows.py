"""Summarises scripts/profile_rows.sh: per kernel the duration statistics of its launches AT ITS LARGEST GRID (the first three dropped:
clocks, caches) and the per-launch mean of every counter -> <out>/summary/<tag>_kernel_stats_rows.csv, <tag>_pmc_rows.json.
HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB units; the gfx950 note of MI355X_MICROARCH.md)."""
import csv
import glob
import json
import os
import sys

import numpy as np

out, tag = sys.argv[1], sys.argv[2]
summ = os.path.join(out, "summary")
os.makedirs(summ, exist_ok=True)


def short(name):
    n = name.replace("void ndp::", "").replace("ndp::", "")
    return n.split("(")[0][:60]


def pick_grid(grids):
    """A kernel's launches are summarised at ONE grid size: the largest one launched at least five times (the driver's helpers launch
    some kernels at batch 1024 as well, and the list's reset fills 101 entries per vehicle once)."""
    from collections import Counter
    c = Counter(grids)
    often = [g for g, n in c.items() if n >= 5]
    return max(often) if often else max(c, key=lambda g: (c[g], g))


stats = {}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    per = {}
    for r in csv.DictReader(open(f)):
        per.setdefault(short(r["Kernel_Name"]), []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size_X"])))
    for k, v in per.items():
        g = pick_grid([x[2] for x in v])
        v = sorted(x for x in v if x[2] == g)
        d = np.array([x[1] for x in v][3:] or [x[1] for x in v], dtype=float)
        stats[k] = {"calls": len(d), "avg_us": d.mean() / 1e3, "min_us": d.min() / 1e3, "max_us": d.max() / 1e3}
pmc = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_f64"):
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        per, grid = {}, {}
        rows = list(csv.DictReader(open(f)))
        for r in rows:                              # (one row per dispatch AND counter: count dispatches)
            grid.setdefault(short(r["Kernel_Name"]), {})[r["Dispatch_Id"]] = int(r["Grid_Size"])
        grid = {k: pick_grid(list(g.values())) for k, g in grid.items()}
        for r in rows:
            k = short(r["Kernel_Name"])
            if int(r["Grid_Size"]) != grid[k]:
                continue
            per.setdefault(k, {}).setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            per[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for k, cs in per.items():
            for c, d in cs.items():
                ids = sorted(d, key=lambda x: int(x))
                vals = [d[i] for i in ids][3:] or [d[i] for i in ids]
                pmc.setdefault(k, {})[c] = float(np.mean(vals))
for k, c in pmc.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        c["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        if k in stats:
            c["hbm_GBps"] = c["hbm_bytes_per_launch"] / (stats[k]["avg_us"] * 1e-6) / 1e9
            c["hbm_frac_of_8TBps"] = c["hbm_GBps"] / 8000.0
    if "SQ_WAIT_INST_ANY" in c and c.get("SQ_WAVE_CYCLES"):
        c["wait_frac_of_wave_cycles"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
    f64 = sum(c.get(n, 0.0) for n in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
    if f64 and c.get("SQ_INSTS_VALU"):
        c["f64_share_of_valu_insts"] = f64 / c["SQ_INSTS_VALU"]
with open(os.path.join(summ, f"{tag}_kernel_stats_rows.csv"), "w") as fh:
    fh.write("kernel,calls,avg_us,min_us,max_us\n")
    for k in sorted(stats):
        s = stats[k]
        fh.write(f"\"{k}\",{s['calls']},{s['avg_us']:.3f},{s['min_us']:.3f},{s['max_us']:.3f}\n")
json.dump(pmc, open(os.path.join(summ, f"{tag}_pmc_rows.json"), "w"), indent=1, sort_keys=True)
for k in sorted(stats):
    c = pmc.get(k, {})
    print(f"{k:44s} {stats[k]['calls']:4d} x {stats[k]['avg_us']:9.2f} us   HBM {c.get('hbm_bytes_per_launch', 0) / 1e6:9.2f} MB  "
          f"{c.get('hbm_frac_of_8TBps', 0):.3f} of 8 TB/s   VALU {c.get('SQ_INSTS_VALU', 0):.3g}  wait {c.get('wait_frac_of_wave_cycles', 0):.2f}  "
          f"f64 share {c.get('f64_share_of_valu_insts', 0):.2f}")
