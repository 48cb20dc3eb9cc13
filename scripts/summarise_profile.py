"""Summarises the rocprofv3 outputs of scripts/profile_round.sh into gpurun_out/prof_<tag>/summary/:
   <tag>_kernel_stats_timed.csv  rti_kernel over the TIMED launches only (the last <steps> dispatches of the kernel trace: the run
                                 replays its captured steps untimed first to bring the clocks up), in rocprofv3's stats columns;
   <tag>_kernel_stats_all.csv    the kernel_stats table as rocprofv3 wrote it (every launch, warm-up included);
   <tag>_pmc_rti_kernel.json     per-dispatch mean / min / max of every counter over the last <steps> rti_kernel dispatches.
usage: summarise_profile.py <out dir> <tag> [steps]"""
import csv
import glob
import json
import os
import shutil
import sys

import numpy as np

out, tag = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
summ = os.path.join(out, "summary")
os.makedirs(summ, exist_ok=True)
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(summ, f"{tag}_kernel_stats_all.csv"))
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "rti_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    name = rows[-1]["Kernel_Name"]
    timed = [r for r in rows if r["Kernel_Name"] == name][-steps:]
    d = np.array([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in timed], dtype=float)
    gaps = np.array([int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(timed[:-1], timed[1:])], dtype=float)
    with open(os.path.join(summ, f"{tag}_kernel_stats_timed.csv"), "w") as fh:
        fh.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n')
        fh.write(f'"{name}",{len(d)},{int(d.sum())},{d.mean():.3f},100.0,{int(d.min())},{int(d.max())},{d.std():.3f}\n')
    print(f"timed launches: {len(d)} x {d.mean() / 1e3:.3f} us (min {d.min() / 1e3:.3f}, max {d.max() / 1e3:.3f}); gap between consecutive launches "
          f"median {np.median(gaps) / 1e3:.3f} us; all {len(rows)} rti launches of the trace: {np.mean([int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows]) / 1e3:.3f} us")
pmc = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            if "rti_kernel" not in r.get("Kernel_Name", ""):
                continue
            per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for name, d in per.items():
            ids = sorted(d, key=lambda x: int(x))
            vals = [d[i] for i in ids][-steps:]                      # the timed launches
            pmc[name] = {"dispatches": len(vals), "mean": sum(vals) / len(vals), "min": min(vals), "max": max(vals)}
json.dump(pmc, open(os.path.join(summ, f"{tag}_pmc_rti_kernel.json"), "w"), indent=1)
print(json.dumps({k: v["mean"] for k, v in pmc.items()}))
