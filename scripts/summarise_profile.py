"""Summarises the rocprofv3 outputs of scripts/profile_round.sh into gpurun_out/prof_<tag>/summary/:
   <tag>_kernel_stats.csv (the kernel_stats table as rocprofv3 wrote it) and <tag>_pmc_rti_kernel.json
   (per-dispatch mean / min / max of every counter for the rti_kernel launches, warm-up launches dropped)."""
import csv
import glob
import json
import os
import shutil
import sys

out, tag = sys.argv[1], sys.argv[2]
summ = os.path.join(out, "summary")
os.makedirs(summ, exist_ok=True)
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(summ, f"{tag}_kernel_stats.csv"))
pmc = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        per = {}
        for r in rows:
            if "rti_kernel" not in r.get("Kernel_Name", ""):
                continue
            per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for name, d in per.items():
            ids = sorted(d, key=lambda x: int(x))
            vals = [d[i] for i in ids][20:] or [d[i] for i in ids]      # drop the warm-up launches
            pmc[name] = {"dispatches": len(vals), "mean": sum(vals) / len(vals), "min": min(vals), "max": max(vals)}
json.dump(pmc, open(os.path.join(summ, f"{tag}_pmc_rti_kernel.json"), "w"), indent=1)
print(json.dumps({k: v["mean"] for k, v in pmc.items()}))
for f in glob.glob(os.path.join(summ, "*kernel_stats.csv")):
    print(open(f).read()[:1500])
