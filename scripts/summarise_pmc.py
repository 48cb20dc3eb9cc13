"""Per-kernel means of the counters of one or more rocprofv3 --pmc output directories, and of a kernel trace if given:
   summarise_pmc.py <kernel name substring> <out.json> <dir> [<dir> ...]"""
import csv, glob, json, sys
key, outp, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
res = {}
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            if key not in r.get("Kernel_Name", ""):
                continue
            per.setdefault((r["Kernel_Name"][:90], r["Counter_Name"]), {}).setdefault(r["Dispatch_Id"], 0.0)
            per[(r["Kernel_Name"][:90], r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for (kn, cn), v in per.items():
            vals = list(v.values())
            res.setdefault(kn, {})[cn] = {"dispatches": len(vals), "mean": sum(vals) / len(vals)}
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        dur = {}
        for r in csv.DictReader(open(f)):
            if key in r["Kernel_Name"]:
                dur.setdefault(r["Kernel_Name"][:90], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for kn, v in dur.items():
            res.setdefault(kn, {})["duration_ns"] = {"dispatches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
json.dump(res, open(outp, "w"), indent=1)
print(json.dumps(res)[:1500])
