#!/bin/bash
# the bench lines kept under profiles/ for tag r04 (run with the r04 rocprofv3 summaries already in profiles/)
O=gpurun_out/r04e; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r04_bench_b1024_fused.json 2> $O/bench.err
timeout 600 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-configs > $O/r04_bench_b1024_fused_300steps.json 2>> $O/bench.err
timeout 600 python bench.py --workload nmpc --no-cpu-baseline --no-configs --steps 300 --warmup 30 > $O/r04_bench_variant_nmpc.json 2>> $O/bench.err
python - <<'PY'
import json
for f in ("r04_bench_b1024_fused","r04_bench_b1024_fused_300steps","r04_bench_variant_nmpc"):
    d=json.loads(open("gpurun_out/r04e/%s.json"%f).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f, "%.2f M"%(d["value"]/1e6), "%.2f us/step"%(d["ms_per_step"]*1e3), "frac %.4f frac_rocprof %s kernel_us %.2f rocprof %s dispatch %.2f tag %s mism %s"%(r["frac"], r["frac_rocprof"], r["kernel_us"], r["kernel_us_rocprof"], r["kernel_us_dispatch_events"], r["profile_tag"], r["profile_mismatch"]))
    if "scaling_baseline" in d: print("   baseline", {k:("%.2f"%(v.get("value",0)/1e6), "%.1f us"%(v.get("ms_per_step",0)*1e3), v.get("ok")) for k,v in d["scaling_baseline"]["forms"].items()})
    if "cpu_baseline" in d: print("   cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["ipm_always_value"], "hostincl", d["value_host_inclusive"]["value"], d["value_host_inclusive"]["one_tick_at_a_time"]["value"], "cfg1", d["config1_single_vehicle"]["gpu_drop_in_update_us"], "ipm", d["ipm_always"]["value"])
PY
