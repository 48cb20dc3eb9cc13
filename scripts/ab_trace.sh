#!/bin/bash
# Kernel-trace A/B of the headline launch on one box (see scripts/ab_headline.sh for the staging of .ab_base/): rocprofv3's own
# per-launch durations of the timed launches, gaps between launches excluded.  gpurun -- 'bash scripts/ab_trace.sh'
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/ab_trace; rm -rf $O; mkdir -p $O
run() {   # name dir lib
  (cd /tmp && NDP_NMPC_LIB=$3 rocprofv3 --kernel-trace --output-format csv -d $O/$1 -- python3 $2/bench.py --only-timed --downwash-form fused --steps 200 --warmup 20 > $O/$1.log 2>&1)
  python3 - $O/$1 $1 <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rti_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
last = rows[-200:]
d = [b - a for a, b in last]
gaps = [last[i + 1][0] - last[i][1] for i in range(len(last) - 1)]
import statistics as S
print("%s: launches %d | duration mean %.1f median %.1f min %d max %d ns | gap between launches median %.1f mean %.1f ns | start-to-start mean %.1f" %
      (sys.argv[2], len(rows), S.mean(d), S.median(d), min(d), max(d), S.median(gaps), S.mean(gaps), (last[-1][0] - last[0][0]) / (len(last) - 1)))
PY
}
run new $R $R/ndp_nmpc_qd_amd/libndp_nmpc_hip_dev.so
run base $R/.ab_base $R/.ab_base/ndp_nmpc_qd_amd/libndp_nmpc_hip.so
run new2 $R $R/ndp_nmpc_qd_amd/libndp_nmpc_hip_dev.so
run base2 $R/.ab_base $R/.ab_base/ndp_nmpc_qd_amd/libndp_nmpc_hip.so
