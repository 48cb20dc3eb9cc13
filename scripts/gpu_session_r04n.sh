#!/bin/bash
# kernel timeline of the peer / peer_ahead forms (config 4 shard, one GPU)
O=$PWD/gpurun_out/r04n; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-configs --clock-warm-ms 0 --exchange peer > $O/trace.log 2>&1 )
find $O/trace -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/trace
O=$O python3 - <<'PY'
import csv,collections,os
rows=list(csv.DictReader(open(os.environ.get("O","gpurun_out/r04n")+"/kernel_trace.csv")))
print(len(rows), rows[0].keys())
PY
tail -n 3 $O/trace.log
