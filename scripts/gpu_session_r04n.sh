#!/bin/bash
# which resources RCCL's all-gather kernel asks for (LDS, registers, workgroup) -- kernel trace of the rccl form
O=$PWD/gpurun_out/r04n; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --config 4 --formations 512 --exchange rccl --graph-exchange off --steps 40 --warmup 4 --no-cpu-baseline --clock-warm-ms 0 > $O/trace.log 2>&1 )
find $O/trace -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/trace
O=$O python3 - <<'PY'
import csv,collections,os
rows=list(csv.DictReader(open(os.environ["O"]+"/kernel_trace.csv")))
seen={}
for r in rows:
    n=r["Kernel_Name"][:70]
    if n not in seen:
        seen[n]=(r["LDS_Block_Size"],r["Scratch_Size"],r["VGPR_Count"],r["Accum_VGPR_Count"],r["SGPR_Count"],r["Workgroup_Size_X"],r["Grid_Size_X"])
for n,v in seen.items(): print(v, n)
PY
