#!/bin/bash
# rocprofv3 evidence for the kernels beside rti_kernel (VERDICT r4 #3b): kernel trace + separate PMC passes of scripts/rows_driver.py.
#   gpurun --timeout 1800 -- 'bash scripts/profile_rows.sh r05'
# -> gpurun_out/prof_rows_<tag>/summary/<tag>_kernel_stats_rows.csv, <tag>_pmc_rows.json (scripts/summarise_rows.py); copy into profiles/.
# The program behind `--` is python3 itself (no env / bash -c hop: the profiler's preloaded library has initialised the GPU).
set -u
TAG=${1:-r05}
OUT=$PWD/gpurun_out/prof_rows_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
R=$PWD
DRV="python3 $R/scripts/rows_driver.py --batch 262144 --reps 30"
cd /tmp
rocprofv3 --list-avail > "$OUT/list_avail.txt" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $DRV > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $DRV > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $DRV > "$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY \
    --output-format csv -d "$OUT/pmc_sq" -- $DRV > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM \
    --output-format csv -d "$OUT/pmc_f64" -- $DRV > "$OUT/pmc_f64.log" 2>&1
cd "$R"
python3 scripts/summarise_rows.py "$OUT" "$TAG"
