#!/bin/bash
# Round profile of the headline workload (run on the GPU box through gpurun, from the repo root):
#   kernel trace + stats, then three separate PMC passes (never combined with tracing domains).
# Outputs under gpurun_out/prof_<tag>/; scripts/summarise_profile.py turns them into the files kept in profiles/.
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --no-cpu-baseline --steps 200 --warmup 20"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY \
    --output-format csv -d "$OUT/pmc_sq" -- $BENCH > "$OUT/pmc_sq.log" 2>&1
cd "$OLDPWD"
python3 scripts/summarise_profile.py "$OUT" "$TAG"
