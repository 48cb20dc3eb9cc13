#!/bin/bash
# Round profile of the headline workload (run on the GPU box through gpurun, from the repo root):
#   kernel trace + stats, then three separate PMC passes (never combined with tracing domains).
# The profiled command is the bench's own timed run WITH its untimed clock-warm replays (--clock-warm-ms 30, the default); the
# summary (scripts/summarise_profile.py) keeps only the LAST --steps launches of rti_kernel -- the timed ones -- so that the committed
# average is of the launches `value` is made of, not of warm-up launches at ramping clocks (VERDICT r4 #3a / weak #9).
# Outputs under gpurun_out/prof_<tag>/; scripts/summarise_profile.py turns them into the files kept in profiles/.
# Every native piece is built BEFORE the first rocprofv3 line and the profiled command is `bench.py --only-timed` (no
# oracle, no child processes): under rocprofv3 the preloaded profiler library initialises the GPU in every process, and a
# GPU-initialised process must not start a make -> sh -> cc chain on this pool.
set -u
TAG=${1:-r02}
EXTRA=${2:-}
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
[ -f "$PWD/ndp_nmpc_qd_amd/libndp_nmpc_hip.so" ] || python3 -c 'import __graft_entry__ as g; g.build()' > "$OUT/build.log" 2>&1
STEPS=200
BENCH="python3 $PWD/bench.py --only-timed --downwash-form fused --steps $STEPS --warmup 20 $EXTRA"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY \
    --output-format csv -d "$OUT/pmc_sq" -- $BENCH > "$OUT/pmc_sq.log" 2>&1
cd "$OLDPWD"
python3 scripts/summarise_profile.py "$OUT" "$TAG" "$STEPS"
