#!/bin/bash
# Round-6 GPU sessions, one stage per gpurun call:  gpurun --timeout 1800 -- 'bash scripts/gpu_r06.sh <stage>'
#   tests    : the whole -m gpu suite
#   bench    : the driver-shaped line (bench.py --steps 20 --warmup 5) + a 300-step run + the mixed / constrained / interior-point --only-timed runs
#   profile  : kernel trace + PMC passes of the headline (scripts/profile_round.sh r06), of the MIXED workload (r06mixed) and of
#              config 5's condensed study kernels (kernel trace + SQ counters: matrix-pipe busy)
#   side     : tick rate, config 5 table (scripts/config5_condensed.py), rows profile
# then: bash scripts/collect_profiles.sh r06 (+ the copies at the end of this file's profile stage)
STAGE=${1:-tests}; shift
export TMPDIR=/tmp
O=$PWD/gpurun_out/r06_$STAGE; mkdir -p $O
R=$PWD
hl() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('value %.4g  ms/step %.5f  kernel_us %.2f  parity %s' % (d['value'] or -1, d['ms_per_step'], r.get('kernel_us') or 0, d.get('parity_max_rel_vs_oracle')))"; }
case $STAGE in
tests)
  timeout 1700 python -m pytest tests -m gpu -q --timeout 600 "$@" 2>&1 | tail -40 > $O/tests.txt; tail -8 $O/tests.txt
  ;;
bench)
  timeout 900 python3 bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; tail -c 300 $O/bench_driver.err; wc -c $O/bench_driver.json; hl < $O/bench_driver.json
  timeout 300 python3 bench.py --only-timed --steps 300 --warmup 30 --downwash-form fused 2>> $O/bench_driver.err | tee $O/bench300.json | hl
  timeout 300 python3 bench.py --only-timed --steps 200 --warmup 20 --perturb mixed 2>> $O/bench_driver.err | tee $O/bench_mixed.json | hl
  timeout 300 python3 bench.py --only-timed --steps 200 --warmup 20 --perturb mixed --as-iter-max 0 2>> $O/bench_driver.err | tee $O/bench_mixed_as_off.json | hl
  timeout 300 python3 bench.py --only-timed --steps 200 --warmup 20 --qp-mode 1 2>> $O/bench_driver.err | tee $O/bench_ipm_always.json | hl
  ;;
profile)
  bash scripts/profile_round.sh r06
  bash scripts/profile_round.sh r06mixed "--perturb mixed"
  for p in 5 6 0; do
    (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/cond_trace_$p -- python3 $R/scripts/cond_driver.py $p > $O/cond_trace_$p.log 2>&1)
    (cd /tmp && rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $O/cond_pmc_$p -- python3 $R/scripts/cond_driver.py $p > $O/cond_pmc_$p.log 2>&1)
  done
  python3 scripts/summarise_pmc.py rti_kernel $O/r06_config5_condensed_kernels.json $O/cond_trace_5 $O/cond_pmc_5 $O/cond_trace_6 $O/cond_pmc_6 $O/cond_trace_0 $O/cond_pmc_0 | head -c 1200
  python3 scripts/tick_stamps.py uniform 2>/dev/null | tee $O/tick_stamps.txt | tail -12
  ;;
side)
  timeout 300 python3 scripts/tick_rate.py 2> $O/tick_rate.err | tee $O/tick_rate.txt
  timeout 300 python3 scripts/tick_rate.py --no-estimator --uniform-t 2>> $O/tick_rate.err | head -3 | tee -a $O/tick_rate.txt
  timeout 900 python3 scripts/config5_condensed.py 4096 40 2>/dev/null | tee $O/config5_condensed.txt
  timeout 300 python3 scripts/config5_condensed.py 1024 20 2>/dev/null | tee -a $O/config5_condensed.txt
  bash scripts/ab_headline.sh 3 2>&1 | tee $O/ab_headline.txt
  ;;
esac
