#!/bin/bash
# Round-5 GPU sessions, one stage per gpurun call:  gpurun --timeout 1800 -- 'bash scripts/gpu_r05.sh <stage> [args]'
#   tick     : tests of the tick and the list it reads + scripts/tick_rate.py + the headline's --only-timed run (regression check)
#   tests    : the whole -m gpu suite
#   bench    : the driver-shaped line (bench.py --steps 20 --warmup 5) + a 300-step run
#   profile  : scripts/profile_round.sh r05 (kernel trace + PMC passes of the headline), scripts/profile_rows.sh r05, HBM ceilings, tick stamps
#   side     : 50 Hz cadence, host latency, stress parity
# then: bash scripts/collect_profiles.sh r05  (gpurun_out/ -> profiles/)
# Everything lands in gpurun_out/r05_<stage>/.
STAGE=${1:-tick}; shift
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05_$STAGE; mkdir -p $O
hl() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('value %.4g  ms/step %.5f  kernel_us %.2f  parity %s' % (d['value'] or -1, d['ms_per_step'], r.get('kernel_us') or 0, d.get('parity_max_rel_vs_oracle')))"; }
case $STAGE in
tick)
  timeout 900 python -m pytest tests/test_tick.py tests/test_round5_gpu.py tests/test_ref_window_row.py tests/test_throttle_row.py tests/test_relay_and_plant_rows.py -m gpu -q --timeout 600 -x 2>&1 | tail -30 > $O/tests.txt; tail -15 $O/tests.txt
  timeout 300 python3 scripts/tick_rate.py 2> $O/tick_rate.err | tee $O/tick_rate.txt; tail -3 $O/tick_rate.err
  timeout 300 python3 scripts/tick_rate.py --no-estimator 2>> $O/tick_rate.err | head -3 | tee -a $O/tick_rate.txt
  timeout 300 python3 scripts/tick_rate.py --no-estimator --uniform-t 2>> $O/tick_rate.err | head -3 | tee -a $O/tick_rate.txt
  R=$PWD; for m in noest noest_uni; do rm -rf $O/trace_$m; (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/trace_$m -- python3 $R/scripts/tick_trace.py run $m > $O/trace_$m.log 2>&1); python3 scripts/tick_trace.py summarise $O/trace_$m | tee -a $O/tick_rate.txt; done
  timeout 300 python3 bench.py --only-timed --steps 300 --warmup 30 --downwash-form fused 2> $O/bench300.err | tee $O/bench300.json | hl
  ;;
tests)
  timeout 1700 python -m pytest tests -m gpu -q --timeout 600 "$@" 2>&1 | tail -40 > $O/tests.txt; tail -15 $O/tests.txt
  ;;
bench)
  timeout 600 python3 bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; tail -c 1500 $O/bench_driver.err; wc -c $O/bench_driver.json; hl < $O/bench_driver.json
  timeout 300 python3 bench.py --only-timed --steps 300 --warmup 30 --downwash-form fused 2>> $O/bench_driver.err | tee $O/bench300.json | hl
  ;;
profile)
  bash scripts/profile_round.sh r05 "$@"
  bash scripts/profile_rows.sh r05
  python3 scripts/hbm_ceiling.py 2>/dev/null | tee $O/hbm_ceiling.txt
  python3 scripts/tick_stamps.py uniform 2>/dev/null | tee $O/tick_stamps.txt
  ;;
side)
  timeout 600 python3 scripts/cadence_50hz.py --ticks 300 --wake 2>/dev/null | tee $O/cadence_50hz.txt
  timeout 300 python3 scripts/host_latency.py 2>/dev/null | tee $O/host_latency.txt
  timeout 900 python3 scripts/stress_parity.py 2>/dev/null | tee $O/stress_parity.txt | tail -3
  ;;
*) echo "unknown stage $STAGE"; exit 2;;
esac
