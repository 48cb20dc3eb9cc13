#!/bin/bash
# Round-4 evidence run on the GPU box (one gpurun call): rocprofv3 trace + PMC of the default configuration, phase stamps, the bench
# lines (driver shape, 300 steps, variants), the one-GPU checks of the N > 1 paths (incl. bench.py launching its own ranks),
# config 5 / mixed-workload side measurements.  Everything lands in gpurun_out/r04/; the summaries are copied into profiles/.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04
rm -rf $O; mkdir -p $O
bash scripts/profile_round.sh r04 "--downwash-form fused --clock-warm-ms 0" > $O/profile_round.log 2>&1
cp gpurun_out/prof_r04/summary/* $O/ 2>/dev/null
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_ipm -- python3 $R/bench.py --only-timed --steps 200 --warmup 20 --qp-mode 1 --clock-warm-ms 0 > $O/trace_ipm.log 2>&1 )
find $O/trace_ipm -name "*kernel_stats.csv" -exec cp {} $O/r04_kernel_stats_ipm_always_b1024.csv \;
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_mixed -- python3 $R/bench.py --only-timed --steps 200 --warmup 20 --perturb mixed --batch 2048 --clock-warm-ms 0 > $O/trace_mixed.log 2>&1 )
find $O/trace_mixed -name "*kernel_stats.csv" -exec cp {} $O/r04_kernel_stats_mixed_b2048_work_list.csv \;
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c4 -- python3 $R/bench.py --only-timed --config 4 --formations 512 --placement formation --steps 200 --warmup 20 --clock-warm-ms 0 > $O/trace_c4.log 2>&1 )
find $O/trace_c4 -name "*kernel_stats.csv" -exec cp {} $O/r04_kernel_stats_config4_1536_formation_major.csv \;
rm -rf $O/trace_ipm $O/trace_mixed $O/trace_c4
python scripts/batch_stamps.py 1024 2>&1 | grep -v amdgpu.ids > $O/r04_phase_stamps_b1024.txt
python scripts/batch_stamps.py 1024 nmpc 2>&1 | grep -v amdgpu.ids >> $O/r04_phase_stamps_b1024.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r04_bench_b1024_fused.json 2> $O/bench.err
timeout 600 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-configs > $O/r04_bench_b1024_fused_300steps.json 2>> $O/bench.err
for bb in 256 4096 16384; do timeout 600 python bench.py --only-timed --batch $bb --steps 300 --warmup 30 > $O/r04_bench_variant_b$bb.json 2>> $O/bench.err; done
for ord in leaders_first interleaved; do
  timeout 600 python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline --instance-order $ord > $O/r04_bench_config4_vehicle_major_1gpu_$ord.json 2>> $O/bench.err
  timeout 600 python bench.py --config 4 --placement formation --steps 100 --warmup 10 --no-cpu-baseline --instance-order $ord > $O/r04_bench_config4_formation_major_1gpu_$ord.json 2>> $O/bench.err
  timeout 600 python bench.py --config 4 --formations 512 --steps 200 --warmup 20 --no-cpu-baseline --instance-order $ord > $O/r04_bench_config4_shard1536_vehicle_major_$ord.json 2>> $O/bench.err
  timeout 600 python bench.py --config 4 --formations 512 --placement formation --steps 200 --warmup 20 --no-cpu-baseline --instance-order $ord > $O/r04_bench_config4_shard1536_formation_major_$ord.json 2>> $O/bench.err
done
# bench.py launching its own ranks (no torchrun around it): two ranks time-slicing this one GPU, peer form (RCCL refuses two ranks on one device)
NDP_BENCH_SAME_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 100 --warmup 10 --no-cpu-baseline > $O/r04_self_launch_2ranks_one_gpu.json 2> $O/r04_self_launch_2ranks_one_gpu.err
NDP_BENCH_SAME_DEVICE=1 timeout 600 python bench.py --gpus 8 --config 4 --steps 60 --warmup 10 --no-cpu-baseline --peer-timeout-us 5000000 > $O/r04_self_launch_config4_8ranks_one_gpu.json 2> $O/r04_self_launch_config4_8ranks_one_gpu.err
timeout 900 python scripts/config5_precision.py 2>/dev/null | tail -1 > $O/r04_config5_precision.json
timeout 600 python scripts/queue_probe.py 2>/dev/null > $O/r04_mixed_workload_work_list.txt
timeout 300 python scripts/host_latency.py 2>/dev/null > $O/r04_host_latency.txt
timeout 300 python scripts/refine_probe.py 2>/dev/null > $O/r04_refine_probe.txt
timeout 300 python scripts/rccl_step_cost.py 2>/dev/null > $O/r04_rccl_step_cost.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/r04_gpu_tests.txt
ls -la $O | head -60
tail -3 $O/bench.err
