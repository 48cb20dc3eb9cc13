#!/bin/bash
# Round-3 evidence run on the GPU box (one gpurun call): rocprofv3 trace + PMC of the default configuration, phase stamps, the bench
# lines, the one-GPU checks of the N > 1 paths, the side measurements.  Everything lands in gpurun_out/r03/; the summaries are
# copied into profiles/ by hand afterwards.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03
rm -rf $O; mkdir -p $O
bash scripts/profile_round.sh r03 "--downwash-form fused --clock-warm-ms 0" > $O/profile_round.log 2>&1
cp gpurun_out/prof_r03/summary/* $O/ 2>/dev/null
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_ipm -- python3 $R/bench.py --only-timed --steps 200 --warmup 20 --qp-mode 1 --clock-warm-ms 0 > $O/trace_ipm.log 2>&1 )
find $O/trace_ipm -name "*kernel_stats.csv" -exec cp {} $O/r03_kernel_stats_ipm_always_b1024.csv \;
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_mixed -- python3 $R/bench.py --only-timed --steps 200 --warmup 20 --perturb mixed --batch 2048 --clock-warm-ms 0 > $O/trace_mixed.log 2>&1 )
find $O/trace_mixed -name "*kernel_stats.csv" -exec cp {} $O/r03_kernel_stats_mixed_b2048_work_list.csv \;
rm -rf $O/trace_ipm $O/trace_mixed
python scripts/batch_stamps.py 1024 2>&1 | grep -v amdgpu.ids > $O/r03_phase_stamps_b1024.txt
python scripts/batch_stamps.py 1024 nmpc 2>&1 | grep -v amdgpu.ids >> $O/r03_phase_stamps_b1024.txt
python scripts/launch_ramp.py 2>&1 | grep -v amdgpu.ids >> $O/r03_phase_stamps_b1024.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r03_bench_b1024_fused.json 2> $O/bench.err
timeout 600 python bench.py --steps 300 --warmup 30 --no-cpu-baseline > $O/r03_bench_b1024_fused_300steps.json 2>> $O/bench.err
timeout 600 python bench.py --workload nmpc --no-cpu-baseline --steps 300 --warmup 30 > $O/r03_bench_variant_nmpc.json 2>> $O/bench.err
for bb in 256 4096 16384; do timeout 600 python bench.py --only-timed --batch $bb --steps 300 --warmup 30 > $O/r03_bench_variant_b$bb.json 2>> $O/bench.err; done
timeout 600 python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline > $O/r03_bench_config4_vehicle_major_1gpu.json 2>> $O/bench.err
timeout 600 python bench.py --config 4 --placement formation --steps 100 --warmup 10 --no-cpu-baseline > $O/r03_bench_config4_formation_major_1gpu.json 2>> $O/bench.err
NDP_BENCH_SAME_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 \
    bench.py --gpus 2 --steps 100 --warmup 10 --no-cpu-baseline > $O/r03_peer_windows_2ranks_one_gpu.json 2>> $O/bench.err
NDP_BENCH_SAME_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29522 \
    bench.py --gpus 3 --config 4 --formations 1024 --steps 60 --warmup 10 --no-cpu-baseline > $O/r03_peer_windows_config4_3ranks_one_gpu.json 2>> $O/bench.err
timeout 900 python scripts/config5_precision.py 2>/dev/null | tail -1 > $O/r03_config5_precision.json
timeout 600 python scripts/queue_probe.py 2>/dev/null > $O/r03_mixed_workload_work_list.txt
( for th in 0 7; do NDP_PACK_THREADS=$th timeout 120 python scripts/host_path_rate.py 1024 2>/dev/null | grep -E "^B=|all loops"; done
  NDP_PACK_THREADS=7 timeout 120 python scripts/host_path_rate.py 4096 2>/dev/null | grep -E "^B=|all loops" ) > $O/r03_host_path.txt
timeout 300 python scripts/host_latency.py 2>/dev/null > $O/r03_host_latency.txt
./scripts/ubench/publish_copy.bin > $O/r03_ubench_publish_copy.txt 2>&1
timeout 300 python scripts/graph_overhead.py 2>/dev/null > $O/r03_graph_overhead.txt
( timeout 120 python scripts/overlap_probe.py 2>&1 | grep -v amdgpu.ids; GPU_MAX_HW_QUEUES=8 timeout 120 python scripts/overlap_probe.py 2>&1 | grep -v amdgpu.ids
  GPU_MAX_HW_QUEUES=8 timeout 120 python scripts/overlap_probe2.py 2>&1 | grep -v amdgpu.ids ) > $O/r03_overlap_probe.txt
ls -la $O | head -50
tail -3 $O/bench.err
