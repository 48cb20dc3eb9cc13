#!/usr/bin/env python3
"""bench.py -- NMPC solves/s of the batched control step on N MI355X GPUs (one process per GPU).

    python bench.py --gpus 1 --steps 300 --warmup 30
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric / configs[2]): per GPU batch = 1024 independent quadrotor OCPs, N = 20,
1 SQP-RTI iteration per step + the downwash MLP (NDP controller), inputs resident in HBM.
A "step" is one control tick of the whole batch: [all-gather of neighbour windows when N > 1] ->
rti_kernel (gate + MLP fused in front of linearise, QP, full step).  Weak scaling: the per-GPU batch is fixed.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F64_MFMA_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix (= vector) peak, datasheet; v_mfma_f64_16x16x4 = 2048 FLOP / 64 clk / SIMD
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8.0 TB/s spec


def algorithmic_flops_per_solve(N, sweeps, downwash):
    """N*F_lin + n_fact*N*F_ric (+ F_mlp); F_lin ~ 4.3 kFLOP, F_ric ~ 8.2 kFLOP, F_mlp = 21*35072."""
    f = N * 4.3e3 + sweeps * N * 8.2e3
    return f, (21 * 35072.0 if downwash else 0.0)


def algorithmic_bytes_per_solve(N, downwash):
    """fp64 storage: read x0, xr, ur, X, U; write X, U, u0 (+ neighbour window and ego xy with downwash)."""
    b = 8 * (10 + 10 * (N + 1) + 4 * N + 2 * (10 * (N + 1) + 4 * N) + 4)
    if downwash:
        b += 8 * 10 * (N + 1) + 16
    return b


def measured_traffic(B, N, fused):
    """HBM bytes per rti_kernel launch from the committed rocprofv3 --pmc passes (separate FETCH_SIZE / WRITE_SIZE runs
    of this same command; KB units; FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md).  None if the
    profiled configuration differs from the one being run."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_rti_kernel_fused_b1024.json")
    if not (fused and B == 1024 and N == 20 and os.path.exists(path)):
        return None
    with open(path) as fh:
        pmc = json.load(fh)
    return (2.0 * pmc["FETCH_SIZE"]["mean"] + pmc["WRITE_SIZE"]["mean"]) * 1024.0


def effective_cores():
    """Host cores this process may really use: min(affinity mask, cgroup CPU quota).  OpenMP's default (all logical
    CPUs) oversubscribes a quota-limited container and gets throttled."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=1024, help="instances per GPU")
    ap.add_argument("--horizon", type=int, default=20)
    ap.add_argument("--workload", default="ndp_downwash", choices=["ndp_downwash", "nmpc"])
    ap.add_argument("--qp-mode", type=int, default=0, help="0 auto (exact early exit), 1 interior point always")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--placement", default="vehicle", choices=["vehicle", "formation"],
                    help="N > 1: vehicle-major (a formation's vehicles on different GPUs: one all-gather per step, the default) "
                         "or formation-major (all vehicles of a formation on one GPU: no exchange)")
    ap.add_argument("--no-graph", action="store_true", help="launch every step from the host instead of replaying a hipGraph (N = 1)")
    ap.add_argument("--cpu-passes", type=int, default=200)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import ndp_nmpc_qd_amd as ndp
    from ndp_nmpc_qd_amd import dist as ndist

    B, N = args.batch, args.horizon
    downwash = args.workload == "ndp_downwash"
    T = 8  # distinct control ticks cycled through (reference window slides by ts_nmpc = 0.02 s per tick)
    ticks = []
    for t in range(T):
        b = ndist.make_formation_shard(B, rank, world, N=N, t0=0.02 * t)
        ticks.append({k: torch.from_numpy(b[k]).to(dev) for k in ("x0", "xr", "ur", "other", "ego_xy")})
    host0 = ndist.make_formation_shard(B, rank, world, N=N, t0=0.0)

    eng = ndp.BatchedNMPC(B, N=N, disturbance=downwash, qp_mode=args.qp_mode, device=local_rank)
    # an explicit non-default stream: torch's default stream has handle 0, which the C-ABI reads as "use the
    # library's own stream" -- with a real handle the all-gather (N > 1) and the kernel are ordered on ONE stream
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    u0 = torch.empty(B, 4, dtype=torch.float64, device=dev)
    # N > 1: two gather buffers; the all-gather of tick i+1's reference windows (functions of time only) is started before
    # tick i's kernel is launched and runs on RCCL's stream beside it -- one all-gather and one kernel per step, overlapped
    gathered = [torch.empty(world, B, N + 1, 10, dtype=torch.float64, device=dev) for _ in range(2)] if world > 1 else None
    pending = {}
    exchange = world > 1 and args.placement == "vehicle"

    def prefetch(i):
        if downwash and exchange:
            pending[i] = ndist.exchange_neighbours_begin(ticks[i % T]["xr"], gathered[i % 2])   # one RCCL all-gather over xGMI

    def step(i):
        d = ticks[i % T]
        other = None
        if downwash:
            if exchange:
                if i not in pending:
                    prefetch(i)
                other = ndist.exchange_neighbours_end(pending.pop(i), gathered[i % 2])
                prefetch(i + 1)
            else:
                other = d["other"]
        eng.update_device(d["x0"], d["xr"], d["ur"], u0, other=other, ego_xy=d["ego_xy"] if downwash else None,
                          stream=stream)

    def fence():
        for w in list(pending.values()):      # a gather started for a tick that is never solved (end of a phase)
            w.wait()
        pending.clear()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- parity spot check against the CPU oracle (rank 0, first tick, 64 instances) before timing
    parity = None
    eng.reset_device(ticks[0]["xr"], ticks[0]["ur"], stream=stream)
    step(0)
    torch.cuda.synchronize()
    if rank == 0:
        from oracle import oracle as O
        ns = min(64, B)
        cfgo = O.default_cfg(N=N, use_fd=downwash)
        f = None
        if downwash:
            blob = np.fromfile(os.path.join(ROOT, "ndp_nmpc_qd_amd", "weights", "downwash_sn4.bin"), dtype="<f4")
            f = O.downwash_batch(blob, host0["other"][:ns], host0["xr"][:ns], host0["ego_xy"][:ns])
        Xo, Uo = host0["xr"][:ns].copy(), host0["ur"][:ns].copy()
        u_or, st_or, _ = O.step_batch(cfgo, host0["x0"][:ns], host0["xr"][:ns], host0["ur"][:ns], f, Xo, Uo)
        u_dev = u0[:ns].cpu().numpy()
        parity = float(np.max(np.abs(u_dev - u_or) / np.maximum(1.0, np.abs(u_or))))

    # ---- warm-up, then EXACTLY --steps timed steps between barrier + synchronize
    eng.reset_device(ticks[0]["xr"], ticks[0]["ur"], stream=stream)
    for i in range(args.warmup):
        step(i)
    fence()
    if downwash and exchange:                 # the first timed tick's windows are in place before the clock starts; every
        prefetch(args.warmup)                 # timed step then starts exactly one gather (the next tick's) and one kernel
        pending[args.warmup].wait()
        torch.cuda.synchronize()
        dist.barrier()
    # N = 1: the K timed steps are a launch-bound chain of dependent kernels -> one cycle through the T input ticks is
    # captured into a hipGraph (T kernel nodes) and replayed; a node of a replayed graph starts 1.6 us after its
    # predecessor ends, a host launch 2.6 us (scripts/ubench/launch_floor.hip).  The graph holds G = a multiple of T steps
    # (at most 256); every step still runs: K // G replays plus K % G host launches.  With the
    # neighbour exchange on (N > 1, vehicle-major) steps are launched from the host: each also starts an RCCL all-gather.
    graph, launch_mode = None, "host launch per step"
    if not exchange and not args.no_graph and args.steps >= T:
        try:
            base = ((args.warmup + T - 1) // T) * T          # a multiple of T: the replayed cycle starts at tick 0
            G = min(256, args.steps) // T * T
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream, capture_error_mode="relaxed"):
                for i in range(G):
                    step(base + i)
            torch.cuda.set_stream(stream)
            graph.replay()                                    # instantiate / upload outside the timed region
            torch.cuda.synchronize()
            launch_mode = f"hipGraph of {G} steps replayed"
        except Exception as e:                                # capture unsupported: fall back, say so
            graph, launch_mode = None, f"host launch per step (graph capture failed: {type(e).__name__})"
            torch.cuda.set_stream(stream)
            torch.cuda.synchronize()
    t0 = time.perf_counter()
    if graph is not None:
        for _ in range(args.steps // G):
            graph.replay()
        for i in range(args.steps % G):
            step(i)
    else:
        eng.timing_enable(8)  # HIP events around every 8th launch (an event pair per launch costs a dispatch gap)
        for i in range(args.steps):
            step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    if graph is not None:     # the dominant kernel's duration (roofline): HIP events around host-launched steps, outside the timed region
        eng.timing_enable(1)
        for i in range(64):
            step(i)
        torch.cuda.synchronize()
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    rti_ms, rti_n = eng.timing_read("rti")
    mlp_ms, mlp_n = eng.timing_read("mlp")
    eng.timing_enable(0)
    st, it = eng.status()
    bad = int((st != 0).sum())
    sweeps = float(np.mean(np.where(it > 0, 1 + 2 * it, 1))) if args.qp_mode == 0 else float(np.mean(2 * it))
    if world > 1:
        agg = torch.tensor([bad], dtype=torch.int64, device=dev)
        dist.all_reduce(agg)
        bad = int(agg.item())

    if rank == 0:
        total = B * world * args.steps
        value = total / elapsed
        f_qp, f_mlp = algorithmic_flops_per_solve(N, sweeps, downwash)
        rti_s = rti_ms * 1e-3 / max(rti_n, 1)
        mlp_s = mlp_ms * 1e-3 / max(mlp_n, 1) if mlp_n else 0.0
        fused = downwash and mlp_n == 0          # gate + MLP run inside rti_kernel (one launch per step)
        ach_tf = f_qp * B / rti_s / 1e12
        abytes = algorithmic_bytes_per_solve(N, downwash)
        # matrix-pipe occupancy estimate: every v_mfma_f64_16x16x4 / v_mfma_f32_32x32x2 holds the SIMD's pipe 64 cycles
        # (measured, scripts/ubench); one instance per SIMD: 266 f64 MFMAs per sweep at N = 20; the MLP tile adds 12 f32 (64 clk)
        # and 96 fp16 (32 clk) MFMAs.  PMC cross-check: SQ_INSTS_MFMA = 374 per instance, SQ_VALU_MFMA_BUSY_CYCLES = 21.4 k per SIMD
        n_f64 = sweeps * (6 + 8 * (N - 1) + N + 16 * ((N - 1) // 8) + 4 * N)
        pipe_cycles = n_f64 * 64 + ((12 * 64 + 96 * 32) if fused else 0)
        out = {
            "metric": "NMPC solves/sec (N=20, 1 RTI iter + downwash MLP) at batch",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"batch={B}/GPU independent quadrotors, N={N}, 1 RTI iter, "
                                   + ("MLP downwash on (NDP controller, gate+MLP fused into the RTI launch)" if fused else
                                      "MLP downwash on (NDP controller)" if downwash else "no downwash (NMPC controller)")
                                   + (", neighbour windows all-gathered over RCCL" if exchange else ", formation-major placement (no exchange)" if world > 1 else ""),
                       "batch_per_gpu": B, "horizon": N, "n_rti": 1, "qp_mode": "auto" if args.qp_mode == 0 else "ipm_always",
                       "launch": launch_mode,
                       "parallelism": f"instances sharded x{world}"},
            "roofline": {"kernel": "rti_kernel", "bound": "mfma", "achieved": ach_tf, "peak": F64_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach_tf / F64_MFMA_PEAK_TFLOPS, "traffic": measured_traffic(B, N, fused),
                         "traffic_note": "HBM bytes per launch, PMC (profiles/r01_pmc_rti_kernel_fused_b1024.json); algorithmic bytes per launch = %d" % (abytes * B),
                         "kernel_us": rti_s * 1e6, "flops_per_solve_f64": f_qp, "riccati_sweeps_per_solve": sweeps,
                         "fused_mlp_flops_per_solve": f_mlp if fused else 0.0,
                         "mfma_pipe_busy_est": pipe_cycles / (rti_s * 2.4e9),
                         "hbm_algorithmic_GBps": abytes * B / (rti_s + mlp_s) / 1e9,
                         "hbm_frac": abytes * B / (rti_s + mlp_s) / 1e9 / HBM_PEAK_GBS,
                         "mlp_kernel_us": mlp_s * 1e6 if mlp_n else None},
            "parity_max_rel_vs_oracle": parity, "instances_not_converged": bad,
        }
        if not args.no_cpu_baseline and world == 1:
            from oracle import oracle as O
            cfgo = O.default_cfg(N=N, use_fd=downwash)
            nthr = min(O.num_threads(), effective_cores())
            blob = np.fromfile(os.path.join(ROOT, "ndp_nmpc_qd_amd", "weights", "downwash_sn4.bin"), dtype="<f4")
            Xo, Uo = host0["xr"].copy(), host0["ur"].copy()
            O.step_batch(cfgo, host0["x0"], host0["xr"], host0["ur"], None, Xo, Uo, nthreads=nthr)       # warm the thread pool
            Xo, Uo = host0["xr"].copy(), host0["ur"].copy()
            tc = time.perf_counter()
            for _ in range(args.cpu_passes):
                f = O.downwash_batch(blob, host0["other"], host0["xr"], host0["ego_xy"], nthreads=nthr) if downwash else None
                O.step_batch(cfgo, host0["x0"], host0["xr"], host0["ur"], f, Xo, Uo, nthreads=nthr)
            tc = time.perf_counter() - tc
            out["cpu_baseline"] = {"value": B * args.cpu_passes / tc, "unit": "solves/s", "cores": nthr, "kind": "port",
                                   "sample": f"{args.cpu_passes} control ticks of the same batch={B} workload "
                                             f"(oracle/ndp_oracle.c: fp64 RTI + IPM always, fp32 MLP), OpenMP over instances"}
            # BASELINE config 1 (one vehicle, N = 20, no downwash, 1 RTI iteration): latency of the CPU restatement on one
            # thread next to the same single instance on the GPU (host call to host return, device-resident inputs)
            c1 = O.default_cfg(N=N, use_fd=False)
            X1, U1 = host0["xr"][:1].copy(), host0["ur"][:1].copy()
            n1 = 300
            t1 = time.perf_counter()
            for _ in range(n1):
                O.step_batch(c1, host0["x0"][:1], host0["xr"][:1], host0["ur"][:1], None, X1, U1, nthreads=1)
            t1 = (time.perf_counter() - t1) / n1
            e1 = ndp.BatchedNMPC(1, N=N, device=local_rank)
            d1 = {k: ticks[0][k][:1].contiguous() for k in ("x0", "xr", "ur")}
            u1 = torch.empty(1, 4, dtype=torch.float64, device=dev)
            e1.reset_device(d1["xr"], d1["ur"], stream=stream)
            for _ in range(20):
                e1.update_device(d1["x0"], d1["xr"], d1["ur"], u1, stream=stream)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(n1):
                e1.update_device(d1["x0"], d1["xr"], d1["ur"], u1, stream=stream)
                torch.cuda.synchronize()
            t2 = (time.perf_counter() - t2) / n1
            out["config1_single_vehicle"] = {"cpu_restatement_us_per_solve_1_thread": t1 * 1e6,
                                             "gpu_us_per_solve_sync_each_call": t2 * 1e6,
                                             "note": "N=%d, no downwash, 1 RTI iteration; CPU = oracle (interior point always)" % N}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
