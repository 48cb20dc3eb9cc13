#!/usr/bin/env python3
"""bench.py -- NMPC solves/s of the batched control step on N MI355X GPUs (one process per GPU).

    python bench.py --gpus 1 --steps 300 --warmup 30
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric / configs[2]): per GPU batch = 1024 independent quadrotor OCPs, N = 20,
1 SQP-RTI iteration per step + the downwash MLP (NDP controller), inputs resident in HBM.
A "step" is one control tick of the whole batch: [the neighbour-window exchange of that tick when N > 1 or --config 4] ->
rti_kernel (gate + MLP fused in front of linearise, QP, full step).  Weak scaling: the per-GPU batch is fixed.
All timed steps are captured in ONE hipGraph (chains of graphs beyond 1024 steps) and timed on the host clock around the
replay (barrier + synchronize on both sides, max over ranks); `roofline.achieved` uses HIP events on the launch stream.
`--exchange both|rccl|peer` (N > 1 / config 4): the per-tick exchange as one RCCL all-gather of the position / velocity
columns, or as publish + epoch flags through peer-mapped windows; both are timed, `value` is the RCCL form.
`--downwash-form both|fused|prefetch` (N = 1): the MLP fused into rti_kernel (= `value`), and the downwash of tick t + 1 on a
second stream beside the control step of tick t (reported beside it in `downwash_forms`).
`--config 4` is BASELINE configs[3]: 4096 three-vehicle formations (12 288 instances) split over the GPUs (strong scaling).
Prints ONE JSON line on rank 0.  Exit code 1 (and "value": null) if the parity spot check fails or instances did not converge.
"""
import argparse
import csv
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F64_MFMA_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix (= vector) peak, datasheet; v_mfma_f64_16x16x4 = 2048 FLOP / 64 clk / SIMD
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8.0 TB/s spec
PEER_FORMS = ("peer", "peer_ahead")   # the publish / subscribe forms of the per-tick neighbour exchange (see mode_names)
PROFILE_TAGS = ("r06", "r05", "r04", "r03", "r02")  # newest first: profiles/<tag>_kernel_stats_fused_b1024.csv + <tag>_pmc_rti_kernel.json = the DEFAULT configuration's profile
PROFILE_TOLERANCE = 0.25       # a committed profile whose kernel duration is further than this from the live HIP-event duration is refused


REPEATS = 12          # extra timed passes over the same K steps behind the timed region (see run_mode)


def compact(o, digits=5):
    """The line must fit the driver's record (the tail of stdout it keeps is ~8 KB): floats to `digits` significant digits, recursively."""
    if isinstance(o, float):
        return o if o != o or o in (float("inf"), float("-inf")) or o == 0.0 else float(f"%.{digits}g" % o)
    if isinstance(o, dict):
        return {k: compact(v, digits) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [compact(v, digits) for v in o]
    if isinstance(o, (np.floating,)):
        return compact(float(o), digits)
    if isinstance(o, (np.integer,)):
        return int(o)
    return o


# What the keys of the line mean.  Printed ONCE on stderr (a JSON object behind "[bench legend]"): the line itself carries numbers.
LEGEND = {
    "value": "whole-job solves/s of the timed steps, inputs resident in HBM (device-resident metric); ms_per_step = wall clock / steps",
    "config.launch": "how the timed steps were issued (all --steps captured into ONE hipGraph when no exchange stream is involved)",
    "config.neighbour_exchange": "fused = one GPU, gate + MLP inside the control-step launch; prefetch = downwash of tick t+1 on a second stream; "
                                 "rccl = one library-issued ncclAllGather of the [B,N+1,6] windows per step; rccl_graph = the same captured; "
                                 "peer = publish into a peer-mapped slot + epoch words, read over xGMI by the kernel",
    "roofline": "rti_kernel against the FP64 matrix / vector peak (78.6 TFLOP/s): achieved = algorithmic f64 FLOPs per launch / kernel_us; "
                "kernel_us = HIP events on the launch stream around the timed region / steps; kernel_us_dispatch = start / stop events on the "
                "dispatch packets of 64 host-launched steps; kernel_us_rocprof / frac_rocprof / traffic (HBM bytes per launch, PMC: 2 x "
                "FETCH_SIZE + WRITE_SIZE, KB units) / clock from the committed profiles/<profile_tag>_* files, only when they describe this "
                "configuration and agree with the live duration (else profile_mismatch); hbm_frac = algorithmic bytes / kernel_us / 8 TB/s",
    "forms": "the same timed steps in the other downwash form (prefetch: second stream, late force); per form value / us per step / parity",
    "scaling_baseline": "the SAME steps in the form every N > 1 line runs (one-rank communicator: rccl host-launched / rccl_graph captured / "
                        "peer): scaling efficiency = value(N) / (N x scaling_baseline.value), not against the exchange-free headline",
    "configs": "every other BASELINE config measured in this run on one GPU, each with its own parity check vs the CPU oracle (64 instances "
               "sampled across the local order): config2 = batch 1024, no downwash; config4_one_gpu = 4096 formations (12288 instances) and "
               "one rank's share of the 8-GPU run (1536), formation-major (no exchange) and vehicle-major (one-rank RCCL all-gather per step); "
               "config5 = N 40, 2 RTI, batch 4096: sweeps on the f64 / fp32 / bf16 matrix instructions, nominal and perturbed starts; "
               "condensed_fp32 / condensed_bf16 = configs[4] as worded: every QP's first solve CONDENSED on v_mfma_f32_16x16x4_f32 / v_mfma_f32_16x16x16_bf16 "
               "+ fp32 Cholesky in LDS (a study mode, csrc/cond_qp.hpp), kept if inside the box in fp64, else the fp64 path (qps_kept_condensed of 2 per step); "
               "err = max rel u0 error vs the fp64 oracle, ipm = fraction of instances in the interior-point loop, bad = status != 0",
    "repeat": "REPEATS more passes over the same --steps steps (same graph) behind the timed region, each timed alone: ms_per_step = [min, median, max]; "
              "value_median = solves/s at the median pass; `value` itself is the contract's one pass",
    "tick_remote": "N > 1: ndp_tick with every vehicle's neighbour on the NEXT rank (ndp_tick_config_remote): per control period list advance + estimator -> "
                   "window columns -> all-gather of the [B, N+1, 6] windows -> control step, on one stream; value = whole-job solves/s over the slowest rank; "
                   "parity = rank 0's u0 against the oracle fed the rows the exchange delivered; rows_ok = those rows are the neighbour rank's windows; "
                   "with the library's communicator the headline figures are the form with the exchange ONE CONTROL PERIOD AHEAD (ndp_xchg_tick_step / _begin: advance, "
                   "columns and all-gather of tick i + 1 on the exchange's stream beside tick i's control step) and `serial` holds the one-stream form's",
    "tick.remote": "ndp_tick with neighbours on OTHER ranks (ndp_tick_config_remote): list advance -> window columns -> [exchange] -> control step, three launches; "
                   "one rank with its own windows as the gathered buffer, odometry in HBM, host-launched; value_one_launch_device_resident = ndp_tick_device on the same inputs",
    "ipm_always": "qp_mode 1: every instance runs the interior-point loop like HPIPM does",
    "mixed": "perturbed starts (0.5 m / 1 m/s / 0.15): ~20 % of the instances have inputs on their bounds (constrained = the fraction at the last tick); "
             "value = the default mode (active-set iterations, sets kept between ticks; ipm = fraction that still needed the interior-point loop), "
             "value_as_off = as_iter_max 0 (rounds 1-5: the interior-point loop on those instances); wq = work list on / off / automatic",
    "constrained": "the mixed workload's constrained instances ONLY (every instance has an input on its bound at tick 0, picked by one tick of the device "
                   "itself from a pool of 8 x batch); value / value_as_off as in `mixed`; sweeps_* = Riccati sweeps of the last tick's QPs, pins_mean = inputs on a bound per instance",
    "host": "SURVEY 8d's host-inclusive metric through ndp_step: pageable numpy x0 + xr + ur + neighbour columns + ego xy in (3.4 KB per "
            "solve across PCIe), u0 out; two = ndp_step_begin / _end with two ticks in flight, one = ndp_step",
    "tick": "the node's control tick on the device (ndp_tick: nmpc_node.py:211-231,251-253): odometry x0[B,10] (+ t) in, actuator command "
            "out, reference list resident in HBM, ONE launch per tick; value_host_inclusive_x0_only = solves/s with two ticks in flight, "
            "pcie_GBps_implied = (80 + 8 + 36) B per solve; est = hover-throttle estimator on; t_scalar = NDP_TICK_T_UNIFORM; "
            "b1_us = one vehicle, one tick at a time, back to back [p50, p99]; parity = max rel u0 error of a tick vs the CPU oracle "
            "fed the windows the reference list held",
    "cpu_baseline": "oracle/ndp_oracle.c (fp64 RTI, fp32 MLP) on the host cores, OpenMP over instances, same QP mode as the timed GPU path; "
                    "ipm_always_value = iterating on every instance like HPIPM",
    "config1": "BASELINE config 1, one vehicle, N 20, no downwash, numpy in / numpy out per tick, microseconds: update = "
               "NMPCBodyRateController.update, step_ex = ndp_step_ex alone, tick = ndp_tick, cpu_* = the CPU restatement on one thread; "
               "b2b = back to back (warm loop), hz50 = one call every 20 ms on a timer grid with the GPU idle in between (the reference's "
               "cadence, nmpc_node.py:94): [p50, p99, max]; deadline 20000 us (nmpc_node.py:216-220)",
    "rows": "the widened rows (SURVEY 8f) on device buffers: f1_window = ref_window_kernel, f1_list = list advance + window copy, f3 = "
            "throttle estimator + actuator command, rollout = closed loop (window + control step + plant) -- value = vehicles (x ticks) / s, "
            "frac = algorithmic bytes / kernel time / 8 TB/s",
}


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


class c_stdout_to_stderr:
    """RCCL prints a version banner to the C library's stdout when a communicator is created (RCCL 2.26: "RCCL version : ...",
    HIP / ROCm version, host name, library path).  This program's stdout carries ONE JSON line: while a communicator is being
    created, file descriptor 1 points at stderr, and the C buffers are flushed before it is put back."""

    def __enter__(self):
        import ctypes
        sys.stdout.flush()
        self._libc = ctypes.CDLL(None)
        self._libc.fflush(None)
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        self._libc.fflush(None)
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def committed_profile(default_cfg, root=None):
    """What the committed rocprofv3 summaries say about the DEFAULT configuration's rti_kernel (batch 1024, N = 20, 1 RTI
    iteration, fused downwash, automatic QP mode, nominal starts): average duration (kernel trace) and HBM bytes per launch
    (separate FETCH_SIZE / WRITE_SIZE passes; KB units; FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md).
    The files are read BY NAME -- profiles/<tag>_kernel_stats_fused_b1024.csv and <tag>_pmc_rti_kernel.json of the newest
    tag in PROFILE_TAGS that has the CSV -- never by a glob: other committed traces (interior point always, mixed workload)
    hold the same kernel instantiation at other durations.  All None when the running configuration is not the profiled one."""
    out = {"kernel_us": None, "traffic": None, "wave_cycles_per_simd": None, "tag": None}
    if not default_cfg:
        return out
    pdir = os.path.join(root or ROOT, "profiles")
    for tag in PROFILE_TAGS:
        path = os.path.join(pdir, tag + "_kernel_stats_fused_b1024.csv")
        if not os.path.exists(path):
            continue
        out["tag"] = tag
        best = None
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if "rti_kernel" in row.get("Name", "") and (best is None or float(row["TotalDurationNs"]) > float(best["TotalDurationNs"])):
                    best = row
        if best is not None:
            out["kernel_us"] = float(best["AverageNs"]) / 1e3
        path = os.path.join(pdir, tag + "_pmc_rti_kernel.json")
        if os.path.exists(path):
            with open(path) as fh:
                pmc = json.load(fh)
            if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
                out["traffic"] = (2.0 * pmc["FETCH_SIZE"]["mean"] + pmc["WRITE_SIZE"]["mean"]) * 1024.0
            if "SQ_WAVE_CYCLES" in pmc and "SQ_WAVES" in pmc:
                out["wave_cycles_per_simd"] = 4.0 * pmc["SQ_WAVE_CYCLES"]["mean"] / pmc["SQ_WAVES"]["mean"]   # quad-cycles -> cycles
        break
    return out


def profile_agrees(prof_us, live_us, tol=PROFILE_TOLERANCE):
    """The committed profile may only annotate a run whose own HIP-event kernel duration it matches."""
    return prof_us is not None and live_us is not None and live_us > 0 and abs(prof_us - live_us) <= tol * live_us


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


def baseline_configs_block(ndp, ndist, synth, O, torch, dev, stream, local_rank, steps=40):
    """Every BASELINE.json config beside the metric's own (configs[2] = the headline), measured in THIS run on one GPU, each with
    its own parity spot check against the CPU oracle (64 instances, first tick; bar 1e-5):
      config2          batch = 1024 independent quadrotors, N = 20, NO downwash (configs[1])
      config4_one_gpu  three-vehicle formations (configs[3]) on ONE GPU: all 4096 formations (12 288 instances) and one rank's share
                       of the 8-GPU run, 512 formations (1 536 instances) -- formation-major (no exchange) and vehicle-major with
                       the library-issued RCCL all-gather per step (one-rank communicator); local order leaders-first
      config5          N = 40, 2 RTI iterations, batch = 4096 (configs[4]): the QP sweeps on the f64 matrix instruction (product
                       path: work list = the default at this size, and in place), on v_mfma_f32_16x16x4_f32 and on
                       v_mfma_f32_16x16x16_bf16; nominal and perturbed starts; u0 error against the fp64 oracle
    Device-resident steps; `steps` timed steps per leg, captured into ONE hipGraph and replayed (host-launched where an exchange
    stream or the precision studies are involved -- `launch` says which)."""
    blob = np.fromfile(os.path.join(ROOT, "ndp_nmpc_qd_amd", "weights", "downwash_sn4.bin"), dtype="<f4")
    out = {}

    def to_dev(b, keys):
        return {k: torch.from_numpy(np.ascontiguousarray(b[k])).to(dev) for k in keys if k in b}

    def run_leg(eng, enqueue, reset, n_inst, graph=True):
        """enqueue(i): one control tick on `stream`.  Returns (ms per step, launch mode)."""
        reset()
        for i in range(8):
            enqueue(i)
        torch.cuda.synchronize()
        mode = "host launch per step"
        g = None
        if graph:
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=stream, capture_error_mode="relaxed"):
                    for i in range(steps):
                        enqueue(8 + i)
                torch.cuda.set_stream(stream)
                g.replay()
                torch.cuda.synchronize()
                mode = f"hipGraph of {steps} steps"
            except Exception as e:
                g, mode = None, f"host launch per step (capture failed: {type(e).__name__})"
                torch.cuda.set_stream(stream)
                torch.cuda.synchronize()
        tw = time.perf_counter()
        while time.perf_counter() - tw < 0.02:       # clocks up
            if g is not None:
                g.replay()
            else:
                for i in range(steps):
                    enqueue(i)
            torch.cuda.synchronize()
        passes = []                                  # (the median of five passes: one 0.7 ms pass is at the mercy of the host's launch latency)
        for _ in range(5):
            ta = time.perf_counter()
            if g is not None:
                g.replay()
            else:
                for i in range(steps):
                    enqueue(i)
            torch.cuda.synchronize()
            passes.append(time.perf_counter() - ta)
        return float(np.median(passes)) / steps * 1e3, mode

    def sample(B, ns=64):
        """Instances the parity checks look at: spread over the whole local order (config 4 puts a rank's leaders in front of its
        followers -- the first 64 would all be leaders and the all-follower workgroups' short cut would never be compared)."""
        return np.unique(np.linspace(0, B - 1, min(ns, B)).astype(np.int64))

    def parity(u_dev, host, N, n_rti, use_fd, f, sel):
        cfgo = O.default_cfg(N=N, n_rti=n_rti, use_fd=use_fd)
        Xo, Uo = host["xr"][sel].copy(), host["ur"][sel].copy()
        u_or, st, _ = O.step_batch(cfgo, host["x0"][sel].copy(), host["xr"][sel].copy(), host["ur"][sel].copy(), f, Xo, Uo)
        ok = st == 0
        return float(np.max(np.abs(u_dev[sel][ok] - u_or[ok]) / np.maximum(1.0, np.abs(u_or[ok])))), u_or, ok

    # ---------------------------------------------------------------- config 2
    B, N = 1024, 20
    host = [synth.make_batch(B, N=N, seed=synth.SEED0 + 2, t0=0.02 * t) for t in range(4)]
    tk = [to_dev(h, ("x0", "xr", "ur")) for h in host]
    eng = ndp.BatchedNMPC(B, N=N, device=local_rank)
    u0 = torch.empty(B, 4, dtype=torch.float64, device=dev)
    rs = lambda: eng.reset_device(tk[0]["xr"], tk[0]["ur"], stream=stream)                                            # noqa: E731
    en = lambda i: eng.update_device(tk[i % 4]["x0"], tk[i % 4]["xr"], tk[i % 4]["ur"], u0, stream=stream)            # noqa: E731
    ms, mode = run_leg(eng, en, rs, B)
    rs(); en(0); torch.cuda.synchronize()
    par, _, _ = parity(u0.cpu().numpy(), host[0], N, 1, False, None, sample(B))
    st, _ = eng.status()
    out["config2"] = {"value": B / ms * 1e3, "us": ms * 1e3, "graph": mode.startswith("hipGraph"), "parity": par, "bad": int((st != 0).sum())}
    del eng
    # ---------------------------------------------------------------- config 4 on one GPU
    c4 = {}
    for F in (4096, 512):
        for placement in ("formation", "vehicle"):
            hs = [ndist.make_config4_shard(0, 1, F, placement, N=N, t0=0.02 * t) for t in range(4)]
            Bl = hs[0]["x0"].shape[0]
            tk = [to_dev(h, ("x0", "xr", "ur", "ego_xy", "other_index")) for h in hs]
            eng = ndp.BatchedNMPC(Bl, N=N, disturbance=True, device=local_rank)
            u0 = torch.empty(Bl, 4, dtype=torch.float64, device=dev)
            rs = lambda: eng.reset_device(tk[0]["xr"], tk[0]["ur"], stream=stream)                                    # noqa: E731
            xchg = None
            if placement == "formation":
                en = lambda i: eng.update_device(tk[i % 4]["x0"], tk[i % 4]["xr"], tk[i % 4]["ur"], u0, other=tk[i % 4]["xr"],          # noqa: E731
                                                 ego_xy=tk[i % 4]["ego_xy"], stream=stream, other_index=tk[i % 4]["other_index"])
                ms, mode = run_leg(eng, en, rs, Bl)
            else:
                try:
                    with c_stdout_to_stderr():
                        xchg = ndist.RcclExchange(Bl, N, local_rank)
                    gathered = [torch.empty(Bl, N + 1, ndist.PV_COLS, dtype=torch.float64, device=dev) for _ in range(2)]

                    primed = {"tick": None}

                    def en(i, _g=gathered, _x=xchg, _p=primed):
                        # as the N > 1 steps do it: tick i + 1's windows (functions of time only) are gathered on the library's stream
                        # while tick i is solved; ordered behind the last reader of the buffer through the compute stream
                        d = tk[i % 4]
                        if _p["tick"] != i:
                            _x.begin(d["xr"], _g[i % 2], stream)
                        _x.end(stream)
                        _x.begin(tk[(i + 1) % 4]["xr"], _g[(i + 1) % 2], stream)
                        _p["tick"] = i + 1
                        eng.update_device(d["x0"], d["xr"], d["ur"], u0, other=_g[i % 2], ego_xy=d["ego_xy"], stream=stream,
                                          other_index=d["other_index"])
                    ms, mode = run_leg(eng, en, rs, Bl, graph=False)
                except Exception as e:
                    c4[f"{3 * F}_{placement}"] = {"error": f"{type(e).__name__}: {e}"[:120]}
                    continue
            rs(); en(0); torch.cuda.synchronize()
            if xchg is not None:
                xchg.end(stream)                     # (the gather begun for a tick that is not solved)
                torch.cuda.synchronize()
            h0 = hs[0]
            allv = ndist.make_config4_all(F, N=N, t0=0.0)
            nb = np.where(h0["gids"] % 3 == 0, h0["gids"] + 1, h0["gids"])
            sel = sample(Bl)                              # leaders AND followers (the local order is leaders first)
            ego = h0["ego_xy"][sel].copy()
            ego[h0["other_index"][sel] < 0] = 1e9
            f = O.downwash_batch(blob, allv["xr"][nb][sel].copy(), h0["xr"][sel].copy(), ego)
            par, _, _ = parity(u0.cpu().numpy(), h0, N, 1, True, f, sel)
            st, _ = eng.status()
            c4[f"{3 * F}_{placement}"] = {"value": Bl / ms * 1e3, "us": ms * 1e3, "graph": mode.startswith("hipGraph"), "parity": par,
                                          "followers_checked": int((h0["other_index"][sel] < 0).sum()), "bad": int((st != 0).sum())}
            if xchg is not None:
                xchg.close()
            del eng
    out["config4_one_gpu"] = c4
    # ---------------------------------------------------------------- config 5
    B, N, NS = 4096, 40, 64
    c5 = {}
    for label, kw in (("nominal", {}), ("perturbed", dict(pos_sigma=0.5, vel_sigma=1.0, quat_sigma=0.15))):
        b = synth.make_batch(B, N=N, seed=synth.SEED0 + 5, **kw)
        t = to_dev(b, ("x0", "xr", "ur"))
        res = {}
        for name, prec, wq in (("fp64_work_list", 0, 1), ("fp64_in_place", 0, 2), ("fp32_mfma", 3, 2), ("bf16_mfma", 4, 2),
                               ("condensed_fp32", 5, 2), ("condensed_bf16", 6, 2)):
            eng = ndp.BatchedNMPC(B, N=N, n_rti=2, qp_precision=prec, work_queue=wq, device=local_rank)
            u0 = torch.empty(B, 4, dtype=torch.float64, device=dev)
            rs = lambda: eng.reset_device(t["xr"], t["ur"], stream=stream)                                            # noqa: E731
            en = lambda i: eng.update_device(t["x0"], t["xr"], t["ur"], u0, stream=stream)                            # noqa: E731
            if prec >= 5:                         # the condensed study: 12-20 ms per step -- three timed steps, not forty
                rs(); en(0); torch.cuda.synchronize()
                tq = time.perf_counter()
                for i in range(3):
                    en(i)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - tq) / 3 * 1e3
            else:
                ms, mode = run_leg(eng, en, rs, B, graph=False)
            rs(); en(0); torch.cuda.synchronize()
            st, it = eng.status()
            par, u_or, ok = parity(u0.cpu().numpy(), b, N, 2, False, None, sample(B, NS))
            res[name] = {"value": B / ms * 1e3, "us": ms * 1e3, "err": par, "bad": int((st != 0).sum()), "ipm": float((it > 0).mean())}
            if prec >= 5:
                sel = sample(B, NS)
                e_ = np.max(np.abs(u0.cpu().numpy()[sel][ok] - u_or[ok]) / np.maximum(1.0, np.abs(u_or[ok])), axis=1)
                res[name].update({"err_median": float(np.median(e_)), "qps_kept_condensed": float(eng.condensed_kept().mean()), "of": 2})
            del eng
        c5[label] = res
    out["config5"] = c5
    return out


def tick_block(ndp, synth, B, N, device):
    """SURVEY 8d's host-inclusive metric through ndp_tick: per tick the odometry rows x0[B,10] (+ the time) cross PCIe inbound and
    the actuator commands outbound; the reference list, the neighbours' windows, the estimator state live in HBM; ONE launch per tick.
    Workload: the metric's figure-eights as trajectories (synth.figure_eight_traj; vehicle i's neighbour = i ^ 1, ~36 % of the gates
    open), odometry = node 0 of each tick's window + SURVEY 8d's noise.  A parity tick is checked against the CPU oracle fed the
    windows the list held."""
    from oracle import oracle as O
    tr = synth.figure_eight_traj(B, seed=synth.SEED0 + 3, n_seg=80, t_seg=0.25, pairs=True)
    eng = ndp.BatchedNMPC(B, N=N, disturbance=True, device=device)
    eng.ref_set_trajectory(tr["coeff_x"], tr["coeff_y"], tr["coeff_z"], tr["coeff_yaw"], tr["time_cum"], tr["time_seg"], tr["final_pt"])
    eng.ref_list_reset()
    oi = np.arange(B, dtype=np.int32) ^ 1
    eng.tick_config(oi, gate=True)
    eng.tick_reset()
    rng = np.random.default_rng(0)
    n_w, n_1, n_2 = 24, 80, 200
    n_r = 24 + 300                                      # the device-resident legs below: warm-up + timed ticks (both legs tick the same times)
    nt = n_w + n_1 + 3 * n_2 + 8 + n_r
    xs = []
    for i in range(nt):
        x = eng.ref_window(np.full(B, 0.02 * i))[0][:, 0, :].copy()
        x[:, 0:3] += rng.normal(0, 0.1, (B, 3))
        x[:, 3:6] += rng.normal(0, 0.2, (B, 3))
        xs.append(x)
    it = iter(range(nt))
    # ---- parity: one tick against the oracle (gate + network on the neighbour's window, then the control step)
    for _ in range(3):
        i = next(it)
        eng.tick(xs[i], t=0.02 * i)
    X, U = eng.get_iterate()
    i = next(it)
    cmd, u0, st, itn = eng.tick(xs[i], t=np.full(B, 0.02 * i), full=True, raise_on_status=False)
    xr, ur = eng.ref_list_window(None)
    sel = np.unique(np.linspace(0, B - 1, 64).astype(np.int64))
    blob = np.fromfile(os.path.join(ROOT, "ndp_nmpc_qd_amd", "weights", "downwash_sn4.bin"), dtype="<f4")
    f = O.downwash_batch(blob, xr[oi[sel]].copy(), xr[sel].copy(), xs[i][sel, 0:2].copy())
    Xo, Uo = X[sel].copy(), U[sel].copy()
    u_or, st_o, _ = O.step_batch(O.default_cfg(N=N, use_fd=True), xs[i][sel].copy(), xr[sel].copy(), ur[sel].copy(), f, Xo, Uo)
    ok = st_o == 0
    par = float(np.max(np.abs(u0[sel][ok] - u_or[ok]) / np.maximum(1.0, np.abs(u_or[ok]))))
    thr_ok = bool(np.allclose(cmd[:, 3], u0[:, 3] * eng.cfg.mass / 50.0, rtol=1e-15, atol=0))     # estimator never ran: k_throttle_init
    for _ in range(n_w - 4):
        i = next(it)
        eng.tick(xs[i], t=0.02 * i)
    t0 = time.perf_counter()
    for _ in range(n_1):
        i = next(it)
        eng.tick(xs[i], t=0.02 * i)
    one = (time.perf_counter() - t0) / n_1
    cmdb = np.empty((B, 4))
    res = {}
    for name, kw in (("value_host_inclusive_x0_only", dict(estimate=False, scalar=True)), ("t_per_vehicle", dict(estimate=False, scalar=False)),
                     ("est", dict(estimate=True, scalar=True))):
        i = next(it)
        eng.tick_begin(xs[i], t=0.02 * i, estimate=kw["estimate"])
        t0 = time.perf_counter()
        for _ in range(n_2):
            i = next(it)
            eng.tick_begin(xs[i], t=(0.02 * i) if kw["scalar"] else np.full(B, 0.02 * i), estimate=kw["estimate"])
            eng.tick_end(out=cmdb)
        res[name] = B * n_2 / (time.perf_counter() - t0)
        eng.tick_end(out=cmdb)
    st, itn = eng.status()
    v = res["value_host_inclusive_x0_only"]
    out = {"value_host_inclusive_x0_only": v, "us": B / v * 1e6, "pcie_GBps_implied": (80 + 8 + 36) * B * v / B / 1e9, "unit": "solves/s",
           "t_per_vehicle": res["t_per_vehicle"], "est": res["est"], "one_at_a_time": B / one, "launches_per_tick": 1,
           "parity": par, "cmd_ok": thr_ok, "bad": int((st != 0).sum()), "ipm": float((itn > 0).mean()),
           "gates_open": float(np.any(eng.device_force().cpu().numpy() != 0, axis=(1, 2)).mean())}
    # ---- the tick when neighbours live on other ranks (ndp_tick_config_remote): advance -> window columns -> [exchange] -> step, three
    # launches per tick.  One rank, its own windows standing for the gathered buffer (no collective: the wire is not what is measured),
    # odometry resident in HBM; beside it the one-launch tick on the same device-resident inputs -- the like-for-like pair an N > 1 run's
    # tick form would be held against.
    import ctypes as C
    import torch
    dev = torch.device("cuda", device)
    i0 = next(it)
    xd = torch.from_numpy(np.stack(xs[i0:i0 + n_r])).to(dev)            # every tick its own odometry, resident in HBM
    xp = [C.c_void_p(xd[k].data_ptr()) for k in range(n_r)]
    cmd_t = torch.empty(B, 4, dtype=torch.float64, device=dev)
    cp = C.c_void_p(cmd_t.data_ptr())
    tv = C.c_double(0.0)
    tp = C.cast(C.pointer(tv), C.c_void_p)
    UNI = ndp._lib.TICK_T_UNIFORM

    def rate(fn, n=300):
        # (the C entry points called with arguments built once: the rate of the boundary, not of the Python wrapper's checks)
        for k in range(n_r - n):
            tv.value = 0.02 * (i0 + k)
            assert fn(k) == 0
        torch.cuda.synchronize()
        ta = time.perf_counter()
        for k in range(n_r - n, n_r):
            tv.value = 0.02 * (i0 + k)
            fn(k)
        torch.cuda.synchronize()
        return B * n / (time.perf_counter() - ta)
    L, h1 = eng._lib, eng._h
    v_one = rate(lambda k: L.ndp_tick_device(h1, xp[k], tp, None, None, UNI, cp, None, None))
    st1, _ = eng.status()
    del eng
    e2 = ndp.BatchedNMPC(B, N=N, disturbance=True, device=device)
    e2.ref_set_trajectory(tr["coeff_x"], tr["coeff_y"], tr["coeff_z"], tr["coeff_yaw"], tr["time_cum"], tr["time_seg"], tr["final_pt"])
    e2.ref_list_reset()
    own = torch.zeros(B, N + 1, 6, dtype=torch.float64, device=dev)
    e2.tick_config_remote(own, oi, gate=True)
    e2.tick_reset()
    h2, op = e2._h, C.c_void_p(own.data_ptr())

    def three(k):
        return (L.ndp_tick_advance_device(h2, xp[k], tp, None, None, UNI, None) or L.ndp_tick_window_pv_device(h2, op, None)
                or L.ndp_tick_step_device(h2, xp[k], cp, None, None))
    v_three = rate(three)
    st2, _ = e2.status()
    out["remote"] = {"value_three_stage_one_rank": v_three, "value_one_launch_device_resident": v_one, "launches_per_tick": 3,
                     "bad": int((st2 != 0).sum()) + int((st1 != 0).sum())}
    del e2
    return out


def tick_remote_ranks(ndp, synth, dist, torch, B, N, rank, world, local_rank, dev, cdev, stream, same_dev, n_ticks=120, n_warm=24, xchg=None,
                      ahead=False):
    """ndp_tick over `world` ranks, every vehicle's neighbour on the NEXT rank (ndp_tick_config_remote): per control period
        ndp_tick_advance_device -> ndp_tick_window_pv_device -> all-gather of the [B, N+1, 6] windows -> ndp_tick_step_device
    on `stream`; with xchg (dist.RcclExchange: the library's own communicator) the two middle stages are ONE call, ndp_xchg_tick_windows
    (pack out of the list + ncclAllGather on `stream`), else torch.distributed's all_gather_into_tensor.  ahead (needs xchg): the
    exchange one control period ahead -- ndp_xchg_tick_step (tick i) then ndp_xchg_tick_begin (advance, columns, all-gather of tick
    i + 1 on the exchange's stream beside tick i's control step; two gather buffers).  Every rank flies the same B figure-eights (vehicle i's neighbour = vehicle i ^ 1 of rank + 1: the gates of the
    metric's workload), odometry = node 0 of the tick's window + SURVEY 8d's noise (per-rank noise).  All ranks call this; set-up
    failures are agreed on by a collective before the first tick, the ticks themselves only launch.  Returns (on every rank) the
    whole-job rate over the slowest rank's time, a parity tick of rank 0 against the CPU oracle fed the rows the exchange
    delivered, and whether those rows are the neighbour rank's windows."""
    import ctypes as C
    res, ok, e = {}, 1, None
    try:
        tr = synth.figure_eight_traj(B, seed=synth.SEED0 + 3, n_seg=80, t_seg=0.25, pairs=True)
        e = ndp.BatchedNMPC(B, N=N, disturbance=True, device=local_rank)
        e.ref_set_trajectory(tr["coeff_x"], tr["coeff_y"], tr["coeff_z"], tr["coeff_yaw"], tr["time_cum"], tr["time_seg"], tr["final_pt"])
        e.ref_list_reset()
        gathered = torch.zeros(world * B, N + 1, 6, dtype=torch.float64, device=dev)
        gathered_b = torch.zeros(world * B, N + 1, 6, dtype=torch.float64, device=dev)      # (ahead: the second and third gather buffers)
        gathered_c = torch.zeros(world * B, N + 1, 6, dtype=torch.float64, device=dev)
        pv = torch.zeros(B, N + 1, 6, dtype=torch.float64, device=dev)
        nbr = (rank + 1) % world
        oi = (nbr * B + (np.arange(B, dtype=np.int64) ^ 1)).astype(np.int32)
        e.tick_config_remote(gathered, oi, gate=True)
        e.tick_reset()
        rng = np.random.default_rng(100 + rank)
        nt = n_warm + n_ticks + 1
        xs = np.empty((nt, B, 10))
        for i in range(nt):
            xs[i] = e.ref_window(np.full(B, 0.02 * i))[0][:, 0, :]
        xs[:, :, 0:3] += rng.normal(0, 0.1, (nt, B, 3))
        xs[:, :, 3:6] += rng.normal(0, 0.2, (nt, B, 3))
        xd = torch.from_numpy(xs).to(dev)
        cmd = torch.empty(B, 4, dtype=torch.float64, device=dev)
        u0d = torch.empty(B, 4, dtype=torch.float64, device=dev)
    except Exception as ex:
        res["error"], ok = f"{type(ex).__name__}: {ex}"[:200], 0
    okt = torch.tensor([ok], dtype=torch.int64, device=cdev)
    dist.all_reduce(okt, op=dist.ReduceOp.MIN)
    if int(okt.item()) == 0:
        res.setdefault("error", "set-up failed on another rank")
        return res
    L, h = e._lib, e._h
    sp = C.c_void_p(stream.cuda_stream)
    xp = [C.c_void_p(xd[i].data_ptr()) for i in range(nt)]
    cp, up, pp = C.c_void_p(cmd.data_ptr()), C.c_void_p(u0d.data_ptr()), C.c_void_p(pv.data_ptr())
    tv = C.c_double(0.0)
    tp = C.cast(C.pointer(tv), C.c_void_p)
    UNI = ndp._lib.TICK_T_UNIFORM
    rcs = 0

    gp, xh = C.c_void_p(gathered.data_ptr()), (xchg._h if xchg is not None else None)

    def gather():
        nonlocal rcs
        if xh is not None:
            rcs |= L.ndp_xchg_tick_windows(xh, h, gp, sp)
        elif same_dev:                                # (mechanics check, gloo: through the host)
            parts = [torch.empty(B, N + 1, 6, dtype=torch.float64) for _ in range(world)]
            with torch.cuda.stream(stream):
                mine = pv.cpu()
            dist.all_gather(parts, mine)
            with torch.cuda.stream(stream):
                gathered.copy_(torch.cat(parts, 0))
        else:
            with torch.cuda.stream(stream):
                dist.all_gather_into_tensor(gathered, pv)             # RCCL on torch's current stream = `stream`

    ahead = bool(ahead and xh is not None)
    gps = [gp, C.c_void_p(gathered_b.data_ptr()), C.c_void_p(gathered_c.data_ptr())]
    gts = [gathered, gathered_b, gathered_c]

    def tick(i, last=False):
        nonlocal rcs
        if ahead:               # tick i's gather was begun two ticks ago (the first two: below); tick i + 2's is begun behind tick i's step,
            #                     into the third buffer -- the one step i - 1 read: that gather never waits for a control step
            rcs |= L.ndp_xchg_tick_step(xh, h, xp[i], None, None, 0, cp, up, gps[i % 3], sp)
            if i + 2 < nt:
                tv.value = 0.02 * (i + 2)
                rcs |= L.ndp_xchg_tick_begin(xh, h, tp, UNI, gps[(i + 2) % 3])
            return
        tv.value = 0.02 * i
        rcs |= L.ndp_tick_advance_device(h, xp[i], tp, None, None, UNI, sp)
        if xh is None:
            rcs |= L.ndp_tick_window_pv_device(h, pp, sp)
        gather()
        rcs |= L.ndp_tick_step_device(h, xp[i], cp, up, sp)
    if ahead:
        rcs |= L.ndp_track_steps(h, 1)
        tv.value = 0.0
        rcs |= L.ndp_xchg_tick_begin(xh, h, tp, UNI, gps[0])
        tv.value = 0.02
        rcs |= L.ndp_xchg_tick_begin(xh, h, tp, UNI, gps[1])
    for i in range(n_warm):
        tick(i)
    torch.cuda.synchronize()
    dist.barrier()
    ta = time.perf_counter()
    for i in range(n_warm, n_warm + n_ticks):
        tick(i)
    torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - ta], dtype=torch.float64, device=cdev)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    st, _ = e.status()
    # ---- one more tick, checked: the rows the exchange delivered are the neighbour rank's windows (every rank flies the same
    # curves: they equal this rank's own), and rank 0's u0 against the oracle fed those rows
    X, U = e.get_iterate()
    i = n_warm + n_ticks
    tick(i, last=True)
    torch.cuda.synchronize()
    g = gts[i % 3 if ahead else 0].cpu().numpy()
    rcs |= L.ndp_tick_window_pv_device(h, pp, sp)                   # (this rank's own columns, for the comparison below)
    torch.cuda.synchronize()
    rows_ok = bool(np.array_equal(g[nbr * B:(nbr + 1) * B], pv.cpu().numpy())) and bool(np.any(g[nbr * B:(nbr + 1) * B] != 0))
    par = None
    if rank == 0:
        from oracle import oracle as O
        xr, ur = e.ref_list_window(None)
        sel = np.unique(np.linspace(0, B - 1, 64).astype(np.int64))
        other = np.zeros((len(sel), N + 1, 10))
        other[:, :, 0:6] = g[oi[sel]]
        other[:, :, 6] = 1.0
        blob = np.fromfile(os.path.join(ROOT, "ndp_nmpc_qd_amd", "weights", "downwash_sn4.bin"), dtype="<f4")
        f = O.downwash_batch(blob, other, xr[sel].copy(), xs[i][sel, 0:2].copy())
        Xo, Uo = X[sel].copy(), U[sel].copy()
        u_or, st_o, _ = O.step_batch(O.default_cfg(N=N, use_fd=True), xs[i][sel].copy(), xr[sel].copy(), ur[sel].copy(), f, Xo, Uo)
        good = st_o == 0
        u0 = u0d.cpu().numpy()
        par = float(np.max(np.abs(u0[sel][good] - u_or[good]) / np.maximum(1.0, np.abs(u_or[good]))))
    agg = torch.tensor([int((st != 0).sum()) + (1 if rcs else 0), 0 if rows_ok else 1], dtype=torch.int64, device=cdev)
    dist.all_reduce(agg)
    elapsed = float(el.item())
    res.update({"value": world * B * n_ticks / elapsed, "us": elapsed / n_ticks * 1e6, "unit": "solves/s", "ticks": n_ticks,
                "launches_per_tick": 3 if xh is None else 2,
                "gather": ("ndp_xchg_tick_begin / _step: advance, columns and ncclAllGather of tick i + 2 on the exchange's stream beside the control steps of ticks i, i + 1; three gather buffers" if ahead else
                           "ndp_xchg_tick_windows (pack + ncclAllGather on the tick's stream)" if xh is not None else
                           "host-staged (gloo, one device)" if same_dev else "torch.distributed all_gather_into_tensor (RCCL)"),
                "parity": par, "bad": int(agg[0].item()), "rows_ok": int(agg[1].item()) == 0})
    del e
    return res


def launch_ranks(n, script, script_args, python=None, out=None, err=None):
    """`bench.py --gpus N` started WITHOUT a torch.distributed environment launches its own N ranks: a CHILD process
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port> script args`
    (never a re-exec; this parent has not imported torch and has not touched the GPU).  Rank 0's JSON line -- the one stdout line that
    parses as a JSON object holding "metric" -- is relayed to `out`; every other line of the children goes to `err`.  Returns the
    child's exit code (3 when the ranks exited with 0 but printed no JSON line)."""
    import socket
    import subprocess
    out, err = out or sys.stdout, err or sys.stderr
    # Under rocprofv3 the preloaded profiler library initialises the GPU in THIS process before main(): the launch chain below
    # (Popen -> torch.distributed.run -> ranks) would then be exec hops out of a GPU-initialised process, which takes this pool's
    # machines down.  Profiled runs are single-rank and --only-timed (scripts/profile_round.sh).
    pre = " ".join(os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY", "HSA_TOOLS_LIB"))
    if "rocprof" in pre.lower():
        print("bench.py: refusing to launch ranks under a profiler preload (%s): profile a single rank with --only-timed" % pre.strip(),
              file=err, flush=True)
        return 6
    # the port stays bound (SO_REUSEADDR) until the child has been started: nobody else is handed it in between
    sk = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    sk.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script] + list(script_args)
    print("bench.py: no WORLD_SIZE in the environment, launching %d ranks: %s" % (n, " ".join(cmd)), file=err, flush=True)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    sk.close()
    relayed = 0
    for line in child.stdout:
        txt = line.strip()
        is_line = False
        if txt.startswith("{") and '"metric"' in txt:
            try:
                is_line = isinstance(json.loads(txt), dict)
            except ValueError:
                is_line = False
        if is_line and relayed == 0:
            out.write(txt + "\n")
            out.flush()
            relayed += 1
        else:
            err.write(line)
            err.flush()
    rc = child.wait()
    if rc == 0 and relayed == 0:
        print("bench.py: the ranks exited with 0 but printed no JSON line", file=err, flush=True)
        rc = 3
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=1024, help="instances per GPU")
    ap.add_argument("--horizon", type=int, default=20)
    ap.add_argument("--workload", default="ndp_downwash", choices=["ndp_downwash", "nmpc"])
    ap.add_argument("--config", type=int, default=3, choices=[3, 4],
                    help="BASELINE.json configs[] (1-based): 3 = batch of independent quadrotors (the metric's configuration), "
                         "4 = 4096 three-vehicle formations split over the GPUs")
    ap.add_argument("--instance-order", default="leaders_first", choices=["leaders_first", "interleaved"],
                    help="--config 4: a rank's local instance order (dist.config4_gids): its leaders (gate + downwash network) in front of its "
                         "followers, or ascending global id")
    ap.add_argument("--formations", type=int, default=4096, help="--config 4: number of three-vehicle formations (whole job)")
    ap.add_argument("--qp-mode", type=int, default=0, help="0 auto (exact early exit), 1 interior point always")
    ap.add_argument("--work-queue", type=int, default=0, help="0 automatic, 1 on, 2 off (ndp_cfg.work_queue)")
    ap.add_argument("--as-iter-max", type=int, default=None,
                    help="ndp_cfg.as_iter_max of the timed engine (default: the library's, 8); 0 = QP_AUTO as rounds 1-5 had it: early exit or interior point")
    ap.add_argument("--perturb", default="nominal", choices=["nominal", "mixed"],
                    help="mixed: 0.5 m / 1 m/s / 0.15 initial errors, ~20 %% of the instances need the interior-point loop")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-repeats", action="store_true", help="skip the extra timed passes over the same steps behind the timed region (`repeat`)")
    ap.add_argument("--no-tick-remote", action="store_true", help="N > 1: skip the `tick_remote` leg (ndp_tick with neighbours on the next rank)")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` block (BASELINE configs 2, 4-on-one-GPU, 5 measured in this run)")
    ap.add_argument("--only-timed", action="store_true",
                    help="no parity check, no extra legs: just warm-up + the timed steps (profiling runs: nothing but the kernel)")
    ap.add_argument("--placement", default="vehicle", choices=["vehicle", "formation"],
                    help="N > 1 (and --config 4): vehicle-major (a formation's vehicles on different GPUs: one all-gather per step, "
                         "the default) or formation-major (all vehicles of a formation on one GPU: no exchange)")
    ap.add_argument("--no-graph", action="store_true", help="launch every step from the host instead of replaying a hipGraph")
    ap.add_argument("--torch-collective", action="store_true",
                    help="rccl form: call torch.distributed.all_gather_into_tensor per step instead of the library's own ncclAllGather (ndp_xchg_*)")
    ap.add_argument("--graph-exchange", nargs="?", const="on", default="auto", choices=["auto", "on", "off"],
                    help="rccl form: off = every step launched from the host (gather on the library's stream + kernel); on = gather + kernel "
                         "of every step captured into the hipGraph; auto (default) = BOTH are timed (host launches first; the captured form "
                         "under a watchdog, never fatal) and `value` is the faster one that passed its checks -- config.launch says which")
    ap.add_argument("--leg-timeout-s", type=float, default=240.0,
                    help="watchdog of every secondary form (captured rccl form, peer form, downwash-ahead form): a leg that has not finished "
                         "after this long is reported as timed out, the line is printed with the forms that did finish, and the ranks exit")
    ap.add_argument("--deadline-s", type=float, default=1500.0,
                    help="whole-run bound: a rank still running after this long (a collective that never returns in set-up or in the "
                         "headline form, which has no watchdog of its own) says on stderr where it is and exits with code 4")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline: bounded sample, about this many seconds")
    ap.add_argument("--cadence-ticks", type=int, default=300,
                    help="config 1 at the reference's 50 Hz cadence: this many paced ticks of NMPCBodyRateController.update (half as many of "
                         "ndp_step_ex and ndp_tick); 20 ms each.  0 = skip")
    ap.add_argument("--exchange", default="both", choices=["both", "peer", "rccl", "peer_ahead"],
                    help="N > 1 (and --config 4), vehicle-major placement: how a rank gets its neighbours' reference windows, every "
                         "step.  rccl: one all-gather of the position/velocity columns per step (the north star's collective); peer: "
                         "one publish launch per step into a peer-mapped slot + epoch word, the control-step kernel reads the "
                         "neighbour's slot over xGMI; both (default): the timed steps are run once in each form, `value` is the rccl "
                         "form's, `exchange` carries both; peer_ahead (opt-in, measured slower on one GPU: five launches per tick on the second stream, "
                         "49 against 27 us): the publish and the gate / MLP launch of tick t+1 on a second stream beside the control step of tick t")
    ap.add_argument("--downwash-form", default="both", choices=["both", "prefetch", "fused"],
                    help="one GPU, config 3, downwash on: how the MLP runs.  fused: gate + MLP inside the control-step launch (the "
                         "product default); prefetch: the force of tick t+1 is predicted by a second launch on a second stream WHILE "
                         "tick t is solved (as the reference's subscriber callback runs beside its control loop); both (default): the "
                         "timed steps run once in each form, `value` is the fused form's, `downwash_forms` carries both")
    ap.add_argument("--clock-warm-ms", type=float, default=30.0,
                    help="after the --warmup steps and the graph's instantiation, replay the captured steps UNTIMED for this long so that the "
                         "GPU's clocks are up when the timed steps start (a 20-step run is 0.5 ms of work after an idle period: without it "
                         "the timed steps run at ramping clocks); the extra untimed steps are reported in config.warmup_untimed_extra_steps")
    ap.add_argument("--peer-timeout-us", type=int, default=20000, help="bound of each wait of the peer form (epoch / acknowledgement)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    # whole-run bound (--deadline-s): says where the rank is stuck, then leaves -- never a silent hang
    import threading as _th
    stage = {"at": "start-up"}

    def _deadline():
        print(f"[bench] rank {rank}: not finished after {args.deadline_s:.0f} s, stuck in: {stage['at']} -- giving up (exit 4)", file=sys.stderr, flush=True)
        os._exit(4)
    _dl = _th.Timer(args.deadline_s, _deadline)
    _dl.daemon = True
    _dl.start()

    # multi-process GPU work on this pool needs dmabuf IPC (RCCL and the peer-window mapping alike); must be set before HIP starts
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # the downwash-one-tick-ahead form runs two launches side by side: they need different hardware queues.  The library's second
    # stream is created at another priority level (its own queues); more queues than the default four as well, so that no two of
    # this process's streams share one by accident (measured: two streams on one queue run their kernels one after the other)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)")
    # NDP_BENCH_SAME_DEVICE=1 (mechanics check on a one-GPU box): every rank on device 0, control-plane collectives over gloo
    # -- RCCL refuses two ranks on one device, so only --exchange peer and --placement formation run in that mode
    same_dev = os.environ.get("NDP_BENCH_SAME_DEVICE", "0") == "1"
    if same_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if same_dev else dev       # where the few control-plane tensors live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if same_dev:
            dist.init_process_group("gloo")
        else:
            stage["at"] = "init_process_group(nccl) + first barrier"
            with c_stdout_to_stderr():
                dist.init_process_group("nccl", device_id=dev)
                dist.barrier()                # (the communicator is created by the first collective at the latest)

    import ndp_nmpc_qd_amd as ndp
    from ndp_nmpc_qd_amd import dist as ndist
    from ndp_nmpc_qd_amd import synth

    N = args.horizon
    cfg4 = args.config == 4
    downwash = args.workload == "ndp_downwash" or cfg4
    T = 8  # distinct control ticks cycled through (reference window slides by ts_nmpc = 0.02 s per tick)
    mixed_kw = dict(pos_sigma=0.5, vel_sigma=1.0, quat_sigma=0.15)

    def make_tick(B, t0, perturb):
        if cfg4:
            return ndist.make_config4_shard(rank, world, args.formations, args.placement, N=N, t0=t0, order=args.instance_order)
        if perturb == "mixed":
            return synth.make_batch(B, N=N, seed=synth.SEED0 + 40 + rank, downwash=True, t0=t0, **mixed_kw)
        return ndist.make_formation_shard(B, rank, world, N=N, t0=t0)

    B = ndist.config4_gids(rank, world, args.formations, args.placement, args.instance_order)[1] if cfg4 else args.batch
    keys = ("x0", "xr", "ur", "ego_xy") + (("other_index",) if cfg4 else ("other",))
    ticks = []
    for t in range(T):
        b = make_tick(B, 0.02 * t, args.perturb)
        ticks.append({k: torch.from_numpy(b[k]).to(dev) for k in keys})
    host0 = make_tick(B, 0.0, args.perturb)

    as_kw = {} if args.as_iter_max is None else {"as_iter_max": args.as_iter_max}
    eng = ndp.BatchedNMPC(B, N=N, disturbance=downwash, qp_mode=args.qp_mode, device=local_rank, work_queue=args.work_queue, **as_kw)
    # an explicit non-default stream: torch's default stream has handle 0, which the C-ABI reads as "use the
    # library's own stream" -- with a real handle the exchange (N > 1) and the kernel are ordered on ONE stream
    stream = torch.cuda.Stream(device=dev)
    stream_b = torch.cuda.Stream(device=dev, priority=-1)     # the downwash-ahead launches (prefetch form): its own hardware queue
    torch.cuda.set_stream(stream)
    u0 = torch.empty(B, 4, dtype=torch.float64, device=dev)
    # ---- neighbour exchange (vehicle-major placement: a leader's neighbour lives on the next rank).  The reference publishes a
    # NEW PredXU every control tick (nmpc_node.py:116-133,229-230) and the leader consumes it (ndp_nmpc_leader_node.py:40,60-76);
    # every timed step therefore contains one exchange of this tick's windows, in one of two forms, BOTH of which are run and
    # reported (`exchange`), the north star's collective first and as the headline `value`:
    #   rccl : ONE all-gather per step of the position / velocity columns of the ranks' reference windows, [B, N+1, 6] fp64 per
    #          rank -- all that the gate and the MLP read (downwash_nn.py:22).  config 3 (ring of ranks): rank r reads rank r+1's
    #          slice of the gathered buffer as it lies; config 4: the kernel picks each leader's neighbour row through
    #          other_index.  Two gather buffers: the gather of tick i+1's windows (functions of time only) is started before tick
    #          i's kernel is launched and runs on RCCL's stream beside it.
    #   peer : ONE publish per step (dist.PeerWindows / csrc/peer_epoch.hpp; a copy launch + a one-wave launch): this tick's
    #          windows into the rank's own peer-mapped slot, epoch word, wait for the neighbour's epoch; the control-step kernel then
    #          reads the neighbour's slot out of the neighbour GPU's HBM over xGMI.  No collective, no host round trip, hipGraph-
    #          replayed like N = 1.
    need_exchange = downwash and args.placement == "vehicle" and (world > 1 or cfg4)
    two_forms = (not need_exchange and downwash and world == 1 and not cfg4 and N + 1 <= 32 and args.qp_mode == 0
                 and not eng.work_queue)
    # rccl form, two launch modes: "rccl" = every step launched from the host, "rccl_graph" = gather + kernel captured (--graph-exchange)
    rccl_modes = {"auto": ["rccl", "rccl_graph"], "on": ["rccl_graph"], "off": ["rccl"]}[args.graph_exchange]
    if args.no_graph:
        rccl_modes = ["rccl"]
    baseline_modes = []
    if two_forms:
        modes = {"both": ["fused", "prefetch"], "prefetch": ["prefetch"], "fused": ["fused"]}[args.downwash_form]
        # LIKE-FOR-LIKE baseline of the 1 -> N scaling curve: the N > 1 lines run the per-tick exchange (one library-issued RCCL
        # all-gather per step); the N = 1 headline above replays a hipGraph with no exchange at all.  The same steps are therefore
        # ALSO run here in exactly the N > 1 form -- a one-rank communicator, the pack + ncclAllGather on the library's stream per
        # step, host-launched and captured -- and reported as `scaling_baseline`: efficiency = value(N) / (N * scaling_baseline).
        if not args.only_timed and not same_dev and args.perturb == "nominal":       # (the exchange forms gather the ranks' OWN reference windows: make_formation_shard's workload)
            baseline_modes = ((list(rccl_modes) if args.exchange in ("both", "rccl") else []) + (["peer"] if args.exchange in ("both", "peer") else [])
                              + (["peer_ahead"] if args.exchange == "peer_ahead" else []))
            modes = modes + baseline_modes
    elif not need_exchange:
        modes = ["none"]
    elif same_dev and world > 1:
        modes = ["peer_ahead"] if args.exchange == "peer_ahead" else ["peer"]      # RCCL refuses two ranks on one device
    else:
        modes = {"both": rccl_modes + ["peer"], "rccl": list(rccl_modes), "peer": ["peer"], "peer_ahead": ["peer_ahead"]}[args.exchange]
    if not (N + 1 <= 32 and args.qp_mode == 0 and not eng.work_queue):
        modes = [m for m in modes if m != "peer_ahead"]       # (the downwash-ahead launch holds one instance's rows in one 32-row tile)
    peer, peer_err = None, None
    if any(m in PEER_FORMS for m in modes):
        stage["at"] = "PeerWindows: peer-mapped window buffers (IPC handles)"
        try:
            peer = ndist.PeerWindows(B, N, local_rank, timeout_us=args.peer_timeout_us)
            if cfg4:
                peer_oidx = torch.from_numpy(ndist.config4_other_index(rank, world, args.formations, "vehicle", peer_rows=True, order=args.instance_order)).to(dev)
        except Exception as e:                    # same outcome on every rank (PeerWindows exchanges the result)
            peer_err = f"{type(e).__name__}: {e}"[:200]
            modes = [m for m in modes if m not in PEER_FORMS]
            if not modes:
                raise
    xchg, xchg_err = None, None
    if any(m in modes for m in ("rccl", "rccl_graph")):
        gathered = [torch.empty(world * B, N + 1, ndist.PV_COLS, dtype=torch.float64, device=dev) for _ in range(2)]
        pv_local = torch.empty(B, N + 1, ndist.PV_COLS, dtype=torch.float64, device=dev)
        # The all-gather issued by the C-ABI library itself (ndp_xchg_*: pack launch + ncclAllGather on a HIP stream of its own, two
        # event operations, ~10 us of host time per tick) instead of torch.distributed's all_gather_into_tensor (~25 us of host time
        # per call: more than a control step lasts).  A real communicator also with ONE rank.  Creation is collective: all ranks
        # use it or none does (--torch-collective forces the torch.distributed call).
        if not args.torch_collective and not same_dev:
            ok_here = 1
            stage["at"] = "RcclExchange: the library's own RCCL communicator (ncclCommInitRank)"
            try:
                with c_stdout_to_stderr():
                    xchg = ndist.RcclExchange(B, N, local_rank)
            except Exception as e:
                xchg_err, ok_here = f"{type(e).__name__}: {e}"[:200], 0
            if world > 1:
                okt = torch.tensor([ok_here], dtype=torch.int64, device=cdev)
                dist.all_reduce(okt, op=dist.ReduceOp.MIN)
                if int(okt.item()) == 0 and xchg is not None:
                    xchg.close()
                    xchg, xchg_err = None, "RcclExchange could not be created on every rank"
    rccl_form = ("rccl all-gather per step, issued by the library on its own HIP stream (ndp_xchg_*)" + ("" if world > 1 else "; one rank: a real communicator, no xGMI traffic")
                 if xchg is not None else
                 "rccl all-gather per step through torch.distributed" + ("" if world > 1 else " (one rank: the pack only, no RCCL call)"))
    mode_names = {"rccl_graph": rccl_form + "; gather + control step of every tick captured into the hipGraph",
                  "prefetch": "none (one GPU); downwash of tick t+1 on a second stream beside the control step of tick t",
                  "fused": "none (one GPU); gate + MLP fused into the control-step launch",
                  "none": "none", "rccl": rccl_form,
                  "peer": "peer windows over xGMI: one publish (copy launch + one-wave epoch launch) per step, read by the control-step kernel",
                  "peer_ahead": "peer windows over xGMI, one tick ahead: publish + gate / MLP launch of tick t+1 (which reads the neighbour's slot over "
                                "xGMI) on a second stream beside the control step of tick t, which takes the force late -- no xGMI wait on the control steps' chain"}

    def host_other(h):
        """The neighbour windows of the host copy of tick 0 (oracle legs)."""
        if not cfg4:
            return h["other"]
        allv = ndist.make_config4_all(args.formations, N=N, t0=0.0)
        nb = np.where(h["gids"] % 3 == 0, h["gids"] + 1, h["gids"])
        return allv["xr"][nb].copy()

    def run_mode(mode, check_parity):
        """Parity spot check, warm-up and EXACTLY --steps timed steps of one exchange form; returns its measurements."""
        exchange = mode in ("rccl", "rccl_graph")
        graph_x = mode == "rccl_graph"
        pending, bound = {}, {}

        # The gather of tick i + 1 overwrites the buffer tick i - 1 read: it is ordered behind THAT control step -- through the step's own
        # completion event (ndp_track_steps: no packet on the compute stream; 34 against 38 us per tick) when the steps are launched
        # from the host, through an event recorded on the compute stream inside a capture.
        track = exchange and xchg is not None and not graph_x
        eng.track_steps(track)

        def prefetch(i):
            if exchange and xchg is not None:
                ev = None
                if track:
                    try:
                        ev = eng.last_step_event()
                    except ndp.NdpError:
                        ev = None                                                                    # no control step launched yet
                if ev is not None:
                    xchg.begin(ticks[i % T]["xr"], gathered[i % 2], None, after_event=ev)            # one RCCL all-gather over xGMI
                else:
                    xchg.begin(ticks[i % T]["xr"], gathered[i % 2], stream)
                pending[i] = None
            elif exchange:
                pending[i] = ndist.exchange_pv_begin(ticks[i % T]["xr"], pv_local, gathered[i % 2])   # the same through torch.distributed

        def step(i, first=True, last=True):
            """One control tick.  first / last: position in a chain of consecutive ticks (prefetch form: the chain's first tick
            predicts its own force before it starts, every tick but the last predicts the next one's beside its control step)."""
            d = ticks[i % T]
            other, oidx = None, None
            if mode == "peer_ahead":
                # stream_b carries, per tick: publish (this rank's windows into its slot, epoch, wait for the neighbour's) and the gate / MLP
                # launch that reads the neighbour's slot; the protocol's "reader is done with the slot" is stream_b's own order
                oi = peer_oidx if cfg4 else None
                if first:
                    eng.downwash_prefetch_device(peer.publish_device(d["xr"], stream_b), d["xr"], ego_xy=d["ego_xy"], other_index=oi,
                                                 after_stream=stream, on_stream=stream_b)
                if not last:
                    n = ticks[(i + 1) % T]
                    eng.downwash_prefetch_device(peer.publish_device(n["xr"], stream_b), n["xr"], ego_xy=n["ego_xy"], other_index=oi, on_stream=stream_b)
                eng.update_device_prefetched(d["x0"], d["xr"], d["ur"], u0, stream=stream)
                if last:
                    stream.wait_stream(stream_b)
                return
            if mode == "prefetch":
                if first:
                    eng.downwash_prefetch_device(d["other"], d["xr"], ego_xy=d["ego_xy"], after_stream=stream)
                if not last:
                    n = ticks[(i + 1) % T]
                    eng.downwash_prefetch_device(n["other"], n["xr"], ego_xy=n["ego_xy"])
                eng.update_device_prefetched(d["x0"], d["xr"], d["ur"], u0, stream=stream)
                if last:
                    eng.prefetch_join(stream)
                return
            if downwash:
                if exchange:
                    if i not in pending:
                        prefetch(i)
                    w = pending.pop(i)
                    if xchg is not None and track:
                        # end(i) + begin(i + 1) in ONE library call (ndp_xchg_tick: two ctypes calls per tick instead of four)
                        xchg.tick(eng, ticks[(i + 1) % T]["xr"], gathered[(i + 1) % 2], stream)
                        pending[i + 1] = None
                    else:
                        if xchg is not None:
                            xchg.end(stream)             # (the gather begun last is tick i's: end(i) always precedes begin(i + 1))
                        else:
                            ndist.exchange_pv_end(w)
                        prefetch(i + 1)
                    # the step on (tick i % T, gathered buffer i % 2): argument checks once, then one ctypes call per launch -- this
                    # loop launches from the host at the control step's own pace
                    key = (i % T, i % 2)
                    if key not in bound:
                        if cfg4:
                            other, oidx = gathered[i % 2], d["other_index"]
                        else:
                            other = gathered[i % 2].view(world, B, N + 1, ndist.PV_COLS)[ndist.neighbour_rank(rank, world)]
                        bound[key] = eng.bind_update_device(d["x0"], d["xr"], d["ur"], u0, other=other, ego_xy=d["ego_xy"], stream=stream,
                                                            other_index=oidx)
                    bound[key]()
                    return
                elif mode == "peer":
                    other = peer.publish_device(d["xr"], stream)     # this tick's publish launch; the neighbour's slot of this tick
                    oidx = peer_oidx if cfg4 else None
                elif cfg4:
                    other, oidx = d["xr"], d["other_index"]      # formation-major: the neighbour's window is a local row of xr
                else:
                    other = d["other"]
            eng.update_device(d["x0"], d["xr"], d["ur"], u0, other=other, ego_xy=d["ego_xy"] if downwash else None,
                              stream=stream, other_index=oidx)

        def fence():
            for w in list(pending.values()):      # a gather started for a tick that is never solved (end of a phase)
                if xchg is not None and exchange:
                    xchg.end(stream)
                else:
                    ndist.exchange_pv_end(w)
            pending.clear()
            if world > 1:
                dist.barrier()
            # plain synchronise: recording an event and polling it first was measured SLOWER on this image (one host-launched
            # step 44.8 us against 37.3 us, a 20-step graph 452.7 against 449.2 us; profiles/r03_launch_floor.txt) -- the wait
            # spins already, and the extra event costs a packet
            torch.cuda.synchronize()

        # ---- parity spot check against the CPU oracle (rank 0, first tick, 64 instances).  Runs AFTER the timed region: the
        # oracle's OpenMP phase in front of it made the 20-step timed region 3-6 % slower (23.9 against 23.2 us per step on the
        # same box: the launching thread comes back from it on a cold core); a failed check still withholds `value`.
        def parity_check():
            eng.reset_device(ticks[0]["xr"], ticks[0]["ur"], stream=stream)
            step(0)
            fence()
            if rank != 0:
                return None
            from oracle import oracle as O
            # 64 instances spread over the whole local order (config 4, leaders first: leaders AND the all-follower workgroups)
            sel = np.unique(np.linspace(0, B - 1, min(64, B)).astype(np.int64))
            cfgo = O.default_cfg(N=N, use_fd=downwash)
            f = None
            if downwash:
                blob = np.fromfile(os.path.join(ROOT, "ndp_nmpc_qd_amd", "weights", "downwash_sn4.bin"), dtype="<f4")
                ego = host0["ego_xy"][sel].copy()
                if cfg4:
                    ego[host0["other_index"][sel] < 0] = 1e9       # followers: no neighbour, gate closed
                f = O.downwash_batch(blob, host_other(host0)[sel].copy(), host0["xr"][sel].copy(), ego)
            Xo, Uo = host0["xr"][sel].copy(), host0["ur"][sel].copy()
            u_or, st_or, _ = O.step_batch(cfgo, host0["x0"][sel].copy(), host0["xr"][sel].copy(), host0["ur"][sel].copy(), f, Xo, Uo)
            u_dev = u0.cpu().numpy()[sel]
            return float(np.max(np.abs(u_dev - u_or) / np.maximum(1.0, np.abs(u_or))))

        # ---- warm-up, then EXACTLY --steps timed steps between barrier + synchronize
        eng.reset_device(ticks[0]["xr"], ticks[0]["ur"], stream=stream)
        for i in range(args.warmup):
            step(i)
        fence()
        if exchange:                              # the first timed tick's windows are in place before the clock starts; every
            prefetch(args.warmup)                 # timed step then starts exactly one gather (the next tick's) and one kernel
            ndist.exchange_pv_end(pending[args.warmup])       # (torch form; the library's form is waited for on the device, in step)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
        # The K timed steps are a launch-bound chain of dependent kernels -> they are captured into a hipGraph and replayed; a
        # node of a replayed graph starts 1.6 us after its predecessor ends, a host launch 2.6 us (scripts/ubench/launch_floor.hip).
        # K <= 1024: ONE graph of all K steps, replayed once (the driver's short run -- 20 steps -- is then the steady state, not
        # 16 replayed + 4 host launches).  K > 1024: a graph of G = 1024 rounded down to whole cycles through the T input ticks,
        # replayed K // G times, and a second graph with the K % G remaining steps.  Every step runs, inside the timed region.
        # rccl form: steps are launched from the host by default (each also starts an all-gather on RCCL's stream); --graph-exchange
        # captures gather + kernel as well.  peer form: publish + control-step launches are captured like the N = 1 steps (the
        # slot parity is baked into a launch, so every graph holds an even number of ticks; an odd remainder is host-launched).
        graphs, tail, launch_mode = [], 0, "host launch per step"

        def replay(gt):
            if gt[2] is not None:
                with torch.cuda.stream(stream_b):
                    gt[2].replay()
            gt[0].replay()
        # (the library-issued gather runs on its own stream: inside ONE hipGraph such a branch executes in line with the kernels on
        # ROCm 7.2 -- measured 305 against 281 us per step of 12 288 instances -- so that form is launched from the host)
        want_graph = not args.no_graph and (not exchange or graph_x or (world == 1 and xchg is None and mode == "rccl"))
        if want_graph:
            try:
                base = ((args.warmup + T - 1) // T) * T          # a multiple of T: the replayed cycle starts at tick 0
                G = args.steps if args.steps <= 1024 else 1024 // T * T
                plan = [(G, args.steps // G)] + ([(args.steps % G, 1)] if args.steps % G else [])
                if mode in PEER_FORMS:                          # even graphs only
                    plan = [(n - (n & 1), r) for n, r in plan if n - (n & 1) > 0]
                    tail = args.steps - sum(n * r for n, r in plan)
                fence()
                first = base
                for n_cap, n_rep in plan:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=stream, capture_error_mode="relaxed"):
                        for i in range(n_cap):
                            if mode in ("prefetch", "peer_ahead"):  # the control steps only; the downwash launches: graph gb below
                                d = ticks[(first + i) % T]
                                eng.update_device_prefetched(d["x0"], d["xr"], d["ur"], u0, stream=stream)
                            else:
                                step(first + i, first=i == 0, last=i == n_cap - 1)
                        for w in list(pending.values()):         # the last step's prefetch belongs to the captured cycle
                            if xchg is not None and exchange:
                                xchg.end(stream)                 # (joins the exchange stream back into the capture)
                            else:
                                ndist.exchange_pv_end(w)
                        pending.clear()
                    torch.cuda.set_stream(stream)
                    gb = None
                    if mode in ("prefetch", "peer_ahead"):
                        # The downwash launches of the same ticks as a SECOND graph on a second stream (another priority level =
                        # its own hardware queue): the two chains order themselves through the device-side tick words, so the two
                        # graphs are replayed side by side.  (Parallel branches inside ONE hipGraph execute one after the other on
                        # ROCm 7.2: 37 us per tick against 19 us this way.)
                        gb = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(gb, stream=stream_b, capture_error_mode="relaxed"):
                            for i in range(n_cap):
                                d = ticks[(first + i) % T]
                                if mode == "peer_ahead":          # (+ the tick's publish in front: copy, epoch, wait for the neighbour's)
                                    eng.downwash_prefetch_device(peer.publish_device(d["xr"], stream_b), d["xr"], ego_xy=d["ego_xy"],
                                                                 other_index=peer_oidx if cfg4 else None, on_stream=stream_b)
                                else:
                                    eng.downwash_prefetch_device(d["other"], d["xr"], ego_xy=d["ego_xy"], on_stream=stream_b)
                        torch.cuda.set_stream(stream)
                        with torch.cuda.stream(stream_b):
                            gb.replay()
                    g.replay()                                    # instantiate / upload outside the timed region
                    torch.cuda.synchronize()
                    graphs.append((g, n_rep, gb))
                    first += n_cap * n_rep
                launch_mode = " + ".join(f"hipGraph of {n} steps x {r}" for n, r in plan) + (f" + {tail} host-launched" if tail else "")
                if mode in ("prefetch", "peer_ahead"):
                    launch_mode += (" (control steps) beside a hipGraph of the same ticks' " + ("publish + " if mode == "peer_ahead" else "")
                                    + "downwash launches on a second stream")
            except Exception as e:                                # capture unsupported: fall back, say so
                graphs, tail, launch_mode = [], 0, f"host launch per step (graph capture failed: {type(e).__name__}: {e})"[:300]
                pending.clear()
                torch.cuda.set_stream(stream)
                torch.cuda.synchronize()
                if exchange:
                    prefetch(args.warmup)
            extra = 0
            if graphs and args.clock_warm_ms > 0:                 # untimed: bring the clocks up (see --clock-warm-ms)
                # The SAME number of replays on every rank (a replay holds this rank's share of collectives / publishes: a rank
                # that replays once more than its neighbour waits for epochs that never come): one replay is timed, the slowest
                # rank's time sets the count.
                torch.cuda.synchronize()
                tw = time.perf_counter()
                replay(graphs[0])
                torch.cuda.synchronize()
                t1 = time.perf_counter() - tw
                extra += plan[0][0]
                if world > 1:
                    tt = torch.tensor([t1], dtype=torch.float64, device=cdev)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    t1 = float(tt.item())
                for _ in range(min(2000, int(args.clock_warm_ms * 1e-3 / max(t1, 1e-6)))):
                    replay(graphs[0])
                    extra += plan[0][0]
                fence()
            if mode in PEER_FORMS:
                peer.tick = peer.stats()["ticks"]                 # host mirror of the device-side tick count (capture ran no kernel)
        else:
            extra = 0
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        # HIP events on the launch stream around the timed region's launches (roofline.kernel_us).  The first one is recorded on the
        # idle stream just BEFORE the clock starts (its packet is not part of the work being timed; it adds the host's launch latency
        # of the first replay to the event time, once per region), the second behind the last launch, under the running kernels.
        ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if graphs:
            ev_a.record(stream)
        t0 = time.perf_counter()
        if graphs:
            for gt in graphs:
                for _ in range(gt[1]):
                    replay(gt)
            for i in range(tail):
                step(args.steps - tail + i, first=i == 0, last=i == tail - 1)
            ev_b.record(stream)
        else:
            eng.timing_enable(8)  # HIP events around every 8th launch (an event pair per launch costs a dispatch gap)
            for i in range(args.steps):
                step(args.warmup + i, first=i == 0, last=i == args.steps - 1)
        fence()
        elapsed = time.perf_counter() - t0
        if mode in PEER_FORMS:
            peer.tick = peer.stats()["ticks"]
        region_ms = ev_a.elapsed_time(ev_b) if graphs else None
        # The timed region above is ONE pass over the K steps (the contract); a 20-step pass is 0.45 ms, so the same graph(s) are replayed
        # REPEATS more times, each pass timed on its own (synchronise, clock, replay, synchronise): min / median / max say how much of
        # `value` is the box's mood.  One rank, graph launches only; not part of `value`.
        repeat_ms = []
        if graphs and world == 1 and not tail and not args.no_repeats:
            for _ in range(REPEATS):
                torch.cuda.synchronize()
                tr = time.perf_counter()
                for gt in graphs:
                    for _ in range(gt[1]):
                        replay(gt)
                torch.cuda.synchronize()
                repeat_ms.append((time.perf_counter() - tr) * 1e3)
        if graphs:                # start / stop events carried by EVERY dispatch packet of 64 host-launched steps, outside the timed region (kernel_us_dispatch_events)
            eng.timing_enable(1)
            for i in range(64):
                step(i, first=i == 0, last=i == 63)
            fence()
        if world > 1:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        rti_ms, rti_n = eng.timing_read("rti")
        mlp_ms, mlp_n = eng.timing_read("mlp")
        eng.timing_enable(0)
        st, it = eng.status()
        parity = None                         # filled in by the caller: the checks of ALL forms run after ALL timed regions
        bad = int((st != 0).sum())
        if world > 1:
            agg = torch.tensor([bad], dtype=torch.int64, device=cdev)
            dist.all_reduce(agg)
            bad = int(agg.item())
        res = {"extra_warm": extra, "elapsed": elapsed, "launch": launch_mode, "parity": parity, "bad": bad, "rti_ms": rti_ms, "rti_n": rti_n,
               "mlp_ms": mlp_ms, "mlp_n": mlp_n, "it": it, "step": step, "name": mode_names[mode], "region_ms": region_ms, "repeat_ms": repeat_ms,
               "parity_fn": parity_check if check_parity else None}
        if mode in ("prefetch", "peer_ahead"):
            res["prefetch_stats"] = eng.prefetch_stats()
        if mode in PEER_FORMS:
            ps = peer.stats()
            if world > 1:                                         # any rank's timed-out wait shows in the line
                agg = torch.tensor([ps["ack_timeouts"], ps["epoch_timeouts"], ps["slot_mismatches"]], dtype=torch.int64, device=cdev)
                dist.all_reduce(agg)
                ps["ack_timeouts"], ps["epoch_timeouts"], ps["slot_mismatches"] = (int(x) for x in agg.tolist())
            res["peer_stats"] = ps
        return res

    results, form_errors = {}, {}
    tick_remote = None                        # (N > 1: filled in behind the forms)

    def finish(partial=False):
        """Builds and prints the line (rank 0) from the forms that have finished.  partial: called by the watchdog of a secondary form that
        did not come back -- nothing that touches the GPU or a collective runs then."""
        nonlocal results
        results = {m: r for m, r in results.items() if m not in form_errors}
        # `value` is the first form's (the north star's collective when an exchange runs); its two launch modes -- host launches,
        # captured -- are one form: the faster one that passed its checks is the headline, config.launch says which
        def form_ok(r):
            ps_, pf_ = r.get("peer_stats"), r.get("prefetch_stats") if r.get("peer_stats") else None
            return ((r["parity"] is None or r["parity"] <= 1e-5) and r["bad"] == 0
                    and not (ps_ and (ps_["ack_timeouts"] or ps_["epoch_timeouts"] or ps_["slot_mismatches"]))
                    and not (pf_ and (pf_["force_timeouts"] or pf_["slot_timeouts"])))
        headline = modes[0]
        if headline in ("rccl", "rccl_graph"):
            cands = [m for m in ("rccl", "rccl_graph") if m in results and m in modes and form_ok(results[m])]
            if cands:
                headline = min(cands, key=lambda m: results[m]["elapsed"])
        is_rccl = headline in ("rccl", "rccl_graph")
        head = results[headline]
        elapsed, launch_mode, parity, bad, it = head["elapsed"], head["launch"], head["parity"], head["bad"], head["it"]
        rti_ms, rti_n, mlp_ms, mlp_n, step = head["rti_ms"], head["rti_n"], head["mlp_ms"], head["mlp_n"], head["step"]
        exchange_mode = head["name"]
        # `value` is the headline form's and stands or falls with ITS checks (parity against the oracle, converged instances, no
        # timed-out wait); every other form carries its own figures and its own verdict ("ok") under exchange / downwash_forms --
        # a defect there is reported there and in `secondary_form_failed`, it does not take the headline's measurement away.
        frac_ipm = float((it > 0).mean()) if args.qp_mode == 0 else 1.0
        sweeps = float(np.mean(np.where(it > 0, 1 + 2 * it, 1))) if args.qp_mode == 0 else float(np.mean(2 * it))

        def timed_leg(e, batch_ticks, n_steps, n_warm=10):
            """Device-resident steps of another engine / workload: (solves/s, ms per step, ipm fraction, sweeps per solve)."""
            uu = torch.empty(e.B, 4, dtype=torch.float64, device=dev)
            e.reset_device(batch_ticks[0]["xr"], batch_ticks[0]["ur"], stream=stream)

            def one(i):
                d = batch_ticks[i % len(batch_ticks)]
                e.update_device(d["x0"], d["xr"], d["ur"], uu, other=d.get("other"), ego_xy=d.get("ego_xy"), stream=stream)
            for i in range(n_warm):
                one(i)
            torch.cuda.synchronize()
            ta = time.perf_counter()
            for i in range(n_steps):
                one(n_warm + i)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - ta) / n_steps
            s_, i_ = e.status()
            return {"value": e.B / dt, "us": dt * 1e6, "ipm": float((i_ > 0).mean()),
                    "sweeps": float(np.mean(np.where(i_ > 0, (0 if e.cfg.qp_mode else 1) + 2 * i_, 1))),
                    "bad": int((s_ != 0).sum()), "wq": bool(e.work_queue)}

        if rank == 0:
            total = B * world * args.steps
            value = total / elapsed
            f_qp, f_mlp = algorithmic_flops_per_solve(N, sweeps, downwash)
            pair_s = rti_ms * 1e-3 / max(rti_n, 1)       # HIP start / stop events carried by the launches' own dispatch packets (hipExtLaunchKernel)
            mlp_s = mlp_ms * 1e-3 / max(mlp_n, 1) if mlp_n else 0.0
            # One launch per step on the launch stream (N = 1, no exchange): the kernel's average duration is the HIP-event time over
            # the timed region's launches / steps -- back to back in a replayed graph, so it includes the ~0.1 us between two nodes
            # and, once per replay, the graph's start-up on the device.  With more launches per step (exchange forms) the pairs stay.
            one_launch = head["region_ms"] is not None and headline in ("none", "fused", "prefetch") and mlp_n == 0
            rti_s = head["region_ms"] * 1e-3 / args.steps if one_launch else pair_s
            prefetch_form = headline == "prefetch"
            fused = downwash and mlp_n == 0 and not prefetch_form   # gate + MLP run inside rti_kernel (one launch per step)
            ach_tf = f_qp * B / rti_s / 1e12
            abytes = algorithmic_bytes_per_solve(N, downwash)
            is_default = (not cfg4 and (fused or prefetch_form) and B == 1024 and N == 20 and args.qp_mode == 0 and args.perturb == "nominal" and world == 1)
            prof = committed_profile(is_default)
            # the committed profile annotates this run only if it describes it: its kernel duration must agree with the duration
            # measured live (HIP events) -- otherwise every figure derived from it is withheld and the mismatch is reported
            profile_mismatch = None
            if prof["kernel_us"] is not None and not profile_agrees(prof["kernel_us"], rti_s * 1e6):
                profile_mismatch = {"tag": prof["tag"], "kernel_us_rocprof": prof["kernel_us"], "kernel_us_live": rti_s * 1e6,
                                    "tolerance": PROFILE_TOLERANCE}
                prof = {"kernel_us": None, "traffic": None, "wave_cycles_per_simd": None, "tag": prof["tag"]}
            # matrix-pipe occupancy estimate: a v_mfma_f64_16x16x4 / v_mfma_f32_32x32x2 holds the SIMD's pipe 64 cycles, the four-block
            # v_mfma_f64_4x4x4 16, a v_mfma_f32_32x32x16_f16 32 (scripts/ubench; PMC SQ_VALU_MFMA_BUSY_CYCLES of the committed profile
            # = 143 * 64 + 119 * 16 + 12 * 64 + 96 * 32 per instance exactly).  One instance per SIMD; per full sweep at horizon N:
            # 6 + 7 (N-1) + 4 per re-symmetrisation on the 16x16x4 form, 2 (N-1) + 1 + 4 N on the four-block form (a corrector
            # solve of the interior-point loop is counted as a full sweep here: an upper estimate for those workloads).  The shader
            # clock is MEASURED: wave-cycles per SIMD of the committed PMC pass over the committed kernel duration when the profile
            # belongs to this configuration, else the in-kernel stamp span of one launch over its HIP-event duration.
            n_16 = sweeps * (6 + 7 * (N - 1) + 4 * ((N - 1) // 10))
            n_4 = sweeps * (2 * (N - 1) + 1 + 4 * N)
            pipe_cycles = (n_16 * 64 + n_4 * 16 + ((12 * 64 + 96 * 32) if fused else 0)) * -(-B // 1024)      # instances per SIMD (1024 SIMDs), in rounds
            clock_hz, clock_src = None, None
            if prof["wave_cycles_per_simd"] and prof["kernel_us"]:
                clock_hz, clock_src = prof["wave_cycles_per_simd"] / (prof["kernel_us"] * 1e-6), f"profiles/{prof['tag']} PMC SQ_WAVE_CYCLES / kernel trace duration"
            if clock_hz is None and not args.only_timed and not partial:
                eng.debug_stamps(True)
                eng.timing_enable(1)
                step(0)
                torch.cuda.synchronize()
                sm = eng.debug_stamps(False, read=True)
                ms1, n1 = eng.timing_read("rti")
                eng.timing_enable(0)
                # stamps 12 / 13: s_memrealtime (100 MHz) at kernel entry / exit of every wave, 14 / 15: s_memtime (shader clock) there
                okw = (sm[:, 13] > sm[:, 12]) & (sm[:, 15] > sm[:, 14])
                if okw.any():
                    clock_hz = float(np.median((sm[okw, 15] - sm[okw, 14]) / (sm[okw, 13] - sm[okw, 12]))) * 100e6
                    clock_src = "in-kernel s_memtime / s_memrealtime (100 MHz) over one stamped launch, median over waves"
            out = {
                "metric": "NMPC solves/sec (N=20, 1 RTI iter + downwash MLP) at batch",
                "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if cfg4 else "weak",
                "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": (f"BASELINE config 4: {args.formations} three-vehicle formations = {3 * args.formations} instances over {world} GPU(s), "
                                        f"{B}/GPU, N={N}, 1 RTI iter, " if cfg4 else f"batch={B}/GPU independent quadrotors, N={N}, 1 RTI iter, ")
                                       + ("MLP downwash on" if downwash else "no downwash")
                                       + (", formation-major (no exchange)" if (world > 1 or cfg4) and not need_exchange else "")
                                       + (", perturbed starts" if args.perturb == "mixed" else ""),
                           "batch_per_gpu": B, "horizon": N, "n_rti": 1, "qp_mode": "auto" if args.qp_mode == 0 else "ipm_always",
                           "work_queue": eng.work_queue, "launch": launch_mode[:80], "warmup_untimed_extra_steps": head["extra_warm"],
                           "neighbour_exchange": headline, **({"instance_order": args.instance_order} if cfg4 else {}),
                           "parallelism": f"instances sharded x{world}"},
                "roofline": {"kernel": "rti_kernel", "bound": "mfma", "achieved": ach_tf, "peak": F64_MFMA_PEAK_TFLOPS,
                             "unit": "TFLOP/s", "frac": ach_tf / F64_MFMA_PEAK_TFLOPS, "traffic": prof["traffic"],
                             "algorithmic_bytes": abytes * B, "profile_tag": prof["tag"],
                             **({"profile_mismatch": profile_mismatch} if profile_mismatch else {}),
                             "kernel_us": rti_s * 1e6, "kernel_us_rocprof": prof["kernel_us"], "kernel_us_dispatch": pair_s * 1e6,
                             "frac_rocprof": (f_qp * B / (prof["kernel_us"] * 1e-6) / 1e12 / F64_MFMA_PEAK_TFLOPS) if prof["kernel_us"] else None,
                             "flops_per_solve_f64": f_qp, "sweeps_per_solve": sweeps, "frac_interior_point": frac_ipm,
                             "clock_ghz": clock_hz / 1e9 if clock_hz else None,
                             "mfma_pipe_busy_est": pipe_cycles / (rti_s * clock_hz) if clock_hz else None,
                             "hbm_frac": abytes * B / (rti_s + mlp_s) / 1e9 / HBM_PEAK_GBS,
                             **({"mlp_kernel_us": mlp_s * 1e6} if mlp_n else {})},
                "parity_max_rel_vs_oracle": parity, "instances_not_converged": bad,
                # SURVEY 8d words the metric over ndp_step calls INCLUDING the H2D of the inputs and the D2H of u0; `value` is the
                # device-resident rate (inputs in HBM when the clock starts, the contract's reading).  The host-inclusive rates of the same
                # run stand beside it: value_host_inclusive (ndp_step_begin / _end, two ticks in flight, all inputs across PCIe) and
                # value_tick (ndp_tick: only what is new per control period crosses PCIe) -- filled in below when those legs ran.
                "metric_variant": "device_resident", "value_host_inclusive": None, "value_tick": None,
            }
            rp = head.get("repeat_ms") or []
            if rp:
                out["repeat"] = {"n": len(rp), "ms_per_step": [min(rp) / args.steps, float(np.median(rp)) / args.steps, max(rp) / args.steps],
                                 "value_median": total / (float(np.median(rp)) * 1e-3)}
            legend = dict(LEGEND)
            legend["config.neighbour_exchange." + headline] = exchange_mode
            legend["roofline.kernel_us"] = ("HIP events on the launch stream around the timed region / steps" if one_launch
                                            else "HIP start / stop events on the dispatch packets of host-launched steps")
            legend["roofline.clock_ghz"] = clock_src
            if partial:
                out["watchdog"] = {"fired": True, "legs": {m: e[:80] for m, e in form_errors.items()}, "headline_parity_checked": parity is not None}
                legend["watchdog"] = ("a secondary form did not come back within --leg-timeout-s: this line was built by the watchdog from the forms that had "
                                      "finished (no extra legs, no CPU baseline); a headline whose parity check had not run yet is reported as value_unchecked; exit code 5")
            if two_forms:
                out["forms"] = {m: {"value": total / r["elapsed"], "us": r["elapsed"] / args.steps * 1e6, "parity": r["parity"], "bad": r["bad"],
                                    **({"late_waves": r["prefetch_stats"]["late_waves"],
                                        "timeouts": r["prefetch_stats"]["force_timeouts"] + r["prefetch_stats"]["slot_timeouts"]} if "prefetch_stats" in r else {})}
                                for m, r in results.items() if m not in baseline_modes}
                for m, e in form_errors.items():
                    if m not in baseline_modes:
                        out["forms"][m] = {"error": e[:120]}
                for m, r in results.items():
                    legend["launch." + m] = r["launch"]
                if baseline_modes:
                    # the N > 1 lines' own form at one rank (see baseline_modes above): what the driver's 1 -> N efficiency should divide by
                    sb = {m: {"value": total / r["elapsed"], "us": r["elapsed"] / args.steps * 1e6, "parity": r["parity"], "bad": r["bad"], "ok": form_ok(r)}
                          for m, r in results.items() if m in baseline_modes}
                    for m, e in form_errors.items():
                        if m in baseline_modes:
                            sb[m] = {"error": e[:120]}
                    okm = [m for m in sb if sb[m].get("ok") and m not in PEER_FORMS]          # (the base of the curve is the headline form's: rccl)
                    best = max(okm, key=lambda m: sb[m]["value"]) if okm else None
                    out["scaling_baseline"] = {"value": sb[best]["value"] if best else None, "ms_per_step": sb[best]["us"] / 1e3 if best else None,
                                               "form": best, "forms": sb}
            if need_exchange:
                # both forms of the per-step neighbour exchange, same steps, same inputs (value = whole-job solves/s)
                out["exchange"] = {m: {"value": total / r["elapsed"], "ms_per_step": r["elapsed"] / args.steps * 1e3,
                                       "parity": r["parity"], "bad": r["bad"],
                                       "ok": form_ok(r), **({"peer_stats": r["peer_stats"]} if "peer_stats" in r else {}),
                                       **({"prefetch_stats": r["prefetch_stats"]} if "prefetch_stats" in r else {})}
                                   for m, r in results.items()}
                for m, r in results.items():
                    legend["exchange." + m] = r["name"] + "; launch: " + r["launch"]
                if any(not form_ok(r) for m, r in results.items() if m != headline):
                    out["secondary_form_failed"] = True
                for m, e in form_errors.items():
                    out["exchange"][m] = {"error": e}
                if peer_err:
                    for m in PEER_FORMS:
                        out["exchange"][m] = {"error": peer_err}
                if world > 1:
                    out["scaling_baseline"] = {"compare_with": "the N = 1 line's scaling_baseline.value", "form": headline}
                if tick_remote is not None:
                    out["tick_remote"] = tick_remote
                if xchg_err and "rccl" in out["exchange"]:
                    out["exchange"]["rccl"]["library_collective_unavailable"] = xchg_err
                out["exchange"]["headline"] = headline
            extras = world == 1 and not args.only_timed and not cfg4 and args.perturb == "nominal" and args.qp_mode == 0 and not partial
            if extras and B == 1024 and N == 20 and not args.no_configs:
                try:
                    from oracle import oracle as O_
                    tcb = time.perf_counter()
                    out["configs"] = baseline_configs_block(ndp, ndist, synth, O_, torch, dev, stream, local_rank)
                    out["configs"]["seconds"] = time.perf_counter() - tcb
                except Exception as e:                      # never fatal to the headline
                    out["configs"] = {"error": f"{type(e).__name__}: {e}"[:200]}
                    torch.cuda.set_stream(stream)
            if extras:
                # ---- what the reference's QP solver actually does (HPIPM always iterates): every instance through the interior-point loop
                e_ipm = ndp.BatchedNMPC(B, N=N, disturbance=downwash, qp_mode=1, device=local_rank)
                nom = [{k: v for k, v in d.items() if k in ("x0", "xr", "ur", "other", "ego_xy")} for d in ticks]
                if not downwash:
                    nom = [{k: d[k] for k in ("x0", "xr", "ur")} for d in ticks]
                out["ipm_always"] = timed_leg(e_ipm, nom, 60)
                del e_ipm
                # ---- mixed workload: ~20 % of the instances hit a bound and run the loop, the rest take the early exit.  At
                # batch = SIMD count the step lasts as long as its slowest instance; with several instances per SIMD the work queue
                # hands the interior-point solves to whichever wave is free (compare work_queue on / off at the larger batch).
                def mixed_ticks(bb):
                    tk = []
                    for t in range(4):
                        m = synth.make_batch(bb, N=N, seed=synth.SEED0 + 40, downwash=downwash, t0=0.02 * t, **mixed_kw)
                        tk.append({k: torch.from_numpy(m[k]).to(dev) for k in (("x0", "xr", "ur", "other", "ego_xy") if downwash else ("x0", "xr", "ur"))})
                    return tk
                out["mixed"] = {}
                mt = mixed_ticks(B)
                e_m = ndp.BatchedNMPC(B, N=N, disturbance=downwash, device=local_rank)
                out["mixed"]["b%d" % B] = timed_leg(e_m, mt, 60)
                sw_m, act_m = e_m.active_set()
                out["mixed"]["b%d" % B].update({"constrained": float(act_m.any(axis=(1, 2)).mean()), "sweeps_max": int(sw_m.max())})
                del e_m
                e_m = ndp.BatchedNMPC(B, N=N, disturbance=downwash, device=local_rank, as_iter_max=0)     # rounds 1-5: early exit or interior point
                out["mixed"]["b%d" % B]["value_as_off"] = timed_leg(e_m, mt, 60)["value"]
                del e_m
                # ---- every instance on an input bound: the mixed workload's CONSTRAINED instances only.  A pool of 8 B perturbed starts,
                # one control tick of the device itself, the first B instances whose QP solution has an input on its bound (no oracle in a
                # timed leg's way); then the same four ticks as above.  as_off: the same through the interior-point loop (what HPIPM does).
                pool = synth.make_batch(8 * B, N=N, seed=synth.SEED0 + 41, downwash=downwash, **mixed_kw)
                e_p = ndp.BatchedNMPC(8 * B, N=N, disturbance=downwash, device=local_rank)
                e_p.reset(pool["xr"], pool["ur"])
                e_p.update(pool["x0"], pool["xr"], pool["ur"], raise_on_status=False,
                           **(dict(other=pool["other"], ego_xy=pool["ego_xy"]) if downwash else {}))
                pick = np.flatnonzero(e_p.active_set()[1].any(axis=(1, 2)))[:B]
                del e_p
                if len(pick) == B:
                    ct = []
                    for t in range(4):
                        m = synth.make_batch(8 * B, N=N, seed=synth.SEED0 + 41, downwash=downwash, t0=0.02 * t, **mixed_kw)
                        ct.append({k: torch.from_numpy(np.ascontiguousarray(m[k][pick])).to(dev)
                                   for k in (("x0", "xr", "ur", "other", "ego_xy") if downwash else ("x0", "xr", "ur"))})
                    e_c = ndp.BatchedNMPC(B, N=N, disturbance=downwash, device=local_rank)
                    out["constrained"] = timed_leg(e_c, ct, 60)
                    sw_c, act_c = e_c.active_set()
                    out["constrained"].update({"constrained": float(act_c.any(axis=(1, 2)).mean()), "sweeps_max": int(sw_c.max()),
                                               "sweeps_mean": float(sw_c.mean()), "pins_mean": float((act_c != 0).sum(axis=(1, 2)).mean())})
                    del e_c
                    e_c = ndp.BatchedNMPC(B, N=N, disturbance=downwash, device=local_rank, as_iter_max=0)
                    lg = timed_leg(e_c, ct, 60)
                    out["constrained"].update({"value_as_off": lg["value"], "as_off_ipm": lg["ipm"], "as_off_bad": lg["bad"]})
                    del e_c, ct
                else:
                    out["constrained"] = {"error": f"only {len(pick)} constrained instances in the pool"}
                del mt
                for Bq in (2 * B, 8 * B):      # two and eight instances per SIMD
                    mt = mixed_ticks(Bq)
                    e_q = ndp.BatchedNMPC(Bq, N=N, disturbance=downwash, device=local_rank)          # the automatic choice
                    auto = timed_leg(e_q, mt, 30, n_warm=40)                              # (the automatic rule decides within 16-24 steps)
                    auto_on = e_q.work_queue
                    del e_q
                    e_q = ndp.BatchedNMPC(Bq, N=N, disturbance=downwash, device=local_rank, work_queue=2 if auto_on else 1)
                    other = timed_leg(e_q, mt, 30, n_warm=40)
                    del e_q, mt
                    out["mixed"]["b%d" % Bq] = {"value": auto["value"], "us": auto["us"], "ipm": auto["ipm"], "bad": auto["bad"], "wq_auto": auto_on,
                                                ("value_wq_off" if auto_on else "value_wq_on"): other["value"]}
                # ---- the metric as SURVEY 8d words it: host arrays in, host arrays out (H2D of the inputs and D2H of u0 inside the
                # time).  Pageable numpy arrays, as the reference's callers hold them (nmpc_body_rate_ctl.py:93-112).  Two forms of the
                # same C-ABI path: ndp_step (one tick at a time: pack -> H2D -> kernel -> D2H -> wait) and ndp_step_begin / _end with
                # two ticks in flight (tick i+1's packing and PCIe transfer run under tick i's kernel; x0 comes from odometry, not
                # from the previous u0: nmpc_node.py:202-226).
                hb = {k: np.ascontiguousarray(host0[k]) for k in ("x0", "xr", "ur", "other", "ego_xy")}
                e_h = ndp.BatchedNMPC(B, N=N, disturbance=downwash, device=local_rank)
                e_h.reset(hb["xr"], hb["ur"])
                kw = dict(other=hb["other"], ego_xy=hb["ego_xy"]) if downwash else {}
                u_pipe = np.empty((B, 4))
                e_h.update_begin(hb["x0"], hb["xr"], hb["ur"], **kw)
                for _ in range(300):                  # the first ~100 ms of host-array steps run slower (link / clock state): untimed
                    e_h.update_begin(hb["x0"], hb["xr"], hb["ur"], **kw)
                    e_h.update_end(out=u_pipe)
                e_h.update_end(out=u_pipe)
                nh = 100
                th = time.perf_counter()
                for _ in range(nh):
                    e_h.update(hb["x0"], hb["xr"], hb["ur"], **kw)
                th = (time.perf_counter() - th) / nh
                e_h.update_begin(hb["x0"], hb["xr"], hb["ur"], **kw)
                for _ in range(5):
                    e_h.update_begin(hb["x0"], hb["xr"], hb["ur"], **kw)
                    e_h.update_end(out=u_pipe)
                tp = time.perf_counter()
                for _ in range(nh):
                    e_h.update_begin(hb["x0"], hb["xr"], hb["ur"], **kw)
                    e_h.update_end(out=u_pipe)
                tp = (time.perf_counter() - tp) / nh
                e_h.update_end(out=u_pipe)
                # host -> device bytes per solve as they cross PCIe: x0, xr, ur, and of the caller's 10-column neighbour windows the 6
                # position / velocity columns the gate and the network read (packed that way into the mirror), ego xy
                in_b = 8 * (10 + 10 * (N + 1) + 4 * N) + ((8 * 6 * (N + 1) + 16) if downwash else 0)
                out["host"] = {"two": {"value": B / tp, "us": tp * 1e6, "pcie_GBps_implied": (in_b + 40) * B / tp / 1e9},
                               "one": {"value": B / th, "us": th * 1e6}, "bytes_in_per_solve": in_b, **e_h.host_info()}
                out["value_host_inclusive"] = B / tp
                del e_h
                # ---- the same metric when only what is NEW crosses PCIe: the node's control tick on the device (ndp_tick)
                try:
                    out["tick"] = tick_block(ndp, synth, B, N, local_rank)
                    out["value_tick"] = out["tick"].get("value_host_inclusive_x0_only")
                except Exception as e:                      # never fatal to the headline
                    out["tick"] = {"error": f"{type(e).__name__}: {e}"[:200]}
                try:
                    sys.path.insert(0, os.path.join(ROOT, "scripts"))
                    import bench_rows
                    out["rows"] = bench_rows.rows_block(torch, ndp, dev)
                except Exception as e:
                    out["rows"] = {"error": f"{type(e).__name__}: {e}"[:200]}
                torch.cuda.set_stream(stream)
            if not args.no_cpu_baseline and not args.only_timed and world == 1 and not cfg4 and not partial:
                from oracle import oracle as O
                nthr = min(O.num_threads(), effective_cores())
                blob = np.fromfile(os.path.join(ROOT, "ndp_nmpc_qd_amd", "weights", "downwash_sn4.bin"), dtype="<f4")

                def cpu_rate(qp_mode, seconds):
                    cfgo = O.default_cfg(N=N, use_fd=downwash)
                    cfgo.qp_mode = qp_mode
                    Xo, Uo = host0["xr"].copy(), host0["ur"].copy()
                    O.step_batch(cfgo, host0["x0"], host0["xr"], host0["ur"], None, Xo, Uo, nthreads=nthr)       # warm the thread pool
                    Xo, Uo = host0["xr"].copy(), host0["ur"].copy()
                    n, t_mlp, tc = 0, 0.0, time.perf_counter()
                    while True:
                        tm = time.perf_counter()
                        f = O.downwash_batch(blob, host0["other"], host0["xr"], host0["ego_xy"], nthreads=nthr) if downwash else None
                        t_mlp += time.perf_counter() - tm
                        O.step_batch(cfgo, host0["x0"], host0["xr"], host0["ur"], f, Xo, Uo, nthreads=nthr)
                        n += 1
                        if time.perf_counter() - tc >= seconds:
                            break
                    tt = time.perf_counter() - tc
                    return B * n / tt, n, t_mlp / tt
                mode = args.qp_mode
                v_same, n_same, mlp_share = cpu_rate(mode, args.cpu_seconds)
                v_ipm, n_ipm, _ = (v_same, n_same, mlp_share) if mode == 1 else cpu_rate(1, args.cpu_seconds * 0.5)
                out["cpu_baseline"] = {"value": v_same, "unit": "solves/s", "cores": nthr, "kind": "port", "per_core": v_same / nthr,
                                       "mlp_share": mlp_share,
                                       "qp_mode": "auto" if mode == 0 else "ipm_always",
                                       "sample": f"{n_same} ticks of the same batch={B} workload, ~{args.cpu_seconds:.0f} s",
                                       "ipm_always_value": v_ipm}
                # BASELINE config 1 (one vehicle, N = 20, no downwash, 1 RTI iteration), the reference's own drop-in shape: the
                # controller object exactly as nmpc_node.py:202-209 calls it -- numpy x0 / xr / ur in, numpy u0 out, every tick --
                # next to the CPU restatement on one thread.  The reference's budget per tick is 20 ms (nmpc_node.py:216-220).
                from ndp_nmpc_qd_amd.nmpc_ctl import NMPCBodyRateController
                c1 = O.default_cfg(N=N, use_fd=False)
                x0_1, xr_1, ur_1 = host0["x0"][0].copy(), host0["xr"][0].copy(), host0["ur"][0].copy()
                n1 = 300
                lat = {}
                for qm, nm in ((1, "cpu_restatement_ipm_always_us"), (0, "cpu_restatement_auto_us")):
                    c1.qp_mode = qm
                    X1, U1 = xr_1[None].copy(), ur_1[None].copy()
                    t1 = time.perf_counter()
                    for _ in range(n1):
                        O.step_batch(c1, x0_1[None], xr_1[None], ur_1[None], None, X1, U1, nthreads=1)
                    lat[nm] = (time.perf_counter() - t1) / n1 * 1e6
                ctl = NMPCBodyRateController(device=local_rank)
                ctl.reset(xr_1, ur_1)
                for _ in range(20):
                    ctl.update(x0_1, xr_1, ur_1)
                t2 = time.perf_counter()
                for _ in range(n1):
                    ctl.update(x0_1, xr_1, ur_1)
                lat["gpu_drop_in_update_us"] = (time.perf_counter() - t2) / n1 * 1e6
                # the same tick without the Python facade: BatchedNMPC(1).update(full=True) on numpy arrays
                e1 = ndp.BatchedNMPC(1, N=N, device=local_rank)
                e1.reset(xr_1[None], ur_1[None])
                for _ in range(20):
                    e1.update(x0_1[None], xr_1[None], ur_1[None], full=True)
                t3 = time.perf_counter()
                for _ in range(n1):
                    e1.update(x0_1[None], xr_1[None], ur_1[None], full=True)
                lat["gpu_ndp_step_ex_us"] = (time.perf_counter() - t3) / n1 * 1e6
                c1o = {"b2b": {"update": lat["gpu_drop_in_update_us"], "step_ex": lat["gpu_ndp_step_ex_us"], "cpu_auto": lat["cpu_restatement_auto_us"],
                               "cpu_ipm": lat["cpu_restatement_ipm_always_us"]}, "deadline_us": 20000.0}
                # ... and at the reference's cadence: one call every 20 ms (rospy.Timer(ts_nmpc), nmpc_node.py:94), the GPU idle in between
                if args.cadence_ticks > 0:
                    try:
                        sys.path.insert(0, os.path.join(ROOT, "scripts"))
                        import cadence_50hz as cad
                        tick1 = cad.single_vehicle_tick(ndp, synth, local_rank)
                        c1o["b2b"]["tick"] = float(np.median(cad.back_to_back(tick1, 200)))
                        pct = lambda v: [float(np.median(v)), float(np.percentile(v, 99)), float(v.max())]      # noqa: E731
                        c1o["hz50"] = {"update": pct(cad.paced(lambda: ctl.update(x0_1, xr_1, ur_1), args.cadence_ticks)),
                                       "step_ex": pct(cad.paced(lambda: e1.update(x0_1[None], xr_1[None], ur_1[None], full=True), args.cadence_ticks // 2)),
                                       "tick": pct(cad.paced(tick1, args.cadence_ticks // 2)), "n": args.cadence_ticks}
                    except Exception as e:
                        c1o["hz50"] = {"error": f"{type(e).__name__}: {e}"[:160]}
                out["config1"] = c1o
            ps = head.get("peer_stats")                               # (only when the peer form is the headline)
            peer_bad = bool(ps and (ps["ack_timeouts"] or ps["epoch_timeouts"] or ps["slot_mismatches"]))
            pf = results["prefetch"].get("prefetch_stats") if "prefetch" in results else None
            if pf and (pf["force_timeouts"] or pf["slot_timeouts"]) and headline == "prefetch":
                peer_bad, ps = True, pf
            fail = (parity is not None and not parity <= 1e-5) or bad > 0 or peer_bad
            if fail:
                out["error"] = (f"parity_max_rel_vs_oracle {parity} (bar 1e-5), instances_not_converged {bad}"
                                + (f", peer exchange waits timed out / slots mismatched {ps}" if peer_bad else "") + ": value withheld")
                out["value_unchecked"], out["value"] = out["value"], None
            elif partial and parity is None and not args.only_timed:
                # the watchdog fired before the headline's parity check had run (one rank: the checks run behind all timed regions):
                # an unchecked number is not published as `value`
                out["error"] = "watchdog: a secondary form hung before the headline's parity check had run: value withheld"
                out["value_unchecked"], out["value"] = out["value"], None
                fail = True
            print("[bench legend] " + json.dumps(legend), file=sys.stderr, flush=True)
            line = json.dumps(compact(out), separators=(",", ":"))
            if len(line) > 6144:
                print(f"[bench] the line is {len(line)} bytes (> 6 KB: the driver's record keeps the tail of stdout)", file=sys.stderr, flush=True)
            print(line, flush=True)
        else:
            fail = False
        return fail

    # Which forms may fail without taking the line away: everything but the first.  A secondary form runs under a watchdog
    # (--leg-timeout-s): a multi-rank capture of the collective, or the peer form's xGMI leg, has never run on hardware -- if such
    # a leg does not come back, every rank's timer prints (rank 0) the line from the forms that DID finish and leaves.  With more
    # than one rank the parity check of a form runs right behind its timed region (so that a finished form is a checked form
    # when a later one hangs); with one rank all checks run behind all timed regions (the oracle's OpenMP phase in front of a
    # 20-step timed region costs it 3-6 %).
    import threading
    check_now = world > 1
    wd_lock = threading.Lock()
    wd_state = {"done": False}

    def watchdog(m):
        with wd_lock:
            if wd_state["done"]:
                return
            wd_state["done"] = True
        form_errors[m] = f"did not finish within {args.leg_timeout_s:.0f} s (watchdog): the line carries the forms that did"
        # a watchdog exit is never a success: the launcher (and the driver) see a non-zero code -- except for the tick_remote leg, which
        # runs behind every exchange form's timed region AND parity check: the line is complete without it, and says that it hung
        code = 0 if m == "tick_remote" else 5
        try:
            if rank == 0:
                finish(partial=True)
        except BaseException:
            import traceback
            traceback.print_exc()
        finally:
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(code)

    def run_checked(m):
        r = run_mode(m, check_parity=not args.only_timed)
        if check_now and r.get("parity_fn") is not None:
            r["parity"] = r.pop("parity_fn")()
        return r

    for k, m in enumerate(modes):
        stage["at"] = f"form '{m}' (warm-up, timed steps, checks)"
        if k == 0:
            results[m] = run_checked(m)
            continue
        timer = threading.Timer(args.leg_timeout_s, watchdog, args=(m,))
        timer.daemon = True
        timer.start()
        try:
            results[m] = run_checked(m)
        except Exception as e:
            form_errors[m] = f"{type(e).__name__}: {e}"[:300]
            torch.cuda.set_stream(stream)
        finally:
            timer.cancel()
        with wd_lock:
            if wd_state["done"]:                   # the timer fired while this leg was being torn down: it prints and exits
                time.sleep(3600)
    # ---- N > 1: the node's control tick (ndp_tick) with every neighbour on the next rank -- a secondary leg like the forms above
    if world > 1 and need_exchange and not cfg4 and N == 20 and args.qp_mode == 0 and not args.only_timed and not args.no_tick_remote:
        stage["at"] = "tick_remote (ndp_tick with neighbours on the next rank)"
        timer = threading.Timer(args.leg_timeout_s, watchdog, args=("tick_remote",))
        timer.daemon = True
        timer.start()
        try:
            tick_remote = tick_remote_ranks(ndp, synth, dist, torch, B, N, rank, world, local_rank, dev, cdev, stream, same_dev, xchg=xchg)
            if xchg is not None and "error" not in tick_remote:      # (the same on every rank) ... and with the exchange one period ahead
                ta_ = tick_remote_ranks(ndp, synth, dist, torch, B, N, rank, world, local_rank, dev, cdev, stream, same_dev, xchg=xchg, ahead=True)
                tick_remote = {**ta_, "serial": {k: tick_remote.get(k) for k in ("value", "us", "parity", "bad", "rows_ok")}}
        except Exception as e:
            tick_remote = {"error": f"{type(e).__name__}: {e}"[:300]}
            torch.cuda.set_stream(stream)
        finally:
            timer.cancel()
        with wd_lock:
            if wd_state["done"]:
                time.sleep(3600)
    for k, (m, r) in enumerate(list(results.items())):     # parity spot checks (CPU oracle, OpenMP): behind every form's timed region
        fn = r.pop("parity_fn", None)
        if fn is None:
            continue
        if k > 0:
            try:
                r["parity"] = fn()
            except Exception as e:
                form_errors[m] = f"{type(e).__name__}: {e}"[:300]
                torch.cuda.set_stream(stream)
        else:
            r["parity"] = fn()


    stage["at"] = "the line's extra legs (configs, interior point always, mixed, host path, CPU baseline)"
    fail = finish()
    stage["at"] = "teardown"
    if xchg is not None:
        xchg.close()
    if peer is not None:
        peer.close()
    if world > 1:
        flag = torch.tensor([int(fail)], dtype=torch.int64, device=cdev)
        dist.broadcast(flag, 0)
        fail = bool(flag.item())
        dist.destroy_process_group()
    if fail:
        sys.exit(1)


if __name__ == "__main__":
    main()
