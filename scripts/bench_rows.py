#!/usr/bin/env python3
"""Measurement of the widened rows -- separate from bench.py, whose line is the headline metric.

    python scripts/bench_rows.py --row throttle --batch 4194304

throttle (f3): one KF + differentiator update + actuator command per vehicle and tick.  Elementwise, HBM-bound:
algorithmic bytes per vehicle = read vz 8 + throttle 8 + state 64, write state 64 + k 8 (152 B) for the estimator,
read u0 32 + k 8, write cmd 32 (72 B) for the actuator command.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _events(torch, fn, steps, warmup):
    """fn(i) `steps` times behind `warmup` untimed calls; returns seconds per call by HIP events on the current stream."""
    for i in range(warmup):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / steps


def _trajectories(B, M=4, seed=7):
    import numpy as np
    from ndp_nmpc_qd_amd.pt_pub import TrajCoefficients
    rng = np.random.default_rng(seed)
    wp = np.zeros((B, 4, M + 1))
    wp[:, 0:2] = np.cumsum(rng.uniform(-1.0, 1.0, (B, 2, M + 1)), axis=2)
    wp[:, 2] = 1.0 + 0.2 * rng.uniform(-1, 1, (B, M + 1))
    wp[:, 3] = np.cumsum(rng.uniform(-0.3, 0.3, (B, M + 1)), axis=1)
    return TrajCoefficients.from_waypoints(wp, rng.uniform(3.0, 5.0, (B, M)))


def rows_block(torch, ndp, dev, B=1 << 18):
    """The widened rows in bench.py's line (`rows`): each on device buffers at batch B, ~0.2 s per row.  value = vehicles per second,
    frac = algorithmic bytes (the docstrings below) / kernel time / 8 TB/s."""
    import ctypes as C
    import numpy as np
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    out = {"batch": B}
    tc = _trajectories(B)
    eng = ndp.BatchedNMPC(B, load_mlp=False)
    eng.ref_set_trajectory(tc.coeff_x, tc.coeff_y, tc.coeff_z, tc.coeff_yaw, tc.traj_time_cum, tc.traj_time_seg, tc.final_pt)
    ts = [torch.full((B,), 0.02 * i, dtype=torch.float64, device=dev) for i in range(8)]
    xr = torch.empty(B, 21, 10, dtype=torch.float64, device=dev)
    ur = torch.empty(B, 20, 4, dtype=torch.float64, device=dev)
    dt = _events(torch, lambda i: eng.ref_window_device(ts[i % 8], xr, ur, stream=st), 30, 5)
    out["f1_window"] = {"value": B / dt, "frac": (8 + 224 + 72 + 1680 + 640) * B / dt / 8e12}
    eng.ref_list_reset()

    def list_tick(i):
        eng.ref_list_advance_device(ts[i % 8], stream=st)
        eng.ref_list_window_device(xr, ur, stream=st)
    dt = _events(torch, list_tick, 30, 5)
    out["f1_list"] = {"value": B / dt, "frac": ((8 + 224 + 72 + 2 * 112) + 2 * (1680 + 640)) * B / dt / 8e12}
    del eng, xr, ur
    B3 = 1 << 22                                          # (224 B per vehicle: a device-filling batch is a few million vehicles)
    e3 = ndp.BatchedNMPC(B3, N=2, load_mlp=False)         # tiny horizon: only the estimator state matters here
    vz = torch.randn(B3, dtype=torch.float64, device=dev) * 0.1
    th = torch.rand(B3, dtype=torch.float64, device=dev) * 0.8 + 0.15
    k = torch.empty(B3, dtype=torch.float64, device=dev)
    u0 = torch.randn(B3, 4, dtype=torch.float64, device=dev)
    cmd = torch.empty(B3, 4, dtype=torch.float64, device=dev)
    lib, h, s = e3._lib, e3._h, C.c_void_p(st.cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731

    def f3(i):
        lib.ndp_throttle_update_device(h, p(vz), p(th), p(k), s)
        lib.ndp_actuator_cmd_device(h, p(u0), p(k), p(cmd), s)
    dt = _events(torch, f3, 50, 5)
    out["f3"] = {"value": B3 / dt, "frac": (152 + 72) * B3 / dt / 8e12, "batch": B3}
    del e3
    Br = 1024
    tcr = _trajectories(Br)
    er = ndp.BatchedNMPC(Br)
    er.ref_set_trajectory(tcr.coeff_x, tcr.coeff_y, tcr.coeff_z, tcr.coeff_yaw, tcr.traj_time_cum, tcr.traj_time_seg, tcr.final_pt)
    x = torch.from_numpy(er.ref_window(np.zeros(Br))[0][:, 0].copy()).to(dev)
    er.rollout_device(20, x)
    er.synchronize()
    t0 = time.perf_counter()
    er.rollout_device(200, x, t0=0.4)
    er.synchronize()
    el = time.perf_counter() - t0
    out["rollout"] = {"value": Br * 200 / el, "us_per_tick": el / 200 * 1e6, "batch": Br}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--row", default="throttle")
    ap.add_argument("--batch", type=int, default=1 << 22)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    a = ap.parse_args()
    import torch
    import ndp_nmpc_qd_amd as ndp
    B = a.batch
    dev = torch.device("cuda:0")
    if a.row == "ref_window":
        return bench_ref_window(a, torch, ndp, dev)
    if a.row == "rollout":
        return bench_rollout(a, torch, ndp, dev)
    if a.row == "ref_list":
        return bench_ref_list(a, torch, ndp, dev)
    eng = ndp.BatchedNMPC(B, N=2, load_mlp=False)       # tiny horizon: only the estimator state matters here
    vz = torch.randn(B, dtype=torch.float64, device=dev) * 0.1
    th = torch.rand(B, dtype=torch.float64, device=dev) * 0.8 + 0.15
    k = torch.empty(B, dtype=torch.float64, device=dev)
    u0 = torch.randn(B, 4, dtype=torch.float64, device=dev)
    cmd = torch.empty(B, 4, dtype=torch.float64, device=dev)
    import ctypes as C
    lib, h = eng._lib, eng._h
    st = torch.cuda.Stream(device=dev)       # non-default: handle 0 would mean "the library's own stream"
    torch.cuda.set_stream(st)
    s = C.c_void_p(st.cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731

    def step():
        assert lib.ndp_throttle_update_device(h, p(vz), p(th), p(k), s) == 0
        assert lib.ndp_actuator_cmd_device(h, p(u0), p(k), p(cmd), s) == 0

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(a.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    dev_s = e0.elapsed_time(e1) * 1e-3 / a.steps
    bytes_per = 152 + 72
    print(json.dumps({"row": "f3 hover-throttle estimator + actuator command", "metric": "vehicle updates/s",
                      "value": B * a.steps / el, "batch": B, "ms_per_step": el / a.steps * 1e3, "dtype": "f64",
                      "roofline": {"bound": "hbm", "achieved": bytes_per * B / dev_s / 1e9, "peak": 8000.0, "unit": "GB/s",
                                   "frac": bytes_per * B / dev_s / 1e9 / 8000.0, "algorithmic_bytes_per_vehicle": bytes_per}}))


def bench_ref_window(a, torch, ndp, dev):
    """f1: reference windows of B vehicles per tick (4-segment minimum-snap trajectories, N = 20).  Algorithmic bytes
    per vehicle: read t 8 + the coefficients of the segments the window touches (one or two: 224..448, counted as 224)
    + 5 time_cum + 4 time_seg doubles 72, write xr 1680 + ur 640 = 2624 B."""
    import numpy as np
    from ndp_nmpc_qd_amd.pt_pub import TrajCoefficients
    B, M = a.batch, 4
    rng = np.random.default_rng(7)
    wp = np.zeros((B, 4, M + 1))
    wp[:, 0:2] = np.cumsum(rng.uniform(-1.0, 1.0, (B, 2, M + 1)), axis=2)
    wp[:, 2] = 1.0 + 0.2 * rng.uniform(-1, 1, (B, M + 1))
    wp[:, 3] = np.cumsum(rng.uniform(-0.3, 0.3, (B, M + 1)), axis=1)
    tc = TrajCoefficients.from_waypoints(wp, rng.uniform(3.0, 5.0, (B, M)))
    eng = ndp.BatchedNMPC(B, load_mlp=False)
    eng.ref_set_trajectory(tc.coeff_x, tc.coeff_y, tc.coeff_z, tc.coeff_yaw, tc.traj_time_cum, tc.traj_time_seg, tc.final_pt)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    ts = [torch.full((B,), 0.02 * i, dtype=torch.float64, device=dev) for i in range(8)]
    xr = torch.empty(B, 21, 10, dtype=torch.float64, device=dev)
    ur = torch.empty(B, 20, 4, dtype=torch.float64, device=dev)
    for i in range(a.warmup):
        eng.ref_window_device(ts[i % 8], xr, ur, stream=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(a.steps):
        eng.ref_window_device(ts[i % 8], xr, ur, stream=st)
    e1.record()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    dev_s = e0.elapsed_time(e1) * 1e-3 / a.steps
    bytes_per = 8 + 224 + 72 + 1680 + 640
    print(json.dumps({"row": "f1 reference window generation (polynomial trajectory + differential flatness)",
                      "metric": "vehicle windows/s", "value": B * a.steps / el, "batch": B, "ms_per_step": el / a.steps * 1e3,
                      "dtype": "f64", "kernel_us": dev_s * 1e6,
                      "roofline": {"bound": "hbm", "achieved": bytes_per * B / dev_s / 1e9, "peak": 8000.0, "unit": "GB/s",
                                   "frac": bytes_per * B / dev_s / 1e9 / 8000.0, "algorithmic_bytes_per_vehicle": bytes_per}}))


def bench_ref_list(a, torch, ndp, dev):
    """f1, the reference's own bookkeeping (NMPCRefPublisher's sliding list, pt_pub/pt_publisher.py:36-103) on the device: per
    tick ONE new reference point per vehicle into the phase-major list (ref_list_fill_kernel: every entry is stored twice so that
    every window is contiguous) and the window as a dense copy (ref_list_window_kernel).  Algorithmic bytes per vehicle and tick:
    advance = read t 8 + coefficients 224 + 72, write one entry twice 224; window = read 1680 + 640, write xr 1680 + ur 640."""
    import numpy as np
    from ndp_nmpc_qd_amd.pt_pub import TrajCoefficients
    B, M = a.batch, 4
    rng = np.random.default_rng(7)
    wp = np.zeros((B, 4, M + 1))
    wp[:, 0:2] = np.cumsum(rng.uniform(-1.0, 1.0, (B, 2, M + 1)), axis=2)
    wp[:, 2] = 1.0 + 0.2 * rng.uniform(-1, 1, (B, M + 1))
    wp[:, 3] = np.cumsum(rng.uniform(-0.3, 0.3, (B, M + 1)), axis=1)
    tc = TrajCoefficients.from_waypoints(wp, rng.uniform(3.0, 5.0, (B, M)))
    eng = ndp.BatchedNMPC(B, load_mlp=False)
    eng.ref_set_trajectory(tc.coeff_x, tc.coeff_y, tc.coeff_z, tc.coeff_yaw, tc.traj_time_cum, tc.traj_time_seg, tc.final_pt)
    eng.ref_list_reset()
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    ts = [torch.full((B,), 0.02 * i, dtype=torch.float64, device=dev) for i in range(8)]
    xr = torch.empty(B, 21, 10, dtype=torch.float64, device=dev)
    ur = torch.empty(B, 20, 4, dtype=torch.float64, device=dev)

    def tick(i):
        eng.ref_list_advance_device(ts[i % 8], stream=st)
        eng.ref_list_window_device(xr, ur, stream=st)
    for i in range(a.warmup):
        tick(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(a.steps):
        tick(i)
    e1.record()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    dev_s = e0.elapsed_time(e1) * 1e-3 / a.steps
    bytes_per = (8 + 224 + 72 + 2 * 112) + 2 * (1680 + 640)      # advance: t + one segment record + one entry stored twice; window: a dense copy
    print(json.dumps({"row": "f1 reference list on the device: one new point per vehicle and tick + window = every 5th entry",
                      "metric": "vehicle windows/s", "value": B * a.steps / el, "batch": B, "ms_per_step": el / a.steps * 1e3,
                      "dtype": "f64", "kernels_us": dev_s * 1e6,
                      "roofline": {"bound": "hbm", "achieved": bytes_per * B / dev_s / 1e9, "peak": 8000.0, "unit": "GB/s",
                                   "frac": bytes_per * B / dev_s / 1e9 / 8000.0, "algorithmic_bytes_per_vehicle": bytes_per}}))


def bench_rollout(a, torch, ndp, dev):
    """f4 + f1 + the control step: closed-loop rollout of B vehicles on their minimum-snap trajectories, a.steps ticks in
    one ndp_rollout_device call (3 launches per tick: reference window, control step, plant step)."""
    import numpy as np
    from ndp_nmpc_qd_amd.pt_pub import TrajCoefficients
    B, M = a.batch, 4
    rng = np.random.default_rng(7)
    wp = np.zeros((B, 4, M + 1))
    wp[:, 0:2] = np.cumsum(rng.uniform(-1.0, 1.0, (B, 2, M + 1)), axis=2)
    wp[:, 2] = 1.0 + 0.2 * rng.uniform(-1, 1, (B, M + 1))
    wp[:, 3] = np.cumsum(rng.uniform(-0.3, 0.3, (B, M + 1)), axis=1)
    tc = TrajCoefficients.from_waypoints(wp, rng.uniform(3.0, 5.0, (B, M)))
    eng = ndp.BatchedNMPC(B)
    eng.ref_set_trajectory(tc.coeff_x, tc.coeff_y, tc.coeff_z, tc.coeff_yaw, tc.traj_time_cum, tc.traj_time_seg, tc.final_pt)
    xr0, _ = eng.ref_window(np.zeros(B))
    x = torch.from_numpy(xr0[:, 0].copy()).to(dev)
    eng.rollout_device(a.warmup, x)
    eng.synchronize()
    t0 = time.perf_counter()
    eng.rollout_device(a.steps, x, t0=a.warmup * 0.02)
    eng.synchronize()
    el = time.perf_counter() - t0
    xr_end, _ = eng.ref_window(np.full(B, (a.warmup + a.steps) * 0.02))
    err = np.linalg.norm(x.cpu().numpy()[:, 0:3] - xr_end[:, 0, 0:3], axis=1)
    print(json.dumps({"row": "closed-loop rollout: reference window + control step (N=20, 1 RTI) + plant step per tick",
                      "metric": "vehicle ticks/s", "value": B * a.steps / el, "batch": B, "ticks": a.steps,
                      "us_per_tick": el / a.steps * 1e6, "simulated_seconds_per_vehicle": a.steps * 0.02,
                      "tracking_error_at_end_m": {"median": float(np.median(err)), "max": float(err.max())}}))


if __name__ == "__main__":
    main()
