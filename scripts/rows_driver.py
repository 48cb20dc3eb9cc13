#!/usr/bin/env python3
"""Launches every secondary kernel of the library a few dozen times at a batch that fills the device -- the program scripts/profile_rows.sh
puts under rocprofv3 (kernel trace, then PMC passes).  No oracle, no child processes.
    python3 scripts/rows_driver.py [--batch 262144] [--reps 30]
Kernels: ref_window_kernel, ref_list_fill_kernel (advance), ref_list_window_kernel, throttle_kernel, actuator_kernel, plant_kernel,
relay_reference_kernel, mlp_kernel, mlp_stream_kernel, pack_pv_kernel, peer_publish_kernel + peer_epoch_kernel, tick_pre_kernel, and the
one-launch tick (rti_kernel<..., TICK>) at batch 1024."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1 << 18)
    ap.add_argument("--reps", type=int, default=30)
    a = ap.parse_args()
    import torch
    import ndp_nmpc_qd_amd as ndp
    from ndp_nmpc_qd_amd import dist as ndist
    from bench_rows import _trajectories
    B, R = a.batch, a.reps
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    s = C.c_void_p(st.cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    tc = _trajectories(B)
    eng = ndp.BatchedNMPC(B, disturbance=True)
    lib, h = eng._lib, eng._h
    eng.ref_set_trajectory(tc.coeff_x, tc.coeff_y, tc.coeff_z, tc.coeff_yaw, tc.traj_time_cum, tc.traj_time_seg, tc.final_pt)
    ts = [torch.full((B,), 0.02 * i, dtype=torch.float64, device=dev) for i in range(8)]
    xr = torch.empty(B, 21, 10, dtype=torch.float64, device=dev)
    ur = torch.empty(B, 20, 4, dtype=torch.float64, device=dev)
    xr2 = torch.empty_like(xr)
    for i in range(R):                                              # f1: windows straight from the polynomials
        eng.ref_window_device(ts[i % 8], xr, ur, stream=st)
    eng.ref_list_reset()
    for i in range(R):                                              # f1: the list -- advance + window copy
        eng.ref_list_advance_device(ts[i % 8], stream=st)
        eng.ref_list_window_device(xr, ur, stream=st)
    vz = torch.randn(B, dtype=torch.float64, device=dev) * 0.1
    th = torch.rand(B, dtype=torch.float64, device=dev) * 0.8 + 0.15
    k = torch.empty(B, dtype=torch.float64, device=dev)
    u0 = torch.randn(B, 4, dtype=torch.float64, device=dev)
    u0[:, 3] += 9.81
    cmd = torch.empty(B, 4, dtype=torch.float64, device=dev)
    x = xr[:, 0, :].clone()
    for i in range(R):                                              # f3, f4, f2
        lib.ndp_throttle_update_device(h, p(vz), p(th), p(k), s)
        lib.ndp_actuator_cmd_device(h, p(u0), p(k), p(cmd), s)
        lib.ndp_plant_step_device(h, p(x), p(u0), None, C.c_double(0.02), 4, s)
        lib.ndp_relay_reference_device(h, p(xr), p(xr2), s)
    f = torch.empty(B, 21, 3, dtype=torch.float32, device=dev)
    other = xr.roll(1, 0).contiguous()
    ego = x[:, 0:2].contiguous()
    for i in range(R):                                              # a7 / a8 as launches of their own
        eng.downwash_device(other, xr, f, ego_xy=ego, stream=st)
        lib.ndp_debug_downwash_stream_device(h, p(other), p(xr), p(ego), p(f), s)
    torch.cuda.synchronize()
    # the exchange's kernels (one rank: its own buffer is the neighbour's)
    Bx = min(B, 16384)
    try:
        xw = xr[:Bx].contiguous()
        xchg = ndist.RcclExchange(Bx, 20, 0)
        g = torch.empty(Bx, 21, ndist.PV_COLS, dtype=torch.float64, device=dev)
        for i in range(R):                                          # pack_pv_kernel (+ the one-rank gather)
            xchg.begin(xw, g, st)
            xchg.end(st)
        torch.cuda.synchronize()
        xchg.close()
    except Exception as e:
        print("rccl exchange skipped:", e, file=sys.stderr)
    try:
        peer = ndist.PeerWindows(Bx, 20, 0, timeout_us=20000)
        for i in range(R):                                          # peer_publish_kernel + peer_epoch_kernel
            peer.publish_device(xw, st)
        torch.cuda.synchronize()
        peer.close()
    except Exception as e:
        print("peer windows skipped:", e, file=sys.stderr)
    # tick_pre_kernel (the two-launch tick of the shapes without a TICK instantiation): N = 10
    Bt = min(B, 65536)
    e10 = ndp.BatchedNMPC(Bt, N=10, load_mlp=False)
    tc10 = _trajectories(Bt)
    e10.ref_set_trajectory(tc10.coeff_x, tc10.coeff_y, tc10.coeff_z, tc10.coeff_yaw, tc10.traj_time_cum, tc10.traj_time_seg, tc10.final_pt)
    e10.ref_list_reset()
    e10.tick_reset()
    xo = torch.from_numpy(e10.ref_list_window(None)[0][:, 0, :].copy()).to(dev)
    cm = torch.empty(Bt, 4, dtype=torch.float64, device=dev)
    for i in range(R):
        e10.tick_device(xo, cm, t=0.02 * i, estimate=True, stream=st)
    torch.cuda.synchronize()
    # the ONE-launch tick (rti_kernel<3, 4, true, 20, 0, 1, 0, true>) on the metric's workload: batch 1024, vehicle pairs, 36 % of the gates open
    from tick_rate import setup
    e1k = setup(1024)
    x1 = [torch.from_numpy(e1k.ref_window(np.full(1024, 0.02 * i))[0][:, 0, :].copy()).to(dev) for i in range(3 * R)]
    c1 = torch.empty(1024, 4, dtype=torch.float64, device=dev)
    for i in range(3 * R):
        e1k.tick_device(x1[i], c1, t=0.02 * i, estimate=True, stream=st)
    torch.cuda.synchronize()
    print("rows_driver done: batch", B, "reps", R)


if __name__ == "__main__":
    main()
