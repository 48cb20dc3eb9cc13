"""SURVEY 8f-2 (follower reference relay) and 8f-4 (plant step for closed-loop rollouts).
relay_golden.npz: filter outputs produced by the reference's own AlphaFilter (tests/golden/make_relay_golden.py)."""
import os

import numpy as np
import pytest

from ndp_nmpc_qd_amd import synth
from ndp_nmpc_qd_amd.params import nmpc_params as CP
from tests import ref_numpy as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "relay_golden.npz"))


def test_oracle_relay_against_reference_fixture(oracle, gold):
    T, V, _ = gold["form"].shape
    st = np.zeros((V, 4))
    for t in range(T):
        off = oracle.relay_formation(st, gold["form"][t])
        assert np.array_equal(off, gold["off"][t])                     # bit-exact: same two multiplies and one add
        assert np.array_equal(oracle.relay_reference(st, gold["xr_lead"]), gold["xr_fol"][t])


def test_oracle_plant_step_vs_numpy_rk4(oracle):
    cfg = oracle.default_cfg()
    rng = np.random.default_rng(0)
    b = synth.make_batch(6, seed=2)
    x = b["x0"].copy()
    u = b["ur"][:, 0, :].copy()
    f = rng.normal(0, 1, (6, 3))
    x1 = oracle.plant_step(cfg, x.copy(), u, f, CP.ts_nmpc, 4)
    for i in range(6):
        xi = x[i].copy()
        for _ in range(4):
            xi = R.rk4(xi, u[i], f[i], CP.ts_nmpc / 4)
        xi[6:10] /= np.linalg.norm(xi[6:10])
        np.testing.assert_allclose(x1[i], xi, atol=1e-13)


@pytest.mark.gpu
def test_gpu_relay_against_reference_fixture(gold):
    import ndp_nmpc_qd_amd as ndp
    T, V, _ = gold["form"].shape
    eng = ndp.BatchedNMPC(V, load_mlp=False)
    for t in range(T):
        off = eng.relay_formation(gold["form"][t])
        assert np.array_equal(off, gold["off"][t])
        if t % 7 == 0 or t == T - 1:
            assert np.array_equal(eng.relay_reference(gold["xr_lead"]), gold["xr_fol"][t])
    eng.relay_reset()
    assert np.array_equal(eng.relay_formation(gold["form"][0]), gold["off"][0])


@pytest.mark.gpu
def test_gpu_plant_step_and_device_closed_loop(oracle):
    import ndp_nmpc_qd_amd as ndp
    cfg = oracle.default_cfg()
    B = 257
    rng = np.random.default_rng(1)
    b = synth.make_batch(B, seed=4)
    eng = ndp.BatchedNMPC(B)
    x, u, f = b["x0"].copy(), b["ur"][:, 0, :].copy(), rng.normal(0, 1, (B, 3))
    for ff in (None, f):
        xg = eng.plant_step(x, u, ff, CP.ts_nmpc, 4)
        xo = oracle.plant_step(cfg, x.copy(), u, ff, CP.ts_nmpc, 4)
        np.testing.assert_allclose(xg, xo, atol=1e-13)
    # 100 closed-loop ticks entirely through the device kernels (controller + plant), figure-eight tracking
    omega, phi = rng.uniform(0.5, 1.0, B), rng.uniform(0, 2 * np.pi, B)

    def windows(t0):
        t = t0 + CP.th_pred * np.arange(21)
        pos, vel, acc, jerk = synth.figure_eight(omega[:, None], phi[:, None], t[None, :])
        xr, ur = synth.diff_flatness(pos, vel, acc, jerk)
        return np.ascontiguousarray(xr), np.ascontiguousarray(ur[:, :20])

    xr, ur = windows(0.0)
    xs = xr[:, 0].copy()
    xs[:, 0:3] += rng.normal(0, 0.1, (B, 3))
    eng.reset(xr, ur)
    for k in range(100):
        xr, ur = windows(k * CP.ts_nmpc)
        u0 = eng.update(xs, xr, ur)
        xs = eng.plant_step(xs, u0, None, CP.ts_nmpc, 4)
    pr = synth.figure_eight(omega, phi, np.full(B, 100 * CP.ts_nmpc))[0]
    assert np.linalg.norm(xs[:, 0:3] - pr, axis=1).max() < 0.04


@pytest.mark.gpu
def test_gpu_closed_loop_rollout_matches_the_tick_by_tick_loop(oracle):
    """ndp_rollout_device (reference window -> control step -> plant step, 3 launches per tick, nothing returns to the
    host) against the same three calls made tick by tick from the host: identical states, bit for bit; and the vehicles
    follow their minimum-snap trajectories."""
    import torch
    import ndp_nmpc_qd_amd as ndp
    from ndp_nmpc_qd_amd.pt_pub import TrajCoefficients
    B, M, K = 96, 3, 60
    rng = np.random.default_rng(3)
    wp = np.zeros((B, 4, M + 1))
    wp[:, 0:2] = np.cumsum(rng.uniform(-1.0, 1.0, (B, 2, M + 1)), axis=2)
    wp[:, 2] = 1.0 + 0.2 * rng.uniform(-1, 1, (B, M + 1))
    wp[:, 3] = np.cumsum(rng.uniform(-0.2, 0.2, (B, M + 1)), axis=1)
    tc = TrajCoefficients.from_waypoints(wp, rng.uniform(3.0, 4.0, (B, M)))
    t0 = 0.4
    engs = [ndp.BatchedNMPC(B) for _ in range(2)]
    for e in engs:
        e.ref_set_trajectory(tc.coeff_x, tc.coeff_y, tc.coeff_z, tc.coeff_yaw, tc.traj_time_cum, tc.traj_time_seg, tc.final_pt)
    xr0, ur0 = engs[0].ref_window(np.full(B, t0))
    x_init = xr0[:, 0].copy()
    x_init[:, 0:3] += rng.normal(0, 0.05, (B, 3))
    # host loop
    e = engs[0]
    e.reset(xr0, ur0)
    xs = x_init.copy()
    states = []
    for k in range(K):
        xr, ur = e.ref_window(np.full(B, t0 + k * CP.ts_nmpc))
        u0 = e.update(xs, xr, ur)
        xs = e.plant_step(xs, u0, None, CP.ts_nmpc, 4)
        states.append(xs.copy())
    # device rollout
    dev = torch.device("cuda", 0)
    xd = torch.from_numpy(x_init).to(dev)
    log = torch.empty(K, B, 10, dtype=torch.float64, device=dev)
    engs[1].rollout_device(K, xd, log, t0=t0, dt=CP.ts_nmpc, substeps=4)
    engs[1].synchronize()
    assert np.array_equal(log.cpu().numpy(), np.stack(states))
    assert np.array_equal(xd.cpu().numpy(), states[-1])
    ref_end, _ = engs[0].ref_window(np.full(B, t0 + K * CP.ts_nmpc))
    assert np.linalg.norm(states[-1][:, 0:3] - ref_end[:, 0, 0:3], axis=1).max() < 0.03
