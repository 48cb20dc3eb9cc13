"""The node's control tick end to end on the device (ndp_tick: odometry in, actuator command out, references resident) --
ControllerNode.nmpc_callback + hover_throttle_callback, nmpc_node.py:211-231,251-253, for every vehicle of a handle.

What pins it:
  * the tick-by-tick composition of the calls that existed before it (reference list advance + window, estimator update,
    control step with the neighbour's window, actuator command): bit-equal u0 / cmd / status over 60 ticks;
  * the fixtures made by running the reference's own code: the 60-tick window sequence of NMPCRefPublisher (flat_golden.npz),
    the estimator sequence of HoverThrottleEstimator (throttle_golden.npz);
  * the CPU oracle (control step + downwash) on the same inputs.
"""
import os

import numpy as np
import pytest

from ndp_nmpc_qd_amd import synth
from ndp_nmpc_qd_amd.params import nmpc_params as CP

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_golden.npz"))


@pytest.fixture(scope="module")
def flat():
    return np.load(os.path.join(ROOT, "tests", "golden", "flat_golden.npz"))


@pytest.fixture(scope="module")
def thr_gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "throttle_golden.npz"))


def _cum(tseg):
    return np.concatenate([np.zeros((tseg.shape[0], 1)), np.cumsum(tseg, axis=1)], axis=1)


def _seq_traj(gold, flat, reps=1):
    """The five trajectories of the reference-run window sequence, `reps` times over (vehicle V r + v flies trajectory v)."""
    c = int(flat["seq_case"])
    coeff, tseg, wpts = gold[f"coeff_{c}"], flat["seq_tseg"], gold[f"wpts_{c}"]
    coeff, tseg, fpt = (np.tile(a, (reps,) + (1,) * (a.ndim - 1)) for a in (coeff, tseg, wpts[:, 0:3, -1].copy()))
    return coeff, tseg, _cum(tseg), fpt


def _set_traj(eng, coeff, tseg, cum, fpt):
    eng.ref_set_trajectory(coeff[:, :, 0:8], coeff[:, :, 8:16], coeff[:, :, 16:24], coeff[:, :, 24:28], cum, tseg, fpt)


def _odometry(rng, xr):
    """x0 = node 0 of the window + measurement noise (SURVEY 8d's recipe), quaternion normalised."""
    x = xr[:, 0, :].copy()
    x[:, 0:3] += rng.normal(0, 0.1, (x.shape[0], 3))
    x[:, 3:6] += rng.normal(0, 0.2, (x.shape[0], 3))
    x[:, 6:10] += rng.normal(0, 0.03, (x.shape[0], 4))
    x[:, 6:10] /= np.linalg.norm(x[:, 6:10], axis=1, keepdims=True)
    return x


def _rel(a, b):
    return np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))


def test_tick_equals_the_composition_the_fixtures_and_the_oracle(oracle, gold, flat, mlp_blob):
    import torch
    import ndp_nmpc_qd_amd as ndp
    V, reps = 5, 2
    B = V * reps
    coeff, tseg, cum, fpt = _seq_traj(gold, flat, reps)
    fpt[V:, 1] += 0.6                      # (the second set lands 0.6 m beside the first: final points differ)
    # vehicles 0..4: NDP leaders whose neighbour is the next vehicle of the set (ring); 5..9: plain vehicles (no neighbour).
    other_index = np.array([(v + 1) % V for v in range(V)] + [-1] * V, dtype=np.int32)
    dev = torch.device("cuda", 0)

    # (as_iter_max = 0: this test holds status and iteration counts against the always-iterating oracle, down to the two ticks on which the
    # interior-point loop gives up; the 260-tick test below runs tick and composition in the default mode, active-set iterations on)
    tk = ndp.BatchedNMPC(B, disturbance=True, as_iter_max=0)          # the tick
    cp = ndp.BatchedNMPC(B, disturbance=True, as_iter_max=0)          # the composition of the existing calls
    for e in (tk, cp):
        _set_traj(e, coeff, tseg, cum, fpt)
        e.ref_list_reset()
        e.throttle_reset()
    tk.tick_config(other_index, gate=True)
    tk.tick_reset()
    xr0, ur0 = cp.ref_list_window(None)
    np.testing.assert_allclose(xr0[:V], flat["seq_xr0"], rtol=0, atol=1e-7)
    cp.reset(xr0, ur0)
    Xt, Ut = tk.get_iterate()
    assert np.array_equal(Xt, xr0) and np.array_equal(Ut, ur0)       # tick_reset = reset(window of the list)

    cfg = oracle.default_cfg(use_fd=True)
    Xo, Uo = xr0.copy(), ur0.copy()
    tcfg = oracle.thr_default_cfg()
    tst = oracle.thr_reset(tcfg, B)
    ko = np.full(B, float(tcfg.k_init))
    idx_t = torch.from_numpy(other_index).to(dev)
    u0_t = torch.empty(B, 4, dtype=torch.float64, device=dev)
    rng = np.random.default_rng(11)
    thrust_prev = np.zeros(B)
    k_prev = np.full(B, 50.0)
    worst_u0 = worst_f = 0.0
    n_open = n_ipm = n_bad = 0
    for i, t in enumerate(flat["seq_t"]):
        tt = np.full(B, float(t))
        est = i % 3 != 2                                  # the estimator's timer is not the control timer: skip every third tick
        # ---- the composition, call by call
        xr, ur = cp.ref_list_window(tt)
        np.testing.assert_allclose(xr[:V], flat["seq_xr"][i], rtol=0, atol=1e-7)       # the reference's own windows
        np.testing.assert_allclose(ur[:V], flat["seq_ur"][i], rtol=0, atol=1e-7)
        x0 = _odometry(rng, xr)
        k = cp.throttle_update(x0[:, 5].copy(), thrust_prev) if est else k_prev
        cp.update_device(torch.from_numpy(x0).to(dev), torch.from_numpy(xr).to(dev), torch.from_numpy(ur).to(dev), u0_t,
                         other=torch.from_numpy(xr).to(dev), other_index=idx_t, ego_xy=torch.from_numpy(x0[:, 0:2].copy()).to(dev))
        cp.synchronize()
        u0_c = u0_t.cpu().numpy()
        st_c, it_c = cp.status()
        cmd_c = cp.actuator_cmd(u0_c, k)
        # ---- the tick
        cmd, u0, st, it = tk.tick(x0, t=tt, estimate=est, full=True, raise_on_status=False)
        assert np.array_equal(u0, u0_c), (i, np.max(np.abs(u0 - u0_c)))
        assert np.array_equal(cmd, cmd_c), (i, np.max(np.abs(cmd - cmd_c)))
        assert np.array_equal(st, st_c) and np.array_equal(it, it_c)
        # ---- the oracle: gate + network on the neighbour's window, the control step, estimator + thrust
        nb = np.where(other_index >= 0, other_index, 0)
        f = oracle.downwash_batch(mlp_blob, xr[nb], xr, x0[:, 0:2].copy())
        f[other_index < 0] = 0.0
        n_open += int(np.any(f[:V] != 0.0, axis=(1, 2)).sum())
        u_or, st_o, it_o = oracle.step_batch(cfg, x0, xr, ur, f, Xo, Uo)
        # trajectory 0 of the fixture saturates the thrust at both bounds; with this noise two of its ticks exhaust the interior-point
        # iterations -- on the device exactly where the oracle does (status 4 = acados' QP failure, never a silent answer)
        assert np.array_equal(st, st_o), (i, st, st_o)
        ok = st == 0
        n_bad += int((~ok).sum())
        n_ipm += int((it > 0).sum())
        worst_u0 = max(worst_u0, _rel(u0[ok], u_or[ok]))
        worst_f = max(worst_f, _rel(tk.device_force().cpu().numpy(), f))
        if est:
            ko = oracle.thr_update(tcfg, tst, x0[:, 5].copy(), thrust_prev)
        thrust_o = oracle.att_thrust(tcfg, u_or[:, 3], ko)
        assert np.max(np.abs(cmd[ok, 3] - thrust_o[ok]) / np.maximum(0.1, np.abs(thrust_o[ok]))) < 1e-5     # (a thrust of ~0.3; 0 when c sits on its lower bound)
        thrust_prev, k_prev = cmd[:, 3].copy(), k
    assert worst_u0 < 1e-5 and worst_f < 1e-5, (worst_u0, worst_f)
    assert n_open > 20                                     # the gate was open for some leaders, closed for others
    assert n_ipm > 60 and n_bad <= 6, (n_ipm, n_bad)        # active bounds were exercised through the tick
    Xt, Ut = tk.get_iterate()
    Xc, Uc = cp.get_iterate()
    assert np.array_equal(Xt, Xc) and np.array_equal(Ut, Uc)
    assert np.all(np.abs(tk.throttle_state() - tst) <= 1e-6 * np.maximum(1.0, np.abs(tst)))      # (the oracle's estimator saw the oracle's thrusts)
    assert np.array_equal(tk.throttle_state(), cp.throttle_state())


def test_tick_estimator_follows_the_reference_fixture(thr_gold):
    """260 ticks of HoverThrottleEstimator.update as the reference computed them (throttle_golden.npz), driven through the tick:
    vz and throttle handed in, the reference list at a fixed point (t = None: it is not advanced)."""
    import ndp_nmpc_qd_amd as ndp
    T, V = thr_gold["vz"].shape
    eng = ndp.BatchedNMPC(V, load_mlp=False)
    x = np.tile(np.array([0, 0, 1.0, 0, 0, 0, 1, 0, 0, 0]), (V, 1))
    eng.ref_list_fix_pt(x)
    eng.tick_reset()
    eng.throttle_reset()
    for i in range(T):
        cmd, u0, st, _ = eng.tick(x, vz=thr_gold["vz"][i], throttle=thr_gold["throttle"][i], estimate=True, full=True)
        s = eng.throttle_state() if (i % 20 == 0 or i in (50, 55, 60, 200, 201, T - 1)) else None
        if s is not None:
            assert np.all(np.abs(s[:, 1] - thr_gold["k"][i]) <= 1e-12 * np.abs(thr_gold["k"][i]))
            assert np.all(np.abs(s[:, 2:6] - thr_gold["P"][i].reshape(-1, 4)) <= 1e-12 * np.maximum(1.0, np.abs(thr_gold["P"][i].reshape(-1, 4))))
            # nmpc_u_2_att_tgt on the tick's own u0 with the reference's k
            assert np.all(np.abs(cmd[:, 3] - u0[:, 3] * float(thr_gold["mass"]) / thr_gold["k"][i]) <= 1e-12 * np.abs(cmd[:, 3]))
        assert np.array_equal(cmd[:, :3], u0[:, :3])
        if i == 0:   # hover at the fixed point with the reference's u_r[3] = mass * g: c below g, the same for every vehicle
            assert abs(u0[0, 3] - 9.79359713) < 1e-6 and np.allclose(u0, u0[0], rtol=0, atol=1e-12)


def test_tick_throttle_default_is_the_previous_command(gold, flat):
    """throttle = None: the estimator reads the thrust this handle commanded one tick earlier (body_rate_cmd.thrust, nmpc_node.py:253),
    vz = None: column 5 of the odometry."""
    import ndp_nmpc_qd_amd as ndp
    coeff, tseg, cum, fpt = _seq_traj(gold, flat)
    B = coeff.shape[0]
    a, b = ndp.BatchedNMPC(B, load_mlp=False), ndp.BatchedNMPC(B, load_mlp=False)
    for e in (a, b):
        _set_traj(e, coeff, tseg, cum, fpt)
        e.ref_list_reset()
        e.tick_reset()
    rng = np.random.default_rng(3)
    prev = np.zeros(B)
    for i, t in enumerate(flat["seq_t"][:25]):
        x0 = _odometry(rng, a.ref_list_window(None)[0])
        x0[:, 5] = rng.normal(0, 0.3, B)
        tt = np.full(B, float(t))
        ca = a.tick(x0, t=tt, estimate=True)
        cb = b.tick(x0, t=tt, vz=x0[:, 5].copy(), throttle=prev, estimate=True)
        assert np.array_equal(ca, cb)
        prev = cb[:, 3].copy()
    assert np.array_equal(a.throttle_state(), b.throttle_state())
    assert not np.array_equal(a.throttle_state()[:, 6], np.zeros(B))          # the differentiator did run


def test_two_ticks_in_flight_and_misuse(gold, flat):
    import ndp_nmpc_qd_amd as ndp
    coeff, tseg, cum, fpt = _seq_traj(gold, flat, 3)
    B = coeff.shape[0]
    seq, pipe = ndp.BatchedNMPC(B, load_mlp=False), ndp.BatchedNMPC(B, load_mlp=False)
    for e in (seq, pipe):
        _set_traj(e, coeff, tseg, cum, fpt)
        e.ref_list_reset()
        e.tick_reset()
    rng = np.random.default_rng(5)
    xs = [_odometry(rng, seq.ref_list_window(None)[0]) for _ in range(8)]
    ts = [np.full(B, float(t)) for t in flat["seq_t"][:8]]
    want = [seq.tick(x, t=t, estimate=True) for x, t in zip(xs, ts)]
    got = []
    pipe.tick_begin(xs[0], t=ts[0], estimate=True)
    for i in range(1, 8):
        pipe.tick_begin(xs[i], t=ts[i], estimate=True)            # tick i is enqueued while tick i-1 may still run
        got.append(pipe.tick_end())
    with pytest.raises(ndp.NdpError, match="no step in flight|is a tick"):
        pipe.update_end()                                          # a tick is drained with tick_end, not update_end
    got.append(pipe.tick_end())
    for w, g in zip(want, got):
        assert np.array_equal(w, g)
    with pytest.raises(ndp.NdpError, match="no tick in flight"):
        pipe.tick_end()
    pipe.tick_begin(xs[0]); pipe.tick_begin(xs[1])
    with pytest.raises(ndp.NdpError, match="already in flight"):
        pipe.tick_begin(xs[2])
    with pytest.raises(ndp.NdpError, match="in flight"):
        pipe.tick_reset()
    pipe.tick_end(); pipe.tick_end()
    with pytest.raises(ndp.NdpError, match="u0 was not requested"):
        pipe.tick_begin(xs[0])
        pipe.tick_end(full=True)
    # no list yet / neighbours without the NDP model: refused with a message
    fresh = ndp.BatchedNMPC(4, load_mlp=False)
    with pytest.raises(ndp.NdpError, match="no reference list"):
        fresh.tick(np.zeros((4, 10)))
    with pytest.raises(ndp.NdpError, match="use_fd"):
        fresh.tick_config(np.array([1, 0, -1, -1]))
    with pytest.raises(ndp.NdpError, match="outside the handle"):
        ndp.BatchedNMPC(4, disturbance=True).tick_config(np.array([1, 0, 4, -1]))


def test_one_clock_for_all_vehicles_and_the_two_launch_form(gold, flat, monkeypatch):
    """t as a scalar (NDP_TICK_T_UNIFORM: it travels in the kernel arguments) = t as an array of equal entries; and the one-launch
    tick (the control step's wave makes the list's newest entry itself) = the two-launch form (tick_pre_kernel in front), bit for
    bit, across segment boundaries and past the end of a trajectory."""
    import subprocess
    import sys
    import ndp_nmpc_qd_amd as ndp
    coeff, tseg, cum, fpt = _seq_traj(gold, flat, 2)
    B = coeff.shape[0]
    other_index = np.array([(i + 3) % B for i in range(B)], dtype=np.int32)
    a, b = ndp.BatchedNMPC(B, disturbance=True), ndp.BatchedNMPC(B, disturbance=True)
    for e in (a, b):
        _set_traj(e, coeff, tseg, cum, fpt)
        e.ref_list_reset()
        e.tick_config(other_index, gate=True)
        e.tick_reset()
    rng = np.random.default_rng(21)
    t_end = float(cum.max()) + 1.0
    xs, ts, outs = [], [], []
    for i in range(40):
        t = i * t_end / 39                                            # big steps: segment changes of more than one, then past the end
        x0 = _odometry(rng, a.ref_list_window(None)[0])
        ca = a.tick(x0, t=float(t), estimate=True, full=True, raise_on_status=False)
        cb = b.tick(x0, t=np.full(B, t), estimate=True, full=True, raise_on_status=False)
        for u, v in zip(ca, cb):
            assert np.array_equal(u, v)
        xs.append(x0); ts.append(t); outs.append(ca[0])
    Xa, Ua = a.get_iterate()
    wa = a.ref_list_window(None)
    # the same ticks through the two-launch form, in a child process (the form is chosen once per process: NDP_TICK_FORM)
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "ndp_tick_ab_%d.npz" % os.getpid())
    np.savez(path, xs=np.array(xs), ts=np.array(ts), coeff=coeff, tseg=tseg, cum=cum, fpt=fpt, oi=other_index)
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import ndp_nmpc_qd_amd as ndp\n"
        "d = np.load(%r)\n"
        "c = d['coeff']; B = c.shape[0]\n"
        "e = ndp.BatchedNMPC(B, disturbance=True)\n"
        "e.ref_set_trajectory(c[:, :, 0:8], c[:, :, 8:16], c[:, :, 16:24], c[:, :, 24:28], d['cum'], d['tseg'], d['fpt'])\n"
        "e.ref_list_reset(); e.tick_config(d['oi'], gate=True); e.tick_reset()\n"
        "out = [e.tick(x, t=float(t), estimate=True, raise_on_status=False) for x, t in zip(d['xs'], d['ts'])]\n"
        "X, U = e.get_iterate(); w = e.ref_list_window(None)\n"
        "np.savez(%r, out=np.array(out), X=X, U=U, wx=w[0], wu=w[1])\n" % (ROOT, path, path + ".out.npz"))
    env = dict(os.environ, NDP_TICK_FORM="pre")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(path + ".out.npz")
    assert np.array_equal(got["out"], np.array(outs))
    assert np.array_equal(got["X"], Xa) and np.array_equal(got["U"], Ua)
    assert np.array_equal(got["wx"], wa[0]) and np.array_equal(got["wu"], wa[1])
    os.remove(path); os.remove(path + ".out.npz")


def test_tick_device_equals_host_tick(gold, flat):
    import torch
    import ndp_nmpc_qd_amd as ndp
    coeff, tseg, cum, fpt = _seq_traj(gold, flat, 2)
    B = coeff.shape[0]
    other_index = np.array([(i + 1) % B for i in range(B)], dtype=np.int32)
    host, devh = ndp.BatchedNMPC(B, disturbance=True), ndp.BatchedNMPC(B, disturbance=True)
    for e in (host, devh):
        _set_traj(e, coeff, tseg, cum, fpt)
        e.ref_list_reset()
        e.tick_config(other_index, gate=False)                     # gate always open: every vehicle evaluates the network
        e.tick_reset()
    dev = torch.device("cuda", 0)
    cmd_t = torch.empty(B, 4, dtype=torch.float64, device=dev)
    u0_t = torch.empty(B, 4, dtype=torch.float64, device=dev)
    rng = np.random.default_rng(9)
    for i, t in enumerate(flat["seq_t"][:12]):
        x0 = _odometry(rng, host.ref_list_window(None)[0])
        tt = np.full(B, float(t))
        cmd, u0, st, it = host.tick(x0, t=tt, estimate=True, full=True)
        devh.tick_device(torch.from_numpy(x0).to(dev), cmd_t, t=torch.from_numpy(tt).to(dev), estimate=True, u0_out=u0_t)
        devh.synchronize()
        assert np.array_equal(cmd_t.cpu().numpy(), cmd) and np.array_equal(u0_t.cpu().numpy(), u0)
        st_d, it_d = devh.status()
        assert np.array_equal(st_d, st) and np.array_equal(it_d, it)
    assert np.any(host.device_force().cpu().numpy() != 0.0)


def test_tick_long_horizon_goes_through_the_separate_downwash_launch(oracle, gold, flat, mlp_blob):
    """N = 40 (T = 4 s): the network does not fit the fused tile (N + 1 > 32) -- mlp_kernel reads the neighbour's window out of the
    list through the same pitches; the tick still equals the composition and the oracle."""
    import torch
    import ndp_nmpc_qd_amd as ndp
    N = 40
    coeff, tseg, cum, fpt = _seq_traj(gold, flat)
    B = coeff.shape[0]
    other_index = np.array([1, 0, 3, 2, -1], dtype=np.int32)
    tk = ndp.BatchedNMPC(B, N=N, disturbance=True)
    cp = ndp.BatchedNMPC(B, N=N, disturbance=True)
    for e in (tk, cp):
        _set_traj(e, coeff, tseg, cum, fpt)
        e.ref_list_reset()
    tk.tick_config(other_index, gate=True)
    tk.tick_reset()
    xr0, ur0 = cp.ref_list_window(None)
    assert xr0.shape == (B, N + 1, 10)
    cp.reset(xr0, ur0)
    cfg = oracle.default_cfg(use_fd=True, N=N)
    Xo, Uo = xr0.copy(), ur0.copy()
    dev = torch.device("cuda", 0)
    u0_t = torch.empty(B, 4, dtype=torch.float64, device=dev)
    idx_t = torch.from_numpy(other_index).to(dev)
    rng = np.random.default_rng(2)
    for i, t in enumerate(flat["seq_t"][:10]):
        tt = np.full(B, float(t))
        xr, ur = cp.ref_list_window(tt)
        x0 = _odometry(rng, xr)
        cp.update_device(torch.from_numpy(x0).to(dev), torch.from_numpy(xr).to(dev), torch.from_numpy(ur).to(dev), u0_t,
                         other=torch.from_numpy(xr).to(dev), other_index=idx_t, ego_xy=torch.from_numpy(x0[:, 0:2].copy()).to(dev))
        cp.synchronize()
        cmd, u0, st, it = tk.tick(x0, t=tt, full=True)
        assert np.array_equal(u0, u0_t.cpu().numpy())
        nb = np.where(other_index >= 0, other_index, 0)
        f = oracle.downwash_batch(mlp_blob, xr[nb], xr, x0[:, 0:2].copy())
        f[other_index < 0] = 0.0
        u_or, st_o, _ = oracle.step_batch(cfg, x0, xr, ur, f, Xo, Uo)
        assert _rel(u0, u_or) < 1e-5 and not st.any() and not st_o.any()


def test_tick_fixed_point_then_trajectory_like_the_node(oracle, gold, flat):
    """The node's life cycle (nmpc_node.py:87-92,148-152,158-162): hover at gen_fix_pt_ref(odom) -> a trajectory arrives:
    reset the list and the controller -> track it tick by tick -> past its end the list fills with final_pt."""
    import ndp_nmpc_qd_amd as ndp
    coeff, tseg, cum, fpt = _seq_traj(gold, flat)
    B = coeff.shape[0]
    eng = ndp.BatchedNMPC(B, load_mlp=False)
    x_odom = flat["fix_x"].copy()
    eng.ref_list_fix_pt(x_odom)
    xr, ur = eng.ref_list_window(None)
    assert np.array_equal(xr, flat["fix_xr"]) and np.array_equal(ur, flat["fix_ur"])
    eng.tick_reset()
    cfg = oracle.default_cfg()
    X, U = xr.copy(), ur.copy()
    for _ in range(3):
        cmd, u0, st, _ = eng.tick(x_odom, full=True)                 # t = None: the list stays where it is
        u_or, _, _ = oracle.step_batch(cfg, x_odom, xr, ur, None, X, U)
        assert _rel(u0, u_or) < 1e-6
        assert np.allclose(cmd[:, 3], u0[:, 3] * CP.mass / 50.0, rtol=1e-15)
    _set_traj(eng, coeff, tseg, cum, fpt)
    eng.ref_list_reset()
    eng.tick_reset()
    xr, ur = eng.ref_list_window(None)
    np.testing.assert_allclose(xr, flat["seq_xr0"], rtol=0, atol=1e-7)
    X, U = xr.copy(), ur.copy()
    rng = np.random.default_rng(1)
    t_end = float(cum.max()) + 0.5
    # (inputs sit on their bounds on this trajectory: the device's active-set answer is the QP's exact solution, which the oracle's
    # interior-point loop reaches at a tight tolerance -- at its default 1e-8 it stops 1.1e-5 short on one of these ticks)
    cfg.tol = 1e-11
    for i in range(12):
        t = np.full(B, i * t_end / 11)
        x0 = _odometry(rng, xr)
        cmd, u0, st, _ = eng.tick(x0, t=t, full=True)
        xr, ur = eng.ref_list_window(None)
        u_or, _, _ = oracle.step_batch(cfg, x0, xr, ur, None, X, U)
        assert _rel(u0, u_or) < 1e-6 and not st.any()
    assert np.allclose(xr[:, -1, 0:3], fpt) and np.allclose(xr[:, -1, 3:10], [0, 0, 0, 1, 0, 0, 0])    # the newest entry: hover at final_pt


def test_tick_through_every_segment_and_past_the_end_equals_the_composition():
    """260 ticks at 20 ms over a 4 s trajectory of eight 0.5 s segments, batch 203 (a ragged last workgroup), vehicle pairs with the
    gate open for some: the in-launch point takes every path of its segment cache (slot 0, the crossing into slot 1, the re-fill,
    the last segment whose slot 1 is empty, the hover at final_pt past the end, the empty cache of the first tick) -- u0, command,
    status and the list itself equal the call-by-call composition's bit for bit at every tick; the estimator runs on two ticks of
    three; one tick in five does not advance the list (t = None)."""
    import torch
    import ndp_nmpc_qd_amd as ndp
    from ndp_nmpc_qd_amd import synth
    B = 203
    tr = synth.figure_eight_traj(B, seed=5, n_seg=8, t_seg=0.5, pairs=True)
    other_index = (np.arange(B) ^ 1).astype(np.int32)
    other_index[other_index >= B] = -1                    # (the odd one out has no neighbour)
    other_index[7::11] = -1                               # (and a few more: plain NMPC vehicles among the leaders)
    dev = torch.device("cuda", 0)
    tk, cp = ndp.BatchedNMPC(B, disturbance=True), ndp.BatchedNMPC(B, disturbance=True)
    for e in (tk, cp):
        e.ref_set_trajectory(tr["coeff_x"], tr["coeff_y"], tr["coeff_z"], tr["coeff_yaw"], tr["time_cum"], tr["time_seg"], tr["final_pt"])
        e.ref_list_reset()
        e.throttle_reset()
    tk.tick_config(other_index, gate=True)
    tk.tick_reset()
    xr, ur = cp.ref_list_window(None)
    cp.reset(xr, ur)
    idx_t = torch.from_numpy(other_index).to(dev)
    u0_t = torch.empty(B, 4, dtype=torch.float64, device=dev)
    rng = np.random.default_rng(3)
    thrust_prev, k_prev = np.zeros(B), np.full(B, 50.0)
    n_ipm = n_open = 0
    t = 0.0
    for i in range(260):
        adv = i % 5 != 4
        est = i % 3 != 2
        if adv:
            t += 0.02
            xr, ur = cp.ref_list_window(np.full(B, t))
        else:
            xr, ur = cp.ref_list_window(None)
        x0 = _odometry(rng, xr)
        k = cp.throttle_update(x0[:, 5].copy(), thrust_prev) if est else k_prev
        cp.update_device(torch.from_numpy(x0).to(dev), torch.from_numpy(xr).to(dev), torch.from_numpy(ur).to(dev), u0_t,
                         other=torch.from_numpy(xr).to(dev), other_index=idx_t, ego_xy=torch.from_numpy(x0[:, 0:2].copy()).to(dev))
        cp.synchronize()
        u0_c = u0_t.cpu().numpy()
        st_c, it_c = cp.status()
        cmd_c = cp.actuator_cmd(u0_c, k)
        cmd, u0, st, it = tk.tick(x0, t=(t if i % 2 else np.full(B, t)) if adv else None, estimate=est, full=True, raise_on_status=False)
        assert np.array_equal(u0, u0_c), (i, np.max(np.abs(u0 - u0_c)))
        assert np.array_equal(cmd, cmd_c) and np.array_equal(st, st_c) and np.array_equal(it, it_c), i
        n_ipm += int((it > 0).sum())
        n_open += int(np.any(tk.device_force().cpu().numpy() != 0.0, axis=(1, 2)).sum())
        thrust_prev, k_prev = cmd[:, 3].copy(), k
    xt, ut = tk.ref_list_window(None)
    xc, uc = cp.ref_list_window(None)
    assert np.array_equal(xt, xc) and np.array_equal(ut, uc)
    assert np.allclose(xt[:, -1, 0:3], tr["final_pt"])                       # past the end: the newest entries hover at final_pt
    Xt, Ut = tk.get_iterate()
    Xc, Uc = cp.get_iterate()
    assert np.array_equal(Xt, Xc) and np.array_equal(Ut, Uc)
    assert np.array_equal(tk.throttle_state(), cp.throttle_state())
    assert n_open > 1000, n_open                                             # gates were open for a fair share of the leader ticks


def test_neighbours_in_other_workgroups_at_a_batch_larger_than_the_device():
    """the one-launch tick reads its NEIGHBOUR's segment-cache record at entry while the neighbour's own wave may
    re-fill that record in the same launch (it does whenever the neighbour crosses into its next segment) -- with the neighbour in
    another workgroup nothing orders the two, and at a batch of several rounds of workgroups the reader can start after the writer.
    The cache now has two copies (read one, write the other, swapped between ticks).  Here: 4096 vehicles (four rounds of 1024 waves),
    every neighbour 1031 vehicles away (another workgroup, another XCD, usually another round), segments of 0.25 s (a crossing every
    12.5 ticks), 40 ticks -- the tick against the composition of the stand-alone calls (which never touches the cache), bit for bit."""
    import torch
    import ndp_nmpc_qd_amd as ndp
    dev = torch.device("cuda", 0)
    B = 4096
    tr = synth.figure_eight_traj(B, seed=5, n_seg=16, t_seg=0.25)
    other_index = ((np.arange(B) + 1031) % B).astype(np.int32)
    tk, cp = ndp.BatchedNMPC(B, disturbance=True), ndp.BatchedNMPC(B, disturbance=True)
    for e in (tk, cp):
        e.ref_set_trajectory(tr["coeff_x"], tr["coeff_y"], tr["coeff_z"], tr["coeff_yaw"], tr["time_cum"], tr["time_seg"], tr["final_pt"])
        e.ref_list_reset()
    tk.tick_config(other_index, gate=False)               # every gate open: the neighbour's node N feeds the network on every vehicle
    tk.tick_reset()
    xr, ur = cp.ref_list_window(None)
    cp.reset(xr, ur)
    idx_t = torch.from_numpy(other_index).to(dev)
    u0_t = torch.empty(B, 4, dtype=torch.float64, device=dev)
    rng = np.random.default_rng(9)
    for i in range(40):
        t = 0.02 * (i + 1)
        xr, ur = cp.ref_list_window(np.full(B, t))
        x0 = xr[:, 0, :].copy()
        x0[:, 0:3] += rng.normal(0.0, 0.03, size=(B, 3))
        cp.update_device(torch.from_numpy(x0).to(dev), torch.from_numpy(xr).to(dev), torch.from_numpy(ur).to(dev), u0_t,
                         other=torch.from_numpy(xr).to(dev), other_index=idx_t)
        cp.synchronize()
        cmd, u0, st, _ = tk.tick(x0, t=t, estimate=False, full=True, raise_on_status=False)
        assert np.array_equal(u0, u0_t.cpu().numpy()), (i, float(np.max(np.abs(u0 - u0_t.cpu().numpy()))))
        assert np.array_equal(tk.device_force().cpu().numpy(), cp.device_force().cpu().numpy()), i
    xt, ut = tk.ref_list_window(None)
    xc, uc = cp.ref_list_window(None)
    assert np.array_equal(xt, xc) and np.array_equal(ut, uc)


def test_tick_with_neighbours_on_another_rank_equals_the_single_handle_tick():
    """the control tick when a vehicle's neighbour lives on ANOTHER rank (nmpc_node.py:116-133,229-230 -> the leader's
    subscriber, ndp_nmpc_leader_node.py:40,60-76).  Two handles of 64 vehicles each stand for two ranks; every vehicle's neighbour is on
    the other one.  Per tick each "rank": ndp_tick_advance_device -> ndp_tick_window_pv_device into its half of the gathered buffer
    (= what one all-gather of [B, N+1, 6] delivers) -> ndp_tick_step_device against the gathered windows.  Held bit for bit against ONE
    handle with all 128 vehicles ticking through the one-launch ndp_tick_device -- u0, actuator command, predicted force, the lists --
    over 30 ticks with segment crossings, estimator on every other tick; and against the same three-stage form on one handle
    (one rank, its own windows as the gathered buffer)."""
    import torch
    import ndp_nmpc_qd_amd as ndp
    dev = torch.device("cuda", 0)
    Bh, B = 64, 128
    tr = synth.figure_eight_traj(B, seed=6, n_seg=16, t_seg=0.25, pairs=True)
    oi_all = ((np.arange(B) + Bh) % B).astype(np.int32)              # vehicle i's neighbour: i + 64 (the other "rank")
    # vehicle i and i + 64 fly close to each other: copy the pair structure of the first half onto the second
    for k in ("coeff_x", "coeff_y", "coeff_z", "coeff_yaw", "time_cum", "time_seg"):
        tr[k][Bh:] = tr[k][:Bh]
    tr["coeff_z"][Bh:, 0::8] += 0.6                                  # 0.6 m above: the gate is open, the force is not zero
    tr["final_pt"][Bh:] = tr["final_pt"][:Bh] + np.array([0.0, 0.0, 0.6])

    def traj(e, sl):
        e.ref_set_trajectory(*(tr[k][sl] for k in ("coeff_x", "coeff_y", "coeff_z", "coeff_yaw", "time_cum", "time_seg", "final_pt")))
        e.ref_list_reset()
        e.throttle_reset()

    one = ndp.BatchedNMPC(B, disturbance=True)
    traj(one, slice(None))
    one.tick_config(oi_all, gate=True)
    one.tick_reset()
    ranks = [ndp.BatchedNMPC(Bh, disturbance=True) for _ in range(2)]
    gathered = torch.zeros(B, 21, 6, dtype=torch.float64, device=dev)
    for r, e in enumerate(ranks):
        traj(e, slice(r * Bh, (r + 1) * Bh))
        e.tick_config_remote(gathered, oi_all[r * Bh:(r + 1) * Bh], gate=True)        # rows of the gathered buffer = global vehicle ids
        e.tick_reset()
    solo = ndp.BatchedNMPC(B, disturbance=True)                      # one rank, three stages
    traj(solo, slice(None))
    own = torch.zeros(B, 21, 6, dtype=torch.float64, device=dev)
    solo.tick_config_remote(own, oi_all, gate=True)
    solo.tick_reset()
    with pytest.raises(ndp.NdpError, match="ndp_tick_advance_device"):
        solo.tick(np.zeros((B, 10)), t=0.02)                         # the one-call tick is refused while neighbours are remote
    rng = np.random.default_rng(4)
    cmd1, u1 = torch.empty(B, 4, dtype=torch.float64, device=dev), torch.empty(B, 4, dtype=torch.float64, device=dev)
    cmd2, u2 = torch.empty(B, 4, dtype=torch.float64, device=dev), torch.empty(B, 4, dtype=torch.float64, device=dev)
    cmd3, u3 = torch.empty(B, 4, dtype=torch.float64, device=dev), torch.empty(B, 4, dtype=torch.float64, device=dev)
    n_force = 0
    for i in range(30):
        t = 0.02 * (i + 1)
        est = i % 2 == 0
        xr, _ = one.ref_list_window(None)
        x0 = xr[:, 1, :].copy()                                      # near the window the tick will use
        x0[:, 0:3] += rng.normal(0.0, 0.03, size=(B, 3))
        x0_t = torch.from_numpy(x0).to(dev)
        one.tick_device(x0_t, cmd1, t=t, estimate=est, u0_out=u1)
        for r, e in enumerate(ranks):
            sl = slice(r * Bh, (r + 1) * Bh)
            e.tick_advance_device(x0_t[sl].contiguous(), t=t, estimate=est)
            e.tick_window_pv_device(gathered[sl])                    # (each rank writes its rows: the all-gather's result)
        torch.cuda.synchronize()
        for r, e in enumerate(ranks):
            sl = slice(r * Bh, (r + 1) * Bh)
            e.tick_step_device(x0_t[sl].contiguous(), cmd2[sl], u0_out=u2[sl])
        solo.tick_advance_device(x0_t, t=t, estimate=est)
        solo.tick_window_pv_device(own)
        solo.tick_step_device(x0_t, cmd3, u0_out=u3)
        torch.cuda.synchronize()
        for a, b_ in ((u1, u2), (cmd1, cmd2), (u1, u3), (cmd1, cmd3)):
            assert torch.equal(a, b_), (i, float((a - b_).abs().max()))
        f1 = one.device_force().cpu().numpy()
        f2 = np.concatenate([e.device_force().cpu().numpy() for e in ranks])
        assert np.array_equal(f1, f2)
        n_force += int(np.any(f1 != 0.0, axis=(1, 2)).sum())
    assert n_force > 30 * B // 4                                     # the gates were open: the exchanged windows fed the network
    x1, w1 = one.ref_list_window(None)
    x2 = np.concatenate([e.ref_list_window(None)[0] for e in ranks])
    assert np.array_equal(x1, x2)
    assert np.array_equal(one.throttle_state(), np.concatenate([e.throttle_state() for e in ranks]))


@pytest.mark.gpu
def test_tick_windows_through_the_library_collective_equals_the_single_handle_tick():
    """ndp_xchg_tick_windows: stage 2 of the remote tick and the exchange in one call (pack out of the list + ncclAllGather on the
    tick's stream), here with a one-rank communicator (a real ncclAllGather, the neighbour "rank" is the rank itself).  Bit-equal
    with the one-launch tick of one handle over 12 ticks; the gathered rows are the list window's position / velocity columns."""
    import torch
    import ndp_nmpc_qd_amd as ndp
    from ndp_nmpc_qd_amd import dist as ndist
    dev = torch.device("cuda", 0)
    B = 128
    tr = synth.figure_eight_traj(B, seed=9, n_seg=16, t_seg=0.25, pairs=True)
    oi = (np.arange(B) ^ 1).astype(np.int32)

    def make():
        e = ndp.BatchedNMPC(B, disturbance=True)
        e.ref_set_trajectory(*(tr[k] for k in ("coeff_x", "coeff_y", "coeff_z", "coeff_yaw", "time_cum", "time_seg", "final_pt")))
        e.ref_list_reset()
        e.throttle_reset()
        return e
    one, rem = make(), make()
    one.tick_config(oi, gate=True)
    one.tick_reset()
    gathered = torch.zeros(B, 21, 6, dtype=torch.float64, device=dev)
    rem.tick_config_remote(gathered, oi, gate=True)
    rem.tick_reset()
    try:
        ex = ndist.RcclExchange(B, 20, 0)
    except RuntimeError as e:
        pytest.skip(f"RCCL could not be bound: {e}")
    stream = torch.cuda.Stream(device=dev)
    cmd1, u1 = torch.empty(B, 4, dtype=torch.float64, device=dev), torch.empty(B, 4, dtype=torch.float64, device=dev)
    cmd2, u2 = torch.empty(B, 4, dtype=torch.float64, device=dev), torch.empty(B, 4, dtype=torch.float64, device=dev)
    rng = np.random.default_rng(5)
    n_force = 0
    for i in range(12):
        t = 0.02 * (i + 1)
        xr, _ = one.ref_list_window(None)
        x0 = xr[:, 1, :].copy()
        x0[:, 0:3] += rng.normal(0.0, 0.03, size=(B, 3))
        x0_t = torch.from_numpy(x0).to(dev)
        torch.cuda.synchronize()
        one.tick_device(x0_t, cmd1, t=t, estimate=True, u0_out=u1)
        rem.tick_advance_device(x0_t, t=t, estimate=True, stream=stream)
        ex.tick_windows(rem, gathered, stream)
        rem.tick_step_device(x0_t, cmd2, u0_out=u2, stream=stream)
        torch.cuda.synchronize()
        assert torch.equal(u1, u2) and torch.equal(cmd1, cmd2), i
        w = rem.ref_list_window(None)[0]
        assert np.array_equal(gathered.cpu().numpy(), w[:, :, 0:6])
        n_force += int(np.any(one.device_force().cpu().numpy() != 0.0, axis=(1, 2)).sum())
    assert n_force > 12 * B // 4
    ex.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tracked,asyn,nbuf", [(True, False, 2), (False, False, 2), (True, True, 2), (True, False, 3), (True, True, 3), (False, True, 3)])
def test_remote_tick_with_the_exchange_one_period_ahead_equals_the_single_handle_tick(tracked, asyn, nbuf):
    """ndp_xchg_tick_begin / _step: the list advance, window columns and all-gather of tick i+1 on the exchange's stream beside the
    control step of tick i (two gather buffers), ordered by events on the device -- with tracked steps the gather waits for exactly the
    step that read its buffer last, without them for that step's stream.  One-rank communicator (a real ncclAllGather).  Bit-equal with
    the one-launch tick of one handle over 14 ticks, estimator on every tick; the protocol's misuse is refused.  asyn: the begins'
    launches made by the exchange's own thread (ndp_xchg_tick_async).  nbuf = 3: the begin of tick i + 2 behind the step of tick i,
    into a third buffer (the gather then never waits for a control step)."""
    import torch
    import ndp_nmpc_qd_amd as ndp
    from ndp_nmpc_qd_amd import dist as ndist
    dev = torch.device("cuda", 0)
    B = 128
    tr = synth.figure_eight_traj(B, seed=11, n_seg=16, t_seg=0.25, pairs=True)
    oi = (np.arange(B) ^ 1).astype(np.int32)

    def make():
        e = ndp.BatchedNMPC(B, disturbance=True)
        e.ref_set_trajectory(*(tr[k] for k in ("coeff_x", "coeff_y", "coeff_z", "coeff_yaw", "time_cum", "time_seg", "final_pt")))
        e.ref_list_reset()
        e.throttle_reset()
        return e
    one, rem = make(), make()
    one.tick_config(oi, gate=True)
    one.tick_reset()
    gathered = [torch.zeros(B, 21, 6, dtype=torch.float64, device=dev) for _ in range(nbuf)]
    rem.tick_config_remote(gathered[0], oi, gate=True)
    rem.tick_reset()
    if tracked:
        rem.track_steps(True)
    try:
        ex = ndist.RcclExchange(B, 20, 0)
    except RuntimeError as e:
        pytest.skip(f"RCCL could not be bound: {e}")
    if asyn:
        ex.tick_async(True)
    stream = torch.cuda.Stream(device=dev)
    cmd1, u1 = torch.empty(B, 4, dtype=torch.float64, device=dev), torch.empty(B, 4, dtype=torch.float64, device=dev)
    cmd2 = [torch.empty(B, 4, dtype=torch.float64, device=dev) for _ in range(14)]
    u2 = [torch.empty(B, 4, dtype=torch.float64, device=dev) for _ in range(14)]
    rng = np.random.default_rng(6)
    n = 14
    with pytest.raises(RuntimeError, match="no gather was begun"):
        ex.tick_step(rem, torch.zeros(B, 10, dtype=torch.float64, device=dev), cmd1, gathered[0], stream)
    # the odometry of every tick up front (the one-handle engine's windows: the same list)
    xs, ref1 = [], []
    for i in range(n):
        t = 0.02 * (i + 1)
        xr, _ = one.ref_list_window(None)
        x0 = xr[:, 1, :].copy()
        x0[:, 0:3] += rng.normal(0.0, 0.03, size=(B, 3))
        x0_t = torch.from_numpy(x0).to(dev)
        xs.append(x0_t)
        one.tick_device(x0_t, cmd1, t=t, estimate=True, u0_out=u1)
        torch.cuda.synchronize()
        ref1.append((cmd1.clone(), u1.clone(), one.device_force().clone()))
    # the pipelined remote form: no host synchronisation inside the loop; tick i's trajectory time is 0.02 (i + 1)
    la = nbuf - 1                              # begins ahead of the steps: 1 (two buffers) or 2 (three)
    for i in range(la):
        ex.tick_begin(rem, gathered[i % nbuf], t=0.02 * (i + 1))
    for i in range(n):
        ex.tick_step(rem, xs[i], cmd2[i], gathered[i % nbuf], stream, estimate=True, u0_out=u2[i])
        if i + la < n:
            ex.tick_begin(rem, gathered[(i + la) % nbuf], t=0.02 * (i + la + 1))
    torch.cuda.synchronize()
    for i in range(n):
        assert torch.equal(cmd2[i], ref1[i][0]) and torch.equal(u2[i], ref1[i][1]), i
    assert np.array_equal(rem.throttle_state(), one.throttle_state())
    # two gathers may be ahead of the steps, not three; and not two into one buffer
    ex.tick_begin(rem, gathered[0], t=None)
    with pytest.raises(RuntimeError, match="not been stepped on"):
        ex.tick_begin(rem, gathered[0], t=None)
    ex.tick_begin(rem, gathered[1], t=None)
    with pytest.raises(RuntimeError, match="already ahead"):
        ex.tick_begin(rem, gathered[0], t=None)
    if asyn:
        with pytest.raises(RuntimeError, match="ahead of the control steps"):     # (not while gathers are ahead)
            ex.tick_async(False)
    torch.cuda.synchronize()
    ex.close()


@pytest.mark.gpu
def test_tick_with_inputs_on_their_bounds_one_launch_equals_three_stages():
    """Large odometry errors (1 m, 2 m/s): a third of the vehicles have inputs on their bounds, kept sets grow and shrink from tick to tick.
    The one-launch tick (rti_kernel<..., TICK>) and the three-stage form (list kernel + plain fused control step) run the same active-set
    iterations: u0, command, sweeps, interior-point iterations and kept sets bit-equal over 10 ticks, no instance unsolved."""
    import torch
    import ndp_nmpc_qd_amd as ndp
    dev = torch.device("cuda", 0)
    B = 256
    tr = synth.figure_eight_traj(B, seed=13, n_seg=16, t_seg=0.25, pairs=True)
    oi = (np.arange(B) ^ 1).astype(np.int32)

    def make():
        e = ndp.BatchedNMPC(B, disturbance=True)
        e.ref_set_trajectory(*(tr[k] for k in ("coeff_x", "coeff_y", "coeff_z", "coeff_yaw", "time_cum", "time_seg", "final_pt")))
        e.ref_list_reset()
        e.throttle_reset()
        return e
    one, three = make(), make()
    one.tick_config(oi, gate=True)
    one.tick_reset()
    own = torch.zeros(B, 21, 6, dtype=torch.float64, device=dev)
    three.tick_config_remote(own, oi, gate=True)
    three.tick_reset()
    cmd1, u1 = torch.empty(B, 4, dtype=torch.float64, device=dev), torch.empty(B, 4, dtype=torch.float64, device=dev)
    cmd2, u2 = torch.empty(B, 4, dtype=torch.float64, device=dev), torch.empty(B, 4, dtype=torch.float64, device=dev)
    rng = np.random.default_rng(8)
    n_con = 0
    for i in range(10):
        t = 0.02 * (i + 1)
        xr, _ = one.ref_list_window(None)
        x0 = xr[:, 1, :].copy()
        big = rng.random(B) < (0.6 if i < 5 else 0.1)                    # the disturbance dies down: sets empty again
        x0[:, 0:3] += rng.normal(0.0, 1.0, size=(B, 3)) * big[:, None] + rng.normal(0.0, 0.03, size=(B, 3))
        x0[:, 3:6] += rng.normal(0.0, 2.0, size=(B, 3)) * big[:, None]
        x0_t = torch.from_numpy(x0).to(dev)
        one.tick_device(x0_t, cmd1, t=t, estimate=True, u0_out=u1)
        three.tick_advance_device(x0_t, t=t, estimate=True)
        three.tick_window_pv_device(own)
        three.tick_step_device(x0_t, cmd2, u0_out=u2)
        torch.cuda.synchronize()
        assert torch.equal(u1, u2) and torch.equal(cmd1, cmd2), i
        (st1, it1), (st2, it2) = one.status(), three.status()
        (sw1, a1), (sw2, a2) = one.active_set(), three.active_set()
        assert not st1.any() and np.array_equal(st1, st2) and np.array_equal(it1, it2) and np.array_equal(sw1, sw2) and np.array_equal(a1, a2), i
        n_con += int(a1.any(axis=(1, 2)).sum())
    assert n_con > B
