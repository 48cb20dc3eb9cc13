"""reference window generation (polynomial trajectory -> flatness -> xr/ur windows).
ref_golden.npz: coefficients and trajectory points produced by the reference's own PolymOptimizer
(tests/golden/make_ref_golden.py).  flat_golden.npz: the flatness map (R_wb, body rates, collective force) and the
101-entry sliding list of NMPCRefPublisher (windows of 60 control ticks, the start-up duplicate, gen_fix_pt_ref) produced by
RUNNING the reference's pt_pub/pt_publisher.py under stand-ins for the ROS modules (tests/golden/make_flatness_golden.py).
The physical identities of the first round stay as a second, independent pin."""
import os

import numpy as np
import pytest

from ndp_nmpc_qd_amd.params import nmpc_params as CP
from ndp_nmpc_qd_amd.pt_pub import MinMethod, PolymOptimizer, TrajCoefficients

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NCASE = 4


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_golden.npz"))


def _cum(tseg):
    return np.concatenate([np.zeros((tseg.shape[0], 1)), np.cumsum(tseg, axis=1)], axis=1)


def test_host_optimizer_matches_reference_coefficients(gold):
    for c in range(NCASE):
        wpts, tseg, coeff = gold[f"wpts_{c}"], gold[f"tseg_{c}"], gold[f"coeff_{c}"]
        V, M = tseg.shape
        for v in range(V):
            for a in range(3):
                got = PolymOptimizer(MinMethod.SNAP).get_coeff(wpts[v, a]).reshape(M, 8)
                np.testing.assert_allclose(got, coeff[v, :, 8 * a:8 * a + 8], rtol=1e-8, atol=1e-8 * np.abs(coeff[v]).max())
            got = PolymOptimizer(MinMethod.ACCEL).get_coeff(wpts[v, 3]).reshape(M, 4)
            np.testing.assert_allclose(got, coeff[v, :, 24:28], rtol=1e-9, atol=1e-10)
        tc = TrajCoefficients.from_waypoints(wpts, tseg)          # the batched solve gives the same coefficients
        np.testing.assert_allclose(tc.coeff_x.reshape(V, M, 8), coeff[:, :, 0:8], rtol=1e-8, atol=1e-8 * np.abs(coeff).max())
        np.testing.assert_allclose(tc.coeff_yaw.reshape(V, M, 4), coeff[:, :, 24:28], rtol=1e-9, atol=1e-10)
        np.testing.assert_array_equal(tc.traj_time_cum, _cum(tseg))


def test_get_poly_params_known_answers():
    o = PolymOptimizer(MinMethod.SNAP)
    np.testing.assert_array_equal(o.get_poly_params(0, 1.0), np.ones((1, 8)))
    np.testing.assert_array_equal(o.get_poly_params(2, 1.0), [[0, 0, 2, 6, 12, 20, 30, 42]])
    np.testing.assert_array_equal(o.get_poly_params(1, 0.0), [[0, 1, 0, 0, 0, 0, 0, 0]])
    np.testing.assert_allclose(o.get_poly_params(3, 0.5), [[0, 0, 0, 6, 12, 15, 15, 13.125]])


def test_oracle_trajectory_points_against_reference_fixture(oracle, gold):
    for c in range(NCASE):
        tseg, coeff, tq = gold[f"tseg_{c}"], gold[f"coeff_{c}"], gold[f"tq_{c}"]
        cum = _cum(tseg)
        for v in range(tseg.shape[0]):
            for s in range(tq.shape[1]):
                pvaj, yaw = oracle.traj_point(coeff[v], cum[v], tseg[v], np.zeros(3), tq[v, s])
                scale = 1.0 + np.abs(gold[f"pvaj_{c}"][v, s])
                assert np.max(np.abs(pvaj - gold[f"pvaj_{c}"][v, s]) / scale) < 1e-11
                np.testing.assert_allclose(yaw, gold[f"yaw_{c}"][v, s], rtol=1e-11, atol=1e-12)


def _rot(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def test_oracle_flatness_identities(oracle, gold):
    """x, u from the flatness map must (i) carry the trajectory's p, v; (ii) have a unit quaternion with w > 0 whose
    thrust axis times c reproduces a + g e3 (the OCP's own dynamics, nmpc_body_rate_ctl.py:151-153); (iii) have
    heading yaw; (iv) have body rates that reproduce the time derivative of the thrust axis (finite differences along
    the reference's own polynomial) and r = yaw_dot * z_b[2] (pt_publisher.py:225)."""  # noqa: D401
    c = 3
    tseg, coeff = gold[f"tseg_{c}"], gold[f"coeff_{c}"]
    cum = _cum(tseg)
    rng = np.random.default_rng(5)
    for v in range(tseg.shape[0]):
        for t in rng.uniform(0.2, cum[v, -1] - 0.2, 12):
            pvaj, yaw = oracle.traj_point(coeff[v], cum[v], tseg[v], np.zeros(3), t)
            x, u = oracle.diff_flatness(pvaj, yaw)
            np.testing.assert_array_equal(x[0:6], pvaj[0:6])
            q = x[6:10]
            R = _rot(q)
            # tf's quaternion_from_matrix only guarantees w > 0 on its trace branch (the sign is not normalised)
            assert abs(np.linalg.norm(q) - 1) < 1e-12 and (q[0] > 0 or np.trace(R) <= 0)
            np.testing.assert_allclose(R[:, 2] * u[3], pvaj[6:9] + np.array([0, 0, CP.gravity]), atol=1e-10)
            # y_b is orthogonal to the heading direction x_c and x_b has a positive component along it (:208-217)
            xc = np.array([np.cos(yaw[0]), np.sin(yaw[0]), 0.0])
            assert abs(R[:, 1] @ xc) < 1e-12 and R[:, 0] @ xc > 0
            h = 1e-5
            zs = []
            for tt in (t - h, t + h):
                pv, yw = oracle.traj_point(coeff[v], cum[v], tseg[v], np.zeros(3), tt)
                xx, _ = oracle.diff_flatness(pv, yw)
                zs.append(_rot(xx[6:10])[:, 2])
            zdot = (zs[1] - zs[0]) / (2 * h)
            # z_b' = omega x z_b with omega = R [p, q, r]: z_b' = q x_b - p y_b
            np.testing.assert_allclose(zdot, u[1] * R[:, 0] - u[0] * R[:, 1], atol=2e-6)
            assert abs(u[2] - yaw[1] * R[2, 2]) < 1e-12


def test_oracle_quaternion_branches_and_hover_after_the_end(oracle):
    # inverted / sideways thrust directions exercise the non-trace branches of quaternion_from_matrix
    for acc, yw in (([0, 0, -30.0], 0.3), ([25.0, 0, -9.81], 2.0), ([0, -25.0, -9.81], -1.0), ([0, 0, 0.0], 3.1)):
        pvaj = np.zeros(12); pvaj[6:9] = acc
        x, u = oracle.diff_flatness(pvaj, np.array([yw, 0.0]))
        R = _rot(x[6:10])
        assert abs(np.linalg.norm(x[6:10]) - 1) < 1e-12
        np.testing.assert_allclose(R[:, 2] * u[3], np.array(acc) + np.array([0, 0, CP.gravity]), atol=1e-9)
    # after the end: final_pt, identity attitude, zero rates, c = g  (base_pt_publisher.py:93-94 + flatness of a rest point)
    coeff = np.random.default_rng(0).normal(size=(2, 28))
    pvaj, yaw = oracle.traj_point(coeff, np.array([0, 1.0, 3.0]), np.array([1.0, 2.0]), np.array([4.0, 5.0, 6.0]), 3.0)
    x, u = oracle.diff_flatness(pvaj, yaw)
    np.testing.assert_allclose(x, [4, 5, 6, 0, 0, 0, 1, 0, 0, 0], atol=1e-15)
    np.testing.assert_allclose(u, [0, 0, 0, CP.gravity], atol=1e-15)


def test_oracle_window_node_spacing(oracle, gold):
    c = 2
    tseg, coeff = gold[f"tseg_{c}"], gold[f"coeff_{c}"]
    cum, V = _cum(tseg), tseg.shape[0]
    fpt = gold[f"wpts_{c}"][:, 0:3, -1]
    t0 = np.linspace(0.0, 1.0, V) * cum[:, -1]
    xr, ur = oracle.ref_window(coeff, cum, tseg, fpt, t0)
    assert xr.shape == (V, 21, 10) and ur.shape == (V, 20, 4)
    for v in range(V):
        for k in (0, 7, 20):
            pvaj, yaw = oracle.traj_point(coeff[v], cum[v], tseg[v], fpt[v], t0[v] + k * CP.th_pred)
            x, u = oracle.diff_flatness(pvaj, yaw)
            np.testing.assert_array_equal(xr[v, k], x)
            if k < 20:
                np.testing.assert_array_equal(ur[v, k], u)
    np.testing.assert_allclose(xr[-1, :, 0:3], np.broadcast_to(fpt[-1], (21, 3)))   # the last vehicle starts at its end


@pytest.mark.gpu
def test_gpu_reference_window_against_oracle_and_fixture(oracle, gold):
    import ndp_nmpc_qd_amd as ndp
    from ndp_nmpc_qd_amd.pt_pub import BatchedNMPCRefPublisher
    for c in range(NCASE):
        wpts, tseg, coeff, tq = gold[f"wpts_{c}"], gold[f"tseg_{c}"], gold[f"coeff_{c}"], gold[f"tq_{c}"]
        V, M = tseg.shape
        cum = _cum(tseg)
        eng = ndp.BatchedNMPC(V, load_mlp=False)
        pub = BatchedNMPCRefPublisher(eng)
        tc = TrajCoefficients(coeff[:, :, 0:8], coeff[:, :, 8:16], coeff[:, :, 16:24], coeff[:, :, 24:28], cum, tseg,
                              wpts[:, 0:3, -1].copy())
        pub.reset(tc)
        for s in range(0, tq.shape[1], 5):
            xr, ur = pub.get_nmpc_pts_direct(tq[:, s])
            xo, uo = oracle.ref_window(coeff, cum, tseg, tc.final_pt, tq[:, s])
            np.testing.assert_allclose(xr, xo, rtol=1e-10, atol=1e-10)
            np.testing.assert_allclose(ur, uo, rtol=1e-9, atol=1e-9)
            # node 0 carries the reference's own trajectory point (fixture)
            np.testing.assert_allclose(xr[:, 0, 0:6], gold[f"pvaj_{c}"][:, s, 0:6], rtol=1e-10, atol=1e-10)
        assert np.all(pub.is_activated(0.0)) and not np.any(pub.is_activated(cum[:, -1] + 2.5, t_pred=2.0))


@pytest.mark.gpu
def test_gpu_reference_window_feeds_the_controller(oracle):
    """reference generation -> control step, both on the device, 1024 vehicles, windows never leave HBM."""
    import torch
    import ndp_nmpc_qd_amd as ndp
    B, M = 1024, 4
    rng = np.random.default_rng(7)
    wp = np.zeros((B, 4, M + 1))
    wp[:, 0:2] = np.cumsum(rng.uniform(-1.0, 1.0, (B, 2, M + 1)), axis=2)
    wp[:, 2] = 1.0 + 0.2 * rng.uniform(-1, 1, (B, M + 1))
    wp[:, 3] = np.cumsum(rng.uniform(-0.3, 0.3, (B, M + 1)), axis=1)
    tseg = rng.uniform(3.0, 5.0, (B, M))
    tc = TrajCoefficients.from_waypoints(wp, tseg)
    eng = ndp.BatchedNMPC(B)
    eng.ref_set_trajectory(tc.coeff_x, tc.coeff_y, tc.coeff_z, tc.coeff_yaw, tc.traj_time_cum, tc.traj_time_seg, tc.final_pt)
    dev = torch.device("cuda", 0)
    t = torch.full((B,), 1.0, dtype=torch.float64, device=dev)
    xr = torch.empty(B, 21, 10, dtype=torch.float64, device=dev)
    ur = torch.empty(B, 20, 4, dtype=torch.float64, device=dev)
    u0 = torch.empty(B, 4, dtype=torch.float64, device=dev)
    eng.ref_window_device(t, xr, ur)
    eng.reset_device(xr, ur)
    x0 = xr[:, 0, :].clone()
    x0[:, 0:3] += 0.05
    eng.update_device(x0, xr, ur, u0)
    eng.synchronize()
    xo, uo = oracle.ref_window(np.concatenate([tc.coeff_x.reshape(B, M, 8), tc.coeff_y.reshape(B, M, 8), tc.coeff_z.reshape(B, M, 8),
                                               tc.coeff_yaw.reshape(B, M, 4)], axis=2), tc.traj_time_cum, tseg, tc.final_pt, np.full(B, 1.0))
    np.testing.assert_allclose(xr.cpu().numpy(), xo, rtol=1e-10, atol=1e-10)
    cfg = oracle.default_cfg()
    X, U = xo.copy(), uo.copy()
    u_or, st, _ = oracle.step_batch(cfg, x0.cpu().numpy(), xo, uo, None, X, U)
    assert np.max(np.abs(u0.cpu().numpy() - u_or) / np.maximum(1.0, np.abs(u_or))) < 1e-5


# ------------------------------------------------------------------------------------------- goldens from the reference's own pt_publisher.py
@pytest.fixture(scope="module")
def flat():
    return np.load(os.path.join(ROOT, "tests", "golden", "flat_golden.npz"))


def test_oracle_flatness_against_the_reference_code(oracle, flat):
    """diff_flatness (pt_publisher.py:188-248) as the reference computed it: R_wb is the matrix the reference's numpy code
    handed to quaternion_from_matrix, p/q/r and collective_force are read off its TrajFullStatePt.  The oracle's quaternion
    is checked THROUGH R_wb (R(q) = R_wb, unit norm), so the test does not lean on the stand-in's quaternion routine."""
    for p, y, R, w, F, q_xyzw in zip(flat["flat_pvaj"], flat["flat_yaw"], flat["flat_R"], flat["flat_rates"], flat["flat_force"], flat["flat_q"]):
        x, u = oracle.diff_flatness(p, y)
        np.testing.assert_allclose(_rot(x[6:10]), R, rtol=0, atol=1e-12)
        assert abs(np.linalg.norm(x[6:10]) - 1) < 1e-13
        np.testing.assert_allclose(u[0:3], w, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(u[3] * CP.mass, F, rtol=1e-13)
        np.testing.assert_array_equal(x[0:6], p[0:6])
        # same branch and sign as ROS geometry's quaternion_from_matrix (restated in the generator): [x, y, z, w]
        np.testing.assert_allclose(x[[7, 8, 9, 6]], q_xyzw, rtol=0, atol=1e-13)


def _seq_traj(gold, flat):
    c = int(flat["seq_case"])
    coeff, tseg, wpts = gold[f"coeff_{c}"], flat["seq_tseg"], gold[f"wpts_{c}"]
    return coeff, tseg, _cum(tseg), wpts[:, 0:3, -1].copy()


def test_sliding_list_semantics_against_the_reference_code(oracle, gold, flat):
    """NMPCRefPublisher's list (pt_publisher.py:57-103) replayed with the oracle's point evaluation: after reset the list
    holds the points at i * 0.02 s (i = 0..99) with the first one duplicated in front; every get_nmpc_pts drops the
    oldest and appends the point at t + T_horizon; the window is every 5th entry.  60 ticks with timer jitter, one
    trajectory ending inside the sequence (hover at final_pt afterwards)."""
    coeff, tseg, cum, fpt = _seq_traj(gold, flat)
    V = coeff.shape[0]

    def point(v, t):
        return np.concatenate(oracle.diff_flatness(*oracle.traj_point(coeff[v], cum[v], tseg[v], fpt[v], t)))
    for v in range(V):
        lst = [point(v, i * CP.ts_nmpc) for i in range(CP.long_list_size - 1)]
        lst.insert(0, lst[0])
        win = np.array(lst[::5])
        np.testing.assert_allclose(win[:, :10], flat["seq_xr0"][v], rtol=0, atol=1e-7)
        np.testing.assert_allclose(win[:-1, 10:], flat["seq_ur0"][v], rtol=0, atol=1e-7)
        for k, t in enumerate(flat["seq_t"]):
            lst.pop(0)
            lst.append(point(v, t + CP.T_horizon))
            win = np.array(lst[::5])
            np.testing.assert_allclose(win[:, :10], flat["seq_xr"][k, v], rtol=0, atol=1e-7)
            np.testing.assert_allclose(win[:-1, 10:], flat["seq_ur"][k, v], rtol=0, atol=1e-7)
    # the start-up duplicate is visible in the first windows: node 1 is 0.08 s after node 0, not 0.1 s
    assert np.allclose(flat["seq_xr0"][:, 0], flat["seq_xr0"][:, 0]) and not np.allclose(flat["seq_xr0"][1, 1, 0:3], point(1, 0.1)[0:3], atol=1e-6)
    assert np.allclose(flat["seq_xr0"][1, 1, 0:3], point(1, 0.08)[0:3], atol=1e-7)


@pytest.mark.gpu
def test_gpu_sliding_list_against_the_reference_code(gold, flat):
    """The device ring (ndp_ref_list_*) behind BatchedNMPCRefPublisher against the windows the reference's NMPCRefPublisher
    produced: right after reset, over 60 control ticks, and gen_fix_pt_ref (u_r[3] = mass * g)."""
    import ndp_nmpc_qd_amd as ndp
    from ndp_nmpc_qd_amd.pt_pub import BatchedNMPCRefPublisher
    coeff, tseg, cum, fpt = _seq_traj(gold, flat)
    V = coeff.shape[0]
    eng = ndp.BatchedNMPC(V, load_mlp=False)
    pub = BatchedNMPCRefPublisher(eng)
    pub.reset(TrajCoefficients(coeff[:, :, 0:8], coeff[:, :, 8:16], coeff[:, :, 16:24], coeff[:, :, 24:28], cum, tseg, fpt))
    xr, ur = pub.get_nmpc_ref_from_long_list()
    np.testing.assert_allclose(xr, flat["seq_xr0"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(ur, flat["seq_ur0"], rtol=0, atol=1e-7)
    for k, t in enumerate(flat["seq_t"]):
        xr, ur = pub.get_nmpc_pts(t)
        np.testing.assert_allclose(xr, flat["seq_xr"][k], rtol=0, atol=1e-7)
        np.testing.assert_allclose(ur, flat["seq_ur"][k], rtol=0, atol=1e-7)
    assert np.allclose(xr[0, -1, 0:3], fpt[0]) and np.allclose(ur[0, -1], [0, 0, 0, CP.gravity])   # vehicle 0 hovers at its final point
    xr, ur = pub.gen_fix_pt_ref(flat["fix_x"])
    np.testing.assert_array_equal(xr, flat["fix_xr"])
    np.testing.assert_array_equal(ur, flat["fix_ur"])
    assert ur[0, 0, 3] == CP.mass * CP.gravity
    xr2, ur2 = pub.gen_fix_pt_ref(flat["fix_x"], quirk_b1=False)
    assert np.array_equal(xr2, xr) and ur2[0, 0, 3] == CP.gravity
