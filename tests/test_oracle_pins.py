"""Pins the CPU oracle (oracle/ndp_oracle.c) by means that share no code with it.

The reference holds no tests or golden vectors and its NMPC back-end (acados) is
absent, so the NMPC half is pinned by: finite differences of the reference's own
symbolic expressions, dense KKT solves, a textbook active-set QP solver, scipy's
bounded least squares, and analytic hover answers.
The MLP half is pinned against fixtures produced by importing the reference.
"""
import numpy as np
import pytest

from ndp_nmpc_qd_amd import synth
from tests import ref_numpy as R


def _rand_xu(rng):
    x = rng.normal(0, 1, 10)
    x[6:10] = rng.normal(0, 1, 4)
    x[6:10] /= np.linalg.norm(x[6:10])
    u = np.array([*rng.normal(0, 1.5, 3), rng.uniform(3, 15)])
    return x, u


def test_dynamics_matches_reference_expressions(oracle):
    rng = np.random.default_rng(0)
    cfg = oracle.default_cfg(use_fd=True)
    for _ in range(20):
        x, u = _rand_xu(rng)
        fd = rng.normal(0, 2, 3)
        np.testing.assert_allclose(oracle.dynamics(cfg, x, u, fd), R.f_dyn(x, u, fd), rtol=0, atol=1e-14)
    cfg0 = oracle.default_cfg(use_fd=False)
    np.testing.assert_allclose(oracle.dynamics(cfg0, x, u, fd), R.f_dyn(x, u, None), rtol=0, atol=1e-14)


def test_jacobians_vs_finite_differences(oracle):
    rng = np.random.default_rng(1)
    for _ in range(10):
        x, u = _rand_xu(rng)
        A, B = oracle.jacobians(x, u)
        Afd = R.fd_jac(lambda z: R.f_dyn(z, u), x)
        Bfd = R.fd_jac(lambda z: R.f_dyn(x, z), u)
        np.testing.assert_allclose(A, Afd, atol=1e-8)
        np.testing.assert_allclose(B, Bfd, atol=1e-8)
        assert np.count_nonzero(A) <= 25 and np.count_nonzero(B) <= 15  # SURVEY A.2 sparsity


def test_rk4_and_sensitivities(oracle):
    rng = np.random.default_rng(2)
    cfg = oracle.default_cfg(use_fd=True)
    for _ in range(10):
        x, u = _rand_xu(rng)
        fd = rng.normal(0, 2, 3)
        xn, A, B = oracle.rk4_sens(cfg, x, u, fd)
        np.testing.assert_allclose(xn, R.rk4(x, u, fd, 0.1), atol=1e-13)
        np.testing.assert_allclose(A, R.fd_jac(lambda z: R.rk4(z, u, fd, 0.1), x), atol=2e-8)
        np.testing.assert_allclose(B, R.fd_jac(lambda z: R.rk4(x, z, fd, 0.1), u), atol=2e-8)
        # block structure the HIP kernel exploits: d(p,v,q)/dp = [I;0;0], d/dv = [hI;I;0], dq+/d(p,v,c) = 0
        np.testing.assert_allclose(A[:, 0:3], np.eye(10)[:, 0:3], atol=1e-15)
        np.testing.assert_allclose(A[:, 3:6], np.vstack([0.1 * np.eye(3), np.eye(3), np.zeros((4, 3))]), atol=1e-15)
        np.testing.assert_allclose(B[6:10, 3], 0, atol=1e-15)


def test_gauss_newton_blocks(oracle):
    rng = np.random.default_rng(3)
    cfg = oracle.default_cfg()
    W = np.diag([300, 300, 400, 10, 10, 10, 0, 10, 10, 100, 10, 10, 10, 5.0])
    b = synth.make_batch(3, seed=5)
    X = b["xr"] + rng.normal(0, 0.05, b["xr"].shape)
    U = b["ur"] + rng.normal(0, 0.1, b["ur"].shape)
    for i in range(3):
        qp = oracle.linearize(cfg, b["x0"][i], b["xr"][i], b["ur"][i], None, X[i], U[i])
        for k in (0, 7, 19):
            H, g = R.gn_blocks(X[i, k], U[i, k], b["xr"][i, k], b["ur"][i, k], W, 0.1)
            np.testing.assert_allclose(qp["Q"][k], H[:10, :10], atol=1e-6)
            np.testing.assert_allclose(np.diag(qp["Rd"][k]), H[10:, 10:], atol=1e-6)
            np.testing.assert_allclose(H[:10, 10:], 0, atol=1e-7)
            np.testing.assert_allclose(qp["q"][k], g[:10], atol=1e-6)
            np.testing.assert_allclose(qp["r"][k], g[10:], atol=1e-6)
        He, ge = R.gn_blocks(X[i, 20], None, b["xr"][i, 20], None, W[:10, :10], 1.0)  # terminal: no dt
        np.testing.assert_allclose(qp["Q"][20], He, atol=1e-6)
        np.testing.assert_allclose(qp["q"][20], ge, atol=1e-6)
        # dynamics defect and initial-state equality
        for k in (0, 19):
            np.testing.assert_allclose(qp["b"][k], R.rk4(X[i, k], U[i, k]) - X[i, k + 1], atol=1e-13)
        np.testing.assert_allclose(qp["dx0"], b["x0"][i] - X[i, 0], atol=0)
        np.testing.assert_allclose(qp["lu"][4], np.array([-6, -6, -6, 0]) - U[i, 4], atol=1e-15)
        np.testing.assert_allclose(qp["uu"][4], np.array([6, 6, 6, 9.81 / 0.36]) - U[i, 4], atol=1e-15)
        np.testing.assert_allclose(qp["lv"][4], -20 - X[i, 4, 3:6], atol=1e-15)


def test_riccati_vs_dense_kkt(oracle):
    cfg = oracle.default_cfg()
    b = synth.make_batch(4, seed=11)
    for i in range(4):
        qp = oracle.linearize(cfg, b["x0"][i], b["xr"][i], b["ur"][i], None, b["xr"][i].copy(), b["ur"][i].copy())
        dx, du = oracle.qp_riccati(qp)
        dxk, duk, _ = R.kkt_solve(qp)
        np.testing.assert_allclose(dx, dxk, atol=1e-10)
        np.testing.assert_allclose(du, duk, atol=1e-10)


def _hard_case(oracle, seed, scale):
    """Large initial error -> several input bounds active at the QP solution."""
    cfg = oracle.default_cfg()
    b = synth.make_batch(1, seed=seed, pos_sigma=scale, vel_sigma=2 * scale, quat_sigma=0.2)
    qp = oracle.linearize(cfg, b["x0"][0], b["xr"][0], b["ur"][0], None, b["xr"][0].copy(), b["ur"][0].copy())
    return cfg, qp


@pytest.mark.parametrize("seed,scale", [(1, 0.1), (2, 1.0), (3, 2.0), (4, 3.0)])
def test_ipm_vs_active_set(oracle, seed, scale):
    cfg, qp = _hard_case(oracle, seed, scale)
    dx, du, st = oracle.qp_solve(cfg, qp)
    assert st.status == 0 and st.ipm_iters <= 30
    dxa, dua, active = R.active_set_solve(qp)
    if scale >= 1.0:
        assert len(active) > 0  # the case really exercises the inequality path
    np.testing.assert_allclose(du, dua, atol=2e-7)
    np.testing.assert_allclose(dx, dxa, atol=2e-7)


def test_ipm_state_bound_active(oracle):
    """Velocity bound (idxbx = 3,4,5, stages 1..N-1) made active by shrinking it."""
    cfg, qp = _hard_case(oracle, 7, 1.0)
    dx_free, _, _ = oracle.qp_solve(cfg, qp)
    vmax = np.abs(dx_free[4:20, 3:6]).max()
    assert vmax > 1.2
    qp["lv"][:4], qp["uv"][:4] = -1e3, 1e3      # keep the problem feasible near the fixed x0
    qp["lv"][4:], qp["uv"][4:] = -0.75 * vmax, 0.75 * vmax
    dx, du, st = oracle.qp_solve(cfg, qp)
    assert st.status == 0
    dxa, dua, active = R.active_set_solve(qp)
    assert any(v < 21 * 10 for v in active)  # a state bound is in the active set
    # weakly active state bounds converge like mu/lambda: 2e-6 abs is still 5x inside the 1e-5 parity bar
    np.testing.assert_allclose(du, dua, atol=2e-6)
    np.testing.assert_allclose(dx, dxa, atol=2e-6)
    assert np.abs(dx[4:20, 3:6]).max() <= 0.75 * vmax + 1e-7
    # stage 0 and the terminal stage are NOT bounded (nmpc_body_rate_ctl.py:59-66)
    assert st.n_active > 0


def test_ipm_vs_scipy_bvls(oracle):
    """Input bounds only (state bounds inactive): condense and solve with scipy's BVLS."""
    from scipy.optimize import lsq_linear
    cfg, qp = _hard_case(oracle, 9, 2.0)
    N = 20
    # condensing: dx = Gx du + cx
    Gx = np.zeros(((N + 1) * 10, N * 4))
    cx = np.zeros((N + 1) * 10)
    cx[:10] = qp["dx0"]
    for k in range(N):
        r0, r1 = slice(k * 10, (k + 1) * 10), slice((k + 1) * 10, (k + 2) * 10)
        Gx[r1] = qp["A"][k] @ Gx[r0]
        Gx[r1, k * 4:(k + 1) * 4] += qp["B"][k]
        cx[r1] = qp["A"][k] @ cx[r0] + qp["b"][k]
    Qb = np.zeros(((N + 1) * 10,) * 2)
    for k in range(N + 1):
        Qb[k * 10:(k + 1) * 10, k * 10:(k + 1) * 10] = qp["Q"][k]
    H = Gx.T @ Qb @ Gx + np.diag(qp["Rd"].ravel())
    g = Gx.T @ (Qb @ cx + qp["q"].ravel()) + qp["r"].ravel()
    L = np.linalg.cholesky(H)
    res = lsq_linear(L.T, -np.linalg.solve(L, g), bounds=(qp["lu"].ravel(), qp["uu"].ravel()), method="bvls",
                     tol=1e-14, max_iter=500)
    dx, du, st = oracle.qp_solve(cfg, qp)
    assert st.status == 0
    assert np.abs(dx[1:20, 3:6]).max() < 19.0  # state bounds inactive, so the condensed problem is equivalent
    np.testing.assert_allclose(du.ravel(), res.x, atol=5e-7)


def test_hover_known_answers(oracle):
    """SURVEY Appendix C.2."""
    cfg = oracle.default_cfg()
    xr, ur = synth.hover_reference()
    X, U = xr.copy(), ur.copy()
    u0, st = oracle.step(cfg, xr[0], xr, ur, None, X, U)
    assert st.status == 0
    np.testing.assert_allclose(u0, [0, 0, 0, 9.81], atol=1e-9)
    np.testing.assert_allclose(X, xr, atol=1e-9)
    # quirk B1 (pt_publisher.py:50): u_r[3] = mass*g although u[3] is an acceleration
    xr, ur = synth.hover_reference(quirk_b1=True)
    X, U = xr.copy(), ur.copy()
    u0, st = oracle.step(cfg, xr[0], xr, ur, None, X, U)
    assert st.status == 0
    # independent check of the quirk value by the dense KKT route
    qp = oracle.linearize(cfg, xr[0], xr, ur, None, xr.copy(), ur.copy())
    _, duk, _ = R.kkt_solve(qp)
    np.testing.assert_allclose(u0, ur[0] + duk[0], atol=1e-8)
    assert abs(u0[3] - 9.79359713) < 1e-6 and np.abs(u0[:3]).max() < 1e-9


def test_full_step_no_shift_and_persistence(oracle):
    """update() keeps the iterate between calls without shifting it."""
    cfg = oracle.default_cfg()
    b = synth.make_batch(1, seed=21)
    X, U = b["xr"][0].copy(), b["ur"][0].copy()
    qp = oracle.linearize(cfg, b["x0"][0], b["xr"][0], b["ur"][0], None, X, U)
    dx, du, _ = oracle.qp_solve(cfg, qp)
    X0, U0 = X.copy(), U.copy()
    u0, st = oracle.step(cfg, b["x0"][0], b["xr"][0], b["ur"][0], None, X, U)
    np.testing.assert_allclose(X, X0 + dx, atol=1e-12)
    np.testing.assert_allclose(U, U0 + du, atol=1e-12)
    np.testing.assert_allclose(u0, U[0], atol=0)
    np.testing.assert_allclose(X[0], b["x0"][0], atol=1e-9)  # x0 equality holds after the full step
    # second call starts from the stored iterate: equals n_rti=2 in one call
    u1, _ = oracle.step(cfg, b["x0"][0], b["xr"][0], b["ur"][0], None, X, U)
    cfg2 = oracle.default_cfg(n_rti=2)
    X2, U2 = b["xr"][0].copy(), b["ur"][0].copy()
    u2, _ = oracle.step(cfg2, b["x0"][0], b["xr"][0], b["ur"][0], None, X2, U2)
    np.testing.assert_allclose(u1, u2, atol=1e-12)


def test_disturbance_enters_dynamics_only(oracle):
    """NDP variant: f/mass added to v-dot; A, B, cost unchanged (ndp_nmpc_body_rate_ctl.py:155-157)."""
    b = synth.make_batch(1, seed=31)
    f = np.random.default_rng(0).normal(0, 2, (21, 3))
    c0, c1 = oracle.default_cfg(use_fd=False), oracle.default_cfg(use_fd=True)
    X, U = b["xr"][0].copy(), b["ur"][0].copy()
    q0 = oracle.linearize(c0, b["x0"][0], b["xr"][0], b["ur"][0], None, X, U)
    q1 = oracle.linearize(c1, b["x0"][0], b["xr"][0], b["ur"][0], f, X, U)
    np.testing.assert_allclose(q0["A"], q1["A"], atol=1e-15)
    np.testing.assert_allclose(q0["B"], q1["B"], atol=1e-15)
    np.testing.assert_allclose(q0["Q"], q1["Q"], atol=0)
    d = q1["b"] - q0["b"]
    # constant acceleration f/m over one RK4 step: dv = h f/m, dp = h^2/2 f/m, dq = 0
    np.testing.assert_allclose(d[:, 3:6], 0.1 * f[:20] / 1.4844, atol=1e-14)
    np.testing.assert_allclose(d[:, 0:3], 0.005 * f[:20] / 1.4844, atol=1e-14)
    np.testing.assert_allclose(d[:, 6:10], 0, atol=1e-15)


def test_mlp_against_reference_fixture(oracle, mlp_blob, mlp_golden):
    f = oracle.mlp_forward(mlp_blob, mlp_golden["z"])
    # fp32 network: tolerance 1e-5 * max(1, |f|)  (reference output is fp32, summation order differs)
    tol = 1e-5 * np.maximum(1.0, np.abs(mlp_golden["f"]))
    assert np.all(np.abs(f - mlp_golden["f"]) <= tol)
    # SURVEY C.1 digits
    np.testing.assert_allclose(mlp_golden["f"][0], [0.5289574, -0.4382870, -4.2395115], atol=2e-6)


def test_downwash_update_and_gate(oracle, mlp_blob, mlp_golden):
    other, ego = mlp_golden["other"], mlp_golden["ego"]
    f = oracle.downwash_batch(mlp_blob, other, ego, None)
    tol = 1e-5 * np.maximum(1.0, np.abs(mlp_golden["f_update"]))
    assert np.all(np.abs(f - mlp_golden["f_update"]) <= tol)
    # gate: ego ODOMETRY xy vs other.x[0] xy, strict '<' (ndp_nmpc_leader_node.py:65-68)
    ego_xy = other[:, 0, 0:2].copy()
    ego_xy[0] += [0.6, 0.79]      # d^2 = 0.9841 < 1  -> on
    ego_xy[1] += [0.6, 0.8]       # d^2 = 1.0 (not < 1) -> off
    ego_xy[2] += [3.0, 0.0]       # far -> off
    fg = oracle.downwash_batch(mlp_blob, other, ego, ego_xy)
    assert np.array_equal(fg[0], f[0]) and np.all(fg[1] == 0) and np.all(fg[2] == 0) and np.array_equal(fg[3], f[3])
