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


# ------------------------------------------------------------------------------------------- wider independent pins (round 3)
# What these tests found, and assert.  The oracle's QP solver is an interior-point method stopped at complementarity
# mu <= tol = 1e-8 (HPIPM's default, nmpc_body_rate_ctl.py:71-79 leaves it there).  Against the EXACT solution of the same QP
# (primal-dual active set on the dense KKT system) it is
#   * within 2e-7 on (all but the most weakly decided of) these problems once it is run to tol = 1e-10 -- the QP data and the
#     algorithm are right;
#   * in general within ~4e-6 (N / 20) (tol / 1e-8) / s, where s = how clearly the solution decides each bound (distance of the
#     nearest inactive bound, magnitude of the weakest active multiplier): 1e-9 on well-separated problems, up to 3e-5 in u0
#     when a bound is 0.01 from becoming active -- ONE more iteration takes such a problem from 3e-5 to 5e-9 (Mehrotra's
#     iteration converges superlinearly, the stopping test is a threshold).  Two correct interior-point codes that stop one
#     iteration apart therefore differ by up to ~3e-5 on such problems: this is the floor under any parity claim against the
#     reference's HPIPM on near-active problems.  The device runs the SAME iteration as the oracle (same counts, asserted on
#     the GPU), which is why device-vs-oracle parity is 1e-9 .. 4e-6 even there.
DEFAULT_TOL_CONST = 4e-6


def _check_qp(oracle, cfg, qp, stats=None, tight=True):
    """Both QP modes of the oracle against the primal-dual active-set answer of the same QP data; returns the active set."""
    dxa, dua, active = R.pdas_solve(qp)
    sep = min(1.0, R.separation(qp, dxa, dua, active))
    tol0 = cfg.tol
    for mode in (1, 0):                         # 1: interior point always (HPIPM's behaviour), 0: the device's early-exit rule
        cfg.qp_mode = mode
        N = qp["A"].shape[0]                    # (the constant grows with the horizon: more bounds, longer error propagation)
        for tol in ((1e-10,) if tight else ()) + (tol0,):
            bar = max(2e-7, DEFAULT_TOL_CONST * (N / 20.0) * (tol / 1e-8) / sep)
            cfg.tol = tol
            dx, du, st = oracle.qp_solve(cfg, qp)
            assert st.status == 0
            err = max(np.abs(du - dua).max(), np.abs(dx - dxa).max())
            assert err <= bar, (mode, tol, err, bar, len(active), sep)
            if stats is not None and mode == 1:
                stats.append((tol, err, sep, len(active)))
    cfg.tol = tol0
    return active


def test_both_qp_modes_against_active_set_on_200_random_qps(oracle):
    """200 QPs of the reference configuration (N = 20) from the two perturbation levels of the GPU suite's large-perturbation
    test (seeds 2 and 4: the batches in which, before the floor under the centring target, rare instances sat 2e-5 apart
    between the early-exit and the always-iterating form), first tick and a warm-started second tick."""
    stats, n_active = [], 0
    for seed, kw in ((2, dict(pos_sigma=0.5, vel_sigma=1.0, quat_sigma=0.15)), (4, dict(pos_sigma=1.0, vel_sigma=2.0, quat_sigma=0.3))):
        b = synth.make_batch(768, seed=seed, **kw)
        cfg = oracle.default_cfg()
        for i in list(range(0, 768, 15))[:50]:
            X, U = b["xr"][i].copy(), b["ur"][i].copy()
            for tick in range(2):
                qp = oracle.linearize(cfg, b["x0"][i], b["xr"][i], b["ur"][i], None, X, U)
                n_active += len(_check_qp(oracle, cfg, qp, stats)) > 0
                cfg.qp_mode = 1
                oracle.step(cfg, b["x0"][i], b["xr"][i], b["ur"][i], None, X, U)      # the iterate the next tick starts from
    st = np.array(stats)
    tight, dflt = st[st[:, 0] == 1e-10], st[st[:, 0] != 1e-10]
    assert len(tight) == 200 and n_active >= 40, (len(tight), n_active)                # a fifth and more of them with active bounds
    assert tight[:, 1].max() <= 2e-7
    assert np.median(dflt[:, 1]) < 1e-8 and (dflt[:, 1] <= 2e-7).mean() > 0.9          # the bulk at the default tolerance
    assert dflt[:, 1].max() < 5e-5                                                      # the near-active tail (see the note above)


def test_input_and_velocity_bounds_active_together(oracle):
    """u- and v-bounds (nmpc_body_rate_ctl.py:56-61: both families) active in the SAME QP: the velocity box is shrunk below the
    unconstrained solution's peak while the large initial error keeps input bounds active."""
    both, stats = 0, []
    for seed in range(40, 80):
        cfg, qp = _hard_case(oracle, seed, 1.5)
        cfg.qp_mode = 1
        dx_free, _, _ = oracle.qp_solve(cfg, qp)
        vmax = np.abs(dx_free[4:20, 3:6]).max()
        qp["lv"][:4], qp["uv"][:4] = -1e3, 1e3          # keep the problem feasible near the fixed x0
        qp["lv"][4:], qp["uv"][4:] = -0.7 * vmax, 0.7 * vmax
        # With ACTIVE STATE bounds the barrier puts lambda / t ~ 1e9 .. 1e11 on the diagonal of Q: the Newton systems get a
        # condition number of that size, and the loop is written in absolute form (an iteration's answer IS its last solve), so
        # through round 3 the answer got WORSE as the tolerance was tightened (1e-3 at tol = 1e-10 on some of these problems).
        # Round 4: two refinement solves with the same factorisation whenever a state bound's barrier term exceeds 1e6
        # (cfg.refine, oracle/ndp_oracle.c: refine_solution) -- and the tight tolerance holds its usual bound here too.  (A factored
        # / square-root Riccati recursion, which round 3 had named as the remedy, does NOT help: measured, see the test below.)
        active = _check_qp(oracle, cfg, qp, stats, tight=True)
        nz = 21 * 10
        has_v, has_u = any(v < nz for v in active), any(v >= nz for v in active)
        both += has_v and has_u
    st = np.array(stats)
    assert both >= 30, both
    tight, dflt = st[st[:, 0] == 1e-10], st[st[:, 0] != 1e-10]
    assert np.median(tight[:, 1]) < 1e-9 and tight[:, 1].max() < 2e-6
    assert np.median(dflt[:, 1]) < 1e-7 and dflt[:, 1].max() < 5e-5


def _shrunk_velocity_box(oracle, seed):
    cfg, qp = _hard_case(oracle, seed, 1.5)
    cfg.qp_mode = 1
    dx_free, _, _ = oracle.qp_solve(cfg, qp)
    vmax = np.abs(dx_free[4:20, 3:6]).max()
    qp["lv"][:4], qp["uv"][:4] = -1e3, 1e3
    qp["lv"][4:], qp["uv"][4:] = -0.7 * vmax, 0.7 * vmax
    return cfg, qp


def test_active_state_bounds_at_a_tight_tolerance_need_refinement_not_a_factored_recursion(oracle):
    """VERDICT r3 #5.  Velocity box shrunk until >= 3 state bounds are active; tol = 1e-10; against the exact (active-set) answer.
      * refinement off (round 3's oracle): the error GROWS with the tightened tolerance, up to 1e-2;
      * refinement on (default): the problems with >= 3 active state bounds within 1e-7 (all but at most two whose active set
        is weakly decided: an interior-point answer stopped at mu <= tol cannot be closer than ~tol / separation), median < 1e-9;
      * the SAME loop with the Riccati recursion in square-root form (a QR factor of P~ propagated instead of P~: numpy, below)
        is no better than the explicit form -- the digits are lost in solving a system of condition lambda / t, not in forming P."""
    rows = []
    for seed in range(40, 80):
        cfg, qp = _shrunk_velocity_box(oracle, seed)
        dxa, dua, active = R.pdas_solve(qp)
        nv = sum(1 for v in active if v < 21 * 10)
        sep = min(1.0, R.separation(qp, dxa, dua, active))
        cfg.tol = 1e-10
        errs = []
        for refine in (0, 2):
            cfg.refine = refine
            dx, du, st = oracle.qp_solve(cfg, qp)
            assert st.status == 0
            errs.append(max(np.abs(du - dua).max(), np.abs(dx - dxa).max()))
        rows.append((nv, sep, errs[0], errs[1]))
    r = np.array(rows)
    many = r[:, 0] >= 3
    assert many.sum() >= 20
    assert r[many, 2].max() > 1e-4                                   # what round 3 saw
    # the usual termination bound of an interior-point answer (see the note above): 4e-6 (tol / 1e-8) / separation, floor 1e-7
    bar = np.maximum(1e-7, DEFAULT_TOL_CONST * 1e-2 / r[:, 1])
    assert (r[:, 3] <= bar).all(), (r[:, 3] / bar).max()
    assert (r[many, 3] <= 1e-7).sum() >= many.sum() - 2              # all but the most weakly decided within 1e-7
    assert np.median(r[many, 3]) < 1e-9 and r[:, 3].max() < 2e-6
    # ---- the square-root recursion in the same loop (numpy): not the remedy
    worse = 0
    for seed in (40, 46, 50, 76):
        cfg, qp = _shrunk_velocity_box(oracle, seed)
        dxa, dua, _ = R.pdas_solve(qp)
        e = {}
        for name, ric in (("explicit", R.riccati_explicit), ("sqrt", R.riccati_sqrt), ("explicit+refine", R.refined(R.riccati_explicit, 2))):
            dx, du, _ = R.ipm_absolute(qp, 1e-10, ric)
            e[name] = max(np.abs(du - dua).max(), np.abs(dx - dxa).max())
        assert e["explicit+refine"] < 1e-8, (seed, e)
        worse += e["sqrt"] > 1e-6
        assert e["sqrt"] > 100 * e["explicit+refine"], (seed, e)
    assert worse >= 3


def test_long_horizon_two_iterations_against_active_set(oracle):
    """BASELINE config 5's shape (N = 40, 2 RTI iterations per step): each iteration's QP -- linearised at the oracle's own
    iterate -- against the active-set answer, in both QP modes, and the step's final u0 (run to tol = 1e-10) against the two
    independent solves chained by hand (full step, no shift)."""
    N = 40
    n_active = 0
    for seed in range(70, 82):
        b = synth.make_batch(1, N=N, seed=seed, pos_sigma=0.7, vel_sigma=1.5, quat_sigma=0.2)
        cfg = oracle.default_cfg(N=N, n_rti=1)
        X, U = b["xr"][0].copy(), b["ur"][0].copy()
        Xa, Ua = X.copy(), U.copy()
        for it in range(2):
            qp = oracle.linearize(cfg, b["x0"][0], b["xr"][0], b["ur"][0], None, X, U)
            n_active += len(_check_qp(oracle, cfg, qp)) > 0
            # the independent chain: active-set solution of the QP linearised at ITS OWN iterate
            qpa = oracle.linearize(cfg, b["x0"][0], b["xr"][0], b["ur"][0], None, Xa, Ua)
            dxa, dua, _ = R.pdas_solve(qpa)
            Xa, Ua = Xa + dxa, Ua + dua
            cfg.qp_mode = 1
            oracle.step(cfg, b["x0"][0], b["xr"][0], b["ur"][0], None, X, U)
        for mode in (1, 0):
            cfg2 = oracle.default_cfg(N=N, n_rti=2)
            cfg2.qp_mode, cfg2.tol = mode, 1e-10
            X2, U2 = b["xr"][0].copy(), b["ur"][0].copy()
            u2, st = oracle.step(cfg2, b["x0"][0], b["xr"][0], b["ur"][0], None, X2, U2)
            assert st.status == 0
            np.testing.assert_allclose(u2, Ua[0], atol=2e-7)
            np.testing.assert_allclose(X2, Xa, atol=2e-7)
    assert n_active >= 8, n_active
