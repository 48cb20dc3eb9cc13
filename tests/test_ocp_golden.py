"""Rows a1-a6 pinned to the REFERENCE'S OWN FILES: tests/golden/ocp_golden.{npz,json} were produced by running
nmpc_ctl/nmpc_body_rate_ctl.py and ndp_nmpc_ctl/ndp_nmpc_body_rate_ctl.py unmodified under stand-ins for casadi / acados_template
(tests/golden/make_ocp_golden.py).  Checked against them here, on the CPU: the oracle's dynamics, Jacobians, ERK4 step with
sensitivities, Gauss-Newton blocks and default configuration; the PRODUCT's default configuration (ndp_default_cfg); the numpy
restatement other pin tests use (tests/ref_numpy.py); and the drop-in classes' staging against the recorded set / solve_for_x0
call sequences of the reference's reset / update.  The device's LDS image against the same fixture: tests/test_gpu_parity.py.

What this does NOT pin (no acados exists here): the SQP-RTI / HPIPM solve itself and the [acados-knowledge] items of SURVEY A.4
(stage cost scaled by the interval, terminal by 1; ERK defaults 4 stages / 1 step; bounds on stages 1..N-1; full step)."""
import json
import os

import numpy as np
import pytest

from tests import ref_numpy as R

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(HERE, "golden", "ocp_golden.npz"))


@pytest.fixture(scope="module")
def M():
    with open(os.path.join(HERE, "golden", "ocp_golden.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


def _close(a, b, tol=1e-14):
    """|a - b| <= tol * max(1, |b|) elementwise (values here are O(1) .. O(400))."""
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b) / np.maximum(1.0, np.abs(b))
    assert err.max() <= tol, float(err.max())


# ------------------------------------------------------------------------------------------- a3: the OCP definition
def test_reference_ocp_definition_is_what_survey_a1_says(G, M):
    for pre in ("nmpc_", "ndp_"):
        W = G[pre + "W"]
        assert W.shape == (14, 14) and np.array_equal(W, np.diag(np.diag(W)))
        assert np.array_equal(np.diag(W), [300, 300, 400, 10, 10, 10, 0, 10, 10, 100, 10, 10, 10, 5])
        assert np.array_equal(G[pre + "W_e"], W[:10, :10])
        assert np.array_equal(G[pre + "lbu"], [-6, -6, -6, 0]) and np.array_equal(G[pre + "ubu"], [6, 6, 6, 9.81 / 0.36])
        assert np.array_equal(G[pre + "idxbu"], [0, 1, 2, 3]) and np.array_equal(G[pre + "idxbx"], [3, 4, 5])
        assert np.array_equal(G[pre + "lbx"], [-20] * 3) and np.array_equal(G[pre + "ubx"], [20] * 3)
        assert int(G[pre + "N"]) == 20 and float(G[pre + "tf"]) == 2.0
        assert not G[pre + "x0"].any() and not G[pre + "yref"].any() and not G[pre + "yref_e"].any()
        o = M[pre + "ocp"]
        assert o["cost_type"] == "NONLINEAR_LS" and o["cost_type_e"] == "NONLINEAR_LS"
        s = o["solver_options_set"]
        assert (s["qp_solver"], s["hessian_approx"], s["integrator_type"], s["nlp_solver_type"]) == \
            ("PARTIAL_CONDENSING_HPIPM", "GAUSS_NEWTON", "ERK", "SQP_RTI")
        assert s["qp_solver_cond_N"] == 20 and s["tf"] == 2 and s["print_level"] == 0
        # the reference leaves warm start, the ERK tableau, tolerances, iteration limits and regularisation at acados' defaults
        for k in ("qp_solver_warm_start", "sim_method_num_stages", "sim_method_num_steps", "qp_solver_iter_max", "levenberg_marquardt"):
            assert k in o["solver_options_not_set_by_the_reference"]
        assert o["status_exception_text"] == "acados acados_ocp_solver returned status 4. Exiting."
        assert o["model_name"] == "qd_body_rate_model"                       # both classes
    assert M["nmpc_ocp"]["solver_options_set"]["hpipm_mode"] == "BALANCE" and "hpipm_mode" not in M["ndp_ocp"]["solver_options_set"]
    assert int(G["nmpc_np"]) == 4 and int(G["ndp_np"]) == 7
    assert M["ndp_ocp"]["param_names"] == ["qwr", "qxr", "qyr", "qzr", "disturb_fx", "disturb_fy", "disturb_fz"]
    assert M["chdir_targets"] == ["nmpc_ctl", "ndp_nmpc_ctl"]                 # the os.chdir side effect, recorded


def test_oracle_and_product_default_cfg_match_the_reference_ocp(G, oracle):
    from ndp_nmpc_qd_amd import _lib
    co = oracle.default_cfg()
    cp = _lib.default_cfg()                                                  # ndp_default_cfg: the product's constants (no GPU needed)
    W = np.diag(G["nmpc_W"])
    for c in (co, cp):
        assert c.N == int(G["nmpc_N"]) and c.n_rti == 1
        assert c.dt == float(G["nmpc_tf"]) / int(G["nmpc_N"])
        assert np.array_equal(np.array(c.Qd), W[:10]) and np.array_equal(np.array(c.Rd), W[10:])
        assert np.array_equal(np.array(c.lbu), G["nmpc_lbu"]) and np.array_equal(np.array(c.ubu), G["nmpc_ubu"])
        assert np.array_equal(np.array(c.lbv), G["nmpc_lbx"]) and np.array_equal(np.array(c.ubv), G["nmpc_ubx"])
    # mass and gravity as the reference's expressions carry them: d(vdot)/d(force) = 1 / mass, f(0-thrust) = -g on vz
    inv_m = G["ndp_dfdp"][0][3, 4]
    assert inv_m == 1.0 / co.mass == 1.0 / cp.mass
    assert np.array_equal(G["ndp_dfdp"][:, 3:6, 4:7], np.broadcast_to(np.eye(3) * inv_m, (64, 3, 3)))
    assert co.g == cp.gravity == 9.81


# ------------------------------------------------------------------------------------------- a1: dynamics and Jacobians
def test_oracle_dynamics_and_jacobians_against_the_reference_expressions(G, oracle):
    cn, cd = oracle.default_cfg(use_fd=False), oracle.default_cfg(use_fd=True)
    for i in range(G["pt_x"].shape[0]):
        x, u, f = G["pt_x"][i], G["pt_u"][i], G["pt_f"][i]
        _close(oracle.dynamics(cn, x, u), G["nmpc_f"][i])
        _close(oracle.dynamics(cd, x, u, f), G["ndp_f"][i])
        A, B = oracle.jacobians(x, u)
        _close(A, G["nmpc_dfdx"][i])
        _close(B, G["nmpc_dfdu"][i])
        _close(A, G["ndp_dfdx"][i])                                           # the disturbance is additive: same A, B
        _close(B, G["ndp_dfdu"][i])
        # the numpy restatement the other pin tests lean on
        _close(R.f_dyn(x, u), G["nmpc_f"][i])
        _close(R.f_dyn(x, u, f), G["ndp_f"][i])
    # structure: 25 + 15 structural non-zeros, the same pattern at every point
    nzA = (np.abs(G["nmpc_dfdx"]) > 0).any(axis=0)
    nzB = (np.abs(G["nmpc_dfdu"]) > 0).any(axis=0)
    assert nzA.sum() == 25 and nzB.sum() == 15
    assert not G["nmpc_dfdp"].any()                                           # the quaternion reference does not enter the dynamics


def test_oracle_erk4_step_and_sensitivities_against_the_reference_expressions(G, oracle):
    """One classical RK4 step of h = tf / N through the reference's f_expl_expr, and the exact derivative of that map (dual numbers
    in the generator) -- what acados' ERK with forward sensitivities computes ([acados-knowledge]: 4 stages, 1 step)."""
    cn, cd = oracle.default_cfg(use_fd=False), oracle.default_cfg(use_fd=True)
    assert float(G["nmpc_rk4_h"]) == cn.dt
    for i in range(G["pt_x"].shape[0]):
        x, u, f = G["pt_x"][i], G["pt_u"][i], G["pt_f"][i]
        xn, A, B = oracle.rk4_sens(cn, x, u)
        _close(xn, G["nmpc_rk4_xn"][i], 1e-13)
        _close(A, G["nmpc_rk4_A"][i], 1e-13)
        _close(B, G["nmpc_rk4_B"][i], 1e-13)
        xn, A, B = oracle.rk4_sens(cd, x, u, f)
        _close(xn, G["ndp_rk4_xn"][i], 1e-13)
        _close(A, G["ndp_rk4_A"][i], 1e-13)
        _close(B, G["ndp_rk4_B"][i], 1e-13)
        _close(R.rk4(x, u, f), G["ndp_rk4_xn"][i], 1e-13)
    # SURVEY section 4's claim behind the device's 8 sensitivity columns: the p and v columns of the RK4 sensitivity are [I;0;0], [hI;I;0]
    h = float(G["nmpc_rk4_h"])
    A = G["ndp_rk4_A"]
    assert np.array_equal(A[:, :, 0:3], np.broadcast_to(np.eye(10)[:, 0:3], A[:, :, 0:3].shape))
    want_v = np.zeros((10, 3))
    want_v[0:3], want_v[3:6] = h * np.eye(3), np.eye(3)
    _close(A[:, :, 3:6], np.broadcast_to(want_v, A[:, :, 3:6].shape), 1e-15)


# ------------------------------------------------------------------------------------------- a2: cost output and Gauss-Newton blocks
def test_oracle_gauss_newton_blocks_against_the_reference_cost_expressions(G, oracle):
    """cost_y_expr / cost_y_expr_e and their Jacobians from the reference's graph; H = s J'WJ, g = s J'W (y - yref) with s = the
    interval for stages and 1 for the terminal node ([acados-knowledge]); p = xr[6:10] as the reference's update sets it."""
    c = oracle.default_cfg()
    W, We = G["nmpc_W"], G["nmpc_W_e"]
    for pre in ("nmpc_", "ndp_"):
        for i in range(G["pt_x"].shape[0]):
            x, u, xr, ur = G["pt_x"][i], G["pt_u"][i], G["pt_xr"][i], G["pt_ur"][i]
            y, Jx, Ju = G[pre + "y"][i], G[pre + "dydx"][i], G[pre + "dydu"][i]
            _close(R.cost_y(x, u, xr[6:10]), y)
            _close(R.cost_y(x, None, xr[6:10]), G[pre + "ye"][i])
            assert np.array_equal(Ju, np.vstack([np.zeros((10, 4)), np.eye(4)]))          # J_u = [0; I]
            assert np.array_equal(Jx[10:], np.zeros((4, 10))) and np.array_equal(G[pre + "dyedx"][i], Jx[:10])
            assert np.array_equal(Jx[:6, :6], np.eye(6)) and not Jx[6].any()              # row 6 (qw_r) has no Jacobian and weight 0
            J = np.hstack([Jx, Ju])
            res = y - np.concatenate([xr, ur])
            assert res[6] == 0.0                                                           # y[6] = qw_r = yref[6]
            H, g = c.dt * J.T @ W @ J, c.dt * J.T @ W @ res
            Q, q, Rd, r = _cost_stage(oracle, c, c.dt, x, u, xr, ur)
            _close(Q, H[:10, :10], 1e-13)
            _close(q, g[:10], 1e-13)
            _close(Rd, np.diag(H[10:, 10:]), 1e-13)
            _close(r, g[10:], 1e-13)
            assert not H[:10, 10:].any()                                                   # S = 0
            He, ge = Jx[:10].T @ We @ Jx[:10], Jx[:10].T @ We @ (G[pre + "ye"][i] - xr)
            Q, q, _, _ = _cost_stage(oracle, c, 1.0, x, None, xr, None)
            _close(Q, He, 1e-13)
            _close(q, ge, 1e-13)


def _cost_stage(oracle, c, scale, x, u, xr, ur):
    import ctypes as C
    f64 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)      # noqa: E731
    p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)                     # noqa: E731
    x, u, xr, ur = f64(x), f64(u), f64(xr), f64(ur)
    Q, q, Rd, r = np.zeros((10, 10)), np.zeros(10), np.zeros(4), np.zeros(4)
    oracle.lib().orc_cost_stage(C.byref(c), C.c_double(scale), p(x), p(u), p(xr), p(ur), p(Q), p(q), p(Rd), p(r))
    return Q, q, Rd, r


def test_oracle_linearisation_of_a_whole_horizon_against_the_reference_expressions(G, oracle):
    """The QP data of one RTI iteration at an iterate away from the reference, N = 20, both classes: A_k, B_k, b_k from the ERK4 map
    of the reference's f; q_k, r_k, Q_k from its cost expressions; the step bounds from its lbu / ubu / lbx / ubx."""
    N = int(G["nmpc_N"])
    for tag, use_fd in (("lin_nmpc_", False), ("lin_ndp_", True)):
        c = oracle.default_cfg(use_fd=use_fd)
        f = G["lin_f"].astype(np.float64) if use_fd else None
        qp = oracle.linearize(c, G["lin_x0"], G["lin_xr"], G["lin_ur"], f, G["lin_X"], G["lin_U"])
        _close(qp["A"], G[tag + "A"], 1e-13)
        _close(qp["B"], G[tag + "B"], 1e-13)
        _close(qp["b"], G[tag + "b"], 1e-13)
        W, We = G["nmpc_W"], G["nmpc_W_e"]
        for k in range(N):
            J, res = G[tag + "Jy"][k], G[tag + "res"][k]
            H, g = c.dt * J.T @ W @ J, c.dt * J.T @ W @ res
            _close(qp["Q"][k], H[:10, :10], 1e-13)
            _close(qp["q"][k], g[:10], 1e-13)
            _close(qp["Rd"][k], np.diag(H[10:, 10:]), 1e-13)
            _close(qp["r"][k], g[10:], 1e-13)
        Jx, res = G[tag + "Jy"][N][:10, :10], G[tag + "res"][N][:10]
        _close(qp["Q"][N], Jx.T @ We @ Jx, 1e-13)
        _close(qp["q"][N], Jx.T @ We @ res, 1e-13)
        _close(qp["dx0"], G["lin_x0"] - G["lin_X"][0])
        _close(qp["lu"], G["nmpc_lbu"][None] - G["lin_U"])
        _close(qp["uu"], G["nmpc_ubu"][None] - G["lin_U"])
        _close(qp["lv"][1:N], G["nmpc_lbx"][None] - G["lin_X"][1:N, 3:6])
        _close(qp["uv"][1:N], G["nmpc_ubx"][None] - G["lin_X"][1:N, 3:6])


# ------------------------------------------------------------------------------------------- a4-a6: reset / update marshalling
FIELD = {0: "x", 1: "u", 2: "yref", 3: "p", 4: "x0"}


def _calls(G, pre, what):
    st, fl, ln, vl = (G[f"{pre}{what}_{k}"] for k in ("stage", "field", "len", "val"))
    return [(int(s), FIELD[int(f)], vl[i, :int(n)].copy()) for i, (s, f, n) in enumerate(zip(st, fl, ln))]


class _FakeEngine:
    """BatchedNMPC(batch = 1) as SolverFacade sees it, recording what reaches the C-ABI."""

    def __init__(self, N):
        self.N, self.updates, self.iterates = N, [], []
        self._X, self._U = np.zeros((1, N + 1, 10)), np.zeros((1, N, 4))

    def get_iterate(self):
        return self._X.copy(), self._U.copy()

    def set_iterate(self, X, U):
        self._X, self._U = np.array(X, dtype=np.float64), np.array(U, dtype=np.float64)
        self.iterates.append((self._X.copy(), self._U.copy()))

    def update(self, x0, xr, ur, f=None, raise_on_status=True, full=False):
        self.updates.append(dict(x0=np.array(x0), xr=np.array(xr), ur=np.array(ur), f=None if f is None else np.array(f)))
        return np.zeros((1, 4)), self._X.copy(), self._U.copy(), np.zeros(1, dtype=np.int32), np.zeros(1, dtype=np.int32)


@pytest.mark.parametrize("pre", ["nmpc_", "ndp_"])
def test_reset_and_update_call_sequences_of_the_reference_and_the_drop_in_staging(G, pre):
    """The recorded sequences are what the reference's reset / update do (nmpc_body_rate_ctl.py:86-112, ndp :84-112): N x (x, u) + a
    terminal x; N x (yref[14], p) + terminal yref[10] + p, then solve_for_x0(x0).  Replayed call by call into a SolverFacade, and
    staged the drop-in's way (set_reference: three array assignments), they must leave the SAME staging -- and the same arrays must
    reach the C-ABI."""
    from ndp_nmpc_qd_amd.solver_facade import SolverFacade
    ndp = pre == "ndp_"
    N, npar = int(G[pre + "N"]), int(G[pre + "np"])
    xr, ur, x0, f = G[pre + "call_xr"], G[pre + "call_ur"], G[pre + "call_x0"], G[pre + "call_f"]
    # ---- the reference's sequences themselves
    rs = _calls(G, pre, "reset")
    assert [(s, fl) for s, fl, _ in rs] == [(i, fl) for i in range(N) for fl in ("x", "u")] + [(N, "x")]
    for s, fl, v in rs:
        assert np.array_equal(v, xr[s] if fl == "x" else ur[s])
    up = _calls(G, pre, "update")
    assert [(s, fl) for s, fl, _ in up] == [(i, fl) for i in range(N) for fl in ("yref", "p")] + [(N, "yref"), (N, "p"), (-1, "x0")]
    for s, fl, v in up[:-1]:
        if fl == "yref":
            assert np.array_equal(v, np.concatenate([xr[s], ur[s]]) if s < N else xr[N])
        else:
            want = np.concatenate([xr[s, 6:10], f[s].astype(np.float64)]) if ndp else xr[s, 6:10]   # fp32 force promoted
            assert v.size == npar and np.array_equal(v, want)
    assert np.array_equal(up[-1][2], x0)
    # ---- replayed into the facade call by call vs. staged the drop-in's way
    a, b = SolverFacade(_FakeEngine(N), disturbance=ndp), SolverFacade(_FakeEngine(N), disturbance=ndp)
    for s, fl, v in rs:
        a.set(s, fl, v)
    for i in range(N):                                   # the drop-in classes' reset loop (nmpc_ctl/nmpc_body_rate_ctl.py)
        b.set(i, "x", xr[i, :])
        b.set(i, "u", ur[i, :])
    b.set(N, "x", xr[N, :])
    assert np.array_equal(a._X, b._X) and np.array_equal(a._U, b._U) and np.array_equal(a._X, xr) and np.array_equal(a._U, ur)
    for s, fl, v in up[:-1]:
        a.set(s, fl, v)
    b.set_reference(xr, ur, f if ndp else None)
    assert np.array_equal(a._yref, b._yref) and np.array_equal(a._p, b._p)
    ua, ub = a.solve_for_x0(up[-1][2]), b.solve_for_x0(x0)
    assert ua.shape == ub.shape == (4,)
    for fac in (a, b):
        eng = fac._eng
        assert len(eng.iterates) == 1 and np.array_equal(eng.iterates[0][0][0], xr) and np.array_equal(eng.iterates[0][1][0], ur)
        (call,) = eng.updates
        assert np.array_equal(call["x0"][0], x0) and np.array_equal(call["xr"][0], xr) and np.array_equal(call["ur"][0], ur)
        if ndp:
            # float64 all the way, like the reference's p (ndp_nmpc_body_rate_ctl.py:97-99): the fp32 values, promoted exactly
            assert call["f"].dtype == np.float64 and np.array_equal(call["f"][0], f.astype(np.float64))
        else:
            assert call["f"] is None
