"""QP_AUTO's active-set iterations on the input bounds, on the device through the C-ABI (run with -m gpu).

The default mode of every handle (ndp_cfg.as_iter_max = 8).  Checked against
  * the oracle's interior-point loop -- the reference's method (HPIPM, nmpc_body_rate_ctl.py:71-74) -- at a tight tolerance, where both
    must give the QP's solution (bar 1e-6 on u0, 1e-7 typical), and at its default tolerance (the north star's 1e-5);
  * the oracle's plain-C restatement of the same rule (qp_mode 0): status, sweeps, kept sets identical, numbers to 1e-8;
  * exact dense-KKT solutions on a sample (tests/ref_numpy.py).
CPU twin on the wave emulator: tests/test_active_set.py.
"""
import numpy as np
import pytest

from ndp_nmpc_qd_amd import synth
from tests import ref_numpy as R

pytestmark = pytest.mark.gpu

MIXED = dict(pos_sigma=0.5, vel_sigma=1.0, quat_sigma=0.15)        # bench.py's `mixed` workload
HARD = dict(pos_sigma=1.5, vel_sigma=3.0, quat_sigma=0.2)          # test_active_bounds_and_infeasible_start's


@pytest.fixture(scope="module")
def ndp():
    import ndp_nmpc_qd_amd
    return ndp_nmpc_qd_amd


def _rel(u, uo):
    return float(np.max(np.abs(u - uo) / np.maximum(1.0, np.abs(uo))))


def _twin_cfg(oracle, **kw):
    c = oracle.default_cfg(**kw)
    c.qp_mode = 0
    return c


def test_mixed_workload_four_ticks_against_twin_and_interior_point(ndp, oracle):
    """bench.py's mixed workload at the headline batch (1024 instances, ~20 % with active input bounds), four control ticks with the
    kept sets carried along.  Device = oracle twin: status, sweeps and sets identical on EVERY instance, iterate to 1e-8.  Device
    against the always-iterating oracle at tol 1e-11: u0 inside 1e-6 on EVERY instance (1.5e-7 measured).  At the oracle's default tol
    1e-8 the interior-point answer itself is 6e-5 .. 2e-4 away from its own converged (tol 1e-11) answer on one to three nearly
    degenerate instances of 1024 per tick: the device is never further from it than that converged answer is, and
    inside the north star's 1e-5 on > 99.5 % of the instances.  No instance needs the interior-point loop; after the first tick most
    constrained instances take ONE sweep."""
    B = 1024
    eng = ndp.BatchedNMPC(B)
    twin, tight, dflt = _twin_cfg(oracle), oracle.default_cfg(), oracle.default_cfg()
    tight.tol = 1e-11
    b0 = synth.make_batch(B, seed=synth.SEED0 + 40, **MIXED)
    eng.reset(b0["xr"], b0["ur"])
    Xo, Uo = b0["xr"].copy(), b0["ur"].copy()
    acto = np.zeros((B, 20, 4), dtype=np.int8)
    worst_tight = 0.0
    one_sweep, frac_dflt = [], []
    for t in range(4):
        b = synth.make_batch(B, seed=synth.SEED0 + 40, t0=0.02 * t, **MIXED)
        Xp, Up = Xo.copy(), Uo.copy()
        u0, X, U, st, it = eng.update(b["x0"], b["xr"], b["ur"], raise_on_status=False, full=True)
        sw, act = eng.active_set()
        uo, sto, ito, swo = oracle.step_batch_as(twin, b["x0"], b["xr"], b["ur"], None, Xo, Uo, acto)
        assert np.array_equal(st, sto) and not st.any() and not it.any() and not ito.any()
        assert np.array_equal(sw, swo) and np.array_equal(act, acto)
        np.testing.assert_allclose(U, Uo, rtol=0, atol=1e-8)
        np.testing.assert_allclose(X, Xo, rtol=0, atol=1e-8)
        ans = {}
        for cfg, name in ((tight, "tight"), (dflt, "dflt")):
            Xi, Ui = Xp.copy(), Up.copy()
            ans[name], sti, _ = oracle.step_batch(cfg, b["x0"], b["xr"], b["ur"], None, Xi, Ui)
            assert not sti.any()
        e_t = np.max(np.abs(u0 - ans["tight"]) / np.maximum(1.0, np.abs(ans["tight"])), axis=1)
        e_d = np.max(np.abs(u0 - ans["dflt"]) / np.maximum(1.0, np.abs(ans["dflt"])), axis=1)
        e_own = np.max(np.abs(ans["tight"] - ans["dflt"]) / np.maximum(1.0, np.abs(ans["dflt"])), axis=1)     # the default tolerance's own error
        worst_tight = max(worst_tight, float(e_t.max()))
        assert (e_d <= e_own + 1e-6).all()                                   # never further from it than its own converged answer
        frac_dflt.append(float((e_d <= 1e-5).mean()))
        con = act.any(axis=(1, 2))
        assert 0.1 < con.mean() < 0.4
        on = act != 0
        lim = np.where(act > 0, np.array([6, 6, 6, 9.81 / 0.36]), np.array([-6, -6, -6, 0.0]))
        assert np.array_equal(U[on], lim[on])                              # pinned inputs sit ON their bounds
        assert (U[..., :3] <= 6).all() and (U[..., :3] >= -6).all() and (U[..., 3] >= 0).all() and (U[..., 3] <= 9.81 / 0.36).all()
        one_sweep.append(float((sw[con] == 1).mean()))
        assert sw.max() <= 4
        Xo[:], Uo[:] = X, U                                                # one trajectory for both
    assert worst_tight < 1e-6 and min(frac_dflt) > 0.995, (worst_tight, frac_dflt)
    assert one_sweep[0] == 0.0 and min(one_sweep[1:]) > 0.5, one_sweep


def test_fused_downwash_step_with_active_bounds(ndp, oracle, mlp_blob):
    """The headline launch (gate + MLP + step in one kernel) on the mixed workload: same answers as the oracle twin fed the oracle's
    force (the network's fp16-split layers leave 4e-6 on the force: bar 1e-6 on u0), sets identical, two ticks."""
    B = 1024
    b = synth.make_batch(B, seed=synth.SEED0 + 41, downwash=True, **MIXED)
    eng = ndp.BatchedNMPC(B, disturbance=True)
    eng.reset(b["xr"], b["ur"])
    twin = _twin_cfg(oracle, use_fd=True)
    f = oracle.downwash_batch(mlp_blob, b["other"], b["xr"], b["ego_xy"])
    Xo, Uo = b["xr"].copy(), b["ur"].copy()
    acto = np.zeros((B, 20, 4), dtype=np.int8)
    for t in range(2):
        u0, X, U, st, it = eng.update(b["x0"], b["xr"], b["ur"], other=b["other"], ego_xy=b["ego_xy"], raise_on_status=False, full=True)
        sw, act = eng.active_set()
        uo, sto, ito, swo = oracle.step_batch_as(twin, b["x0"], b["xr"], b["ur"], f, Xo, Uo, acto)
        assert not st.any() and not sto.any() and not it.any()
        same = (act == acto).all(axis=(1, 2))
        assert same.mean() > 0.995                     # (a multiplier within the force's 4e-6 of zero may fall either way)
        assert _rel(u0[same], uo[same]) < 1e-6
        assert act.any(axis=(1, 2)).mean() > 0.1
        Xo[:], Uo[:] = X, U
        acto[:] = act


def test_hard_starts_against_exact_solutions(ndp, oracle):
    """test_active_bounds_and_infeasible_start's inputs (large initial errors, iterates outside the box) in the default mode.  The
    oracle's interior-point loop gives up on the iterates with a thrust of -1 (status 4 after 50 iterations); the QPs are feasible
    and the active-set iterations solve them: every instance the device reports solved WITHOUT the interior-point loop is held to 1e-8
    of the exact dense-KKT solution; the twin agrees on status, sweeps and sets; where the interior-point oracle converged (tol 1e-11)
    the two agree to 1e-6."""
    B = 64
    b = synth.make_batch(B, seed=77, **HARD)
    U0 = b["ur"].copy()
    U0[::4, :, 0] = 7.5
    U0[1::4, 3, 3] = -1.0
    eng = ndp.BatchedNMPC(B)
    eng.set_iterate(b["xr"], U0)
    u0, X, U, st, it = eng.update(b["x0"], b["xr"], b["ur"], raise_on_status=False, full=True)
    sw, act = eng.active_set()
    Xo, Uo = b["xr"].copy(), U0.copy()
    acto = np.zeros((B, 20, 4), dtype=np.int8)
    uo, sto, ito, swo = oracle.step_batch_as(_twin_cfg(oracle), b["x0"], b["xr"], b["ur"], None, Xo, Uo, acto)
    assert np.array_equal(st, sto) and np.array_equal(sw, swo) and np.array_equal(it, ito) and np.array_equal(act, acto)
    tight = oracle.default_cfg()
    tight.tol = 1e-11
    Xi, Ui = b["xr"].copy(), U0.copy()
    ui, sti, _ = oracle.step_batch(tight, b["x0"], b["xr"], b["ur"], None, Xi, Ui)
    solved = (st == 0) & (it == 0)
    assert solved.mean() > 0.9 and (sti != 0).sum() >= 1 and solved[sti != 0].any()     # solved where the interior-point loop gave up
    both = solved & (sti == 0)
    assert _rel(u0[both], ui[both]) < 1e-6
    cfgl = oracle.default_cfg()
    for i in np.flatnonzero(solved)[::3]:
        qp = oracle.linearize(cfgl, b["x0"][i], b["xr"][i], b["ur"][i], None, b["xr"][i], U0[i])
        dxa, dua, active = R.pdas_solve(qp)
        assert np.abs(U[i] - U0[i] - dua).max() < 1e-8 and np.abs(X[i] - b["xr"][i] - dxa).max() < 1e-8, i
        assert (act[i] != 0).sum() == len(active)
    assert U[solved][..., :3].max() <= 6 and U[solved][..., 3].min() >= 0
    assert sw.max() >= 5                                                   # (the all-stages-outside instances: 20+ pins, five or six sweeps)


def test_reset_and_set_iterate_empty_the_kept_sets(ndp):
    B = 256
    b = synth.make_batch(B, seed=synth.SEED0 + 40, **MIXED)
    eng = ndp.BatchedNMPC(B)
    eng.reset(b["xr"], b["ur"])
    u_cold = eng.update(b["x0"], b["xr"], b["ur"])
    sw_cold, act = eng.active_set()
    assert act.any()
    eng.update(b["x0"], b["xr"], b["ur"])
    sw_warm, _ = eng.active_set()
    con = act.any(axis=(1, 2))
    assert sw_warm[con].mean() < sw_cold[con].mean()
    for how in ("reset", "set_iterate", "reset_device"):
        if how == "reset":
            eng.reset(b["xr"], b["ur"])
        elif how == "set_iterate":
            eng.set_iterate(b["xr"], b["ur"])
        else:
            import torch
            eng.reset_device(torch.from_numpy(b["xr"]).cuda(), torch.from_numpy(b["ur"]).cuda())
        assert not eng.active_set()[1].any(), how
        u = eng.update(b["x0"], b["xr"], b["ur"])
        assert np.array_equal(u, u_cold) and np.array_equal(eng.active_set()[0], sw_cold), how
        eng.update(b["x0"], b["xr"], b["ur"])


def test_work_list_and_in_place_agree(ndp, oracle):
    """The work list's producer runs the active-set iterations itself and lists only what needs the interior-point loop: same status,
    sweeps, sets and (to rounding) numbers as the in-place kernel -- at the reference shape and at config 5's (N = 40, 2 RTI
    iterations, the five-slot kernels with the set parked in LDS), which is also held against the twin."""
    for N, n_rti, B, kw in ((20, 1, 2051, MIXED), (40, 2, 515, HARD)):
        b = synth.make_batch(B, N=N, seed=synth.SEED0 + 5, **kw)
        res = {}
        for wq in (1, 2):
            eng = ndp.BatchedNMPC(B, N=N, n_rti=n_rti, work_queue=wq)
            assert eng.work_queue == (wq == 1)
            eng.reset(b["xr"], b["ur"])
            outs = []
            for _ in range(2):
                o = eng.update(b["x0"], b["xr"], b["ur"], raise_on_status=False, full=True)
                outs.append(o + eng.active_set())
            res[wq] = outs
            eng.close()
        for a, c in zip(res[1], res[2]):
            for k in (3, 4, 5, 6):
                assert np.array_equal(a[k], c[k]), (N, k)
            np.testing.assert_allclose(a[0], c[0], rtol=0, atol=1e-8)
        u0, X, U, st, it, sw, act = res[2][0]
        assert act.any(axis=(1, 2)).mean() > 0.1
        NS = 128
        Xo, Uo = b["xr"][:NS].copy(), b["ur"][:NS].copy()
        acto = np.zeros((NS, N, 4), dtype=np.int8)
        uo, sto, ito, swo = oracle.step_batch_as(_twin_cfg(oracle, N=N, n_rti=n_rti), b["x0"][:NS], b["xr"][:NS], b["ur"][:NS], None, Xo, Uo, acto)
        assert np.array_equal(st[:NS], sto) and np.array_equal(sw[:NS], swo) and np.array_equal(act[:NS], acto)
        easy = ito == 0
        assert easy.mean() > 0.5
        np.testing.assert_allclose(U[:NS][easy], Uo[easy], rtol=0, atol=1e-7)


def test_interior_point_always_is_untouched(ndp, oracle):
    """qp_mode 1 (what HPIPM does on every tick) neither reads nor writes a set: iteration for iteration the oracle's."""
    B = 256
    b = synth.make_batch(B, seed=synth.SEED0 + 40, **MIXED)
    eng = ndp.BatchedNMPC(B, qp_mode=1)
    eng.reset(b["xr"], b["ur"])
    u0 = eng.update(b["x0"], b["xr"], b["ur"], raise_on_status=False)
    st, it = eng.status()
    Xo, Uo = b["xr"].copy(), b["ur"].copy()
    uo, sto, ito = oracle.step_batch(oracle.default_cfg(), b["x0"], b["xr"], b["ur"], None, Xo, Uo)
    assert np.array_equal(st, sto) and np.array_equal(it, ito) and (it > 0).all()
    assert _rel(u0[st == 0], uo[st == 0]) < 1e-6
    sw, act = eng.active_set()
    assert not act.any() and not sw.any()


def test_weak_multipliers_fixture(ndp, oracle):
    """tests/golden/as_weak_multiplier_cases.npz: QPs with a bound whose multiplier is small, out of closed-loop recoveries, on which
    round 6's first form of the pins kept a wrongly signed one or cycled (tests/test_active_set.py has the story).  One batch, the kept
    sets handed in (ndp_set_active_set): every instance to 1e-9 of the exact dense-KKT solution, the exact number of active bounds, no
    interior-point iteration; the twin agrees on sweeps and sets."""
    g = np.load("tests/golden/as_weak_multiplier_cases.npz")
    B = len(g["n_active"])
    eng = ndp.BatchedNMPC(B)
    eng.set_iterate(g["X"], g["U"])
    eng.set_active_set(g["act"])
    assert np.array_equal(eng.active_set()[1], g["act"])
    u0, X, U, st, it = eng.update(g["x0"], g["xr"], g["ur"], raise_on_status=False, full=True)
    sw, act = eng.active_set()
    assert not st.any() and not it.any()
    assert np.abs(U - g["U_exact"]).max() < 1e-9 and np.abs(X - g["X_exact"]).max() < 1e-9
    assert np.array_equal((act != 0).sum(axis=(1, 2)), g["n_active"])
    Xo, Uo, acto = g["X"].copy(), g["U"].copy(), g["act"].copy()
    uo, sto, ito, swo = oracle.step_batch_as(_twin_cfg(oracle), g["x0"], g["xr"], g["ur"], None, Xo, Uo, acto)
    assert np.array_equal(sw, swo) and np.array_equal(act, acto) and not sto.any() and not ito.any()
    with pytest.raises(ndp.NdpError):
        eng.set_active_set(np.full((B, 20, 4), 2, dtype=np.int8))


def test_closed_loop_recovery_matches_twin_on_every_instance(ndp, oracle):
    """scripts/stress_active_set.py in small: 1024 vehicles recover from large initial errors over eight control periods, the plant
    (oracle RK4) driven by the DEVICE's u0, kept sets growing, shrinking and emptying.  Device and oracle twin agree on status,
    sweeps, interior-point iterations and sets on EVERY instance of every tick and on the iterate to 1e-9 (the first form of the pins
    differed on ~1 instance in 3000 here: a multiplier's sign decided by rounding noise); against the interior point at tol 1e-11:
    1e-5 (that answer's own accuracy on nearly degenerate instances, see test_mixed_workload_...)."""
    B = 1024
    kw = dict(pos_sigma=1.5, vel_sigma=3.0, quat_sigma=0.2)
    b = synth.make_batch(B, seed=1002, **kw)
    eng = ndp.BatchedNMPC(B)
    eng.reset(b["xr"], b["ur"])
    twin, tight = _twin_cfg(oracle), oracle.default_cfg()
    tight.tol = 1e-11
    Xo, Uo = b["xr"].copy(), b["ur"].copy()
    acto = np.zeros((B, 20, 4), dtype=np.int8)
    x = b["x0"].copy()
    n_con = n_ipm = 0
    for t in range(8):
        bt = synth.make_batch(B, seed=1002, t0=0.02 * t, **kw)
        Xp, Up = Xo.copy(), Uo.copy()
        u0, X, U, st, it = eng.update(x, bt["xr"], bt["ur"], raise_on_status=False, full=True)
        sw, act = eng.active_set()
        uo, sto, ito, swo = oracle.step_batch_as(twin, x, bt["xr"], bt["ur"], None, Xo, Uo, acto)
        assert not st.any() and np.array_equal(st, sto) and np.array_equal(sw, swo) and np.array_equal(it, ito) and np.array_equal(act, acto), t
        asv = it == 0                                   # (an interior-point fallback: that loop's device / oracle parity, 1e-8 typical)
        assert np.abs(U[asv] - Uo[asv]).max() < 1e-9 and np.abs(X[asv] - Xo[asv]).max() < 1e-9 and np.abs(U - Uo).max() < 1e-6, t
        ui, sti, _ = oracle.step_batch(tight, x, bt["xr"], bt["ur"], None, Xp, Up)
        ok = sti == 0
        assert _rel(u0[ok], ui[ok]) < 1e-5, t
        n_con += int(act.any(axis=(1, 2)).sum())
        n_ipm += int((it > 0).sum())
        Xo[:], Uo[:] = X, U
        x = oracle.plant_step(twin, x.copy(), u0, np.zeros((B, 3)), 0.02)
    assert n_con > 2 * B and n_ipm <= 8
