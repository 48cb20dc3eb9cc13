"""QP_AUTO's active-set iterations on the input bounds (ndp_cfg.as_iter_max; rti_wave.hpp: as_check), CPU side.

The product's wave program runs on the host wave emulator (tests/emu) and is held against
  * the EXACT solution of the same box-constrained QP -- dense KKT systems + a primal-dual active-set loop written for the tests
    (tests/ref_numpy.py: pdas_solve), which shares no code with the device or the oracle;
  * the oracle's interior-point loop (qp_mode 1: what the reference's HPIPM does, nmpc_body_rate_ctl.py:71-74), at a tight tolerance
    where the two must coincide, and at the default tolerance where the interior-point answer is the less accurate one;
  * the oracle's restatement of the same active-set rule (qp_mode 0): same sweeps, same sets, same numbers.
tests/test_active_set_gpu.py repeats this through the C-ABI on the device.
"""
import numpy as np
import pytest

from ndp_nmpc_qd_amd import synth
from tests import ref_numpy as R
from tests.emu import emu as E

MIXED = dict(pos_sigma=0.5, vel_sigma=1.0, quat_sigma=0.15)        # bench.py's `mixed` workload: ~20 % of the instances hit an input bound


def _exact(oracle, b, i, X, U, N=20, f=None, use_fd=False):
    cfgo = oracle.default_cfg(N=N, use_fd=use_fd)
    qp = oracle.linearize(cfgo, b["x0"][i], b["xr"][i], b["ur"][i], None if f is None else f[i], X, U)
    dxa, dua, active = R.pdas_solve(qp)
    return qp, dxa, dua, active


def test_mixed_workload_exact_and_warm_started(oracle):
    """Four control ticks of the mixed workload, the kept set carried from tick to tick: every constrained instance is solved by the
    active-set iterations (no interior-point fallback), to 1e-9 of the exact QP solution; from an empty set in two or three sweeps,
    with the previous tick's set mostly in ONE; and the kept set IS the exact solution's active set."""
    B = 40
    cfg = E.default_cfg()
    X = U = None
    acts = [E.act_record(20) for _ in range(B)]
    cold, warm, n_con = [], [], 0
    for t in range(4):
        b = synth.make_batch(B, seed=synth.SEED0 + 40, t0=0.02 * t, **MIXED)
        if X is None:
            X, U = b["xr"].copy(), b["ur"].copy()
        for i in range(B):
            X0, U0 = X[i].copy(), U[i].copy()
            u0, st, it, *_ = E.rti_step(cfg, b["x0"][i], b["xr"][i], b["ur"][i], None, X[i], U[i], act=acts[i])
            sw, aset = E.act_view(acts[i])
            assert st == 0 and it == 0
            if sw > 1 or aset.any():
                n_con += 1
                qp, dxa, dua, active = _exact(oracle, b, i, X0, U0)
                assert max(np.abs(U[i] - U0 - dua).max(), np.abs(X[i] - X0 - dxa).max()) < 1e-9
                want = np.zeros((20, 4), dtype=np.int8)
                for v, side in active.items():
                    assert v >= 21 * 10                                   # input bounds only in this envelope
                    want.reshape(-1)[v - 21 * 10] = 1 if side == "hi" else -1
                assert np.array_equal(aset, want)
                on = aset != 0
                assert np.array_equal((U[i] == np.where(aset > 0, cfg.ubu, cfg.lbu))[on], np.ones(on.sum(), bool))   # pinned inputs sit ON the bound
                (cold if t == 0 else warm).append(sw)
    assert n_con > 25 and min(cold) >= 2 and max(cold) <= 3
    assert max(warm) <= 3 and np.mean(np.array(warm) == 1) > 0.5, np.bincount(warm)


def test_against_the_interior_point_oracle(oracle):
    """The same QPs through the oracle's interior-point loop (the reference's method).  At tol 1e-11 the two answers coincide to 1e-7
    (1e-8 typically) -- same QP, same solution.  At the default tol 1e-8 the interior-point answer itself is up to ~1e-5 / separation
    away from the exact solution on nearly degenerate instances, the active-set answer is not: it is never the
    worse of the two, and the north star's 1e-5 on u0 holds on every instance."""
    B = 64
    b = synth.make_batch(B, seed=synth.SEED0 + 40, **MIXED)
    cfg = E.default_cfg()
    worst_tight, worse, n = 0.0, 0, 0
    for i in range(B):
        X, U = b["xr"][i].copy(), b["ur"][i].copy()
        act = E.act_record(20)
        u0, st, it, *_ = E.rti_step(cfg, b["x0"][i], b["xr"][i], b["ur"][i], None, X, U, act=act)
        if E.act_view(act)[0] == 1:
            continue
        n += 1
        qp, dxa, dua, _ = _exact(oracle, b, i, b["xr"][i], b["ur"][i])
        res = {}
        for tol in (1e-11, 1e-8):
            c = oracle.default_cfg()
            c.tol = tol
            dxo, duo, sto = oracle.qp_solve(c, qp)
            assert sto.status == 0
            res[tol] = duo
        du = U - b["ur"][i]
        worst_tight = max(worst_tight, np.abs(du - res[1e-11]).max())
        worse += np.abs(du - dua).max() > np.abs(res[1e-8] - dua).max() + 1e-10
        assert np.all(np.abs(du[0] - res[1e-8][0]) <= 1e-5 * np.maximum(1.0, np.abs(U[0])))
    assert n >= 10 and worst_tight < 1e-7 and worse == 0, (n, worst_tight, worse)


def test_oracle_twin_takes_the_same_sweeps(oracle):
    """oracle qp_mode 0 restates the rule in plain C (riccati_solve with the pins' weights): sweep for sweep and set for set the
    emulated wave program's, numbers to 1e-9 -- over three ticks with the sets carried along, hard starts included."""
    B = 24
    b0 = synth.make_batch(B, seed=77, pos_sigma=1.5, vel_sigma=3.0, quat_sigma=0.2)
    cfg = E.default_cfg()
    cfgo = oracle.default_cfg()
    cfgo.qp_mode = 0
    X, U = b0["xr"].copy(), b0["ur"].copy()
    U[::4, :, 0] = 7.5                    # iterates outside the box
    U[1::4, 3, 3] = -1.0
    Xo, Uo = X.copy(), U.copy()
    acts = [E.act_record(20) for _ in range(B)]
    acto = np.zeros((B, 20, 4), dtype=np.int8)
    for t in range(3):
        b = synth.make_batch(B, seed=77, t0=0.02 * t, pos_sigma=1.5, vel_sigma=3.0, quat_sigma=0.2)
        uo, sto, ito, swo = oracle.step_batch_as(cfgo, b["x0"], b["xr"], b["ur"], None, Xo, Uo, acto)
        for i in range(B):
            u0, st, it, *_ = E.rti_step(cfg, b["x0"][i], b["xr"][i], b["ur"][i], None, X[i], U[i], act=acts[i])
            sw, aset = E.act_view(acts[i])
            assert (st, it, sw) == (sto[i], ito[i], swo[i]), (t, i)
            assert np.array_equal(aset, acto[i])
            if it == 0:
                np.testing.assert_allclose(u0, uo[i], rtol=0, atol=1e-9)
                np.testing.assert_allclose(U[i], Uo[i], rtol=0, atol=1e-9)
            else:                                                        # (the interior-point fallback: its own tests' tolerance)
                np.testing.assert_allclose(u0, uo[i], rtol=0, atol=1e-6)
            X[i], U[i] = Xo[i], Uo[i]                                    # keep the pair on one trajectory
    assert swo.max() >= 3


def test_iterate_outside_the_box_is_solved_where_the_interior_point_loop_gives_up(oracle):
    """reset() with an input far outside its box (thrust -1 at one stage): the step variable's box does not contain 0.  The oracle's
    interior-point loop runs out of iterations on these (status 4, as rounds 1-5's device did); the QP is feasible, the active-set
    iterations solve it in a few sweeps -- checked against the exact solution."""
    hit = 0
    for seed in (5, 77, 78):
        b = synth.make_batch(1, seed=seed, pos_sigma=1.5, vel_sigma=3.0, quat_sigma=0.2)
        U0 = b["ur"][0].copy()
        U0[3, 3] = -1.0
        X, U = b["xr"][0].copy(), U0.copy()
        act = E.act_record(20)
        u0, st, it, *_ = E.rti_step(E.default_cfg(), b["x0"][0], b["xr"][0], b["ur"][0], None, X, U, act=act)
        qp, dxa, dua, active = _exact(oracle, b, 0, b["xr"][0], U0)
        if any(v < 210 for v in active):
            continue                                                     # (a velocity bound in play: the fallback's case)
        sw, aset = E.act_view(act)
        assert st == 0 and it == 0 and 2 <= sw <= 6
        assert np.abs(U - U0 - dua).max() < 1e-9 and U[3, 3] >= 0.0
        cfgo = oracle.default_cfg()
        Xo, Uo = b["xr"][0].copy(), U0.copy()
        _, sto = oracle.step(cfgo, b["x0"][0], b["xr"][0], b["ur"][0], None, Xo, Uo)
        hit += sto.status == 4
    assert hit >= 1


def test_velocity_bound_hands_over_to_the_interior_point_loop(oracle):
    """A velocity box shrunk until a STATE bound is active: not this method's case -- the interior-point loop takes the QP exactly as in
    rounds 1-5 (same iterations, same answer as with the active set switched off), and the kept set is emptied."""
    b = synth.make_batch(1, seed=46, pos_sigma=1.5, vel_sigma=3.0, quat_sigma=0.2)
    out = {}
    for as_max in (8, 0):
        cfg = E.default_cfg(as_iter_max=as_max)
        for i in range(3):
            cfg.lbv[i], cfg.ubv[i] = -3.0, 3.0
        X, U = b["xr"][0].copy(), b["ur"][0].copy()
        act = E.act_record(20)
        act[4:12] = 1                                                    # a stale kept set
        u0, st, it, *_ = E.rti_step(cfg, b["x0"][0], b["xr"][0], b["ur"][0], None, X, U, act=act)
        out[as_max] = (u0, st, it, X, U)
        if as_max:
            assert not E.act_view(act)[1].any()
    assert out[8][1] == out[0][1] == 0 and out[8][2] == out[0][2] > 0
    np.testing.assert_allclose(out[8][0], out[0][0], rtol=0, atol=1e-12)
    np.testing.assert_allclose(out[8][3], out[0][3], rtol=0, atol=1e-12)


def test_a_wrong_kept_set_is_repaired(oracle):
    """The warm start is a guess: pins that have nothing to do with the QP (thrust pinned high over half the horizon, a rate pinned low)
    are released where their multipliers say so and the exact solution comes out all the same.  A set so wrong that its solution
    leaves the VELOCITY box (full thrust over the whole horizon) is the fallback's case: the interior-point loop, same solution."""
    b = synth.make_batch(4, seed=synth.SEED0 + 40, **MIXED)
    for i in range(4):
        qp, dxa, dua, active = _exact(oracle, b, i, b["xr"][i], b["ur"][i])
        for wrong in ("some", "all"):
            X, U = b["xr"][i].copy(), b["ur"][i].copy()
            act = E.act_record(20)
            aset = E.act_view(act)[1]
            aset[(slice(2, 12) if wrong == "some" else slice(None)), 3] = 1
            aset[5:9, 0] = -1
            u0, st, it, *_ = E.rti_step(E.default_cfg(), b["x0"][i], b["xr"][i], b["ur"][i], None, X, U, act=act)
            assert st == 0
            if wrong == "some":
                assert it == 0 and E.act_view(act)[0] >= 2
                assert np.abs(U - b["ur"][i] - dua).max() < 1e-9
                assert (E.act_view(act)[1] != 0).sum() == len(active)
            else:
                assert it > 0 and not E.act_view(act)[1].any()
                assert np.abs(U - b["ur"][i] - dua).max() < 1e-5


@pytest.mark.parametrize("N", [2, 5, 17, 27, 31, 45])
def test_any_horizon(oracle, N):
    """Run-time horizons: three-slot kernels (N <= 27: the set in registers) and five-slot kernels (the set parked in LDS), partial last
    rounds of the 4N input bounds; hard starts (2 .. 14 inputs on their bounds at the solution, up to six sweeps)."""
    n = 0
    for seed in (2, 3, 4, 5):
        b = synth.make_batch(1, N=N, seed=seed, pos_sigma=1.5, vel_sigma=3.0, quat_sigma=0.25)
        X, U = b["xr"][0].copy(), b["ur"][0].copy()
        act = E.act_record(N)
        u0, st, it, *_ = E.rti_step(E.default_cfg(N=N), b["x0"][0], b["xr"][0], b["ur"][0], None, X, U, act=act)
        qp, dxa, dua, active = _exact(oracle, b, 0, b["xr"][0], b["ur"][0], N=N)
        assert st == 0 and it == 0
        assert np.abs(U - b["ur"][0] - dua).max() < 1e-8
        assert (E.act_view(act)[1] != 0).sum() == len(active)
        n += len(active)
    assert n >= 7


def test_two_rti_iterations_carry_the_set(oracle):
    """BASELINE config 5's shape (N = 40, 2 RTI iterations, the five-slot kernel): the second iteration's QP starts from the first one's
    set; both iterations against the oracle twin."""
    N = 40
    cfgo = oracle.default_cfg(N=N, n_rti=2)
    cfgo.qp_mode = 0
    n = 0
    for seed in (2, 3, 4):
        b = synth.make_batch(1, N=N, seed=seed, pos_sigma=1.5, vel_sigma=3.0, quat_sigma=0.25)
        X, U = b["xr"][0].copy(), b["ur"][0].copy()
        Xo, Uo = b["xr"].copy(), b["ur"].copy()
        act, acto = E.act_record(N), np.zeros((1, N, 4), dtype=np.int8)
        u0, st, it, *_ = E.rti_step(E.default_cfg(N=N, n_rti=2), b["x0"][0], b["xr"][0], b["ur"][0], None, X, U, act=act)
        uo, sto, ito, swo = oracle.step_batch_as(cfgo, b["x0"], b["xr"], b["ur"], None, Xo, Uo, acto)
        sw, aset = E.act_view(act)
        assert (st, it, sw) == (sto[0], ito[0], swo[0]) and np.array_equal(aset, acto[0])
        np.testing.assert_allclose(U, Uo[0], rtol=0, atol=1e-8)
        n += int(aset.any()) + (sw >= 4)            # (>= 4 sweeps over two QPs: the second one re-solved with a changed set)
    assert n >= 3


def test_work_list_producer_defers_what_its_first_solve_does_not_settle(oracle):
    """The work list's producer (run<DEFER = true>) makes ONE solve, with the kept set's pins: an instance that settles it is finished
    in place exactly as the in-place kernel finishes it (a constrained instance whose set still holds included); any other is handed
    back untouched -- the consumer (run<false>) then iterates on the set, and on to the interior-point loop if need be."""
    b = synth.make_batch(24, seed=synth.SEED0 + 40, **MIXED)
    n_def = n_warm = 0
    for i in range(24):
        X0, U0 = b["xr"][i].copy(), b["ur"][i].copy()
        act = E.act_record(20)
        Xi, Ui = X0.copy(), U0.copy()
        u0i, sti, iti, *_ = E.rti_step(E.default_cfg(), b["x0"][i], b["xr"][i], b["ur"][i], None, Xi, Ui, act=act)
        sw = E.act_view(act)[0]
        Xd, Ud = X0.copy(), U0.copy()
        deferred, u0d, std, itd = E.rti_step_defer(E.default_cfg(), b["x0"][i], b["xr"][i], b["ur"][i], None, Xd, Ud)
        assert deferred == (sw > 1 or iti > 0)                            # (no kept set here: cold)
        if deferred:
            n_def += 1
            assert np.array_equal(Xd, X0) and np.array_equal(Ud, U0) and np.isnan(u0d).all() and (std, itd) == (-7, -7)
        else:
            assert (std, itd) == (0, 0) and np.array_equal(Ud, Ui) and np.array_equal(Xd, Xi)
        if E.act_view(act)[1].any():                                      # second tick, the set kept: the producer finishes it itself
            Xw, Uw = Xi.copy(), Ui.copy()
            Xp, Up = Xi.copy(), Ui.copy()
            actw = act.copy()
            u0w, stw, itw, *_ = E.rti_step(E.default_cfg(), b["x0"][i], b["xr"][i], b["ur"][i], None, Xw, Uw, act=actw)
            dfr, u0p, stp, itp = E.rti_step_defer(E.default_cfg(), b["x0"][i], b["xr"][i], b["ur"][i], None, Xp, Up, act=act.copy())
            assert dfr == (E.act_view(actw)[0] > 1 or itw > 0)
            if not dfr:
                n_warm += 1
                assert np.array_equal(Up, Uw) and np.array_equal(u0p, u0w)
    assert n_def >= 3 and n_warm >= 2


def test_switched_off_is_rounds_1_to_5(oracle):
    """as_iter_max = 0: early exit only auto_margin inside the box, else the interior-point loop, iteration for iteration the
    always-iterating oracle's."""
    b = synth.make_batch(8, seed=synth.SEED0 + 40, **MIXED)
    cfgo = oracle.default_cfg()
    n = 0
    for i in range(8):
        X, U = b["xr"][i].copy(), b["ur"][i].copy()
        Xo, Uo = X.copy(), U.copy()
        act = E.act_record(20)
        u0, st, it, *_ = E.rti_step(E.default_cfg(as_iter_max=0), b["x0"][i], b["xr"][i], b["ur"][i], None, X, U, act=act)
        uo, sto = oracle.step(cfgo, b["x0"][i], b["xr"][i], b["ur"][i], None, Xo, Uo)
        assert not E.act_view(act)[1].any()                              # nothing kept, nothing written
        if it > 0:
            n += 1
            assert it == sto.ipm_iters
            np.testing.assert_allclose(u0, uo, rtol=0, atol=1e-7)
    assert n >= 1


def test_weak_multipliers_fixture_twin_and_emulator(oracle):
    """tests/golden/as_weak_multiplier_cases.npz (scripts/make_as_weak_fixture.py): QPs out of closed-loop recoveries with a bound whose
    multiplier is small (|lambda| ~ 1e-3 .. 1e-1).  Round 6's first form read a pin's multiplier off du - d at as_gamma = 1e12 -- a
    difference of two numbers that agree to 12 digits, right to ~1e-3 in lambda -- and on five of these eleven kept a pin whose
    multiplier was negative (status 0, 2.5e-4 .. 1.6e-2 off the QP's solution) or cycled into the interior-point loop.  The
    re-centred pins (rti_wave.hpp: as_apply; oracle: qp_solve_ws) return delta = lambda / as_gamma itself: oracle twin and wave
    program on every case to 1e-9 of the exact solution, the exact active set, no interior-point iteration, the same sweeps."""
    g = np.load("tests/golden/as_weak_multiplier_cases.npz")
    n = len(g["n_active"])
    twin = oracle.default_cfg()
    twin.qp_mode = 0
    cfg = E.default_cfg()
    cfgl = oracle.default_cfg()
    for k in range(n):
        Xo, Uo, acto = g["X"][k][None].copy(), g["U"][k][None].copy(), g["act"][k][None].copy()
        uo, sto, ito, swo = oracle.step_batch_as(twin, g["x0"][k][None], g["xr"][k][None], g["ur"][k][None], None, Xo, Uo, acto)
        assert sto[0] == 0 and ito[0] == 0, k
        assert np.abs(Uo[0] - g["U_exact"][k]).max() < 1e-9 and np.abs(Xo[0] - g["X_exact"][k]).max() < 1e-9, k
        assert (acto != 0).sum() == g["n_active"][k], k
        X, U = g["X"][k].copy(), g["U"][k].copy()
        act = E.act_record(20)
        E.act_view(act)[1][:] = g["act"][k]
        u0, st, it, *_ = E.rti_step(cfg, g["x0"][k], g["xr"][k], g["ur"][k], None, X, U, act=act)
        sw, aset = E.act_view(act)
        assert st == 0 and it == 0 and sw == swo[0], (k, st, it, sw, swo[0])
        assert np.abs(U - g["U_exact"][k]).max() < 1e-9 and np.abs(X - g["X_exact"][k]).max() < 1e-9, k
        assert np.array_equal(aset, acto[0]), k
        if k % 4 == 0:                         # the fixture's exact solutions, made again
            qp = oracle.linearize(cfgl, g["x0"][k], g["xr"][k], g["ur"][k], None, g["X"][k], g["U"][k])
            dxa, dua, active = R.active_set_solve(qp)
            assert np.abs(g["U"][k] + dua - g["U_exact"][k]).max() < 1e-12 and len(active) == g["n_active"][k]
