"""Closed-loop behaviour: u0 from the controller drives an independent plant (numpy RK4 of the reference's
dynamics expressions, tests/ref_numpy.py) along the figure-eight reference, sliding the window by ts_nmpc = 0.02 s
per tick exactly as the reference node does (nmpc_node.py:158-183, pt_publisher.py:78-103).  Independent evidence
that the restated OCP (weights, dt scaling, sign conventions, horizon indexing) is a working tracker -- the absent
dop_sim would have provided this."""
import numpy as np
import pytest

from ndp_nmpc_qd_amd import synth
from ndp_nmpc_qd_amd.params import nmpc_params as CP
from tests import ref_numpy as R


def _windows(omega, phi, t0, N=20):
    t = t0 + CP.th_pred * np.arange(N + 1)
    pos, vel, acc, jerk = synth.figure_eight(omega[:, None], phi[:, None], t[None, :])
    xr, ur = synth.diff_flatness(pos, vel, acc, jerk)
    return np.ascontiguousarray(xr), np.ascontiguousarray(ur[:, :N])


def _plant_step(x, u, h=CP.ts_nmpc, sub=4):
    for _ in range(sub):
        x = R.rk4(x, u, None, h / sub)
    x[6:10] /= np.linalg.norm(x[6:10])
    return x


def _fly(step_fn, reset_fn, B=3, ticks=200, seed=5):
    rng = np.random.default_rng(seed)
    omega, phi = rng.uniform(0.5, 1.0, B), rng.uniform(0, 2 * np.pi, B)
    xr, ur = _windows(omega, phi, 0.0)
    x = xr[:, 0].copy()
    x[:, 0:3] += rng.normal(0, 0.15, (B, 3))          # start 15 cm off the path
    reset_fn(xr, ur)
    err = []
    for k in range(ticks):
        xr, ur = _windows(omega, phi, k * CP.ts_nmpc)
        u0 = step_fn(x, xr, ur)
        assert np.isfinite(u0).all()
        for i in range(B):
            x[i] = _plant_step(x[i].copy(), u0[i])
        pr = synth.figure_eight(omega, phi, np.full(B, (k + 1) * CP.ts_nmpc))[0]
        err.append(np.linalg.norm(x[:, 0:3] - pr, axis=1))
    return np.array(err)


def test_oracle_tracks_the_figure_eight(oracle):
    cfg = oracle.default_cfg()
    state = {}

    def reset(xr, ur):
        state["X"], state["U"] = xr.copy(), ur.copy()

    def step(x, xr, ur):
        u0, st, _ = oracle.step_batch(cfg, x, xr, ur, None, state["X"], state["U"])
        assert (st == 0).all()
        return u0

    err = _fly(step, reset)
    assert err[0].max() > 0.05                      # it did start off the path
    assert err[-50:].max() < 0.03                   # and settles to < 3 cm tracking error at up to 2 m/s
    assert err.max() < 0.4                          # without overshooting


@pytest.mark.gpu
def test_gpu_closed_loop_matches_oracle_closed_loop(oracle):
    import ndp_nmpc_qd_amd as ndp
    B = 3
    eng = ndp.BatchedNMPC(B)
    cfg = oracle.default_cfg()
    st8 = {}

    def reset_o(xr, ur):
        st8["X"], st8["U"] = xr.copy(), ur.copy()

    def step_o(x, xr, ur):
        return oracle.step_batch(cfg, x, xr, ur, None, st8["X"], st8["U"])[0]

    err_o = _fly(step_o, reset_o, B=B)
    err_g = _fly(lambda x, xr, ur: eng.update(x, xr, ur), lambda xr, ur: eng.reset(xr, ur), B=B)
    assert err_g[-50:].max() < 0.03
    np.testing.assert_allclose(err_g, err_o, atol=1e-6)     # 200 closed-loop ticks, errors do not accumulate
