"""SURVEY 8f-3 -- the step after the path: hover-throttle estimator + actuator command.
Fixture tests/golden/throttle_golden.npz was produced by importing the reference's
hv_throttle_est package (tests/golden/make_throttle_golden.py): real reference outputs."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-12      # fp64 KF: relative tolerance on k_throttle / state (operation order follows the numpy expressions;
                 # numpy's 2x2 matmuls go through BLAS, whose fusion choices are not specified, hence not bit-exact)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "throttle_golden.npz"))


def _check(k, st, g, i):
    assert np.all(np.abs(k - g["k"][i]) <= TOL * np.abs(g["k"][i]))
    assert np.all(np.abs(st[:, 0:2] - g["x"][i]) <= TOL * np.maximum(1.0, np.abs(g["x"][i])))
    assert np.all(np.abs(st[:, 2:6] - g["P"][i].reshape(-1, 4)) <= TOL * np.maximum(1.0, np.abs(g["P"][i].reshape(-1, 4))))


def test_oracle_throttle_estimator_against_reference_fixture(oracle, gold):
    cfg = oracle.thr_default_cfg()
    assert (cfg.k_init, cfg.ts, cfg.R, cfg.mass, cfg.g) == (float(gold["k_init"]), float(gold["ts"]), float(gold["R"]),
                                                            float(gold["mass"]), float(gold["gravity"]))
    T, V = gold["vz"].shape
    st = oracle.thr_reset(cfg, V)
    for i in range(T):
        k = oracle.thr_update(cfg, st, gold["vz"][i], gold["throttle"][i])
        _check(k, st, gold, i)
        thrust = oracle.att_thrust(cfg, gold["c"][i], k)
        assert np.all(np.abs(thrust - gold["thrust"][i]) <= TOL * np.abs(gold["thrust"][i]))
    # the gate really was exercised: closed phases leave x and P untouched while the differentiator keeps running
    assert np.array_equal(gold["x"][55, 3], gold["x"][49, 3]) and not np.array_equal(gold["x"][60, 3], gold["x"][49, 3])
    assert np.array_equal(gold["x"][200, 7], gold["x"][199, 7]) and np.array_equal(gold["x"][201, 7], gold["x"][199, 7])


@pytest.mark.gpu
def test_gpu_throttle_estimator_against_reference_fixture(gold):
    import ndp_nmpc_qd_amd as ndp
    T, V = gold["vz"].shape
    eng = ndp.BatchedNMPC(V, load_mlp=False)
    eng.throttle_reset()
    for i in range(T):
        k = eng.throttle_update(gold["vz"][i], gold["throttle"][i])
        if i % 20 == 0 or i in (50, 55, 60, 100, 105, 200, 201, 202, T - 1):
            _check(k, eng.throttle_state(), gold, i)
        else:
            assert np.all(np.abs(k - gold["k"][i]) <= TOL * np.abs(gold["k"][i]))
        u0 = np.stack([gold["vz"][i], -gold["vz"][i], 0.5 * gold["vz"][i], gold["c"][i]], axis=1)
        cmd = eng.actuator_cmd(u0, k)
        assert np.array_equal(cmd[:, :3], u0[:, :3])
        assert np.all(np.abs(cmd[:, 3] - gold["thrust"][i]) <= TOL * np.abs(gold["thrust"][i]))
    assert np.array_equal(eng.actuator_cmd(u0, np.zeros(V))[:, 3], np.zeros(V))      # k_throttle == 0 -> thrust 0


@pytest.mark.gpu
def test_gpu_throttle_drop_in_class_and_large_batch(oracle, gold):
    import ndp_nmpc_qd_amd as ndp
    from ndp_nmpc_qd_amd.hv_throttle_est import HoverThrottleEstimator
    est = HoverThrottleEstimator(0.02)
    for i in range(40):
        k, x, P = est.update(float(gold["vz"][i, 0]), float(gold["throttle"][i, 0]))
        assert abs(k - gold["k"][i, 0]) <= TOL * abs(gold["k"][i, 0])
    assert x.shape == (2, 1) and P.shape == (2, 2)
    # B = 4099 vehicles against the oracle (ragged last workgroup), 30 ticks
    B = 4099
    rng = np.random.default_rng(0)
    eng = ndp.BatchedNMPC(B, load_mlp=False)
    cfg = oracle.thr_default_cfg()
    st = oracle.thr_reset(cfg, B)
    vz = np.zeros(B)
    for i in range(30):
        vz = vz + rng.normal(0, 0.05, B)
        th = rng.uniform(0.05, 1.05, B)
        k = eng.throttle_update(vz, th)
        ko = oracle.thr_update(cfg, st, vz, th)
        assert np.all(np.abs(k - ko) <= TOL * np.abs(ko))
    assert np.all(np.abs(eng.throttle_state() - st) <= TOL * np.maximum(1.0, np.abs(st)))
