"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C-ABI, against the
CPU oracle on identical seeded inputs, against the committed reference-generated MLP fixtures, and
through size-independent properties at BASELINE.json's full batch size.

Tolerance (BASELINE.json north_star): |du0| <= 1e-5 * max(1, |u0|) after the same number of SQP-RTI
iterations; the fp64 device path is expected (and asserted) to sit orders of magnitude inside that.
"""
import ctypes as C
import os
import time

import numpy as np
import pytest

from ndp_nmpc_qd_amd import synth

pytestmark = pytest.mark.gpu

RTOL_U = 1e-5
# QP_AUTO as rounds 1-5 had it (early exit auto_margin inside the box, else the interior-point loop) -- the tests below that are ABOUT
# the interior-point code in the automatic mode (iteration counts against the always-iterating oracle, the work list that re-balances
# interior-point solves) switch the active-set iterations off; tests/test_active_set_gpu.py covers the default.
LEGACY = dict(as_iter_max=0)


def _assert_u(u, uo, tol=RTOL_U):
    bad = np.abs(u - uo) > tol * np.maximum(1.0, np.abs(uo))
    assert not bad.any(), f"{bad.sum()} control entries outside {tol}: max abs {np.abs(u - uo).max()}"


@pytest.fixture(scope="module")
def ndp():
    import ndp_nmpc_qd_amd
    return ndp_nmpc_qd_amd


def _oracle_batch(oracle, b, N=20, n_rti=1, use_fd=False, f=None, X=None, U=None):
    cfg = oracle.default_cfg(N=N, n_rti=n_rti, use_fd=use_fd)
    X = b["xr"].copy() if X is None else X
    U = b["ur"].copy() if U is None else U
    u0, st, it = oracle.step_batch(cfg, b["x0"], b["xr"], b["ur"], f, X, U)
    return u0, st, it, X, U


def test_mfma_register_maps(ndp):
    """v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64 operand / result layouts are what rti_wave.hpp (and the emulator) assume."""
    from ndp_nmpc_qd_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(0)
    A, Bm, Cm = rng.integers(-9, 9, (16, 4)).astype(float), rng.integers(-9, 9, (4, 16)).astype(float), \
        rng.integers(-9, 9, (16, 16)).astype(float)
    a = np.array([A[l & 15, l >> 4] for l in range(64)])
    b = np.array([Bm[l >> 4, l & 15] for l in range(64)])
    c = np.array([[Cm[(l >> 4) + 4 * r, l & 15] for l in range(64)] for r in range(4)])
    d = np.zeros(832)
    assert lib.ndp_debug_mfma_probe(_lib.ptr(a), _lib.ptr(b), _lib.ptr(np.ascontiguousarray(c)), _lib.ptr(d)) == 0
    D = A @ Bm + Cm
    got = d[:256].reshape(4, 64)
    for r in range(4):
        for l in range(64):
            assert got[r, l] == D[(l >> 4) + 4 * r, l & 15], (r, l)
    np.testing.assert_allclose(d[256:320], a[37] + b.sum() + a.min() + a.max(), atol=0)
    # v_mfma_f64_4x4x4_4b_f64 on the same registers: block q = (l >> 2) & 3 multiplies A rows 4q..4q+3 (lane i + 4q + 16k)
    # with B columns 4q..4q+3 (lane j + 4q + 16k); D_q[i][j] lands in lane j + 4q + 16i (tests/emu/wave_emu.hpp: Wave::mfma4)
    for l in range(64):
        j, q, i = l & 3, (l >> 2) & 3, l >> 4
        assert d[320 + l] == A[4 * q + i, :] @ Bm[:, 4 * q + j] + c[0][l], l
    for q in range(4):      # DPP row_newbcast:4q
        for l in range(64):
            assert d[384 + 64 * q + l] == a[(l & 48) + 4 * q], (q, l)
    for n in (1, 2, 3):     # DPP row_ror:4n -- lane l takes the value of lane (l - 4n) mod 16 of its row (tests/emu/wave_emu.hpp: Wave::rowror4)
        for l in range(64):
            assert d[640 + 64 * (n - 1) + l] == a[(l & 48) | ((l - 4 * n) & 15)], (n, l)


@pytest.mark.parametrize("qp_mode", [0, 1])
def test_batch_matches_oracle(ndp, oracle, qp_mode):
    B = 256
    b = synth.make_batch(B, seed=synth.SEED0 + 2)
    eng = ndp.BatchedNMPC(B, qp_mode=qp_mode)
    eng.reset(b["xr"], b["ur"])
    u0 = eng.update(b["x0"], b["xr"], b["ur"])
    u0o, sto, ito, Xo, Uo = _oracle_batch(oracle, b)
    st, it = eng.status()
    assert (st == 0).all() and (sto == 0).all()
    _assert_u(u0, u0o, 1e-8)
    X, U = eng.get_iterate()
    np.testing.assert_allclose(X, Xo, atol=1e-8)
    np.testing.assert_allclose(U, Uo, atol=1e-8)
    if qp_mode == 1:
        assert (it == ito).all()
    # second and third control tick from the persistent device iterate (no shift)
    for _ in range(2):
        u0 = eng.update(b["x0"], b["xr"], b["ur"])
        u0o, *_ = _oracle_batch(oracle, b, X=Xo, U=Uo)
        _assert_u(u0, u0o, 1e-8)


def test_active_bounds_and_infeasible_start(ndp, oracle):
    """Large initial errors (input bounds active) and iterates outside the box: interior point on the device."""
    B = 64
    b = synth.make_batch(B, seed=77, pos_sigma=1.5, vel_sigma=3.0, quat_sigma=0.2)
    U0 = b["ur"].copy()
    U0[::4, :, 0] = 7.5         # > w_max for a quarter of the instances
    U0[1::4, 3, 3] = -1.0       # < c_min
    eng = ndp.BatchedNMPC(B, **LEGACY)
    eng.set_iterate(b["xr"], U0)
    u0 = eng.update(b["x0"], b["xr"], b["ur"], raise_on_status=False)
    u0o, sto, ito, Xo, Uo = _oracle_batch(oracle, b, U=U0.copy())
    st, it = eng.status()
    assert (ito > 0).sum() > B // 2          # the case exercises the inequality path
    assert np.array_equal(st, sto)
    ok = sto == 0
    assert ok.sum() > B // 2
    ran = ok & (it > 0)                      # auto mode: 0 iterations where the exact early exit fired
    assert ran.sum() > B // 2 and np.array_equal(it[ran], ito[ran])
    _assert_u(u0[ok], u0o[ok], 1e-6)
    X, U = eng.get_iterate()
    np.testing.assert_allclose(U[ok], Uo[ok], atol=1e-5)
    assert U[ok][..., :3].max() <= 6 + 1e-6 and U[ok][..., 3].min() >= -1e-6


def test_active_state_bounds_at_a_tight_tolerance_on_the_device(ndp, oracle):
    """VERDICT r3 #5 through the C-ABI: velocity box shrunk until state bounds are active (barrier terms 1e9 .. 1e13).  While a state
    bound's barrier term exceeds refine_gamma the kernel factorises Lam = L D L' in every lane, applies Lam^-1 by substitution and
    refines every solve twice (ndp_cfg.ipm_refine).  On all ten feasible problems of seeds 40..79, at the default tolerance and at
    1e-10: status 0, the oracle's iteration count, the oracle's step to 1e-9, the exact active-set answer of the same QP within the
    termination bound.  Round 3's loop (ipm_refine = 0) ends seven of them in status 4 and the rest 4e-6 .. 3e-5 off.
    (CPU twin on the emulator: tests/test_wave_program_emulated.py; the whole table: scripts/refine_probe.py.)"""
    from tests import ref_numpy as R
    worse = 0
    for seed in (46, 47, 50, 54, 57, 58, 64, 69, 77, 78):
        b = synth.make_batch(1, seed=seed, pos_sigma=1.5, vel_sigma=3.0, quat_sigma=0.2)
        x0, xr, ur = b["x0"][0], b["xr"][0], b["ur"][0]
        cfgo = oracle.default_cfg()
        cfgo.qp_mode = 1
        qp = oracle.linearize(cfgo, x0, xr, ur, None, xr.copy(), ur.copy())
        dxf, _, _ = oracle.qp_solve(cfgo, qp)
        box = 0.8 * np.abs((xr + dxf)[4:20, 3:6]).max()
        for tol in (1e-8, 1e-10):
            err = {}
            for refine in (2, 0):
                eng = ndp.BatchedNMPC(1, qp_mode=1, tol=tol, ipm_refine=refine, lbv=[-box] * 3, ubv=[box] * 3)
                eng.reset(b["xr"], b["ur"])
                u0, X, U, st, it = eng.update(b["x0"], b["xr"], b["ur"], raise_on_status=False, full=True)
                eng.close()
                cfgo = oracle.default_cfg()
                cfgo.qp_mode, cfgo.tol, cfgo.refine = 1, tol, refine
                for i in range(3):
                    cfgo.lbv[i], cfgo.ubv[i] = -box, box
                qpb = oracle.linearize(cfgo, x0, xr, ur, None, xr.copy(), ur.copy())
                dxa, dua, active = R.pdas_solve(qpb)
                assert sum(1 for v in active if v < 21 * 10) >= 1                     # a state bound IS active at the solution
                if st[0] != 0:
                    assert refine == 0 and st[0] == 4, (seed, tol, refine, st[0])
                    assert np.array_equal(X[0], xr) and np.array_equal(U[0], ur)      # reported, iterate untouched
                    err[refine] = np.inf
                    continue
                err[refine] = max(np.abs(X[0] - xr - dxa).max(), np.abs(U[0] - ur - dua).max())
                if refine:
                    Xo, Uo = xr.copy(), ur.copy()
                    u0o, sto = oracle.step(cfgo, x0, xr, ur, None, Xo, Uo)
                    assert sto.status == 0 and it[0] == sto.ipm_iters, (seed, tol, it[0], sto.ipm_iters)
                    assert max(np.abs(X[0] - Xo).max(), np.abs(U[0] - Uo).max()) <= 1e-9, (seed, tol)
                    sep = min(1.0, R.separation(qpb, dxa, dua, active))
                    assert err[2] <= max(1e-7, 40 * 4e-6 * (tol / 1e-8) / sep), (seed, tol, err[2], sep)
            worse += err[0] > 100 * max(err[2], 1e-11)
    assert worse >= 14, worse


def test_ndp_update_with_force(ndp, oracle):
    B = 64
    b = synth.make_batch(B, seed=5)
    f = np.random.default_rng(3).normal(0, 2.0, (B, 21, 3)).astype(np.float32)
    eng = ndp.BatchedNMPC(B, disturbance=True, load_mlp=False)
    eng.reset(b["xr"], b["ur"])
    u0 = eng.update(b["x0"], b["xr"], b["ur"], f=f)
    u0o, *_ = _oracle_batch(oracle, b, use_fd=True, f=f)
    _assert_u(u0, u0o, 1e-8)


def test_downwash_mlp_against_reference_fixture(ndp, mlp_golden):
    """DownwashNN.update-shaped fixture produced by importing the reference network (tests/golden)."""
    other, ego, want = mlp_golden["other"], mlp_golden["ego"], mlp_golden["f_update"]
    eng = ndp.BatchedNMPC(other.shape[0], disturbance=True)
    f = eng.downwash(other, ego)
    assert f.dtype == np.float32
    tol = 1e-5 * np.maximum(1.0, np.abs(want))      # fp32 network, summation order differs from torch's
    assert np.all(np.abs(f - want) <= tol), np.abs(f - want).max()
    # the 512 single rows (incl. SURVEY C.1) through the same kernel: put z into the position/velocity slots
    z, fz = mlp_golden["z"], mlp_golden["f"]
    rows = z.shape[0]
    Bz = (rows + 20) // 21
    o2 = np.zeros((Bz * 21, 10))
    o2[:rows, 0:6] = z
    eng2 = ndp.BatchedNMPC(Bz, disturbance=True)
    f2 = eng2.downwash(o2.reshape(Bz, 21, 10), np.zeros((Bz, 21, 10))).reshape(-1, 3)[:rows]
    assert np.all(np.abs(f2 - fz) <= 1e-5 * np.maximum(1.0, np.abs(fz)))


def _mlp_fp64(blob, z):
    """The network evaluated in float64 with the shipped fp32 weights: the exact value both fp32 evaluations approximate."""
    o, out = 0, []
    for n_out, n_in in ((128, 6), (64, 128), (128, 64), (3, 128)):
        W = blob[o:o + n_out * n_in].reshape(n_out, n_in).astype(np.float64)
        o += n_out * n_in
        out.append((W, blob[o:o + n_out].astype(np.float64)))
        o += n_out
    h = z.astype(np.float64)
    for i, (W, bias) in enumerate(out):
        h = h @ W.T + bias
        if i < 3:
            h = np.maximum(h, 0.0)
    return h


@pytest.mark.parametrize("scale", [1, 10, 100])
def test_downwash_mlp_error_against_the_fp32_noise_floor(ndp, mlp_golden, mlp_blob, scale):
    """How far the device MLP (fp16 pair splitting on layers 2-3) is from the reference network, measured against what fp32
    arithmetic itself can deliver.  Ground truth = the shipped weights evaluated in float64.  Inside the training envelope
    (scale 1) the reference's own torch-fp32 output is 3.7e-6 from the truth and the device sits inside the 1e-5 bar against
    torch.  At 10x / 100x the envelope (hidden activations up to ~60 / ~660, two orders below the device's fp16 activation cap
    of 65000) the reference's fp32 output itself is 3e-5 / 7e-5 from the truth (cancellation in the last layer): a 1e-5 bar
    against it is below its own rounding noise there -- two fp32 evaluations in different summation order (torch, the plain-C
    oracle) differ by 1e-5 / 6e-5.  What is asserted instead: the device is no further from the truth than 2.5x the
    reference's own fp32 error, at every scale."""
    key = "" if scale == 1 else f"_x{scale}"
    z, f_torch = mlp_golden["z" + key], mlp_golden["f" + key]
    rows = (z.shape[0] // 21) * 21
    z, f_torch = z[:rows], f_torch[:rows]
    Bz = rows // 21
    o2 = np.zeros((Bz, 21, 10))
    o2[:, :, 0:6] = z.reshape(Bz, 21, 6)
    f = ndp.BatchedNMPC(Bz, disturbance=True).downwash(o2, np.zeros((Bz, 21, 10))).reshape(-1, 3)
    truth = _mlp_fp64(mlp_blob, z)
    den = np.maximum(1.0, np.abs(truth))
    e_dev, e_torch = np.abs(f - truth) / den, np.abs(f_torch - truth) / den
    e_vs_torch = np.abs(f - f_torch) / np.maximum(1.0, np.abs(f_torch))
    print(f"scale {scale}: device vs truth {e_dev.max():.2e}, torch fp32 vs truth {e_torch.max():.2e}, device vs torch {e_vs_torch.max():.2e}")
    assert e_dev.max() <= 2.5 * e_torch.max()
    if scale == 1:
        assert e_vs_torch.max() <= 1e-5
    else:
        assert e_vs_torch.max() <= 2e-4 and float(mlp_golden[f"hmax_x{scale}"].max()) < 6500.0


def test_fused_downwash_step_and_gate(ndp, oracle, mlp_blob):
    """update with neighbour windows: gate (ego odometry xy, strict <) + MLP + NDP solve, all on the device."""
    B = 128
    b = synth.make_batch(B, seed=synth.SEED0 + 3, downwash=True)
    eng = ndp.BatchedNMPC(B, disturbance=True)
    eng.reset(b["xr"], b["ur"])
    u0 = eng.update(b["x0"], b["xr"], b["ur"], other=b["other"], ego_xy=b["ego_xy"])
    f_or = oracle.downwash_batch(mlp_blob, b["other"], b["xr"], b["ego_xy"])
    gate_on = (np.abs(f_or).max(axis=(1, 2)) > 0)
    assert 0.15 < gate_on.mean() < 0.6           # roughly a third of the instances pass the 1 m gate
    u0o, *_ = _oracle_batch(oracle, b, use_fd=True, f=f_or)
    _assert_u(u0, u0o, 1e-6)                     # force is fp32: device and oracle MLP differ by ~1e-6 relative
    f_dev = eng.downwash(b["other"], b["xr"], b["ego_xy"])
    assert np.array_equal(f_dev[~gate_on], np.zeros_like(f_dev[~gate_on]))
    assert np.all(np.abs(f_dev - f_or) <= 1e-5 * np.maximum(1.0, np.abs(f_or)))
    # exactly-on-the-rim case is gated OFF (strict <, ndp_nmpc_leader_node.py:65-68); values exact in binary
    oth = b["other"].copy()
    oth[0:2, 0, 0:2] = [1.0, 2.0]
    exy = oth[:, 0, 0:2].copy()
    exy[0] = [1.0, 3.0]                      # d^2 == 1.0 exactly -> off
    exy[1] = [1.0, 2.9999999999999996]       # one ulp inside -> on
    fr = eng.downwash(oth, b["xr"], exy)
    assert np.all(fr[0] == 0)
    assert np.abs(fr[1]).max() > 0
    fro = oracle.downwash_batch(mlp_blob, oth, b["xr"], exy)
    assert np.array_equal(fr == 0, fro == 0)


def test_long_horizon_two_rti_iterations(ndp, oracle):
    """BASELINE config 5 shape: N=40, 2 RTI iterations per call."""
    B = 32
    b = synth.make_batch(B, N=40, seed=11)
    eng = ndp.BatchedNMPC(B, N=40, n_rti=2)
    eng.reset(b["xr"], b["ur"])
    u0 = eng.update(b["x0"], b["xr"], b["ur"])
    u0o, sto, *_ = _oracle_batch(oracle, b, N=40, n_rti=2)
    assert (sto == 0).all()
    _assert_u(u0, u0o, 1e-8)


def test_full_size_batch_1024(ndp, oracle):
    """BASELINE configs[1]/[2] size: every one of the 1024 instances against the oracle, plus properties."""
    B = 1024
    b = synth.make_batch(B, seed=synth.SEED0 + 3, downwash=True)
    eng = ndp.BatchedNMPC(B, disturbance=True)
    eng.reset(b["xr"], b["ur"])
    f = eng.downwash(b["other"], b["xr"], b["ego_xy"])
    u0 = eng.update(b["x0"], b["xr"], b["ur"], other=b["other"], ego_xy=b["ego_xy"])
    u0o, sto, ito, Xo, Uo = _oracle_batch(oracle, b, use_fd=True, f=f)
    _assert_u(u0, u0o, 1e-8)                     # same fp32 force on both sides -> fp64-level agreement
    X, U = eng.get_iterate()
    # properties independent of any oracle: x0 equality holds after the full step; u0 is U[0]; inputs in the box
    np.testing.assert_allclose(X[:, 0, :], b["x0"], atol=1e-9)
    assert np.array_equal(u0, U[:, 0, :])
    assert (U[..., :3] <= 6 + 1e-9).all() and (U[..., :3] >= -6 - 1e-9).all()
    assert (U[..., 3] >= -1e-9).all() and (U[..., 3] <= 9.81 / 0.36 + 1e-9).all()
    # permutation equivariance: instances are independent
    perm = np.random.default_rng(0).permutation(B)
    eng2 = ndp.BatchedNMPC(B, disturbance=True)
    eng2.reset(b["xr"][perm], b["ur"][perm])
    u0p = eng2.update(b["x0"][perm], b["xr"][perm], b["ur"][perm], other=b["other"][perm], ego_xy=b["ego_xy"][perm])
    assert np.array_equal(u0p, u0[perm])


def test_ragged_batches(ndp, oracle):
    """Batch sizes that do not fill a 4-wave workgroup or a 32-row MLP tile."""
    for B in (1, 3, 5, 33):
        b = synth.make_batch(B, seed=100 + B, downwash=True)
        eng = ndp.BatchedNMPC(B, disturbance=True)
        eng.reset(b["xr"], b["ur"])
        f = eng.downwash(b["other"], b["xr"], b["ego_xy"])
        u0 = eng.update(b["x0"], b["xr"], b["ur"], other=b["other"], ego_xy=b["ego_xy"])
        u0o, *_ = _oracle_batch(oracle, b, use_fd=True, f=f)
        _assert_u(u0, u0o, 1e-8)


def test_lds_image_matches_oracle_linearisation(ndp, oracle):
    b = synth.make_batch(1, seed=31)
    N = 20
    rng = np.random.default_rng(0)
    X = b["xr"] + rng.normal(0, 0.05, (1, N + 1, 10))
    U = b["ur"] + rng.normal(0, 0.2, (1, N, 4))
    eng = ndp.BatchedNMPC(1)
    eng.set_iterate(X, U)
    _, lds = eng.update_debug(b["x0"], b["xr"], b["ur"])
    qp = oracle.linearize(oracle.default_cfg(), b["x0"][0], b["xr"][0], b["ur"][0], None, X[0], U[0])
    from ndp_nmpc_qd_amd import _lib
    L = _lib.lds_layout(N)
    MB, CB, MS, CS = L["MB"], L["CB"], L["MB_STRIDE"], L["CB_STRIDE"]
    for k in range(N):
        blk = lds[MB + k * MS: MB + (k + 1) * MS]
        np.testing.assert_allclose(blk[0:48].reshape(6, 8)[:, 0:4], qp["A"][k][0:6, 6:10], atol=1e-12)
        np.testing.assert_allclose(blk[0:48].reshape(6, 8)[:, 4:8], qp["B"][k][0:6, :], atol=1e-12)
        np.testing.assert_allclose(blk[48:76].reshape(4, 7)[:, 0:4], qp["A"][k][6:10, 6:10], atol=1e-12)
        np.testing.assert_allclose(blk[76:86], qp["b"][k], atol=1e-12)
        cb = lds[CB + k * CS: CB + (k + 1) * CS]
        np.testing.assert_allclose(cb[16:26], qp["q"][k], atol=1e-10)
        np.testing.assert_allclose(cb[26:30], qp["r"][k], atol=1e-10)


@pytest.mark.parametrize("cls", ["nmpc", "ndp"])
def test_lds_image_matches_the_reference_expressions(ndp, cls):
    """The same image against tests/golden/ocp_golden.npz -- QP data computed by RUNNING THE REFERENCE'S OWN model expressions
    (nmpc_body_rate_ctl.py:147-195, ndp_nmpc_body_rate_ctl.py:151-197, under casadi / acados_template stand-ins:
    tests/golden/make_ocp_golden.py): the ERK4 map of its f_expl_expr with exact sensitivities (A_k, B_k, b_k), and the
    Gauss-Newton gradients from its cost_y_expr and W (stage cost scaled by the interval: [acados-knowledge]).  No oracle in between."""
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ocp_golden.npz"))
    N, tag, fd = int(G["nmpc_N"]), "lin_%s_" % cls, cls == "ndp"
    eng = ndp.BatchedNMPC(1, disturbance=fd, load_mlp=False)
    eng.set_iterate(G["lin_X"][None], G["lin_U"][None])
    _, lds = eng.update_debug(G["lin_x0"][None], G["lin_xr"][None], G["lin_ur"][None], f=G["lin_f"][None] if fd else None)
    from ndp_nmpc_qd_amd import _lib
    L = _lib.lds_layout(N)
    MB, CB, MS, CS = L["MB"], L["CB"], L["MB_STRIDE"], L["CB_STRIDE"]
    W, dt = G["nmpc_W"], float(G["nmpc_tf"]) / N
    A, Bm, b = G[tag + "A"], G[tag + "B"], G[tag + "b"]
    for k in range(N):
        blk = lds[MB + k * MS: MB + (k + 1) * MS]
        np.testing.assert_allclose(blk[0:48].reshape(6, 8)[:, 0:4], A[k][0:6, 6:10], atol=1e-12)
        np.testing.assert_allclose(blk[0:48].reshape(6, 8)[:, 4:8], Bm[k][0:6, :], atol=1e-12)
        np.testing.assert_allclose(blk[48:76].reshape(4, 7)[:, 0:4], A[k][6:10, 6:10], atol=1e-12)
        np.testing.assert_allclose(blk[48:76].reshape(4, 7)[:, 4:7], Bm[k][6:10, 0:3], atol=1e-12)
        np.testing.assert_allclose(blk[76:86], b[k], atol=1e-12)
        J, res = G[tag + "Jy"][k], G[tag + "res"][k]
        g = dt * J.T @ W @ res
        cb = lds[CB + k * CS: CB + (k + 1) * CS]
        np.testing.assert_allclose(cb[16:26], g[:10], atol=1e-10)
        np.testing.assert_allclose(cb[26:30], g[10:], atol=1e-10)
    # the part of [A B] the image does not store is structural: p / v columns [I; 0; 0], [hI; I; 0]; no dependence of q+ on (p, v, c)
    assert not A[:, 6:10, 0:6].any() and not Bm[:, 6:10, 3].any()


def test_reference_api_drop_in(ndp, oracle, mlp_golden):
    """The three reference classes, used exactly as nmpc_node.py / ndp_nmpc_leader_node.py use them."""
    from ndp_nmpc_qd_amd.dnwash_nn_est import DownwashNN
    from ndp_nmpc_qd_amd.ndp_nmpc_ctl import NDPNMPCBodyRateController
    from ndp_nmpc_qd_amd.nmpc_ctl import NMPCBodyRateController
    b = synth.make_batch(1, seed=9)
    x0, xr, ur = b["x0"][0], b["xr"][0], b["ur"][0]
    ctl = NMPCBodyRateController(is_build_acados=True)
    assert ctl.solver.N == 20
    ctl.reset(xr, ur)
    u0 = ctl.update(x0, xr, ur)
    assert isinstance(u0, np.ndarray) and u0.shape == (4,) and u0.dtype == np.float64
    cfg = oracle.default_cfg()
    Xo, Uo = xr.copy(), ur.copy()
    u0o, _ = oracle.step(cfg, x0, xr, ur, None, Xo, Uo)
    _assert_u(u0, u0o, 1e-8)
    x1 = ctl.solver.get(1, "x")
    np.testing.assert_allclose(x1, Xo[1], atol=1e-8)
    x1[:] = 0                                           # caller mutates its copy (nmpc_node.py:238)
    np.testing.assert_allclose(ctl.solver.get(1, "x"), Xo[1], atol=1e-8)
    assert ctl.solver.status == 0
    # hover fixed point
    xh, uh = synth.hover_reference()
    ctl.reset(xh, uh)
    np.testing.assert_allclose(ctl.update(xh[0], xh, uh), [0, 0, 0, 9.81], atol=1e-10)
    # NDP controller + DownwashNN
    nn = DownwashNN()
    f = nn.update(mlp_golden["other"][0], mlp_golden["ego"][0])
    assert f.shape == (21, 3) and f.dtype == np.float32
    assert np.all(np.abs(f - mlp_golden["f_update"][0]) <= 1e-5 * np.maximum(1, np.abs(mlp_golden["f_update"][0])))
    nctl = NDPNMPCBodyRateController()
    nctl.reset(xr, ur)
    u0n = nctl.update(x0, xr, ur, f)
    Xo, Uo = xr.copy(), ur.copy()
    u0no, _ = oracle.step(oracle.default_cfg(use_fd=True), x0, xr, ur, f, Xo, Uo)
    _assert_u(u0n, u0no, 1e-8)
    assert isinstance(ctl, NMPCBodyRateController) and not isinstance(ctl, NDPNMPCBodyRateController)  # nmpc_node.py:203-208


def test_status_raises_like_reference(ndp):
    """Infeasible QP -> status 4 -> the reference's exception text (nmpc_body_rate_ctl.py:109-110)."""
    b = synth.make_batch(2, seed=41)
    eng = ndp.BatchedNMPC(2, lbv=(-1e-3,) * 3, ubv=(1e-3,) * 3, iter_max=12)
    eng.reset(b["xr"], b["ur"])
    with pytest.raises(Exception, match="acados acados_ocp_solver returned status"):
        eng.update(b["x0"], b["xr"], b["ur"])
    st, it = eng.status()
    assert (st != 0).all() and (it > 0).all() and (it <= 12).all()


def test_device_resident_path_matches_host_path(ndp):
    import torch
    B = 64
    b = synth.make_batch(B, seed=3, downwash=True)
    eng_h = ndp.BatchedNMPC(B, disturbance=True)
    eng_h.reset(b["xr"], b["ur"])
    u_host = eng_h.update(b["x0"], b["xr"], b["ur"], other=b["other"], ego_xy=b["ego_xy"])
    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(b[k]).to(dev) for k in ("x0", "xr", "ur", "other", "ego_xy")}
    eng_d = ndp.BatchedNMPC(B, disturbance=True)
    eng_d.reset_device(t["xr"], t["ur"])
    u_dev = torch.empty(B, 4, dtype=torch.float64, device=dev)
    eng_d.update_device(t["x0"], t["xr"], t["ur"], u_dev, other=t["other"], ego_xy=t["ego_xy"])
    eng_d.synchronize()
    assert np.array_equal(u_dev.cpu().numpy(), u_host)


def test_fused_downwash_with_two_rti_iterations_and_ipm_always(ndp, oracle, mlp_blob):
    """Fused gate+MLP launch: the force must survive the recycling of its LDS staging slot across RTI iterations;
    and the interior-point path must see the same force."""
    B = 48
    b = synth.make_batch(B, seed=123, downwash=True)
    f_or = oracle.downwash_batch(mlp_blob, b["other"], b["xr"], b["ego_xy"])
    for n_rti, qp_mode in ((2, 0), (1, 1), (3, 1)):
        eng = ndp.BatchedNMPC(B, disturbance=True, n_rti=n_rti, qp_mode=qp_mode)
        eng.reset(b["xr"], b["ur"])
        f_dev = eng.downwash(b["other"], b["xr"], b["ego_xy"])
        u0 = eng.update(b["x0"], b["xr"], b["ur"], other=b["other"], ego_xy=b["ego_xy"])
        u0o, sto, *_ = _oracle_batch(oracle, b, n_rti=n_rti, use_fd=True, f=f_dev)
        assert (sto == 0).all()
        _assert_u(u0, u0o, 1e-8)
        assert np.all(np.abs(f_dev - f_or) <= 1e-5 * np.maximum(1.0, np.abs(f_or)))


def test_minimal_and_odd_horizons(ndp, oracle):
    for N in (2, 3, 7, 27, 31):
        b = synth.make_batch(5, N=N, seed=200 + N, downwash=True)
        eng = ndp.BatchedNMPC(5, N=N, disturbance=True)
        eng.reset(b["xr"], b["ur"])
        f = eng.downwash(b["other"], b["xr"], b["ego_xy"])
        u0 = eng.update(b["x0"], b["xr"], b["ur"], other=b["other"], ego_xy=b["ego_xy"])
        u0o, sto, *_ = _oracle_batch(oracle, b, N=N, use_fd=True, f=f)
        assert (sto == 0).all()
        _assert_u(u0, u0o, 1e-8)


def test_nan_input_reports_status_1_for_that_instance_only(ndp, oracle):
    B = 8
    b = synth.make_batch(B, seed=77)
    x0 = b["x0"].copy()
    x0[3, 4] = np.nan
    eng = ndp.BatchedNMPC(B)
    eng.reset(b["xr"], b["ur"])
    u0 = eng.update(x0, b["xr"], b["ur"], raise_on_status=False)
    st, _ = eng.status()
    assert st[3] != 0 and (np.delete(st, 3) == 0).all()
    u0o, *_ = _oracle_batch(oracle, b)
    _assert_u(np.delete(u0, 3, axis=0), np.delete(u0o, 3, axis=0), 1e-8)       # the other instances are untouched


def test_large_batch_and_multiple_handles(ndp, oracle):
    """More instances than SIMDs (waves run in rounds) and two live handles with different configurations."""
    B = 4096 + 3
    b = synth.make_batch(B, seed=5)
    eng = ndp.BatchedNMPC(B)
    eng2 = ndp.BatchedNMPC(16, N=10)
    b2 = synth.make_batch(16, N=10, seed=6)
    eng.reset(b["xr"], b["ur"])
    eng2.reset(b2["xr"], b2["ur"])
    u0 = eng.update(b["x0"], b["xr"], b["ur"])
    u2 = eng2.update(b2["x0"], b2["xr"], b2["ur"])
    idx = np.random.default_rng(0).choice(B, 256, replace=False)
    sub = {k: np.ascontiguousarray(b[k][idx]) for k in ("x0", "xr", "ur")}
    u0o, *_ = _oracle_batch(oracle, sub)
    _assert_u(u0[idx], u0o, 1e-8)
    u2o, *_ = _oracle_batch(oracle, b2, N=10)
    _assert_u(u2, u2o, 1e-8)


def test_concurrent_callers_like_rospy_threads(ndp, oracle):
    """update (control timer), reset (action thread) and solver.get (viz timer) from different threads, no caller
    locks (nmpc_node.py:94,152,237): the handle serialises; every result is one of the legal interleavings."""
    import threading
    from ndp_nmpc_qd_amd.nmpc_ctl import NMPCBodyRateController
    b = synth.make_batch(1, seed=9)
    x0, xr, ur = b["x0"][0], b["xr"][0], b["ur"][0]
    ctl = NMPCBodyRateController()
    ctl.reset(xr, ur)
    errors, outs = [], []

    def control():
        try:
            for _ in range(30):
                outs.append(ctl.update(x0, xr, ur))
        except Exception as e:     # noqa: BLE001
            errors.append(e)

    def action():
        try:
            for _ in range(10):
                ctl.reset(xr, ur)
        except Exception as e:     # noqa: BLE001
            errors.append(e)

    def viz():
        try:
            for _ in range(60):
                x = ctl.solver.get(3, "x")
                assert x.shape == (10,) and np.isfinite(x).all()
        except Exception as e:     # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=f) for f in (control, action, viz)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors
    # every returned u0 is finite and inside the input box; the first solve after a reset equals the oracle's first step
    U = np.array(outs)
    assert np.isfinite(U).all() and (np.abs(U[:, :3]) <= 6 + 1e-9).all() and (U[:, 3] >= -1e-9).all()
    ctl.reset(xr, ur)
    u = ctl.update(x0, xr, ur)
    Xo, Uo = xr.copy(), ur.copy()
    uo, _ = oracle.step(oracle.default_cfg(), x0, xr, ur, None, Xo, Uo)
    _assert_u(u, uo, 1e-8)


@pytest.mark.gpu
def test_auto_mode_against_the_interior_point_oracle_under_large_perturbations(ndp, oracle):
    """Large initial errors (up to 1 m, 2 m/s, 0.3 in the quaternion): hundreds of instances with active bounds, three
    consecutive ticks on the warm-started iterate, everything at DEFAULT settings.  QP_AUTO (the default) returns the exact
    equality-constrained minimiser when it is auto_margin inside every bound and runs the interior-point loop otherwise;
    the oracle always runs the loop, as HPIPM does.  Every instance, early exit or not, has to sit inside the north-star's
    1e-5: what makes that hold is the floor under the centring target (cfg.mu_floor) -- without it rare instances drive mu
    to 1e-12 and below, the slacks (differences) lose their digits, and two correct implementations drift 2e-5 apart."""
    B = 768
    for seed, kw in ((2, dict(pos_sigma=0.5, vel_sigma=1.0, quat_sigma=0.15)), (4, dict(pos_sigma=1.0, vel_sigma=2.0, quat_sigma=0.3))):
        b = synth.make_batch(B, seed=seed, **kw)
        eng = ndp.BatchedNMPC(B, **LEGACY)
        cfg = oracle.default_cfg()
        eng.reset(b["xr"], b["ur"])
        X, U = b["xr"].copy(), b["ur"].copy()
        n_ipm = 0
        for _ in range(3):
            u0 = eng.update(b["x0"], b["xr"], b["ur"], raise_on_status=False)
            uo, sto, _ = oracle.step_batch(cfg, b["x0"], b["xr"], b["ur"], None, X, U)
            st, it = eng.status()
            assert np.array_equal(st, sto)
            ok = (st == 0) & (sto == 0)
            assert ok.mean() > 0.97
            n_ipm += int((it > 0).sum())
            _assert_u(u0[ok], uo[ok], RTOL_U)
        assert n_ipm > 100
        # the oracle's own early-exit mode (cfg.qp_mode = 0) is the same decision rule: identical iteration counts
        cfg0 = oracle.default_cfg()
        cfg0.qp_mode, cfg0.as_iter_max = 0, 0
        X0, U0 = b["xr"].copy(), b["ur"].copy()
        eng.reset(b["xr"], b["ur"])
        u0 = eng.update(b["x0"], b["xr"], b["ur"], raise_on_status=False)
        uo, sto, ito = oracle.step_batch(cfg0, b["x0"], b["xr"], b["ur"], None, X0, U0)
        st, it = eng.status()
        assert np.array_equal(st, sto) and np.array_equal(it, ito)
        _assert_u(u0[sto == 0], uo[sto == 0], 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("N", [5, 12, 27, 31, 40])
def test_downwash_step_at_other_horizons(ndp, oracle, mlp_blob, N):
    """The gate + MLP + NDP step away from the reference horizon: N <= 27 takes the fused launch of the run-time-horizon
    kernel (2 or 4 instances per workgroup), N = 31 the two-launch form (its 214 box constraints need the 5-slot kernel),
    N = 40 the multi-tile standalone MLP kernel; a ragged batch size on top."""
    B = 37
    b = synth.make_batch(B, N=N, seed=100 + N, downwash=True)
    eng = ndp.BatchedNMPC(B, N=N, disturbance=True)
    eng.reset(b["xr"], b["ur"])
    u0 = eng.update(b["x0"], b["xr"], b["ur"], other=b["other"], ego_xy=b["ego_xy"])
    f_or = oracle.downwash_batch(mlp_blob, b["other"], b["xr"], b["ego_xy"])
    assert 0 < (np.abs(f_or).max(axis=(1, 2)) > 0).sum() < B
    u0o, *_ = _oracle_batch(oracle, b, N=N, use_fd=True, f=f_or)
    _assert_u(u0, u0o, 2e-6)


@pytest.mark.gpu
def test_three_vehicle_formations_in_one_batch(ndp, oracle, mlp_blob):
    """BASELINE config 4 semantics, formation-major placement (all three vehicles of a formation on one GPU, no exchange):
    vehicle 0 = NDP controller whose neighbour is vehicle 1 (ndp_nmpc_leader_node.py:40,60-76); vehicles 1 and 2 = plain
    NMPC followers tracking the leader's window shifted by the filtered formation offsets (0, +-1, 0)
    (nmpc_leader_node.py:42-43, nmpc_follower_node.py:44-77).  One batch of 3R instances: the followers ride in the NDP
    engine with their gate closed, which is the NMPC model exactly (f = 0)."""
    R, N = 96, 20
    lead = synth.make_batch(R, seed=77)
    rel = ndp.BatchedNMPC(R, load_mlp=False)
    xr_f = []
    for off in ((0.0, 1.0, 0.0), (0.0, -1.0, 0.0)):
        rel.relay_reset()
        rel.relay_formation(np.tile(off, (R, 1)))
        xr_f.append(rel.relay_reference(lead["xr"]))
    rng = np.random.default_rng(5)
    # leaders fly 0.4 m above follower 1's track displaced sideways: about half of them inside the 1 m gate
    xr_l = lead["xr"].copy()
    xr_l[:, :, 1] += rng.uniform(0.0, 4.0, (R, 1))
    xr_l[:, :, 2] += 0.4
    xr = np.concatenate([xr_l, xr_f[0], xr_f[1]])
    ur = np.concatenate([lead["ur"]] * 3)
    x0 = xr[:, 0].copy()
    x0[:, 0:3] += rng.normal(0, 0.05, (3 * R, 3))
    other = np.concatenate([xr_f[0], xr_l, xr_l])                 # followers: any window, their gate is closed below
    ego_xy = np.concatenate([x0[:R, 0:2], np.full((2 * R, 2), 1e6)])
    eng = ndp.BatchedNMPC(3 * R, disturbance=True)
    eng.reset(xr, ur)
    u0 = eng.update(x0, xr, ur, other=other, ego_xy=ego_xy)
    f = oracle.downwash_batch(mlp_blob, other, xr, ego_xy)
    assert np.all(f[R:] == 0) and 0.2 < (np.abs(f[:R]).max(axis=(1, 2)) > 0).mean() < 0.8
    X, U = xr.copy(), ur.copy()
    uo, st, _ = oracle.step_batch(oracle.default_cfg(use_fd=True), x0, xr, ur, f, X, U)
    _assert_u(u0, uo, 2e-6)
    Xn, Un = xr[R:].copy(), ur[R:].copy()
    un, *_ = oracle.step_batch(oracle.default_cfg(use_fd=False), x0[R:], xr[R:], ur[R:], None, Xn, Un)
    _assert_u(u0[R:], un, 1e-7)                                    # followers: the plain NMPC controller's answer


def test_step_ex_returns_iterate_and_status_from_the_same_call(ndp, oracle):
    """ndp_step_ex: u0, the new iterate, status and interior-point iterations from ONE call (what the reference's callers
    read after solve_for_x0: solver.get / solver.status), through the packed pinned path (small batch) and the direct path."""
    for B in (3, 700):                                  # 3: inputs fit the 1 MiB pinned mirror; 700: they do not
        b = synth.make_batch(B, seed=31, pos_sigma=0.5, vel_sigma=1.0, quat_sigma=0.15)
        eng = ndp.BatchedNMPC(B, **LEGACY)
        eng.reset(b["xr"], b["ur"])
        u0, X, U, st, it = eng.update(b["x0"], b["xr"], b["ur"], raise_on_status=False, full=True)
        X2, U2 = eng.get_iterate()
        st2, it2 = eng.status()
        assert np.array_equal(X, X2) and np.array_equal(U, U2) and np.array_equal(st, st2) and np.array_equal(it, it2)
        assert np.array_equal(u0, U[:, 0])
        uo, sto, ito, Xo, Uo = _oracle_batch(oracle, b)
        assert np.array_equal(st, sto) and (it > 0).any()
        _assert_u(u0[sto == 0], uo[sto == 0])
        np.testing.assert_allclose(X[sto == 0], Xo[sto == 0], atol=2e-5)


def test_work_queue_gives_the_same_answers(ndp, oracle, mlp_blob):
    """The interior-point work queue only changes WHICH wave solves an instance: status and iteration counts are identical
    to the in-place form and the iterates agree to rounding (the queue launch is a separate instantiation of the same
    text: the compiler contracts a few multiply-adds differently), for the plain NMPC launch and for the fused gate + MLP
    launch (whose force reaches the solving wave through the queue's write-through copy); three warm-started ticks; a
    ragged batch size."""
    B = 1500 + 3
    for disturbance in (False, True):
        b = synth.make_batch(B, seed=55, downwash=disturbance, pos_sigma=0.5, vel_sigma=1.0, quat_sigma=0.15)
        kw = dict(other=b["other"], ego_xy=b["ego_xy"]) if disturbance else {}
        res = {}
        for wq in (1, 2):
            eng = ndp.BatchedNMPC(B, disturbance=disturbance, work_queue=wq, **LEGACY)
            assert eng.work_queue == (wq == 1)
            eng.reset(b["xr"], b["ur"])
            outs = []
            for _ in range(3):
                outs.append(eng.update(b["x0"], b["xr"], b["ur"], raise_on_status=False, full=True, **kw))
            res[wq] = outs
        for a, c in zip(res[1], res[2]):
            assert np.array_equal(a[3], c[3]) and np.array_equal(a[4], c[4])
            for x, y in zip(a[:3], c[:3]):
                np.testing.assert_allclose(x, y, rtol=0, atol=1e-8)
        u0, X, U, st, it = res[1][0]
        assert 0.05 < (it > 0).mean() < 0.6                      # the queue really carried a share of the batch
        f = oracle.downwash_batch(mlp_blob, b["other"], b["xr"], b["ego_xy"]) if disturbance else None
        uo, sto, *_ = _oracle_batch(oracle, b, use_fd=disturbance, f=f)
        assert np.array_equal(st, sto)
        _assert_u(u0[sto == 0], uo[sto == 0])
    # the automatic rule: never below two instances per SIMD; from there on IN PLACE to begin with (the N = 40 / 2-iteration shape: always
    # the list) -- what the steps do then switches it (test_work_list_switches_itself_on_when_the_steps_iterate)
    assert not ndp.BatchedNMPC(1024).work_queue and not ndp.BatchedNMPC(2047).work_queue
    assert not ndp.BatchedNMPC(2048).work_queue and not ndp.BatchedNMPC(4096, qp_mode=1).work_queue
    assert ndp.BatchedNMPC(256, N=40, n_rti=2).work_queue


def test_work_list_switches_itself_on_when_the_steps_iterate(ndp, oracle):
    """cfg.work_queue = 0 at two instances per SIMD: the handle starts in place; the device's counts of interior-point instances and of
    executed steps come back every 8th launch, and the list goes on once >= 4 % of a window's instances iterated, off again at <= 1.5 %.
    Same answers in either form (status, iteration counts equal; controls to rounding) -- here: perturbed starts (a fifth iterate) switch
    it on within 24 steps, nominal starts switch it off again, and the controls of the automatic engine equal those of the forced forms."""
    B = 2048
    hard = synth.make_batch(B, seed=57, downwash=False, pos_sigma=0.5, vel_sigma=1.0, quat_sigma=0.15)
    easy = synth.make_batch(B, seed=57, downwash=False)
    eng = ndp.BatchedNMPC(B, **LEGACY)
    ref = {wq: ndp.BatchedNMPC(B, work_queue=wq, **LEGACY) for wq in (1, 2)}
    assert not eng.work_queue and ref[1].work_queue and not ref[2].work_queue

    def run(b, n):
        states = []
        for e in (eng, ref[1], ref[2]):
            e.reset(b["xr"], b["ur"])
        for _ in range(n):
            outs = [e.update(b["x0"], b["xr"], b["ur"], raise_on_status=False, full=True) for e in (eng, ref[1], ref[2])]
            for o in outs[1:]:
                assert np.array_equal(outs[0][3], o[3]) and np.array_equal(outs[0][4], o[4])
                np.testing.assert_allclose(outs[0][0], o[0], rtol=0, atol=1e-8)
            states.append(eng.work_queue)
            for e in (eng, ref[1], ref[2]):       # every tick from the same start: the workload stays what it is
                e.reset(b["xr"], b["ur"])
        return states, outs[0]

    states, out = run(hard, 28)
    assert 0.05 < (out[4] > 0).mean() < 0.6
    assert not states[0] and states[-1] and sum(states) >= 4, states
    first_on = states.index(True)
    assert 8 <= first_on <= 24 and all(states[first_on:]), states
    states, out = run(easy, 28)
    assert (out[4] > 0).mean() < 0.01
    assert states[0] and not states[-1] and not any(states[states.index(False):]), states
    for e in (eng, ref[1], ref[2]):
        e.close()


def test_neighbour_rows_by_index_and_six_column_windows(ndp):
    """ndp_step_device_ex: the neighbour windows as a multi-GPU exchange leaves them -- [rows, N+1, 6] position / velocity
    columns (all the gate and the MLP read) picked per instance through other_index, -1 = no neighbour -- give exactly
    the forces and controls of the plain [B, N+1, 10] form (N = 20: fused launch; N = 40: standalone MLP kernel)."""
    import torch
    dev = torch.device("cuda:0")
    for N, B in ((20, 96), (40, 33)):
        b = synth.make_batch(B, N=N, seed=77 + N, downwash=True)
        rng = np.random.default_rng(3)
        perm = rng.permutation(B)
        none = rng.random(B) < 0.25                       # a quarter of the instances have no neighbour at all
        rows = np.concatenate([b["other"][perm][:, :, :6], rng.normal(size=(5, N + 1, 6))])      # permuted, plus unrelated rows
        inv = np.empty(B, dtype=np.int32)
        inv[perm] = np.arange(B, dtype=np.int32)
        idx = np.where(none, -1, inv).astype(np.int32)
        ego = b["ego_xy"].copy()
        ego[none] = 1e9                                    # plain form: the same instances gated off by distance
        t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in
             dict(x0=b["x0"], xr=b["xr"], ur=b["ur"], other=b["other"], ego=ego, ego_idx=b["ego_xy"], rows=rows, idx=idx).items()}
        t["pv"] = t["other"][:, :, :6].contiguous()      # row i = instance i, six columns, no index: the ring exchange's slice
        outs = []
        for form in ("plain", "indexed", "six_columns"):
            eng = ndp.BatchedNMPC(B, N=N, disturbance=True)
            eng.reset_device(t["xr"], t["ur"])
            u = torch.empty(B, 4, dtype=torch.float64, device=dev)
            if form == "plain":
                eng.update_device(t["x0"], t["xr"], t["ur"], u, other=t["other"], ego_xy=t["ego"])
            elif form == "six_columns":
                eng.update_device(t["x0"], t["xr"], t["ur"], u, other=t["pv"], ego_xy=t["ego"])
            else:
                eng.update_device(t["x0"], t["xr"], t["ur"], u, other=t["rows"], ego_xy=t["ego_idx"], other_index=t["idx"])
            st, _ = eng.status()                           # no explicit synchronise: the getter waits for the foreign-free stream
            assert (st == 0).all()
            outs.append(u.cpu().numpy())
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_getters_wait_for_the_callers_stream(ndp, oracle):
    """A step enqueued on a caller's (non-blocking) stream: ndp_get_status / ndp_get_iterate wait for THAT stream."""
    import torch
    dev = torch.device("cuda:0")
    B = 2048
    b = synth.make_batch(B, seed=8, pos_sigma=0.5, vel_sigma=1.0, quat_sigma=0.15)
    t = {k: torch.from_numpy(b[k]).to(dev) for k in ("x0", "xr", "ur")}
    eng = ndp.BatchedNMPC(B, qp_mode=1)                   # long kernel: every instance iterates
    s = torch.cuda.Stream(device=dev)
    u = torch.empty(B, 4, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    eng.reset_device(t["xr"], t["ur"], stream=s)
    eng.update_device(t["x0"], t["xr"], t["ur"], u, stream=s)
    st, it = eng.status()                                  # must not read the previous (zeroed) status
    assert (it > 0).all()
    X, U = eng.get_iterate()
    uo, sto, ito, Xo, Uo = _oracle_batch(oracle, b)
    assert np.array_equal(st, sto)
    np.testing.assert_allclose(U[sto == 0], Uo[sto == 0], atol=2e-5)


def test_host_entry_points_from_concurrent_threads(ndp, oracle):
    """The host-pointer forms of the step and of the rows f1-f4 share device staging areas: each call holds the handle's
    lock from its first staging copy to its read-back, so calls from different threads cannot corrupt each other
    (rospy runs every timer / subscriber / action callback on its own thread).  Each thread checks its own results against
    what it gets single-threaded."""
    import threading
    from ndp_nmpc_qd_amd.pt_pub import TrajCoefficients
    B = 64
    b = synth.make_batch(B, seed=12)
    eng = ndp.BatchedNMPC(B, load_mlp=False)
    rng = np.random.default_rng(1)
    wp = np.zeros((B, 4, 4))
    wp[:, 0:3] = np.cumsum(rng.uniform(-1, 1, (B, 3, 4)), axis=2)
    tc = TrajCoefficients.from_waypoints(wp, rng.uniform(2.0, 3.0, (B, 3)))
    eng.ref_set_trajectory(tc.coeff_x, tc.coeff_y, tc.coeff_z, tc.coeff_yaw, tc.traj_time_cum, tc.traj_time_seg, tc.final_pt)
    tq = rng.uniform(0.0, 5.0, B)
    vz, thr = rng.normal(0, 0.3, B), rng.uniform(0.2, 0.9, B)
    xs, us = b["x0"].copy(), b["ur"][:, 0].copy()
    lead = b["xr"]
    eng.relay_formation(np.tile((0.0, 1.0, 0.0), (B, 1)))
    want = dict(ref=eng.ref_window(tq), plant=eng.plant_step(xs, us), relay=eng.relay_reference(lead),
                act=eng.actuator_cmd(b["ur"][:, 0], np.full(B, 0.4)))
    eng.reset(b["xr"], b["ur"])
    want["step"] = eng.update(b["x0"], b["xr"], b["ur"])
    errors = []

    def run(name, fn, check):
        try:
            for _ in range(40):
                check(fn(), want[name])
        except Exception as e:     # noqa: BLE001
            errors.append((name, e))

    def eq(a, w):
        if isinstance(w, tuple):
            for x, y in zip(a, w):
                assert np.array_equal(x, y)
        else:
            assert np.array_equal(a, w)

    def step():
        eng.reset(b["xr"], b["ur"])
        return eng.update(b["x0"], b["xr"], b["ur"])

    def step_ok(a, w):            # reset + update are two calls: another thread's reset may fall between them, never a torn result
        assert np.isfinite(a).all() and np.abs(a - w).max() < 1.0

    jobs = [("step", step, step_ok), ("ref", lambda: eng.ref_window(tq), eq), ("plant", lambda: eng.plant_step(xs, us), eq),
            ("relay", lambda: eng.relay_reference(lead), eq), ("act", lambda: eng.actuator_cmd(b["ur"][:, 0], np.full(B, 0.4)), eq),
            ("thr", lambda: eng.throttle_update(vz, thr), lambda a, w: None)]
    want["thr"] = None
    ts = [threading.Thread(target=run, args=j) for j in jobs]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors
    # the estimator state saw exactly 40 clean updates: the same as 40 single-threaded updates on a fresh engine
    ref = ndp.BatchedNMPC(B, load_mlp=False)
    for _ in range(40):
        k = ref.throttle_update(vz, thr)
    np.testing.assert_array_equal(eng.throttle_state(), ref.throttle_state())


def test_f32_and_bf16_mfma_register_maps(ndp):
    """v_mfma_f32_16x16x4_f32 and v_mfma_f32_16x16x16_bf16 as the config-5 backends issue them: A lane l = A[l&15][k], B lane l =
    B[k][l&15] with k = l>>4 (fp32) or 4 (l>>4) + i (bf16, packed element i); accumulator register r of lane l = D[4 (l>>4) + r][l&15]
    -- what rti_wave.hpp's column renumbering (lcol) and the emulator assume.  Also the row sum over lanes 4 apart."""
    from ndp_nmpc_qd_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(1)
    # fp32: integers, exact
    A, Bm, Cm = rng.integers(-9, 9, (16, 4)).astype(np.float32), rng.integers(-9, 9, (4, 16)).astype(np.float32), \
        rng.integers(-9, 9, (16, 16)).astype(np.float32)
    a, b = np.zeros((4, 64), np.float32), np.zeros((4, 64), np.float32)
    a[0] = [A[l & 15, l >> 4] for l in range(64)]
    b[0] = [Bm[l >> 4, l & 15] for l in range(64)]
    c = np.array([[Cm[4 * (l >> 4) + r, l & 15] for l in range(64)] for r in range(4)], np.float32)
    d = np.zeros(320, np.float32)
    assert lib.ndp_debug_mfma_probe_f32(_lib.ptr(a), _lib.ptr(b), _lib.ptr(c), _lib.ptr(d), 0) == 0
    D = A @ Bm + Cm
    got = d[:256].reshape(4, 64)
    for r in range(4):
        for l in range(64):
            assert got[r, l] == D[4 * (l >> 4) + r, l & 15], (r, l)
    want = np.array([sum(a[0][(l & 48) | ((l + 4 * t) & 15)] for t in range(4)) for l in range(64)])
    np.testing.assert_array_equal(d[256:], want)
    # bf16, K = 16: small integers are exact in bf16
    A16, B16 = rng.integers(-9, 9, (16, 16)).astype(np.float32), rng.integers(-9, 9, (16, 16)).astype(np.float32)
    for i in range(4):
        a[i] = [A16[l & 15, 4 * (l >> 4) + i] for l in range(64)]
        b[i] = [B16[4 * (l >> 4) + i, l & 15] for l in range(64)]
    assert lib.ndp_debug_mfma_probe_f32(_lib.ptr(a), _lib.ptr(b), _lib.ptr(c), _lib.ptr(d), 1) == 0
    D = A16 @ B16 + Cm
    got = d[:256].reshape(4, 64)
    for r in range(4):
        for l in range(64):
            assert got[r, l] == D[4 * (l >> 4) + r, l & 15], (r, l)


@pytest.mark.parametrize("N,n_rti", [(40, 2), (20, 1), (13, 1)])
def test_config5_qp_on_the_fp32_and_bf16_matrix_instructions(ndp, oracle, N, n_rti):
    """BASELINE config 5 ("fp32 vs bf16 MFMA on the QP"): the Riccati sweeps on v_mfma_f32_16x16x4_f32 (qp_precision 3) and on
    v_mfma_f32_16x16x16_bf16 (4), everything else fp64.  Nominal starts (no active bounds): the fp32 sweeps stay inside the
    north-star's 1e-5, the bf16 sweeps miss it by two to three orders of magnitude -- reported, not offered as a product
    mode.  (40, 2) is the configuration's own shape (compile-time kernel, two instances per workgroup)."""
    B = 512 if N == 40 else 96
    b = synth.make_batch(B, N=N, seed=20231213 + 5)
    uo, sto, *_ = _oracle_batch(oracle, b, N=N, n_rti=n_rti)
    assert (sto == 0).all()
    err = {}
    for prec in (0, 3, 4):
        eng = ndp.BatchedNMPC(B, N=N, n_rti=n_rti, qp_precision=prec)
        eng.reset(b["xr"], b["ur"])
        u0 = eng.update(b["x0"], b["xr"], b["ur"], raise_on_status=False)
        st, it = eng.status()
        assert (st == 0).all()
        err[prec] = float(np.max(np.abs(u0 - uo) / np.maximum(1.0, np.abs(uo))))
    assert err[0] < 1e-8 and err[3] < 1e-5 and 1e-4 < err[4] < 0.5, err


@pytest.mark.parametrize("B", [512, 4096])
def test_config5_shape_with_perturbed_starts_runs_the_interior_point_kernels(ndp, oracle, B):
    """BASELINE config 5's shape (N = 40, 2 RTI iterations) with PERTURBED starts, so that the instantiations that carry the
    interior-point loop really execute it (>= 15 % of the instances iterate): qp_mode = 1 in place (rti_kernel<5,2,false,40,0,2,0>),
    the automatic mode through the work list (producer <..,1> + consumer <..,2>) and the automatic mode in place.  Each against the
    oracle in the SAME mode on 256 sampled instances (1e-5 bar, status and iteration counts equal on the sample), the three device
    forms against each other on EVERY instance, and the size-independent properties of test_full_size_batch_1024 at full size."""
    N, n_rti, NS = 40, 2, 256
    b = synth.make_batch(B, N=N, seed=synth.SEED0 + 5, pos_sigma=0.5, vel_sigma=1.0, quat_sigma=0.15)
    sample = np.sort(np.random.default_rng(7).choice(B, NS, replace=False))
    bs = {k: np.ascontiguousarray(b[k][sample]) for k in ("x0", "xr", "ur")}
    res = {}
    for name, qp_mode, wq in (("ipm_in_place", 1, 2), ("auto_work_list", 0, 1), ("auto_in_place", 0, 2)):
        eng = ndp.BatchedNMPC(B, N=N, n_rti=n_rti, qp_mode=qp_mode, work_queue=wq, **LEGACY)
        assert eng.work_queue == (wq == 1)
        eng.reset(b["xr"], b["ur"])
        u0, X, U, st, it = eng.update(b["x0"], b["xr"], b["ur"], raise_on_status=False, full=True)
        res[name] = (u0, X, U, st, it)
        cfg = oracle.default_cfg(N=N, n_rti=n_rti)
        cfg.qp_mode, cfg.as_iter_max = qp_mode, 0
        Xo, Uo = bs["xr"].copy(), bs["ur"].copy()
        uo, sto, ito = oracle.step_batch(cfg, bs["x0"], bs["xr"], bs["ur"], None, Xo, Uo)
        assert np.array_equal(st[sample], sto), name
        assert np.array_equal(it[sample], ito), name
        ok = sto == 0
        assert ok.mean() > 0.9, name
        _assert_u(u0[sample][ok], uo[ok], RTOL_U)
        np.testing.assert_allclose(X[sample][ok], Xo[ok], rtol=0, atol=1e-5)
        assert (it > 0).mean() >= 0.15, (name, (it > 0).mean())           # the interior-point code really ran
        good = st == 0
        # properties at full size: x0 equality after the full step, u0 = U[0], inputs inside the box, velocities inside theirs
        np.testing.assert_allclose(X[good][:, 0, :], b["x0"][good], atol=1e-9)
        assert np.array_equal(u0, U[:, 0, :])
        assert (U[good][..., :3] <= 6 + 1e-7).all() and (U[good][..., :3] >= -6 - 1e-7).all()
        assert (U[good][..., 3] >= -1e-7).all() and (U[good][..., 3] <= 9.81 / 0.36 + 1e-7).all()
        assert (np.abs(X[good][:, 1:N, 3:6]) <= 20 + 1e-7).all()
        eng.close()
    # one algorithm, three launch forms: the work list only changes WHICH wave solves an instance
    a, c = res["auto_work_list"], res["auto_in_place"]
    assert np.array_equal(a[3], c[3]) and np.array_equal(a[4], c[4])
    np.testing.assert_allclose(a[0], c[0], rtol=0, atol=1e-8)
    both = (res["ipm_in_place"][3] == 0) & (c[3] == 0)
    # early exit vs always iterating ACROSS modes: each form sits inside 1e-5 of the oracle in its own mode (asserted above); two
    # correct solves that stop one interior-point iteration apart differ by ~4e-6 * (N / 20) per RTI iteration on near-active
    # problems (DESIGN "Oracle"): measured 1.3e-5 on ONE of 16 384 entries at batch 4096 -- the bar between the modes is 5e-5 here
    _assert_u(res["ipm_in_place"][0][both], c[0][both], 5e-5)


def test_config2_full_size_without_downwash(ndp, oracle):
    """BASELINE configs[1] exactly: batch = 1024 independent quadrotors, N = 20, NO downwash (NMPC controller), one RTI iteration
    -- every instance against the oracle, the properties of the full-size test, and permutation equivariance."""
    B = 1024
    b = synth.make_batch(B, seed=synth.SEED0 + 2)
    eng = ndp.BatchedNMPC(B)
    eng.reset(b["xr"], b["ur"])
    u0, X, U, st, it = eng.update(b["x0"], b["xr"], b["ur"], full=True)
    uo, sto, ito, Xo, Uo = _oracle_batch(oracle, b)
    assert (st == 0).all() and (sto == 0).all()
    _assert_u(u0, uo, 1e-8)
    np.testing.assert_allclose(X, Xo, rtol=0, atol=1e-8)
    np.testing.assert_allclose(U, Uo, rtol=0, atol=1e-8)
    np.testing.assert_allclose(X[:, 0, :], b["x0"], atol=1e-9)
    assert np.array_equal(u0, U[:, 0, :])
    assert (U[..., :3] <= 6 + 1e-9).all() and (U[..., :3] >= -6 - 1e-9).all()
    assert (U[..., 3] >= -1e-9).all() and (U[..., 3] <= 9.81 / 0.36 + 1e-9).all()
    perm = np.random.default_rng(1).permutation(B)
    eng2 = ndp.BatchedNMPC(B)
    eng2.reset(b["xr"][perm], b["ur"][perm])
    assert np.array_equal(eng2.update(b["x0"][perm], b["xr"][perm], b["ur"][perm]), u0[perm])
    # a second tick on the warm-started iterate (the reference never shifts it, nmpc_body_rate_ctl.py:86-112)
    u1 = eng.update(b["x0"], b["xr"], b["ur"])
    u1o, st1, *_ = _oracle_batch(oracle, b, X=Xo, U=Uo)
    _assert_u(u1, u1o, 1e-8)


def test_peer_window_buffer_as_neighbour_source(ndp):
    """dist.PeerWindows on one rank (the neighbour is the rank's own buffer: same launches, same protocol words): per tick ONE
    publish launch puts this tick's windows into the slot of the tick's parity and returns the neighbour's slot as a raw device
    address (dist.DevWindows); the control step that takes it gives the same answers as the tensor path, with and without
    other_index; epochs count the ticks, no wait times out, the host's slot parity matches the device's.  (Two processes
    mapping each other's buffers: test_peer_exchange_between_two_processes; the protocol alone: tests/test_peer_epoch.py.)"""
    import torch
    from ndp_nmpc_qd_amd import dist as ndist
    B, N = 96, 20
    dev = torch.device("cuda", 0)
    pw = ndist.PeerWindows(B, N, 0)
    try:
        engs = {m: ndp.BatchedNMPC(B, N=N, disturbance=True) for m in ("tensor", "raw", "raw_indexed")}
        u0 = {m: torch.empty(B, 4, dtype=torch.float64, device=dev) for m in engs}
        idx = torch.arange(B, dtype=torch.int32, device=dev)
        for tick in range(1, 6):
            b = synth.make_batch(B, seed=synth.SEED0 + 61, downwash=True, t0=0.02 * tick)
            d = {k: torch.from_numpy(b[k]).to(dev) for k in ("x0", "xr", "ur", "other", "ego_xy")}
            if tick == 1:
                for e in engs.values():
                    e.reset_device(d["xr"], d["ur"])
            other = pw.publish_device(d["other"])             # "this rank's windows" = the neighbour windows of the synthetic batch
            assert other.dev_ptr == pw.neighbour[tick & 1].dev_ptr
            torch.cuda.synchronize()
            assert torch.equal(pw.local[tick & 1], d["other"])
            engs["tensor"].update_device(d["x0"], d["xr"], d["ur"], u0["tensor"], other=d["other"], ego_xy=d["ego_xy"])
            engs["raw"].update_device(d["x0"], d["xr"], d["ur"], u0["raw"], other=other, ego_xy=d["ego_xy"])
            engs["raw_indexed"].update_device(d["x0"], d["xr"], d["ur"], u0["raw_indexed"], other=other, ego_xy=d["ego_xy"], other_index=idx)
            for e in engs.values():
                e.synchronize()
                assert (e.status()[0] == 0).all()
            assert torch.equal(u0["tensor"], u0["raw"]) and torch.equal(u0["tensor"], u0["raw_indexed"])
        st = pw.stats()
        assert st == dict(ticks=5, ack_timeouts=0, epoch_timeouts=0, slot_mismatches=0), st
        with pytest.raises(ValueError):
            engs["raw"].update_device(d["x0"], d["xr"], d["ur"], u0["raw"], other=ndist.DevWindows(pw.neighbour[1].dev_ptr, (B, N, 10)))
        for e in engs.values():
            e.close()
    finally:
        pw.close()


def test_peer_publish_replayed_from_a_graph(ndp):
    """Publish + control step captured into a hipGraph (an even number of ticks) and replayed: the tick number is read from the
    device-side epoch words, not baked into the launch, so every replay publishes new ticks into the right slots."""
    import torch
    from ndp_nmpc_qd_amd import dist as ndist
    B, N, T = 64, 20, 4
    dev = torch.device("cuda", 0)
    ticks = []
    for t in range(T):
        b = synth.make_batch(B, seed=synth.SEED0 + 62, downwash=True, t0=0.02 * t)
        ticks.append({k: torch.from_numpy(b[k]).to(dev) for k in ("x0", "xr", "ur", "other", "ego_xy")})
    pw = ndist.PeerWindows(B, N, 0)
    try:
        stream = torch.cuda.Stream(device=dev)
        outs = {}
        for mode in ("host", "graph"):
            eng = ndp.BatchedNMPC(B, N=N, disturbance=True)
            u0 = torch.empty(B, 4, dtype=torch.float64, device=dev)
            eng.reset_device(ticks[0]["xr"], ticks[0]["ur"], stream=stream)

            def step(i):
                d = ticks[i % T]
                eng.update_device(d["x0"], d["xr"], d["ur"], u0, other=pw.publish_device(d["other"], stream), ego_xy=d["ego_xy"], stream=stream)
            with torch.cuda.stream(stream):
                if mode == "host":
                    for i in range(3 * T):
                        step(i)
                else:
                    g = torch.cuda.CUDAGraph()
                    torch.cuda.synchronize()
                    with torch.cuda.graph(g, stream=stream, capture_error_mode="relaxed"):
                        for i in range(T):
                            step(i)
                    for _ in range(3):
                        g.replay()
            torch.cuda.synchronize()
            outs[mode] = u0.cpu().numpy().copy()
            eng.close()
        assert np.array_equal(outs["host"], outs["graph"])
        st = pw.stats()
        assert st["ticks"] == 6 * T and st["epoch_timeouts"] == 0 and st["ack_timeouts"] == 0 and st["slot_mismatches"] == 0, st
    finally:
        pw.close()


def _peer_proc(rank, world, port, q, leave_after):
    """One of two processes on the SAME GPU (a one-GPU box): each maps the other's window buffer over IPC and runs the per-tick
    publish launch + a copy of the neighbour's slot (standing in for the control step's read)."""
    import os
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ndp_nmpc_qd_amd import dist as ndist
    try:
        B, N = 256, 20
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        pw = ndist.PeerWindows(B, N, 0, timeout_us=50000 if leave_after else 2000000)
        ticks = 60
        mine = leave_after if (leave_after and rank == 1) else ticks
        bad = stale = 0
        seen = torch.empty(B, N + 1, 10, dtype=torch.float64, device=dev)
        seen_raw = ndist._DevMem(0, (B, N + 1, 10))
        for t in range(1, mine + 1):
            src = torch.full((B, N + 1, 10), 1000.0 * rank + t, dtype=torch.float64, device=dev)
            other = pw.publish_device(src)
            nb = torch.as_tensor(ndist._DevMem(other.dev_ptr, other.shape), device=dev)     # the neighbour's slot of this tick
            seen.copy_(nb)
            torch.cuda.synchronize()
            v = seen.unique()
            want = 1000.0 * ((rank + 1) % world) + t
            if v.numel() != 1:
                bad += 1                              # a torn window
            elif float(v[0]) != want:
                if leave_after and rank == 0 and t > leave_after:
                    stale += 1                        # the publisher has left: its last windows
                else:
                    bad += 1
        st = pw.stats()
        q.put((rank, bad, stale, st))
        pw.close(collective=not leave_after)
        if not leave_after:
            dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("leave_after", [0, 25])
def test_peer_exchange_between_two_processes(ndp, leave_after):
    """Two processes on this GPU, each reading the other's published windows through the IPC mapping, one exchange per tick:
    every read of tick t returns the windows of tick t.  leave_after = 25: rank 1 stops publishing and closes its buffer after
    25 ticks while rank 0 still has it mapped -- rank 0 keeps stepping on the last windows (its epoch waits time out and are
    counted), nothing crashes."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_peer_proc, args=(r, 2, port, q, leave_after)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (r0, bad0, stale0, st0), (r1, bad1, stale1, st1) = res
    assert bad0 == 0 and bad1 == 0, res
    if not leave_after:
        for st in (st0, st1):
            assert st["ticks"] == 60 and st["epoch_timeouts"] == 0 and st["ack_timeouts"] == 0 and st["slot_mismatches"] == 0, st
    else:
        assert st1["ticks"] == 25 and st0["ticks"] == 60
        assert stale0 >= 30 and st0["epoch_timeouts"] >= 30, (stale0, st0)


def test_host_step_in_two_halves_and_pipelined(ndp, oracle):
    """ndp_step_begin / ndp_step_end: the same answers as ndp_step tick by tick -- one tick in flight, two ticks in flight
    (the packing and H2D of tick i+1 under tick i's kernel) -- on the page-locked zero-copy path (small batch) and on the
    packed DMA path (inputs over 1 MiB); misuse is refused."""
    for B in (5, 700):
        ticks = [synth.make_batch(B, seed=77, downwash=True, t0=0.02 * t) for t in range(6)]
        ref = ndp.BatchedNMPC(B, disturbance=True)
        ref.reset(ticks[0]["xr"], ticks[0]["ur"])
        want = [ref.update(t["x0"], t["xr"], t["ur"], other=t["other"], ego_xy=t["ego_xy"], full=True) for t in ticks]
        for depth in (1, 2):
            eng = ndp.BatchedNMPC(B, disturbance=True)
            eng.reset(ticks[0]["xr"], ticks[0]["ur"])
            got = []
            for i, t in enumerate(ticks):
                arrs = {k: t[k].copy() for k in ("x0", "xr", "ur", "other", "ego_xy")}
                eng.update_begin(arrs["x0"], arrs["xr"], arrs["ur"], other=arrs["other"], ego_xy=arrs["ego_xy"], want_iterate=True)
                for a in arrs.values():
                    a[...] = np.nan                       # the caller's arrays are free again as soon as begin returns
                if i >= depth - 1:
                    got.append(eng.update_end(full=True))
            while len(got) < len(ticks):
                got.append(eng.update_end(full=True))
            for g, w in zip(got, want):
                for x, y in zip(g, w):
                    assert np.array_equal(x, y)
            X, U = eng.get_iterate()
            assert np.array_equal(X, want[-1][1]) and np.array_equal(U, want[-1][2])
            # misuse: a third step in flight, an end without a begin, the iterate of a step begun without the flag
            t = ticks[0]
            eng.update_begin(t["x0"], t["xr"], t["ur"], other=t["other"], ego_xy=t["ego_xy"])
            eng.update_begin(t["x0"], t["xr"], t["ur"], other=t["other"], ego_xy=t["ego_xy"])
            with pytest.raises(ndp.batched.NdpError):
                eng.update_begin(t["x0"], t["xr"], t["ur"], other=t["other"], ego_xy=t["ego_xy"])
            with pytest.raises(ndp.batched.NdpError):
                eng.update(t["x0"], t["xr"], t["ur"], other=t["other"], ego_xy=t["ego_xy"])
            with pytest.raises(ndp.batched.NdpError):
                eng.update_end(full=True)                 # X / U were not requested at begin
            eng.update_end()
            eng.update_end()
            with pytest.raises(ndp.batched.NdpError):
                eng.update_end()                          # nothing in flight any more
            eng.close()
        uo, sto, *_ = _oracle_batch(oracle, ticks[0], use_fd=True,
                                    f=oracle.downwash_batch(np.fromfile(ndp._lib.WEIGHTS_PATH, dtype="<f4"), ticks[0]["other"], ticks[0]["xr"], ticks[0]["ego_xy"]))
        _assert_u(want[0][0][sto == 0], uo[sto == 0])


def test_small_handles_keep_the_iterate_in_hbm(ndp):
    """Handles whose inputs fit the page-locked mirror: host steps and device steps interleave on ONE persistent iterate that
    lives in HBM (round 2 kept it in page-locked host memory, so device-resident steps of small handles crossed PCIe)."""
    import torch
    B = 8
    dev = torch.device("cuda", 0)
    ticks = [synth.make_batch(B, seed=78, t0=0.02 * t) for t in range(4)]
    a, b2 = ndp.BatchedNMPC(B), ndp.BatchedNMPC(B)
    a.reset(ticks[0]["xr"], ticks[0]["ur"])
    b2.reset(ticks[0]["xr"], ticks[0]["ur"])
    u_dev = torch.empty(B, 4, dtype=torch.float64, device=dev)
    for i, t in enumerate(ticks):
        ua = a.update(t["x0"], t["xr"], t["ur"])
        if i % 2 == 0:
            ub = b2.update(t["x0"], t["xr"], t["ur"])
        else:
            d = {k: torch.from_numpy(t[k]).to(dev) for k in ("x0", "xr", "ur")}
            b2.update_device(d["x0"], d["xr"], d["ur"], u_dev)
            b2.synchronize()
            ub = u_dev.cpu().numpy()
        assert np.array_equal(ua, ub), i
        assert (b2.status()[0] == 0).all()
    Xa, Ua = a.get_iterate()
    Xb, Ub = b2.get_iterate()
    assert np.array_equal(Xa, Xb) and np.array_equal(Ua, Ub)
    import ctypes
    attr_ptr = a._lib.ndp_device_iterate_x(a._h)
    assert torch.cuda.is_available() and attr_ptr            # a device pointer (HBM); the step's mirror is a different block


@pytest.mark.parametrize("scale", [10, 100])
def test_control_error_from_the_mlp_outside_its_training_envelope(ndp, mlp_golden, mlp_blob, scale):
    """What the MLP's rounding outside the training envelope does to the CONTROL: the 10x / 100x fixture rows (forces from the
    imported reference network, torch fp32) as neighbour offsets of a batch; u0 from (a) the fused device step (device MLP), (b)
    the same step fed the torch-fixture forces, (c) fed the float64 evaluation of the shipped weights.  The device's u0 is no
    further from (c) than 2.5x what the reference's own fp32 forces are, and (a) equals the unfused device path."""
    key = f"_x{scale}"
    z, f_torch = mlp_golden["z" + key], mlp_golden["f" + key]
    B = z.shape[0] // 21
    z, f_torch = z[:B * 21].reshape(B, 21, 6), f_torch[:B * 21].reshape(B, 21, 3)
    b = synth.make_batch(B, seed=91)
    other = b["xr"].copy()
    other[:, :, 0:6] += z.astype(np.float64)
    z_seen = (other - b["xr"])[:, :, 0:6].astype(np.float32)           # downwash_nn.py:22-23: what the network is fed
    f_true = _mlp_fp64(mlp_blob, z_seen.reshape(-1, 6)).reshape(B, 21, 3)

    def u0_with(f=None, **kw):
        eng = ndp.BatchedNMPC(B, disturbance=True)
        eng.reset(b["xr"], b["ur"])
        u = eng.update(b["x0"], b["xr"], b["ur"], f=f, raise_on_status=False, **kw)
        st, _ = eng.status()
        return u, st

    u_fused, st_a = u0_with(other=other)                               # gate always open (no ego_xy)
    eng = ndp.BatchedNMPC(B, disturbance=True)
    f_dev = eng.downwash(other, b["xr"])
    u_dev, _ = u0_with(f=f_dev)
    u_fix, st_b = u0_with(f=f_torch)
    u_true, st_c = u0_with(f=f_true.astype(np.float32))
    ok = (st_a == 0) & (st_b == 0) & (st_c == 0)
    assert np.array_equal(st_a, st_b) and np.array_equal(st_a, st_c)      # whichever evaluation of the network: the same solver verdict
    if scale == 100:
        # forces of ~1.4 kN on a 1.5 kg vehicle: no QP of the batch converges inside the input box -- with ANY of the three
        # force arrays; there is no control to compare (the 10x case, up to 142 N, is the informative one)
        assert np.abs(f_true).max() > 500.0 and ok.sum() == 0
        return
    assert ok.sum() >= B // 2
    np.testing.assert_allclose(u_fused[ok], u_dev[ok], rtol=0, atol=1e-12)
    den = np.maximum(1.0, np.abs(u_true[ok]))
    e_dev = (np.abs(u_fused[ok] - u_true[ok]) / den).max()
    e_fix = (np.abs(u_fix[ok] - u_true[ok]) / den).max()
    e_df = (np.abs(u_fused[ok] - u_fix[ok]) / den).max()
    print(f"scale {scale}: |f| up to {np.abs(f_true).max():.1f} N; u0 device-MLP vs fp64-MLP {e_dev:.2e}, torch-fixture vs fp64-MLP {e_fix:.2e}, "
          f"device vs torch-fixture {e_df:.2e}")
    assert e_dev <= 2.5 * e_fix + 1e-9
    assert e_df <= 1e-3


def test_downwash_one_tick_ahead_on_the_second_stream(ndp, oracle, mlp_blob):
    """ndp_downwash_prefetch_device + ndp_step_device_prefetched: the force of tick t+1 is predicted by a second launch while
    tick t is being solved; the control step takes it late (after its linearisation) and corrects the dynamics defects.  Same
    controls as the fused single launch to rounding, and the oracle's; host-launched and replayed from a hipGraph (fork at the
    first prediction, join at the end); ragged batch; the protocol's counters add up."""
    import torch
    B, N, T = 300, 20, 6
    dev = torch.device("cuda", 0)
    ticks, host = [], []
    for t in range(T):
        b = synth.make_batch(B, seed=synth.SEED0 + 71, downwash=True, t0=0.02 * t)
        host.append(b)
        ticks.append({k: torch.from_numpy(b[k]).to(dev) for k in ("x0", "xr", "ur", "other", "ego_xy")})
    stream = torch.cuda.Stream(device=dev)
    fused = ndp.BatchedNMPC(B, disturbance=True)
    uf = torch.empty(T, B, 4, dtype=torch.float64, device=dev)
    fused.reset_device(ticks[0]["xr"], ticks[0]["ur"], stream=stream)
    for t, d in enumerate(ticks):
        fused.update_device(d["x0"], d["xr"], d["ur"], uf[t], other=d["other"], ego_xy=d["ego_xy"], stream=stream)
    fused.synchronize()
    assert (fused.status()[0] == 0).all()
    want = uf.cpu().numpy()

    def run(eng, out, first_after):
        d = ticks[0]
        eng.downwash_prefetch_device(d["other"], d["xr"], ego_xy=d["ego_xy"], after_stream=first_after)
        for t, d in enumerate(ticks):
            if t + 1 < T:
                n = ticks[t + 1]
                eng.downwash_prefetch_device(n["other"], n["xr"], ego_xy=n["ego_xy"])      # beside this tick's control step
            eng.update_device_prefetched(d["x0"], d["xr"], d["ur"], out[t], stream=stream)

    for mode in ("host", "graph"):
        eng = ndp.BatchedNMPC(B, disturbance=True)
        up = torch.empty(T, B, 4, dtype=torch.float64, device=dev)
        eng.reset_device(ticks[0]["xr"], ticks[0]["ur"], stream=stream)
        with torch.cuda.stream(stream):
            if mode == "host":
                run(eng, up, stream)
            else:
                eng.downwash_prefetch_device(ticks[0]["other"], ticks[0]["xr"], ego_xy=ticks[0]["ego_xy"])   # allocate outside the capture
                eng.update_device_prefetched(ticks[0]["x0"], ticks[0]["xr"], ticks[0]["ur"], up[0], stream=stream)
                eng.synchronize()
                eng.reset_device(ticks[0]["xr"], ticks[0]["ur"], stream=stream)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=stream, capture_error_mode="relaxed"):
                    run(eng, up, stream)
                    eng.prefetch_join(stream)
                g.replay()
        eng.prefetch_join(stream)
        torch.cuda.synchronize()                 # (the replay ran on torch's stream: the engine's own synchronize does not know it)
        st = eng.prefetch_stats()
        extra = 1 if mode == "graph" else 0
        late = st.pop("late_waves")
        assert st == dict(predictions=T + extra, steps=T + extra, force_timeouts=0, slot_timeouts=0), st
        assert (eng.status()[0] == 0).all()
        got = up.cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)
        f_last = eng.force_slot((T + extra) & 1).cpu().numpy()
        f_or = oracle.downwash_batch(mlp_blob, host[-1]["other"], host[-1]["xr"], host[-1]["ego_xy"])
        assert np.all(np.abs(f_last - f_or) <= 1e-5 * np.maximum(1.0, np.abs(f_or)))
        eng.close()
    f0 = oracle.downwash_batch(mlp_blob, host[0]["other"], host[0]["xr"], host[0]["ego_xy"])
    uo, sto, *_ = _oracle_batch(oracle, host[0], use_fd=True, f=f0)
    _assert_u(want[0][sto == 0], uo[sto == 0], 1e-6)


def test_peer_windows_one_tick_ahead_beside_the_control_steps(ndp):
    """bench.py's exchange.peer_ahead on one rank: a second stream carries, per tick, the publish of the tick's windows into the peer
    slot (copy launch, epoch launch that waits for the neighbour's epoch) and the gate / MLP launch that reads the NEIGHBOUR'S SLOT (a
    raw device address) -- one tick ahead of the control steps on the first stream, which take the force late (the lean instantiation:
    it shares the SIMDs with the downwash launch).  Host-launched and as two hipGraphs replayed side by side (an even number of ticks:
    the slot parity is baked into a launch).  Same controls as the fused launch that is handed the windows as a tensor; epochs count the
    ticks; nothing times out."""
    import torch
    from ndp_nmpc_qd_amd import dist as ndist
    B, N, T = 512, 20, 8
    dev = torch.device("cuda", 0)
    ticks = []
    for t in range(T):
        b = synth.make_batch(B, seed=synth.SEED0 + 83, downwash=True, t0=0.02 * t)
        ticks.append({k: torch.from_numpy(b[k]).to(dev) for k in ("x0", "xr", "ur", "other", "ego_xy")})
    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev, priority=-1)
    fused = ndp.BatchedNMPC(B, disturbance=True)
    uf = torch.empty(T, B, 4, dtype=torch.float64, device=dev)
    fused.reset_device(ticks[0]["xr"], ticks[0]["ur"], stream=sa)
    for t, d in enumerate(ticks):
        fused.update_device(d["x0"], d["xr"], d["ur"], uf[t], other=d["other"], ego_xy=d["ego_xy"], stream=sa)
    fused.synchronize()
    assert (fused.status()[0] == 0).all()
    fused.close()
    pw = ndist.PeerWindows(B, N, 0)
    try:
        def ahead(eng, d, first=False):       # "this rank's windows" = the synthetic batch's neighbour windows (one rank: own buffer = neighbour's)
            eng.downwash_prefetch_device(pw.publish_device(d["other"], sb), d["xr"], ego_xy=d["ego_xy"], on_stream=sb,
                                         after_stream=sa if first else None)
        for mode in ("host", "graphs"):
            eng = ndp.BatchedNMPC(B, disturbance=True)
            up = torch.empty(T, B, 4, dtype=torch.float64, device=dev)
            eng.reset_device(ticks[0]["xr"], ticks[0]["ur"], stream=sa)
            base = pw.stats()["ticks"]
            pw.tick = base
            if mode == "host":
                ahead(eng, ticks[0], first=True)
                for t, d in enumerate(ticks):
                    if t + 1 < T:
                        ahead(eng, ticks[t + 1])
                    eng.update_device_prefetched(d["x0"], d["xr"], d["ur"], up[t], stream=sa)
                extra = 0
            else:
                ahead(eng, ticks[0], first=True)                        # allocations outside the captures; two ticks: the parity stays
                ahead(eng, ticks[1])
                for t in (0, 1):
                    eng.update_device_prefetched(ticks[t]["x0"], ticks[t]["xr"], ticks[t]["ur"], up[t], stream=sa)
                torch.cuda.synchronize()
                eng.reset_device(ticks[0]["xr"], ticks[0]["ur"], stream=sa)
                torch.cuda.synchronize()
                ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(ga, stream=sa, capture_error_mode="relaxed"):
                    for t, d in enumerate(ticks):
                        eng.update_device_prefetched(d["x0"], d["xr"], d["ur"], up[t], stream=sa)
                with torch.cuda.graph(gb, stream=sb, capture_error_mode="relaxed"):
                    for d in ticks:
                        ahead(eng, d)
                with torch.cuda.stream(sb):
                    gb.replay()
                with torch.cuda.stream(sa):
                    ga.replay()
                extra = 2
            torch.cuda.synchronize()
            ps = pw.stats()
            assert ps == dict(ticks=base + T + extra, ack_timeouts=0, epoch_timeouts=0, slot_mismatches=0), (mode, ps)
            st = eng.prefetch_stats()
            st.pop("late_waves")
            assert st == dict(predictions=T + extra, steps=T + extra, force_timeouts=0, slot_timeouts=0), (mode, st)
            assert (eng.status()[0] == 0).all()
            np.testing.assert_allclose(up.cpu().numpy(), uf.cpu().numpy(), rtol=0, atol=1e-9)
            eng.close()
    finally:
        pw.close()


def test_downwash_prediction_enqueued_after_its_control_step_takes_the_epoch_path(ndp):
    """The LATE path of the downwash-ahead protocol on purpose: every tick's control step is launched BEFORE the
    prediction it consumes, so its waves find PF_MLP_DONE < t at start, wait on their tiles' epoch words and load the force rows
    past the L2 -- the path on which an epoch word published before its rows had completed would hand over the force of tick t - 2.
    The forces differ from tick to tick (the neighbour moves), so a stale row shows as a control error.  Same controls as the
    fused launch; late waves counted; no timeout."""
    import torch
    B, N, T = 512, 20, 8
    dev = torch.device("cuda", 0)
    ticks = []
    for t in range(T):
        b = synth.make_batch(B, seed=synth.SEED0 + 72, downwash=True, t0=0.02 * t)
        b["other"][:, :, 0:3] += 0.08 * t          # the neighbour drifts: the force of tick t - 2 is visibly not the force of tick t
        ticks.append({k: torch.from_numpy(b[k]).to(dev) for k in ("x0", "xr", "ur", "other", "ego_xy")})
    stream = torch.cuda.Stream(device=dev)
    fused = ndp.BatchedNMPC(B, disturbance=True)
    uf = torch.empty(T, B, 4, dtype=torch.float64, device=dev)
    ff = torch.empty(T, B, N + 1, 3, dtype=torch.float32, device=dev)
    fused.reset_device(ticks[0]["xr"], ticks[0]["ur"], stream=stream)
    for t, d in enumerate(ticks):
        fused.update_device(d["x0"], d["xr"], d["ur"], uf[t], other=d["other"], ego_xy=d["ego_xy"], stream=stream)
        with torch.cuda.stream(stream):
            ff[t].copy_(fused.device_force())
    fused.synchronize()
    torch.cuda.synchronize()
    fh = ff.cpu().numpy()
    assert np.abs(fh[2:] - fh[:-2]).max() > 1e-2                      # the test can tell tick t's force from tick t - 2's
    eng = ndp.BatchedNMPC(B, disturbance=True)
    up = torch.empty(T, B, 4, dtype=torch.float64, device=dev)
    eng.reset_device(ticks[0]["xr"], ticks[0]["ur"], stream=stream)
    # allocate the protocol state, then start from a clean count: one ordinary pair first
    d = ticks[0]
    eng.downwash_prefetch_device(d["other"], d["xr"], ego_xy=d["ego_xy"], after_stream=stream)
    eng.update_device_prefetched(d["x0"], d["xr"], d["ur"], up[0], stream=stream)
    eng.prefetch_join(stream)
    torch.cuda.synchronize()
    eng.reset_device(ticks[0]["xr"], ticks[0]["ur"], stream=stream)
    torch.cuda.synchronize()
    late0 = eng.prefetch_stats()["late_waves"]
    for t, d in enumerate(ticks):
        eng.update_device_prefetched(d["x0"], d["xr"], d["ur"], up[t], stream=stream)    # the step first: it has to wait for ...
        time.sleep(0.002)
        eng.downwash_prefetch_device(d["other"], d["xr"], ego_xy=d["ego_xy"])           # ... the prediction enqueued behind it
        eng.prefetch_join(stream)
        torch.cuda.synchronize()
    st = eng.prefetch_stats()
    assert st["force_timeouts"] == 0 and st["slot_timeouts"] == 0, st
    assert st["late_waves"] - late0 >= T * B // 2, st                 # the epoch path really ran (every wave of every tick, ideally)
    assert (eng.status()[0] == 0).all()
    np.testing.assert_allclose(up.cpu().numpy(), uf.cpu().numpy(), rtol=0, atol=1e-9)
    eng.close()


def test_downwash_prefetch_misuse_is_bounded_and_reported(ndp):
    """A control step whose prediction was never enqueued gives up after the bounded wait: zero force, status 5 on every instance,
    counted; a third prediction ahead of the steps waits for a free slot and is counted too.  Nothing hangs."""
    import torch
    B = 64
    dev = torch.device("cuda", 0)
    b = synth.make_batch(B, seed=9, downwash=True)
    d = {k: torch.from_numpy(b[k]).to(dev) for k in ("x0", "xr", "ur", "other", "ego_xy")}
    eng = ndp.BatchedNMPC(B, disturbance=True)
    u = torch.empty(B, 4, dtype=torch.float64, device=dev)
    eng.reset_device(d["xr"], d["ur"])
    eng.downwash_prefetch_device(d["other"], d["xr"], ego_xy=d["ego_xy"])
    eng.update_device_prefetched(d["x0"], d["xr"], d["ur"], u)
    eng.synchronize()
    assert (eng.status()[0] == 0).all()
    eng.update_device_prefetched(d["x0"], d["xr"], d["ur"], u)          # no prediction for it
    eng.synchronize()
    st = eng.prefetch_stats()
    assert (eng.status()[0] == 5).all() and st["force_timeouts"] == B and st["steps"] == 2 and st["predictions"] == 1, st
    plain = ndp.BatchedNMPC(B)                                          # zero force = the plain NMPC step from the same iterate
    X, U = eng.get_iterate()
    for _ in range(3):                                                   # predictions 2, 3 fill the slots; 4 has to wait for step 2 ... which was taken
        eng.downwash_prefetch_device(d["other"], d["xr"], ego_xy=d["ego_xy"])
    eng.downwash_prefetch_device(d["other"], d["xr"], ego_xy=d["ego_xy"])    # prediction 5: its slot's reader (step 3) never ran
    st = eng.prefetch_stats()
    assert st["predictions"] == 5 and st["slot_timeouts"] >= 1, st
    with pytest.raises(ndp.batched.NdpError):
        ndp.BatchedNMPC(B).downwash_prefetch_device(d["other"], d["xr"])           # plain NMPC model: no force input
    big = ndp.BatchedNMPC(4096, disturbance=True, work_queue=1)                    # work list forced on: not combined (the automatic rule
    with pytest.raises(ndp.batched.NdpError):                                      # keeps the late-force step in place instead)
        big.downwash_prefetch_device(d["other"], torch.zeros(4096, 21, 10, dtype=torch.float64, device=dev),
                                     other_index=torch.zeros(4096, dtype=torch.int32, device=dev))


_RCCL_ONE_RANK = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", sys.argv[2]
import numpy as np, torch, torch.distributed as dist
import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import dist as ndist
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
B, N = 96, 20
b = ndist.make_formation_shard(B, 0, 1, N=N)
t = {k: torch.from_numpy(b[k]).to(dev) for k in ("x0", "xr", "ur", "ego_xy", "other")}
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
# the exchange through RCCL (one rank: the gathered buffer must equal the packed position / velocity columns) ...
pv = torch.empty(B, N + 1, ndist.PV_COLS, dtype=torch.float64, device=dev)
gathered = torch.full((B, N + 1, ndist.PV_COLS), float("nan"), dtype=torch.float64, device=dev)
w = ndist.exchange_pv_begin(t["other"], pv, gathered, force_collective=True)
assert w is not None, "the collective was not called"
ndist.exchange_pv_end(w)
u_a = torch.empty(B, 4, dtype=torch.float64, device=dev)
eng = ndp.BatchedNMPC(B, N=N, disturbance=True)
eng.reset_device(t["xr"], t["ur"], stream=stream)
# ... consumed by the control step launched next on the same stream (no host synchronisation in between)
eng.update_device(t["x0"], t["xr"], t["ur"], u_a, other=gathered, ego_xy=t["ego_xy"], stream=stream)
torch.cuda.synchronize()
assert torch.equal(gathered, t["other"][:, :, :ndist.PV_COLS])
u_b = torch.empty(B, 4, dtype=torch.float64, device=dev)
eng.reset_device(t["xr"], t["ur"], stream=stream)
eng.update_device(t["x0"], t["xr"], t["ur"], u_b, other=t["other"], ego_xy=t["ego_xy"], stream=stream)
torch.cuda.synchronize()
assert torch.equal(u_a, u_b), float((u_a - u_b).abs().max())
# the same exchange + step captured into a hipGraph and replayed with other windows (bench.py --graph-exchange)
g = torch.cuda.CUDAGraph()
src = t["other"].clone()
eng.reset_device(t["xr"], t["ur"], stream=stream)
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=stream, capture_error_mode="relaxed"):
    w = ndist.exchange_pv_begin(src, pv, gathered, force_collective=True)
    ndist.exchange_pv_end(w)
    eng.update_device(t["x0"], t["xr"], t["ur"], u_a, other=gathered, ego_xy=t["ego_xy"], stream=stream)
torch.cuda.set_stream(stream)
src[:, :, 2] += 0.25                      # other neighbour heights: the replay must gather and use the NEW windows
eng.reset_device(t["xr"], t["ur"], stream=stream)
g.replay()
torch.cuda.synchronize()
assert torch.equal(gathered, src[:, :, :ndist.PV_COLS])
eng.reset_device(t["xr"], t["ur"], stream=stream)
eng.update_device(t["x0"], t["xr"], t["ur"], u_b, other=src, ego_xy=t["ego_xy"], stream=stream)
torch.cuda.synchronize()
assert torch.equal(u_a, u_b), float((u_a - u_b).abs().max())
assert float((u_a - u_b).abs().max()) == 0.0 and bool(torch.isfinite(u_a).all())
dist.destroy_process_group()
print("RCCL-ONE-RANK-OK")
"""


@pytest.mark.gpu
def test_rccl_exchange_call_path_one_rank_group():
    """The RCCL form of the per-tick exchange (dist.exchange_pv_begin / _end: pack, all_gather_into_tensor started async, the
    compute stream made to wait, the control step launched behind it) through a REAL RCCL communicator -- one rank, all a
    one-GPU box offers: the gathered buffer equals the pack, the step that consumes it equals the step given the windows
    directly, and the pair replays inside a hipGraph with new windows.  In a child process (a process group is global state)."""
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK, root, str(port)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL-ONE-RANK-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


_XCHG_ONE_RANK = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np, torch
import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import dist as ndist
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
B, N = 96, 20
b = ndist.make_formation_shard(B, 0, 1, N=N)
t = {k: torch.from_numpy(b[k]).to(dev) for k in ("x0", "xr", "ur", "ego_xy", "other")}
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
ex = ndist.RcclExchange(B, N, 0)                     # one rank: no torch.distributed needed, the communicator is real
gathered = torch.full((B, N + 1, ndist.PV_COLS), float("nan"), dtype=torch.float64, device=dev)
src = t["other"].clone()
ex.begin(src, gathered, stream)
ex.end(stream)
eng = ndp.BatchedNMPC(B, N=N, disturbance=True)
u_a = torch.empty(B, 4, dtype=torch.float64, device=dev)
u_b = torch.empty(B, 4, dtype=torch.float64, device=dev)
eng.reset_device(t["xr"], t["ur"], stream=stream)
eng.update_device(t["x0"], t["xr"], t["ur"], u_a, other=gathered, ego_xy=t["ego_xy"], stream=stream)   # no host wait in between
torch.cuda.synchronize()
assert torch.equal(gathered, src[:, :, :ndist.PV_COLS])
eng.reset_device(t["xr"], t["ur"], stream=stream)
eng.update_device(t["x0"], t["xr"], t["ur"], u_b, other=src, ego_xy=t["ego_xy"], stream=stream)
torch.cuda.synchronize()
assert torch.equal(u_a, u_b), float((u_a - u_b).abs().max())
# begin + end + control step captured into a hipGraph, replayed with other windows
g = torch.cuda.CUDAGraph()
eng.reset_device(t["xr"], t["ur"], stream=stream)
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=stream, capture_error_mode="relaxed"):
    ex.begin(src, gathered, stream)
    ex.end(stream)
    eng.update_device(t["x0"], t["xr"], t["ur"], u_a, other=gathered, ego_xy=t["ego_xy"], stream=stream)
torch.cuda.set_stream(stream)
src[:, :, 2] += 0.25
eng.reset_device(t["xr"], t["ur"], stream=stream)
g.replay()
torch.cuda.synchronize()
assert torch.equal(gathered, src[:, :, :ndist.PV_COLS])
eng.reset_device(t["xr"], t["ur"], stream=stream)
eng.update_device(t["x0"], t["xr"], t["ur"], u_b, other=src, ego_xy=t["ego_xy"], stream=stream)
torch.cuda.synchronize()
assert torch.equal(u_a, u_b) and bool(torch.isfinite(u_a).all())
# the pipelined form bench.py uses: two gathered buffers, the gather of tick i + 1 begun beside tick i's step and ordered behind the
# step that read its target buffer last through that step's own completion event (ndp_track_steps: no packet on the compute stream)
eng.track_steps(True)
bufs = [torch.empty_like(gathered) for _ in range(2)]
wins = [t["other"] + 0.01 * k for k in range(8)]
outs = [torch.empty(B, 4, dtype=torch.float64, device=dev) for _ in range(8)]
eng.reset_device(t["xr"], t["ur"], stream=stream)
ex.begin(wins[0], bufs[0], stream)
for i in range(8):
    ex.end(stream)
    if i + 1 < 8:
        try:
            ev = eng.last_step_event()
        except ndp.NdpError:
            ev = None
        if ev is None:
            ex.begin(wins[i + 1], bufs[(i + 1) % 2], stream)
        else:
            ex.begin(wins[i + 1], bufs[(i + 1) % 2], None, after_event=ev)
    eng.update_device(t["x0"], t["xr"], t["ur"], outs[i], other=bufs[i % 2], ego_xy=t["ego_xy"], stream=stream)
torch.cuda.synchronize()
# the same ticks with end(i) + begin(i + 1) as ONE call (ndp_xchg_tick picks the last tracked step's event itself)
outs2 = [torch.empty(B, 4, dtype=torch.float64, device=dev) for _ in range(8)]
eng.reset_device(t["xr"], t["ur"], stream=stream)
ex.begin(wins[0], bufs[0], stream)
for i in range(8):
    if i + 1 < 8:
        ex.tick(eng, wins[i + 1], bufs[(i + 1) % 2], stream)
    else:
        ex.end(stream)
    eng.update_device(t["x0"], t["xr"], t["ur"], outs2[i], other=bufs[i % 2], ego_xy=t["ego_xy"], stream=stream)
torch.cuda.synchronize()
eng.track_steps(False)
eng.reset_device(t["xr"], t["ur"], stream=stream)
for i in range(8):
    eng.update_device(t["x0"], t["xr"], t["ur"], u_b, other=wins[i], ego_xy=t["ego_xy"], stream=stream)
    torch.cuda.synchronize()
    assert torch.equal(outs[i], u_b), (i, float((outs[i] - u_b).abs().max()))
    assert torch.equal(outs2[i], u_b), (i, float((outs2[i] - u_b).abs().max()))
ex.close()
print("XCHG-ONE-RANK-OK")
"""


@pytest.mark.gpu
def test_library_issued_rccl_all_gather_one_rank_communicator():
    """ndp_xchg_* (the all-gather issued by the C-ABI library on its own stream): a REAL RCCL communicator with one rank -- the gathered
    buffer equals the packed position / velocity columns, the control step launched behind ndp_xchg_end (no host wait) equals the
    step given the windows directly, and begin + end + step replay from a hipGraph with new windows.  In a child process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _XCHG_ONE_RANK, root], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "XCHG-ONE-RANK-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
