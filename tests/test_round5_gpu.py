"""Round-5 GPU parity cases beside the tick: the float64 force entry (ndp_step_ex_f64), workgroups none of whose instances has a
neighbour (the fused kernel's short cut), the refinement getter."""
import numpy as np
import pytest

from ndp_nmpc_qd_amd import synth

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))


def test_float64_force_is_not_rounded_to_float32(oracle):
    """ndp_nmpc_body_rate_ctl.py:93-104: p_k = [q_r, f_k] is a float64 vector.  A force that is NOT fp32-representable must reach the
    dynamics with all its digits -- through BatchedNMPC.update(f = float64 array) and through the drop-in class's facade."""
    import ndp_nmpc_qd_amd as ndp
    from ndp_nmpc_qd_amd.ndp_nmpc_ctl import NDPNMPCBodyRateController
    B = 64
    b = synth.make_batch(B, seed=5)
    rng = np.random.default_rng(8)
    f = rng.normal(0, 2.0, (B, 21, 3)) * (1.0 + 1e-3 * rng.normal(size=(B, 21, 3)))          # float64, 53 significant bits
    assert np.max(np.abs(f - f.astype(np.float32))) > 1e-8
    cfg = oracle.default_cfg(use_fd=True)
    cfg.qp_mode = 0                                         # the device's rule (exact early exit): no interior-point termination error in between
    X, U = b["xr"].copy(), b["ur"].copy()
    u_or, st, _ = oracle.step_batch(cfg, b["x0"], b["xr"], b["ur"], f, X, U)
    eng = ndp.BatchedNMPC(B, disturbance=True, load_mlp=False)
    eng.reset(b["xr"], b["ur"])
    u64, X64, U64, st64, _ = eng.update(b["x0"], b["xr"], b["ur"], f=f, full=True)
    assert not st.any() and not st64.any()
    e64 = _rel(u64, u_or)
    eng.reset(b["xr"], b["ur"])
    u32 = eng.update(b["x0"], b["xr"], b["ur"], f=f.astype(np.float32))
    e32 = _rel(u32, u_or)
    assert e64 < 1e-9, e64
    assert e32 > 5 * e64, (e32, e64)                       # the fp32 entry does round (that is what it is for: DownwashNN's output)
    assert _rel(X64, X) < 1e-9
    # fp32-representable forces: the two entries agree bit for bit
    f32 = f.astype(np.float32)
    eng.reset(b["xr"], b["ur"])
    ua = eng.update(b["x0"], b["xr"], b["ur"], f=f32)
    eng.reset(b["xr"], b["ur"])
    ub = eng.update(b["x0"], b["xr"], b["ur"], f=f32.astype(np.float64))
    assert np.array_equal(ua, ub)
    # the drop-in class (reference call shape: update(x0, xr, ur, f) with f float64)
    ctl = NDPNMPCBodyRateController()
    ctl.reset(b["xr"][3], b["ur"][3])
    u1 = ctl.update(b["x0"][3], b["xr"][3], b["ur"][3], f[3])
    assert _rel(u1, u_or[3]) < 1e-9


def test_workgroups_without_any_neighbour_take_the_short_cut_and_match(oracle, mlp_blob):
    """Config 4's local order puts a rank's followers behind its leaders: whole workgroups (4 instances) whose other_index is -1
    skip the weight transfer, both barriers and the network (rti_kernel: FUSED && !wg_nb).  Bit-compared with the same
    instances run with a closed gate (the ordinary fused path), and against the oracle; a ragged batch so that the last
    workgroup is partly idle, and one workgroup that mixes both kinds."""
    import torch
    import ndp_nmpc_qd_amd as ndp
    B = 4 * 9 + 2                                              # 9 full workgroups + a ragged one
    b = synth.make_batch(B, seed=17, downwash=True)
    dev = torch.device("cuda", 0)
    idx = np.arange(B, dtype=np.int32)                         # neighbour windows = b["other"] rows
    idx[8:24] = -1                                             # workgroups 2..5: nobody has a neighbour
    idx[25] = -1                                               # workgroup 6: mixed
    idx[36:] = -1                                              # the ragged last workgroup: nobody
    t = {k: torch.from_numpy(np.ascontiguousarray(b[k])).to(dev) for k in ("x0", "xr", "ur", "other", "ego_xy")}
    u_short = torch.empty(B, 4, dtype=torch.float64, device=dev)
    u_gate = torch.empty(B, 4, dtype=torch.float64, device=dev)
    eng = ndp.BatchedNMPC(B, disturbance=True)
    eng.reset_device(t["xr"], t["ur"])
    eng.update_device(t["x0"], t["xr"], t["ur"], u_short, other=t["other"], ego_xy=t["ego_xy"], other_index=torch.from_numpy(idx).to(dev))
    eng.synchronize()
    f_short = eng.device_force().cpu().numpy().copy()
    Xs, Us = eng.get_iterate()
    # the same instances through the ordinary path: everybody has a row, the gate of the no-neighbour ones is shut (ego far away)
    ego = b["ego_xy"].copy()
    ego[idx < 0] = 1e9
    eng.reset_device(t["xr"], t["ur"])
    eng.update_device(t["x0"], t["xr"], t["ur"], u_gate, other=t["other"], ego_xy=torch.from_numpy(ego).to(dev),
                      other_index=torch.from_numpy(np.arange(B, dtype=np.int32)).to(dev))
    eng.synchronize()
    Xg, Ug = eng.get_iterate()
    assert np.array_equal(u_short.cpu().numpy(), u_gate.cpu().numpy())
    assert np.array_equal(f_short, eng.device_force().cpu().numpy())
    assert np.array_equal(Xs, Xg) and np.array_equal(Us, Ug)
    assert not f_short[idx < 0].any() and f_short[idx >= 0].any()
    f = oracle.downwash_batch(mlp_blob, b["other"], b["xr"], ego)
    cfg = oracle.default_cfg(use_fd=True)
    X, U = b["xr"].copy(), b["ur"].copy()
    u_or, st, _ = oracle.step_batch(cfg, b["x0"], b["xr"], b["ur"], f, X, U)
    ok = st == 0
    assert _rel(u_short.cpu().numpy()[ok], u_or[ok]) < 1e-5 and ok.sum() >= B - 2


def test_refine_getter_says_where_ipm_refine_acts():
    import ndp_nmpc_qd_amd as ndp
    assert ndp.BatchedNMPC(8, load_mlp=False).refine_active                       # N = 20: three-slot kernels
    assert not ndp.BatchedNMPC(8, N=40, load_mlp=False).refine_active             # five-slot kernels ignore it
    assert not ndp.BatchedNMPC(8, load_mlp=False, ipm_refine=0).refine_active


@pytest.mark.parametrize("B", [1, 49, 1000])
def test_relay_reference_dense_piece_copy_at_ragged_sizes(B):
    """relay_reference_kernel is a dense copy in 16-byte pieces, four per thread: batches whose piece count is not a multiple of a
    block's 1024 (and one below a single wave) against the definition -- positions + the filtered offset, everything else untouched
    (nmpc_follower_node.py:58-74), bit for bit."""
    import ndp_nmpc_qd_amd as ndp
    rng = np.random.default_rng(B)
    eng = ndp.BatchedNMPC(B, load_mlp=False)
    form = rng.normal(size=(B, 3))
    off = eng.relay_formation(form)                       # first message: alpha u + (1 - alpha) u (alpha_filter.py:19), u to a rounding
    np.testing.assert_allclose(off, form, rtol=1e-15, atol=0)
    lead = rng.normal(size=(B, eng.N + 1, 10))
    want = lead.copy()
    want[:, :, 0:3] += off[:, None, :]
    assert np.array_equal(eng.relay_reference(lead), want)


@pytest.mark.parametrize("B", [1, 3, 61, 1000])
def test_reference_windows_with_streaming_stores_at_ragged_sizes(B):
    """ref_window_kernel transposes 64 rows through LDS and streams them out; B (N + 1) not a multiple of 64, and fewer rows than one
    wave: against the list path's windows (ring fill + dense copy), which share the point evaluation bit for bit."""
    import ndp_nmpc_qd_amd as ndp
    from ndp_nmpc_qd_amd import synth
    tr = synth.figure_eight_traj(B, seed=7 + B, n_seg=12, t_seg=0.5)
    eng = ndp.BatchedNMPC(B, load_mlp=False)
    eng.ref_set_trajectory(tr["coeff_x"], tr["coeff_y"], tr["coeff_z"], tr["coeff_yaw"], tr["time_cum"], tr["time_seg"], tr["final_pt"])
    t = np.linspace(0.0, 3.0, B)
    xr, ur = eng.ref_window(t)
    assert np.isfinite(xr).all() and np.isfinite(ur).all()
    # the same points one at a time (rows of other windows): node k of the window at t = node 0 of the window at t + k dt
    k = eng.N // 2
    from ndp_nmpc_qd_amd.params import nmpc_params as CP
    xr_k, ur_k = eng.ref_window(t + k * CP.th_pred)
    # (the node time is (t + T_offset) + k dt in the kernel and (t + k dt) + T_offset here: an ulp of time, hence not array_equal)
    np.testing.assert_allclose(xr[:, k, :], xr_k[:, 0, :], rtol=0, atol=1e-11)
    np.testing.assert_allclose(ur[:, k, :], ur_k[:, 0, :], rtol=0, atol=1e-9)
