"""CPU tests of the host side: C-ABI exports, config constants, the `.solver` facade, synthetic inputs,
and the multi-process neighbour exchange (gloo, world_size 2).  No compute call needs a GPU here."""
import ctypes as C
import os
import re
import socket

import numpy as np
import pytest

from ndp_nmpc_qd_amd import _lib, synth
from ndp_nmpc_qd_amd.params import downwash_params as DP
from ndp_nmpc_qd_amd.params import nmpc_params as CP

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ndp_nmpc.h")).read()
    declared = set(re.findall(r"\b(ndp_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = _lib.load()
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing
    assert declared == set(_lib.EXPORTS)


def test_default_cfg_carries_the_reference_constants():
    cfg = _lib.default_cfg()
    assert (cfg.N, cfg.n_rti, cfg.use_fd, cfg.iter_max) == (CP.N_node, 1, 0, 50)
    assert cfg.dt == CP.th_pred == 0.1 and cfg.mass == CP.mass == 1.4844 and cfg.gravity == 9.81
    assert cfg.r_horiz == DP.r_horiz
    assert list(cfg.Qd) == [CP.Qp_xy, CP.Qp_xy, CP.Qp_z, CP.Qv_xy, CP.Qv_xy, CP.Qv_z, 0, CP.Qq_xy, CP.Qq_xy, CP.Qq_z]
    assert list(cfg.Rd) == [CP.Rw, CP.Rw, CP.Rw, CP.Rc]
    assert list(cfg.lbu) == [CP.w_min] * 3 + [CP.c_min] and list(cfg.ubu) == [CP.w_max] * 3 + [CP.c_max]
    assert list(cfg.lbv) == [CP.v_min] * 3 and list(cfg.ubv) == [CP.v_max] * 3
    assert CP.c_max == 9.81 / 0.36
    assert C.sizeof(_lib.NdpCfg) == 12 * 4 + 8 * (4 + 10 + 4 + 4 + 4 + 3 + 3 + 9)      # (+ as_iter_max, a pad word, as_gamma: round 6)
    assert (cfg.as_iter_max, cfg.as_gamma) == (8, 1e12)
    assert cfg.ipm_refine == 2 and cfg.refine_gamma == 1e4
    assert cfg.auto_margin == 0.1 and cfg.qp_precision == 0 and cfg.work_queue == 0 and cfg.ts_nmpc == CP.ts_nmpc == 0.02
    # interior-point constants are the same on both sides
    from oracle import oracle as O
    oc = O.default_cfg()
    assert (cfg.tol, cfg.mu_floor, cfg.mu0, cfg.thr0, cfg.tau) == (oc.tol, oc.mu_floor, oc.mu0, oc.thr0, oc.tau) == (1e-8, 0.1, 10.0, 0.1, 0.995)
    # horizon indexing of the reference generator (nmpc_params.py:40-43): 21 states / 20 controls out of 101
    assert CP.long_list_size == 101 and list(range(101))[CP.xr_list_index] == list(range(0, 101, 5))


@pytest.mark.skipif(_has_gpu(), reason="only meaningful on a machine without a GPU")
def test_no_cpu_fallback():
    import ndp_nmpc_qd_amd as ndp
    with pytest.raises(ndp.NdpError, match="no usable HIP device"):
        ndp.BatchedNMPC(4)
    from ndp_nmpc_qd_amd.nmpc_ctl import NMPCBodyRateController
    with pytest.raises(ndp.NdpError):
        NMPCBodyRateController()


def test_refinement_setting_is_refused_where_the_kernels_have_no_such_path():
    """ipm_refine > 0 for a five-slot shape (N >= 28) or a precision study is an error of ndp_create (argument check: no
    device needed), not a setting that is silently ignored; the Python engine creates such shapes with ipm_refine = 0 itself."""
    import ndp_nmpc_qd_amd as ndp
    for kw in (dict(N=40, n_rti=2, ipm_refine=2), dict(N=28, ipm_refine=1), dict(N=20, qp_precision=3, ipm_refine=2)):
        with pytest.raises(ndp.NdpError, match="ipm_refine > 0 is not served for this shape"):
            ndp.BatchedNMPC(4, **kw)
    if not _has_gpu():                       # the same shapes with the engine's own choice pass the argument checks (and then find no device)
        for kw in (dict(N=40, n_rti=2), dict(N=20, qp_precision=3)):
            with pytest.raises(ndp.NdpError, match="no usable HIP device"):
                ndp.BatchedNMPC(4, **kw)
        with pytest.raises(ndp.NdpError, match="no usable HIP device"):
            ndp.BatchedNMPC(4, N=27, ipm_refine=2)         # three-slot: served


def test_weights_blob():
    w = _lib.load_weights()
    assert w.dtype == np.float32 and w.size == 17859
    assert abs(np.linalg.norm(w[:768].reshape(128, 6), 2) - 4.0) < 1e-3   # spectral norm of layer 1 (SN=4 model)


class _FakeEngine:
    """Records what the facade sends to the device engine."""
    def __init__(self, N=20):
        self.N, self.calls = N, []
        self.X, self.U = np.zeros((1, N + 1, 10)), np.zeros((1, N, 4))

    def get_iterate(self):
        return self.X.copy(), self.U.copy()

    def set_iterate(self, X, U):
        self.calls.append(("set_iterate",))
        self.X, self.U = X.copy(), U.copy()

    def update(self, x0, xr, ur, f=None, raise_on_status=True, full=False):
        self.calls.append(("update", x0.copy(), xr.copy(), ur.copy(), None if f is None else f.copy()))
        self.X = self.X + 1.0
        u0 = np.array([[1.0, 2.0, 3.0, 4.0]])
        return (u0, self.X.copy(), self.U.copy(), np.zeros(1, dtype=np.int32), np.zeros(1, dtype=np.int32)) if full else u0

    def status(self):
        return np.array([0], dtype=np.int32), np.array([0], dtype=np.int32)


def test_solver_facade_marshals_like_acados_template():
    """reset / update of the reference classes expressed through solver.set / solve_for_x0 (nmpc_body_rate_ctl.py:86-112)."""
    from ndp_nmpc_qd_amd.ndp_nmpc_ctl.ndp_nmpc_body_rate_ctl import NDPNMPCBodyRateController
    from ndp_nmpc_qd_amd.solver_facade import SolverFacade
    b = synth.make_batch(1, seed=1)
    xr, ur, x0 = b["xr"][0], b["ur"][0], b["x0"][0]
    f = np.random.default_rng(0).normal(size=(21, 3)).astype(np.float32)
    ctl = NDPNMPCBodyRateController.__new__(NDPNMPCBodyRateController)
    ctl._engine = _FakeEngine()
    ctl.solver = SolverFacade(ctl._engine, disturbance=True)
    ctl.reset(xr, ur)
    u0 = ctl.update(x0, xr, ur, f)
    assert np.array_equal(u0, [1.0, 2.0, 3.0, 4.0])
    kinds = [c[0] for c in ctl._engine.calls]
    assert kinds == ["set_iterate", "update"]                      # reset is pushed lazily, right before the solve; the solve is ONE engine call
    _, x0s, xrs, urs, fs = ctl._engine.calls[1]
    assert np.array_equal(x0s[0], x0) and np.array_equal(xrs[0], xr) and np.array_equal(urs[0], ur)
    # the force reaches the engine as float64 -- the reference's p is a float64 vector (ndp_nmpc_body_rate_ctl.py:97-99) -- holding
    # DownwashNN's float32 values exactly
    assert fs.dtype == np.float64 and np.array_equal(fs[0], f.astype(np.float64))
    np.testing.assert_array_equal(ctl.solver.get(3, "x"), xr[3] + 1.0)   # iterate mirrored back after the solve
    g = ctl.solver.get(3, "x")
    g[:] = 0
    assert ctl.solver.get(3, "x")[0] != 0                                  # get() returns a copy (nmpc_node.py:237-238)
    assert ctl.solver.N == 20 and ctl.solver.status == 0
    with pytest.raises(Exception):
        ctl.solver.set(0, "p", np.zeros(4))                                # NDP model has 7 parameters
    ctl.solver.set(2, "p", np.concatenate([xr[2, 6:10] + 1e-3, f[2]]))
    with pytest.raises(Exception, match="p\\[0:4\\] == yref\\[6:10\\]"):
        ctl.solver.solve_for_x0(x0)


def test_synthetic_references_are_consistent():
    b = synth.make_batch(5, seed=3, downwash=True)
    xr, ur = b["xr"], b["ur"]
    assert xr.shape == (5, 21, 10) and ur.shape == (5, 20, 4) and b["other"].shape == (5, 21, 10)
    np.testing.assert_allclose(np.linalg.norm(xr[..., 6:10], axis=-1), 1.0, atol=1e-12)
    assert (xr[..., 6] > 0).all()                                          # "ROS convention, w > 0" (pt_publisher.py:236)
    # collective acceleration = |a + g e3| (pt_publisher.py:195-201,145) and thrust direction matches the quaternion
    q, c = xr[:, :20, 6:10], ur[..., 3]
    qw, qx, qy, qz = (q[..., i] for i in range(4))
    _, _, acc, _ = synth.figure_eight(b["omega"][:, None], b["phi"][:, None], 0.1 * np.arange(20)[None, :])
    thrust = np.stack([2 * (qx * qz + qw * qy) * c, 2 * (qy * qz - qw * qx) * c, (1 - 2 * qx ** 2 - 2 * qy ** 2) * c - 9.81], -1)
    np.testing.assert_allclose(thrust, acc, atol=1e-12)
    xh, uh = synth.hover_reference(quirk_b1=True)
    assert uh[0, 3] == CP.mass * CP.gravity


# ------------------------------------------------------------------------------------------- multi-process (gloo)
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from ndp_nmpc_qd_amd import dist as ndist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B, N = 6, 20
        shard = ndist.make_formation_shard(B, rank, world, N=N)
        xr = torch.from_numpy(shard["xr"])
        other, gathered = ndist.exchange_neighbours(xr)
        ok = np.array_equal(other.numpy(), shard["other"])                  # what the exchange delivers is the neighbour's window
        ok = ok and np.array_equal(gathered[rank].numpy(), shard["xr"])
        other2, g2 = ndist.exchange_neighbours(xr, gathered)                # steady state: reuse the gather buffer
        ok = ok and g2.data_ptr() == gathered.data_ptr() and np.array_equal(other2.numpy(), shard["other"])
        # split form used by bench.py: the gather of the next tick is started before the current tick is solved
        nxt = ndist.make_formation_shard(B, rank, world, N=N, t0=0.02)
        buf2 = torch.empty_like(gathered)
        work = ndist.exchange_neighbours_begin(torch.from_numpy(nxt["xr"]), buf2)
        ok = ok and np.array_equal(other2.numpy(), shard["other"])          # the current tick's windows are untouched meanwhile
        other3 = ndist.exchange_neighbours_end(work, buf2)
        ok = ok and np.array_equal(other3.numpy(), nxt["other"]) and other3.data_ptr() == buf2[(rank + 1) % world].data_ptr()
        # the form bench.py uses: only the position / velocity columns travel; rank r reads the slice of rank (r+1) % W as it lies
        pv = torch.empty(B, N + 1, ndist.PV_COLS, dtype=torch.float64)
        gpv = torch.empty(world * B, N + 1, ndist.PV_COLS, dtype=torch.float64)
        ndist.exchange_pv_end(ndist.exchange_pv_begin(xr, pv, gpv))
        nb = gpv.view(world, B, N + 1, ndist.PV_COLS)[ndist.neighbour_rank(rank, world)]
        ok = ok and np.array_equal(nb.numpy(), shard["other"][:, :, :ndist.PV_COLS]) and nb.is_contiguous()
        # the gate statistic of the synthetic formation: a sensible fraction of neighbours inside r_horiz
        d2 = ((shard["other"][:, 0, 0:2] - shard["ego_xy"]) ** 2).sum(axis=1)
        q.put((rank, bool(ok), float((d2 < 1.0).mean())))
    finally:
        dist.destroy_process_group()


def test_neighbour_exchange_world_size_2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1] and all(r[1] for r in res)


def test_formation_shards_are_mutually_consistent():
    from ndp_nmpc_qd_amd import dist as ndist
    W, B = 4, 16
    shards = [ndist.make_formation_shard(B, r, W) for r in range(W)]
    for r in range(W):
        assert np.array_equal(shards[r]["other"], shards[ndist.neighbour_rank(r, W)]["xr"])
        assert np.array_equal(shards[r]["ego_xy"], shards[r]["x0"][:, 0:2])
    frac = np.mean([((s["other"][:, 0, 0:2] - s["ego_xy"]) ** 2).sum(axis=1) < 1.0 for s in shards])
    assert 0.1 < frac < 0.9


# ------------------------------------------------------------------------------------------- BASELINE config 4 (three-vehicle formations)
def test_config4_placements_are_consistent():
    """Every placement / world size hands each leader the window of vehicle 1 of ITS formation and no neighbour to the
    followers: what the all-gathered [W * B_local, N+1, 6] buffer (vehicle-major) or the local xr (formation-major) holds
    at other_index is the position/velocity part of the neighbour's reference window."""
    from ndp_nmpc_qd_amd import dist as ndist
    F = 40
    allv = ndist.make_config4_all(F)
    for W, placement, order in [(w, p, o) for w in (1, 2, 4, 8) for p in ("vehicle", "formation") for o in ("leaders_first", "interleaved")]:
        if True:
            shards = [ndist.make_config4_shard(r, W, F, placement, order=order) for r in range(W)]
            for s in shards:                                  # the local order: leaders in front of followers, or ascending ids
                lead = s["gids"] % 3 == 0
                if order == "leaders_first":
                    assert lead[:lead.sum()].all() and not lead[lead.sum():].any()
                else:
                    assert (np.diff(s["gids"]) > 0).all()
            assert sorted(np.concatenate([s["gids"] for s in shards]).tolist()) == list(range(3 * F))
            gathered = np.concatenate([s["xr"][:, :, :ndist.PV_COLS] for s in shards])
            for r, s in enumerate(shards):
                lead = s["gids"] % 3 == 0
                assert (s["other_index"][~lead] == -1).all() and (s["other_index"][lead] >= 0).all()
                buf = gathered if placement == "vehicle" else s["xr"][:, :, :ndist.PV_COLS]
                assert np.array_equal(buf[s["other_index"][lead]], allv["xr"][s["gids"][lead] + 1][:, :, :ndist.PV_COLS])
                if placement == "vehicle" and W > 1:      # the neighbour really lives on another rank
                    assert ((s["other_index"][lead] // (3 * F // W)) != r).all()
                if placement == "vehicle":                # peer windows: the neighbour is a row of rank (r + 1) % W's own buffer
                    prow = ndist.config4_other_index(r, W, F, "vehicle", peer_rows=True, order=order)
                    assert (prow[~lead] == -1).all()
                    nxt = shards[ndist.neighbour_rank(r, W)]
                    assert np.array_equal(nxt["xr"][prow[lead]], allv["xr"][s["gids"][lead] + 1])
    with pytest.raises(ValueError):
        ndist.config4_gids(0, 7, F, "vehicle")
    d2 = ((allv["xr"][1::3, 0, 0:2] - allv["ego_xy"][0::3]) ** 2).sum(axis=1)
    assert 0.25 < (d2 < 1.0).mean() < 0.55                # a sensible share of the leaders inside the r_horiz gate


def _worker_cfg4(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from ndp_nmpc_qd_amd import dist as ndist
    from oracle import oracle as O
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        F, N = 8, 20
        sh = ndist.make_config4_shard(rank, world, F, "vehicle", N=N)
        B = sh["xr"].shape[0]
        xr = torch.from_numpy(sh["xr"])
        pv = torch.empty(B, N + 1, ndist.PV_COLS, dtype=torch.float64)
        gathered = torch.empty(world * B, N + 1, ndist.PV_COLS, dtype=torch.float64)
        work = ndist.exchange_pv_begin(xr, pv, gathered)          # ONE all-gather of the position/velocity columns
        ndist.exchange_pv_end(work)
        g = gathered.numpy()
        idx = sh["other_index"]
        allv = ndist.make_config4_all(F, N=N)
        lead = sh["gids"] % 3 == 0
        ok = np.array_equal(g[idx[lead]], allv["xr"][sh["gids"][lead] + 1][:, :, :ndist.PV_COLS])
        ok = ok and (idx[~lead] == -1).all()
        # the step each rank would run on its shard, on the CPU oracle: leaders = NDP controller reading the gathered window
        # of vehicle 1, followers = the same model with no force.  The result must equal the single-process solution of the
        # whole formation set (parity semantics of SURVEY 8e), bit for bit: the exchange delivers exact copies.
        blob = np.fromfile(os.path.join(ROOT, "ndp_nmpc_qd_amd", "weights", "downwash_sn4.bin"), dtype="<f4")
        other = np.zeros((B, N + 1, 10))
        other[lead, :, :ndist.PV_COLS] = g[idx[lead]]
        ego = sh["ego_xy"].copy()
        ego[~lead] = 1e9
        f = O.downwash_batch(blob, other, sh["xr"], ego)
        cfg = O.default_cfg(N=N, use_fd=True)
        X, U = sh["xr"].copy(), sh["ur"].copy()
        u_loc, st, _ = O.step_batch(cfg, sh["x0"], sh["xr"], sh["ur"], f, X, U, nthreads=1)
        oth_all = allv["xr"][np.where(np.arange(3 * F) % 3 == 0, np.arange(3 * F) + 1, np.arange(3 * F))]
        ego_all = allv["ego_xy"].copy()
        ego_all[np.arange(3 * F) % 3 != 0] = 1e9
        f_all = O.downwash_batch(blob, oth_all, allv["xr"], ego_all)
        Xa, Ua = allv["xr"].copy(), allv["ur"].copy()
        u_all, *_ = O.step_batch(cfg, allv["x0"], allv["xr"], allv["ur"], f_all, Xa, Ua, nthreads=1)
        ok = ok and np.array_equal(u_loc, u_all[sh["gids"]]) and (st == 0).all()
        ok = ok and np.all(f[~lead] == 0) and (not lead.any() or (np.abs(f[lead]).max(axis=(1, 2)) > 0).any())
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_config4_exchange_world_size_3():
    """Three ranks, vehicle-major placement of 8 three-vehicle formations: every formation's vehicles sit on three different
    ranks; one all-gather of the [B_local, N+1, 6] columns, then each rank's NDP / NMPC steps (CPU oracle) reproduce the
    single-process solution."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_cfg4, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1, 2] and all(r[1] for r in res)
