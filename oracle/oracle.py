"""ctypes binding of oracle/libndp_oracle.so (the CPU fp64 restatement).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from the ndp_nmpc_qd_amd package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libndp_oracle.so")

NX, NU = 10, 4
MLP_NPARAM = 17859


class OrcCfg(C.Structure):
    _fields_ = [
        ("N", C.c_int), ("n_rti", C.c_int), ("dt", C.c_double), ("mass", C.c_double), ("g", C.c_double),
        ("Qd", C.c_double * 10), ("Rd", C.c_double * 4),
        ("lbu", C.c_double * 4), ("ubu", C.c_double * 4), ("lbv", C.c_double * 3), ("ubv", C.c_double * 3),
        ("use_fd", C.c_int),
        ("mu0", C.c_double), ("thr0", C.c_double), ("tol", C.c_double), ("tau", C.c_double),
        ("iter_max", C.c_int), ("qp_mode", C.c_int), ("auto_margin", C.c_double), ("mu_floor", C.c_double),
        ("refine", C.c_int), ("refine_gamma", C.c_double),
        ("as_iter_max", C.c_int), ("as_gamma", C.c_double),
    ]


class OrcStats(C.Structure):
    _fields_ = [("status", C.c_int), ("ipm_iters", C.c_int), ("n_active", C.c_int), ("mu", C.c_double), ("as_sweeps", C.c_int)]


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "ndp_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    """Loads the prebuilt oracle.  It never builds implicitly: a build spawns make -> sh -> cc, and a process that has
    initialised the GPU (bench.py under rocprofv3 in particular) must not start such a chain on the GPU pool.  Build with
    `make -C oracle`, oracle.build() (tests/conftest.py does) or __graft_entry__.build()."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise RuntimeError("oracle/libndp_oracle.so is missing: run `make -C oracle` (or __graft_entry__.build()) first")
        _lib = C.CDLL(_SO)
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def default_cfg(N=20, n_rti=1, use_fd=False, T=None):
    c = OrcCfg()
    lib().orc_default_cfg(C.byref(c))
    c.N = N
    c.n_rti = n_rti
    c.use_fd = int(use_fd)
    # reference: T_horizon=2, N_node=20 -> 0.1 s; longer horizons keep the 0.1 s interval (SURVEY 8d cfg 5)
    c.dt = (T / N) if T is not None else 0.1
    return c


VAR_TERMINAL_TIMES_DT, VAR_BOUNDS_STAGE_N, VAR_ERK_2_STEPS, VAR_NO_DT_SCALING = 1, 2, 4, 8


def set_variant(bits):
    """[acados-knowledge] assumptions as switches (ndp_oracle.h: ORC_VAR_*), for scripts/acados_sensitivity.py only; 0 = off."""
    lib().orc_set_variant(int(bits))


def dynamics(cfg, x, u, fd=None):
    x, u, fd = _f64(x), _f64(u), _f64(fd)
    out = np.zeros(NX)
    lib().orc_dynamics(C.byref(cfg), _p(x), _p(u), _p(fd), _p(out))
    return out


def jacobians(x, u):
    x, u = _f64(x), _f64(u)
    A, B = np.zeros((NX, NX)), np.zeros((NX, NU))
    lib().orc_jacobians(_p(x), _p(u), _p(A), _p(B))
    return A, B


def rk4_sens(cfg, x, u, fd=None):
    x, u, fd = _f64(x), _f64(u), _f64(fd)
    xn, A, B = np.zeros(NX), np.zeros((NX, NX)), np.zeros((NX, NU))
    lib().orc_rk4_sens(C.byref(cfg), _p(x), _p(u), _p(fd), _p(xn), _p(A), _p(B))
    return xn, A, B


def linearize(cfg, x0, xr, ur, f, X, U):
    N = cfg.N
    a = [_f64(v) for v in (x0, xr, ur, f, X, U)]
    out = dict(A=np.zeros((N, NX, NX)), B=np.zeros((N, NX, NU)), b=np.zeros((N, NX)),
               Q=np.zeros((N + 1, NX, NX)), q=np.zeros((N + 1, NX)), Rd=np.zeros((N, NU)), r=np.zeros((N, NU)),
               dx0=np.zeros(NX), lu=np.zeros((N, NU)), uu=np.zeros((N, NU)), lv=np.zeros((N + 1, 3)),
               uv=np.zeros((N + 1, 3)))
    lib().orc_linearize(C.byref(cfg), *[_p(v) for v in a],
                        *[_p(out[k]) for k in ("A", "B", "b", "Q", "q", "Rd", "r", "dx0", "lu", "uu", "lv", "uv")])
    return out


def qp_solve(cfg, qp):
    N = qp["A"].shape[0]
    dx, du = np.zeros((N + 1, NX)), np.zeros((N, NU))
    st = OrcStats()
    lib().orc_qp_solve(C.byref(cfg), C.c_int(N),
                       *[_p(_f64(qp[k])) for k in ("A", "B", "b", "Q", "q", "Rd", "r", "dx0", "lu", "uu", "lv", "uv")],
                       _p(dx), _p(du), C.byref(st))
    return dx, du, st


def qp_solve_as(cfg, qp, act=None):
    """qp_solve with the active set of the input bounds handed in and out (act: int8[N,4] or None); qp_mode 0."""
    N = qp["A"].shape[0]
    dx, du = np.zeros((N + 1, NX)), np.zeros((N, NU))
    st = OrcStats()
    assert act is None or (act.dtype == np.int8 and act.flags.c_contiguous and act.size == N * NU)
    lib().orc_qp_solve_as(C.byref(cfg), C.c_int(N),
                          *[_p(_f64(qp[k])) for k in ("A", "B", "b", "Q", "q", "Rd", "r", "dx0", "lu", "uu", "lv", "uv")],
                          _p(dx), _p(du), C.byref(st), _p(act))
    return dx, du, st


def qp_riccati(qp):
    N = qp["A"].shape[0]
    dx, du = np.zeros((N + 1, NX)), np.zeros((N, NU))
    lib().orc_qp_riccati(C.c_int(N), *[_p(_f64(qp[k])) for k in ("A", "B", "b", "Q", "q", "Rd", "r", "dx0")],
                         _p(dx), _p(du))
    return dx, du


def step(cfg, x0, xr, ur, f, X, U):
    """One update(): returns u0, stats; X, U (float64, C-contiguous) updated in place."""
    assert X.dtype == np.float64 and U.dtype == np.float64 and X.flags.c_contiguous and U.flags.c_contiguous
    x0, xr, ur, f = _f64(x0), _f64(xr), _f64(ur), _f64(f)
    u0 = np.zeros(NU)
    st = OrcStats()
    lib().orc_step(C.byref(cfg), _p(x0), _p(xr), _p(ur), _p(f), _p(X), _p(U), _p(u0), C.byref(st))
    return u0, st


def step_batch(cfg, x0, xr, ur, f, X, U, nthreads=0):
    B = x0.shape[0]
    assert X.dtype == np.float64 and U.dtype == np.float64 and X.flags.c_contiguous and U.flags.c_contiguous
    x0, xr, ur, f = _f64(x0), _f64(xr), _f64(ur), _f64(f)
    u0 = np.zeros((B, NU))
    status = np.zeros(B, dtype=np.int32)
    iters = np.zeros(B, dtype=np.int32)
    lib().orc_step_batch(C.byref(cfg), C.c_int(B), _p(x0), _p(xr), _p(ur), _p(f), _p(X), _p(U), _p(u0),
                         _p(status), _p(iters), C.c_int(nthreads))
    return u0, status, iters


def step_batch_as(cfg, x0, xr, ur, f, X, U, act, nthreads=0):
    """step_batch with the instances' kept active sets (act: int8[B,N,4], updated in place; qp_mode 0): returns u0, status,
    ipm_iters, sweeps."""
    B = x0.shape[0]
    assert X.dtype == np.float64 and U.dtype == np.float64 and X.flags.c_contiguous and U.flags.c_contiguous
    assert act is None or (act.dtype == np.int8 and act.flags.c_contiguous and act.shape == (B, cfg.N, NU))
    x0, xr, ur, f = _f64(x0), _f64(xr), _f64(ur), _f64(f)
    u0 = np.zeros((B, NU))
    status = np.zeros(B, dtype=np.int32)
    iters = np.zeros(B, dtype=np.int32)
    sweeps = np.zeros(B, dtype=np.int32)
    lib().orc_step_batch_as(C.byref(cfg), C.c_int(B), _p(x0), _p(xr), _p(ur), _p(f), _p(X), _p(U), _p(u0),
                            _p(status), _p(iters), C.c_int(nthreads), _p(act), _p(sweeps))
    return u0, status, iters, sweeps


def num_threads():
    return lib().orc_num_threads()


def mlp_forward(blob, x):
    blob = np.ascontiguousarray(blob, dtype=np.float32)
    assert blob.size == MLP_NPARAM
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 6)
    out = np.zeros((x.shape[0], 3), dtype=np.float32)
    lib().orc_mlp_forward(_p(blob), C.c_int(x.shape[0]), _p(x), _p(out))
    return out


def downwash_batch(blob, other, ego_ref, ego_xy, r_horiz=1.0, nthreads=0):
    blob = np.ascontiguousarray(blob, dtype=np.float32)
    other, ego_ref, ego_xy = _f64(other), _f64(ego_ref), _f64(ego_xy)
    B, Np1 = other.shape[0], other.shape[1]
    out = np.zeros((B, Np1, 3), dtype=np.float32)
    lib().orc_downwash_batch(_p(blob), C.c_int(B), C.c_int(Np1 - 1), C.c_double(r_horiz), _p(other), _p(ego_ref),
                             _p(ego_xy), _p(out), C.c_int(nthreads))
    return out


class OrcThrCfg(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("ts", "tau", "mass", "g", "R", "Q0", "Q1", "k_init")]


def thr_default_cfg():
    c = OrcThrCfg()
    lib().orc_thr_default_cfg(C.byref(c))
    return c


def thr_reset(cfg, V):
    st = np.zeros((V, 8))
    lib().orc_thr_reset(C.byref(cfg), C.c_int(V), _p(st))
    return st


def thr_update(cfg, state, vz, throttle):
    vz, throttle = _f64(vz), _f64(throttle)
    k = np.zeros(state.shape[0])
    lib().orc_thr_update(C.byref(cfg), C.c_int(state.shape[0]), _p(state), _p(vz), _p(throttle), _p(k))
    return k


def att_thrust(cfg, cacc, k):
    cacc, k = _f64(cacc), _f64(k)
    out = np.zeros_like(k)
    lib().orc_att_thrust(C.byref(cfg), C.c_int(k.size), _p(cacc), _p(k), _p(out))
    return out


def relay_formation(state, form, alpha=0.8):
    form = _f64(form)
    lib().orc_relay_formation(C.c_double(alpha), C.c_int(state.shape[0]), _p(state), _p(form))
    return state[:, 0:3].copy()


def relay_reference(state, xr_lead):
    xr_lead = _f64(xr_lead)
    out = np.zeros_like(xr_lead)
    lib().orc_relay_reference(C.c_int(xr_lead.shape[0]), C.c_int(xr_lead.shape[1] - 1), _p(state), _p(xr_lead), _p(out))
    return out


def plant_step(cfg, x, u, f, dt, sub=4):
    assert x.dtype == np.float64 and x.flags.c_contiguous
    u, f = _f64(u), _f64(f)
    lib().orc_plant_step(C.byref(cfg), C.c_int(x.shape[0]), _p(x), _p(u), _p(f), C.c_double(dt), C.c_int(sub))
    return x


# ---- f1: reference window generation
def traj_point(coeff, t_cum, t_seg, final_pt, t):
    """coeff[n_seg,28] of ONE vehicle -> (pvaj[12], yaw[2]) at trajectory time t."""
    coeff, t_cum, t_seg, final_pt = _f64(coeff), _f64(t_cum), _f64(t_seg), _f64(final_pt)
    pvaj, yaw = np.zeros(12), np.zeros(2)
    lib().orc_traj_point(C.c_int(coeff.shape[0]), _p(coeff), _p(t_cum), _p(t_seg), _p(final_pt), C.c_double(t),
                         _p(pvaj), _p(yaw))
    return pvaj, yaw


def diff_flatness(pvaj, yaw, mass=1.4844, g=9.81):
    pvaj, yaw = _f64(pvaj), _f64(yaw)
    x, u = np.zeros(10), np.zeros(4)
    lib().orc_diff_flatness(C.c_double(mass), C.c_double(g), _p(pvaj), _p(yaw), _p(x), _p(u))
    return x, u


def ref_window(coeff, t_cum, t_seg, final_pt, t, N=20, dt=0.1, mass=1.4844, g=9.81):
    """coeff[V,n_seg,28], t_cum[V,n_seg+1], t_seg[V,n_seg], final_pt[V,3], t[V] -> xr[V,N+1,10], ur[V,N,4]."""
    coeff, t_cum, t_seg, final_pt, t = _f64(coeff), _f64(t_cum), _f64(t_seg), _f64(final_pt), _f64(t)
    V, n_seg = coeff.shape[0], coeff.shape[1]
    xr, ur = np.zeros((V, N + 1, 10)), np.zeros((V, N, 4))
    lib().orc_ref_window(C.c_int(V), C.c_int(N), C.c_double(dt), C.c_double(mass), C.c_double(g), C.c_int(n_seg),
                         _p(coeff), _p(t_cum), _p(t_seg), _p(final_pt), _p(t), _p(xr), _p(ur))
    return xr, ur
