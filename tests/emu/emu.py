"""ctypes binding of tests/emu/libndp_emu.so: the product's wave program on a host wave emulator (tests only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libndp_emu.so")


from ndp_nmpc_qd_amd._lib import NdpCfg  # noqa: E402  (the one mirror of include/ndp_nmpc.h's ndp_cfg)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
    return _lib


def default_cfg(N=20, n_rti=1, use_fd=False, qp_mode=0, as_iter_max=None):
    """as_iter_max: None = the product's default (active-set iterations on); 0 = rounds 1-5's QP_AUTO (early exit or interior point)."""
    c = NdpCfg()
    lib().emu_default_cfg(C.byref(c))
    c.N, c.n_rti, c.use_fd, c.qp_mode = N, n_rti, int(use_fd), qp_mode
    if as_iter_max is not None:
        c.as_iter_max = int(as_iter_max)
    return c


def lds_layout(N):
    out = (C.c_int * 8)()
    lib().emu_lds_layout(int(N), out)
    return dict(zip(("XI", "MB", "CB", "MB_STRIDE", "CB_STRIDE", "total", "stamps"), list(out)[:7]))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def act_record(N):
    """An empty kept-active-set record of one instance as the tests carry it: an int32 (sweeps of the last step, filled in by rti_step
    from the step's iteration word) followed by RtiIo::act -- one signed byte per input bound."""
    return np.zeros(4 + lib().emu_act_pitch(int(N)), dtype=np.int8)


def rti_step(cfg, x0, xr, ur, f, X, U, dump=False, act=None):
    x0, xr, ur = (np.ascontiguousarray(a, dtype=np.float64) for a in (x0, xr, ur))
    f = None if f is None else np.ascontiguousarray(f, dtype=np.float32)
    assert X.dtype == np.float64 and U.dtype == np.float64 and X.flags.c_contiguous and U.flags.c_contiguous
    u0 = np.zeros(4)
    status, iters = C.c_int(-1), C.c_int(-1)
    counters = (C.c_long * 5)()
    n = lib().emu_lds_doubles(cfg.N)
    lds = np.zeros(n) if dump else None
    assert act is None or (act.dtype == np.int8 and act.flags.c_contiguous)
    rc = lib().emu_rti_step_act(C.byref(cfg), _p(x0), _p(xr), _p(ur), _p(f), _p(X), _p(U), _p(u0),
                                C.byref(status), C.byref(iters), _p(lds), counters, _p(None if act is None else act[4:]))
    assert rc == 0
    if act is not None:
        act[:4].view(np.int32)[0] = iters.value >> 16          # rti_wave.hpp: ITERS_SWEEP_SHIFT
    iters.value &= 0xffff
    return u0, status.value, iters.value, lds, dict(mfma=counters[0], lds_ld=counters[1], lds_st=counters[2],
                                                   readlane=counters[3], mfma4=counters[4])


def act_view(act):
    """(sweeps, set[N,4]) of a kept-active-set record."""
    return int(act[:4].view(np.int32)[0]), act[4:].reshape(-1, 4)


def rti_step_defer(cfg, x0, xr, ur, f, X, U, act=None):
    """The work list's producer view (run<DEFER = true>): returns (deferred, u0, status, iters); u0 / status / iters / X / U
    are meaningful only when deferred is False."""
    x0, xr, ur = (np.ascontiguousarray(a, dtype=np.float64) for a in (x0, xr, ur))
    f = None if f is None else np.ascontiguousarray(f, dtype=np.float32)
    u0 = np.full(4, np.nan)
    status, iters = C.c_int(-7), C.c_int(-7)
    rc = lib().emu_rti_step_defer_act(C.byref(cfg), _p(x0), _p(xr), _p(ur), _p(f), _p(X), _p(U), _p(u0), C.byref(status), C.byref(iters),
                                      _p(None if act is None else act[4:]))
    assert rc in (0, 1), rc
    return bool(rc), u0, status.value, (iters.value & 0xffff) if iters.value >= 0 else iters.value


def rti_step_late(cfg, x0, xr, ur, f, X, U, ready=True):
    """The late-force path (RtiIo::f_late): returns (u0, status, iters, missed)."""
    x0, xr, ur = (np.ascontiguousarray(a, dtype=np.float64) for a in (x0, xr, ur))
    f = np.ascontiguousarray(f, dtype=np.float32)
    u0 = np.zeros(4)
    status, iters, missed = C.c_int(-1), C.c_int(-1), C.c_int(0)
    rc = lib().emu_rti_step_late(C.byref(cfg), _p(x0), _p(xr), _p(ur), _p(f), int(bool(ready)), _p(X), _p(U), _p(u0),
                                 C.byref(status), C.byref(iters), C.byref(missed))
    assert rc == 0
    return u0, status.value, iters.value & 0xffff, missed.value
