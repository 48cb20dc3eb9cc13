"""ctypes binding of tests/emu/libndp_emu.so: the product's wave program on a host wave emulator (tests only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libndp_emu.so")


class NdpCfg(C.Structure):
    """Mirror of include/ndp_nmpc.h ndp_cfg."""
    _fields_ = [
        ("batch", C.c_int32), ("N", C.c_int32), ("n_rti", C.c_int32), ("use_fd", C.c_int32),
        ("qp_mode", C.c_int32), ("iter_max", C.c_int32), ("device", C.c_int32), ("qp_precision", C.c_int32),
        ("dt", C.c_double), ("mass", C.c_double), ("gravity", C.c_double), ("r_horiz", C.c_double),
        ("Qd", C.c_double * 10), ("Rd", C.c_double * 4),
        ("lbu", C.c_double * 4), ("ubu", C.c_double * 4), ("lbv", C.c_double * 3), ("ubv", C.c_double * 3),
        ("mu0", C.c_double), ("thr0", C.c_double), ("tol", C.c_double), ("tau", C.c_double), ("auto_margin", C.c_double),
    ]


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


def default_cfg(N=20, n_rti=1, use_fd=False, qp_mode=0):
    c = NdpCfg()
    lib().emu_default_cfg(C.byref(c))
    c.N, c.n_rti, c.use_fd, c.qp_mode = N, n_rti, int(use_fd), qp_mode
    return c


def lds_layout(N):
    out = (C.c_int * 8)()
    lib().emu_lds_layout(int(N), out)
    return dict(zip(("XI", "MB", "CB", "MB_STRIDE", "CB_STRIDE", "total", "stamps"), list(out)[:7]))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def rti_step(cfg, x0, xr, ur, f, X, U, dump=False):
    x0, xr, ur = (np.ascontiguousarray(a, dtype=np.float64) for a in (x0, xr, ur))
    f = None if f is None else np.ascontiguousarray(f, dtype=np.float32)
    assert X.dtype == np.float64 and U.dtype == np.float64 and X.flags.c_contiguous and U.flags.c_contiguous
    u0 = np.zeros(4)
    status, iters = C.c_int(-1), C.c_int(-1)
    counters = (C.c_long * 4)()
    n = lib().emu_lds_doubles(cfg.N)
    lds = np.zeros(n) if dump else None
    rc = lib().emu_rti_step(C.byref(cfg), _p(x0), _p(xr), _p(ur), _p(f), _p(X), _p(U), _p(u0),
                            C.byref(status), C.byref(iters), _p(lds), counters)
    assert rc == 0
    return u0, status.value, iters.value, lds, dict(mfma=counters[0], lds_ld=counters[1], lds_st=counters[2],
                                                   readlane=counters[3])
