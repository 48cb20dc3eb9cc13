"""`.solver` facade: the subset of acados_template.AcadosOcpSolver the reference's callers touch.

nmpc_node.py reaches into `nmpc_ctl.solver.N` (:119,125,235) and `nmpc_ctl.solver.get(i, "x")` (:237);
the controllers use `solver.set(i, "x"|"u"|"yref"|"p", v)`, `solver.solve_for_x0(x0)` and `solver.status`
(nmpc_body_rate_ctl.py:88-110).  The facade keeps a host mirror of one instance's iterate and stages
yref / p exactly as the reference passes them, then runs the step on the device.
"""
import threading

import numpy as np


class SolverFacade:
    def __init__(self, engine, disturbance):
        self._eng = engine            # BatchedNMPC with batch = 1
        self.N = engine.N
        self.status = 0
        self._np = 7 if disturbance else 4
        self._lock = threading.Lock()  # rospy calls update / reset / get from different threads (nmpc_node.py:94,152,237)
        self._yref = np.zeros((self.N + 1, 14))
        self._p = np.zeros((self.N + 1, self._np))
        self._dirty_iter = False
        self._X, self._U = engine.get_iterate()
        self._X, self._U = self._X[0], self._U[0]

    # acados: solver.set(stage, field, value)
    def set(self, stage, field, value):
        value = np.asarray(value, dtype=np.float64).ravel()
        with self._lock:
            if field == "x":
                self._X[stage, :] = value
                self._dirty_iter = True
            elif field == "u":
                self._U[stage, :] = value
                self._dirty_iter = True
            elif field == "yref":
                self._yref[stage, :value.size] = value
            elif field == "p":
                if value.size != self._np:
                    raise Exception(f"set: p has dimension {self._np}, got {value.size}")
                self._p[stage, :] = value
            else:
                raise Exception(f"AcadosOcpSolver.set(): {field} is not supported by this drop-in")

    def set_reference(self, xr, ur, f=None):
        """What the reference's update() does with 2 (N + 1) solver.set calls (nmpc_body_rate_ctl.py:95-104,
        ndp_nmpc_body_rate_ctl.py:93-104), as three array assignments: yref_k = [xr_k, ur_k] (terminal: xr_N), p_k = [xr_k[6:10]
        (, f_k)].  The per-stage solver.set stays available to callers that use it."""
        xr = np.asarray(xr, dtype=np.float64)
        ur = np.asarray(ur, dtype=np.float64)
        N = self.N
        with self._lock:
            self._yref[:, 0:10] = xr[:N + 1]
            self._yref[:N, 10:14] = ur[:N]
            self._p[:, 0:4] = xr[:N + 1, 6:10]
            if self._np == 7:
                self._p[:, 4:7] = np.asarray(f, dtype=np.float64)[:N + 1]

    # acados: solver.get(stage, field) -> fresh array the caller may mutate (nmpc_node.py:237-238)
    def get(self, stage, field):
        with self._lock:
            if field == "x":
                return self._X[stage].copy()
            if field == "u":
                return self._U[stage].copy()
        raise Exception(f"AcadosOcpSolver.get(): {field} is not supported by this drop-in")

    # acados: u0 = solver.solve_for_x0(x0)
    def solve_for_x0(self, x0):
        with self._lock:
            N = self.N
            if self._dirty_iter:
                self._eng.set_iterate(self._X[None], self._U[None])
                self._dirty_iter = False
            xr = self._yref[:, 0:10].copy()
            ur = self._yref[:N, 10:14].copy()
            # The reference always sets p_k[0:4] = xr_k[6:10] (nmpc_body_rate_ctl.py:99-104); the kernel
            # reads the reference quaternion from xr, so a differing p is not representable.
            if not np.array_equal(self._p[:, 0:4], xr[:, 6:10]):
                raise Exception("this drop-in requires p[0:4] == yref[6:10] (as nmpc_body_rate_ctl.py:99-104 sets it)")
            # the force stays float64, as the reference's p is (ndp_nmpc_body_rate_ctl.py:97-99): ndp_step_ex_f64
            f = np.ascontiguousarray(self._p[:, 4:7])[None] if self._np == 7 else None
            # one C-ABI call: inputs in, u0 + new iterate + status out (ndp_step_ex); nothing else touches the device per tick
            u0, X, U, st, _ = self._eng.update(np.asarray(x0, dtype=np.float64)[None], xr[None], ur[None], f=f,
                                               raise_on_status=False, full=True)
            self.status = int(st[0])
            self._X, self._U = X[0], U[0]
            return u0[0]
