"""Minimum-derivative piecewise polynomials through waypoints, in normalised segment time.

Mirror of the reference's PolymOptimizer (ndp_nmpc/scripts/pt_pub/polym_optimizer.py:22-139): same class name,
constructor argument, `get_coeff(wpt_seq)` -> column vector of (N+1) coefficients per segment, ascending powers, and
`get_poly_params(deriv_num, t)` -> row vector.  This runs on the host once per trajectory (the reference does the same);
the per-tick evaluation is the device kernel behind BatchedNMPCRefPublisher.
"""
from enum import Enum, unique

import numpy as np


@unique
class MinMethod(Enum):
    SNAP = "snap"
    JERK = "jerk"
    ACCEL = "acceleration"
    VEL = "velocity"


_ORDER = {MinMethod.SNAP: 4, MinMethod.JERK: 3, MinMethod.ACCEL: 2, MinMethod.VEL: 1}


class PolymOptimizer:
    def __init__(self, method: MinMethod) -> None:
        if method not in _ORDER:
            raise ValueError("non-existent trajectory generation method")   # the reference only prints (polym_optimizer.py:34)
        self.ord_deriv = _ORDER[method]           # Nd
        self.ord_polym = 2 * self.ord_deriv - 1   # N
        self.num_wpt = 0                          # M (segments)

    def get_poly_params(self, deriv_num: int, t: float) -> np.ndarray:
        """d^k/dt^k of [1, t, ..., t^N] as a 1 x (N+1) row (polym_optimizer.py:104-139)."""
        i = np.arange(self.ord_polym + 1, dtype=np.float64)
        fall = np.ones_like(i)
        for j in range(deriv_num):
            fall *= np.maximum(i - j, 0.0)
        return (fall * np.power(float(t), np.maximum(i - deriv_num, 0.0)))[None, :]

    def constraint_matrix(self, m: int) -> np.ndarray:
        """The (N+1)M x (N+1)M system of get_coeff; it depends on the segment count only (time is normalised)."""
        n1 = self.ord_polym + 1
        a = np.zeros((m * n1, m * n1))
        at0 = [self.get_poly_params(k, 0.0)[0] for k in range(n1)]
        at1 = [self.get_poly_params(k, 1.0)[0] for k in range(n1)]
        row = 0
        for i in range(m):                        # p_i(0) = w_i                        polym_optimizer.py:55-63
            a[row, i * n1:(i + 1) * n1] = at0[0]; row += 1
        for i in range(m):                        # p_i(1) = w_{i+1}                    :65-72
            a[row, i * n1:(i + 1) * n1] = at1[0]; row += 1
        for k in range(1, self.ord_deriv):        # rest at the start                   :74-80
            a[row, 0:n1] = at0[k]; row += 1
        for k in range(1, self.ord_deriv):        # rest at the end                     :82-88
            a[row, (m - 1) * n1:m * n1] = at1[k]; row += 1
        for i in range(m - 1):                    # derivatives 1..N-1 continuous       :92-100
            for k in range(1, self.ord_polym):
                a[row, i * n1:(i + 1) * n1] = at1[k]
                a[row, (i + 1) * n1:(i + 2) * n1] = -at0[k]
                row += 1
        return a

    def rhs(self, wpt_seq) -> np.ndarray:
        w = np.asarray(wpt_seq, dtype=np.float64).reshape(-1)
        m = len(w) - 1
        b = np.zeros((m * (self.ord_polym + 1), 1))
        b[0:m, 0] = w[:-1]
        b[m:2 * m, 0] = w[1:]
        return b

    def get_coeff(self, wpt_seq) -> np.ndarray:
        """wpt_seq: 1-D waypoints (M+1) -> coefficients [(N+1)M, 1] (polym_optimizer.py:38-102)."""
        self.num_wpt = len(wpt_seq) - 1
        return np.linalg.solve(self.constraint_matrix(self.num_wpt), self.rhs(wpt_seq))
