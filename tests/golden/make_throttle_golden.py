"""Generates the hover-throttle-estimator fixture by IMPORTING the reference
(/root/reference/ndp_nmpc/scripts/hv_throttle_est, pure numpy).  Run in the build container only.

Output (committed): tests/golden/throttle_golden.npz
  vz[T,V], throttle[T,V]  inputs of HoverThrottleEstimator.update (hover_throttle_estimator.py:37-53), one
                          estimator object per vehicle V, called T times
  k[T,V], x[T,V,2], P[T,V,2,2]   its three return values after every call
  c[T,V], thrust[T,V]     nmpc_u_2_att_tgt (nmpc_node.py:273-283): thrust = c*mass/k_throttle if k_throttle != 0 else 0
"""
import os
import sys

sys.dont_write_bytecode = True    # importing from /root/reference must not leave __pycache__ there (the tree is read-only by contract)

import numpy as np

S = "/root/reference/ndp_nmpc/scripts"
sys.path.insert(0, S)
from hv_throttle_est import HoverThrottleEstimator  # noqa: E402
from params import estimator_params as EP  # noqa: E402
from params import nmpc_params as CP  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.Generator(np.random.PCG64(20231213 + 9))
T, V = 260, 8
t = np.arange(T)[:, None] * EP.ts_est
# climbing / descending vehicles with different true throttle gains, plus phases where the gate (0.1 < throttle < 1) is closed
k_true = rng.uniform(30, 70, V)
c = 9.81 + 2.0 * np.sin(2 * np.pi * rng.uniform(0.2, 1.0, V) * t + rng.uniform(0, 6.28, V)) + rng.normal(0, 0.05, (T, V))
throttle = c * CP.mass / k_true
throttle[50:60, 3] = 0.05          # gate closed (too low)
throttle[100:105, 5] = 1.2         # gate closed (too high)
throttle[200, 7] = 0.1             # exactly on the rim -> closed (strict <)
throttle[201, 7] = 1.0             # exactly on the rim -> closed
az = c - 9.81 + rng.normal(0, 0.1, (T, V))
vz = np.cumsum(az, axis=0) * EP.ts_est

est = [HoverThrottleEstimator(EP.ts_est) for _ in range(V)]
k = np.zeros((T, V)); x = np.zeros((T, V, 2)); P = np.zeros((T, V, 2, 2)); thrust = np.zeros((T, V))
for i in range(T):
    for v in range(V):
        kk, xx, PP = est[v].update(float(vz[i, v]), float(throttle[i, v]))
        k[i, v], x[i, v], P[i, v] = kk, xx[:, 0], PP
        thrust[i, v] = c[i, v] * CP.mass / kk if kk != 0 else 0      # nmpc_node.py:281
np.savez_compressed(os.path.join(HERE, "throttle_golden.npz"), vz=vz, throttle=throttle, k=k, x=x, P=P, c=c, thrust=thrust,
                    k_init=EP.k_throttle_init, ts=EP.ts_est, R=EP.R, Q=np.asarray(EP.Q), mass=EP.mass, gravity=EP.gravity)
print("k range", k.min(), k.max(), "final k vs true", np.c_[k[-1], k_true][:4])
