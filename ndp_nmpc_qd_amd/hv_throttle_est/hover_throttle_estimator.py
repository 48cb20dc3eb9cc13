"""HoverThrottleEstimator on MI355X -- same constructor / update as the reference class
(ndp_nmpc/scripts/hv_throttle_est/hover_throttle_estimator.py:15-53); one vehicle per object.
For many vehicles use BatchedNMPC.throttle_update (one kernel launch for the whole batch)."""
import numpy as np

from ..batched import BatchedNMPC
from ..params import nmpc_params as CP


class HoverThrottleEstimator:
    def __init__(self, ts: float, device=0) -> None:
        if abs(ts - 0.02) > 1e-12:
            raise ValueError("the device estimator is built for ts_est = 0.02 s (params/estimator_params.py:15)")
        self._engine = BatchedNMPC(batch=1, N=CP.N_node, device=device, load_mlp=False)
        self.K = None

    def update(self, vz: float, throttle: float) -> tuple:
        k = self._engine.throttle_update(np.array([vz]), np.array([throttle]))
        st = self._engine.throttle_state()[0]
        self.x = st[0:2].reshape(2, 1)
        self.P_mtx = st[2:6].reshape(2, 2)
        return float(k[0]), self.x, self.P_mtx
