"""DownwashNN on MI355X -- same constructor / update as the reference class
(ndp_nmpc/scripts/dnwash_nn_est/downwash_nn.py:10-29); the 6-128-64-128-3 MLP (nn_net.py:7-18) runs as
one fused f32-MFMA HIP kernel.  Weights: the shipped SN=4 state dict exported to weights/downwash_sn4.bin
(located relative to this package, not the cwd -- SURVEY B9)."""
import numpy as np

from ..batched import BatchedNMPC
from ..params import nmpc_params as CP


class DownwashNN:
    def __init__(self, device=0):
        self._engine = BatchedNMPC(batch=1, N=CP.N_node, disturbance=True, device=device, load_mlp=True)

    def update(self, other_pred_x: np.array, ego_pred_x: np.array):
        # input = (other - ego)[:, 0:6] cast to fp32 (downwash_nn.py:22-23); output fp32 [N+1, 3]
        other = np.asarray(other_pred_x, dtype=np.float64)[None]
        ego = np.asarray(ego_pred_x, dtype=np.float64)[None]
        return self._engine.downwash(other, ego, None)[0]
