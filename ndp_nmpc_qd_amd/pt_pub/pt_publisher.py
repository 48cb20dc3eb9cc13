"""Batched reference-window publisher.

Reference: NMPCRefPublisher (ndp_nmpc/scripts/pt_pub/pt_publisher.py:33-103) keeps, per vehicle, a list of 101 reference
points 0.02 s apart and hands the controller every fifth one (params/nmpc_params.py:40-43): 21 states 0.1 s apart and the
first 20 controls.  Each list entry is get_traj_full_state_pt at its own time, so the window the controller sees at
trajectory time t is {point(t + k * 0.1)}, k = 0..20 -- which is what the device kernel evaluates directly, for every
vehicle at once, without the list.  (The reference's first tick after reset repeats point(0) once, pt_publisher.py:74-75;
that start-up duplicate is not reproduced.)
"""
from dataclasses import dataclass, field

import numpy as np

from .polym_optimizer import MinMethod, PolymOptimizer


@dataclass
class TrajCoefficients:
    """Arrays of msg/TrajCoefficients.msg for B vehicles with the same number of segments."""
    coeff_x: np.ndarray
    coeff_y: np.ndarray
    coeff_z: np.ndarray
    coeff_yaw: np.ndarray
    traj_time_cum: np.ndarray
    traj_time_seg: np.ndarray
    final_pt: np.ndarray = field(default=None)

    @staticmethod
    def from_waypoints(wpts_xyz_yaw, time_seg, xyz_method=MinMethod.SNAP, yaw_method=MinMethod.ACCEL):
        """wpts_xyz_yaw[B,4,M+1], time_seg[B,M]: one linear solve per axis for the whole batch -- the constraint matrix
        depends on the segment count only, so B vehicles share one factorisation."""
        w = np.asarray(wpts_xyz_yaw, dtype=np.float64)
        time_seg = np.asarray(time_seg, dtype=np.float64)
        B, _, m1 = w.shape
        m = m1 - 1
        ox, oyaw = PolymOptimizer(xyz_method), PolymOptimizer(yaw_method)
        ax, ayaw = ox.constraint_matrix(m), oyaw.constraint_matrix(m)
        rhs_xyz = np.concatenate([ox.rhs(w[b, a]) for b in range(B) for a in range(3)], axis=1)     # [8M, 3B]
        rhs_yaw = np.concatenate([oyaw.rhs(w[b, 3]) for b in range(B)], axis=1)
        cxyz = np.linalg.solve(ax, rhs_xyz).T.reshape(B, 3, -1)
        cyaw = np.linalg.solve(ayaw, rhs_yaw).T
        cum = np.concatenate([np.zeros((B, 1)), np.cumsum(time_seg, axis=1)], axis=1)
        return TrajCoefficients(cxyz[:, 0], cxyz[:, 1], cxyz[:, 2], cyaw, cum, time_seg, w[:, 0:3, -1].copy())


class BatchedNMPCRefPublisher:
    """reset(traj_coeff) / get_nmpc_pts(t) for every instance of a BatchedNMPC engine."""

    def __init__(self, engine):
        self.engine = engine
        self.t_all = None

    def reset(self, traj_coeff: TrajCoefficients):
        """NMPCRefPublisher.reset (pt_publisher.py:57-60): the trajectory clock of every vehicle restarts at 0."""
        c = traj_coeff
        self.engine.ref_set_trajectory(c.coeff_x, c.coeff_y, c.coeff_z, c.coeff_yaw, c.traj_time_cum, c.traj_time_seg, c.final_pt)
        self.t_all = np.asarray(c.traj_time_cum)[:, -1].copy()

    def get_nmpc_pts(self, t):
        """t[B]: seconds since reset.  Returns xr[B,N+1,10], ur[B,N,4] (pt_publisher.py:79-103)."""
        return self.engine.ref_window(np.broadcast_to(np.asarray(t, dtype=np.float64), (self.engine.B,)))

    def is_activated(self, t, t_pred=0.0):
        """base_pt_publisher.py:93-96: a vehicle's trajectory is finished once t - t_pred passes its end."""
        return np.asarray(t) - t_pred < self.t_all
