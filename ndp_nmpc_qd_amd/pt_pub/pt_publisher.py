"""Batched reference-window publisher.

Reference: NMPCRefPublisher (ndp_nmpc/scripts/pt_pub/pt_publisher.py:33-103) keeps, per vehicle, a list of 101 reference
points 0.02 s apart and hands the controller every fifth one (params/nmpc_params.py:40-43): 21 states and the first 20
controls.  `reset` fills the list with the points at i * 0.02 s, i = 0..99, and duplicates the first one in front (:62-76);
every `get_nmpc_pts(ros_t)` drops the oldest entry and appends the point at ros_t + T_horizon (:79-97); `gen_fix_pt_ref`
fills it with the current odometry and u = [0, 0, 0, mass * g] (:40-55).  BatchedNMPCRefPublisher keeps that very list on
the device, one ring per vehicle (ndp_ref_list_*): one new point per vehicle and tick, like the reference.
`get_nmpc_pts_direct(t)` is the list-free idealisation {point(t + k * 0.1)}, k = 0..N, evaluated on the spot
(ndp_ref_window): what the list converges to once the start-up duplicate has been shifted out.
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
    """gen_fix_pt_ref(x_odom) / reset(traj_coeff) / get_nmpc_pts(t) for every instance of a BatchedNMPC engine; times are
    seconds since the trajectory start, (ros_t - start_ros_t).to_sec() in the reference."""

    def __init__(self, engine):
        self.engine = engine
        self.t_all = None

    def gen_fix_pt_ref(self, x_odom, quirk_b1=True):
        """pt_publisher.py:40-55: first reference, every node equal to the current odometry state x_odom[B,10]
        (odom_2_nmpc_x), u_r = [0, 0, 0, mass * g] (the reference's value although u[3] is an acceleration: SURVEY B1)."""
        self.engine.ref_list_fix_pt(x_odom, quirk_b1=quirk_b1)
        return self.engine.ref_list_window()

    def reset(self, traj_coeff: TrajCoefficients):
        """NMPCRefPublisher.reset (pt_publisher.py:57-76): the trajectory clock of every vehicle restarts at 0 and the list
        is rebuilt from the trajectory (with the start-up duplicate)."""
        c = traj_coeff
        self.engine.ref_set_trajectory(c.coeff_x, c.coeff_y, c.coeff_z, c.coeff_yaw, c.traj_time_cum, c.traj_time_seg, c.final_pt)
        self.engine.ref_list_reset()
        self.t_all = np.asarray(c.traj_time_cum)[:, -1].copy()

    def get_nmpc_ref_from_long_list(self):
        """pt_publisher.py:99-103 (nmpc_node.py:151 calls it right after reset)."""
        return self.engine.ref_list_window()

    def get_nmpc_pts(self, t):
        """t[B]: seconds since reset.  Returns xr[B,N+1,10], ur[B,N,4] (pt_publisher.py:79-103)."""
        return self.engine.ref_list_window(np.broadcast_to(np.asarray(t, dtype=np.float64), (self.engine.B,)))

    def get_nmpc_pts_direct(self, t):
        """The window {point(t + k * th_pred)} evaluated directly, no list state (ndp_ref_window)."""
        return self.engine.ref_window(np.broadcast_to(np.asarray(t, dtype=np.float64), (self.engine.B,)))

    def is_activated(self, t, t_pred=0.0):
        """base_pt_publisher.py:93-96: a vehicle's trajectory is finished once t - t_pred passes its end."""
        return np.asarray(t) - t_pred < self.t_all
