"""Synthetic hot-path inputs: figure-eight reference trajectories run
through the reference's differential-flatness map, perturbed initial states and
neighbour trajectories for the downwash predictor.

The flatness map restates pt_pub/pt_publisher.py:188-248 (yaw = 0) in batched
numpy; the reference window indexing (21 states / 20 controls, 0.1 s apart)
follows params/nmpc_params.py:40-43.
"""
import numpy as np

from .params import nmpc_params as CP

SEED0 = 20231213


def _quat_from_rot(R):
    """Rotation matrices [...,3,3] -> quaternions [...,(w,x,y,z)] with w >= 0
    (pt_publisher.py:236 "ROS convention, w > 0")."""
    m00, m01, m02 = R[..., 0, 0], R[..., 0, 1], R[..., 0, 2]
    m10, m11, m12 = R[..., 1, 0], R[..., 1, 1], R[..., 1, 2]
    m20, m21, m22 = R[..., 2, 0], R[..., 2, 1], R[..., 2, 2]
    # K matrix eigen-problem is what tf's quaternion_from_matrix solves; for proper
    # rotations it equals the closed form below (trace branch is always valid here
    # because thrust points upward, so w stays well away from 0).
    tr = m00 + m11 + m22
    w = 0.5 * np.sqrt(np.maximum(1.0 + tr, 1e-300))
    x = (m21 - m12) / (4.0 * w)
    y = (m02 - m20) / (4.0 * w)
    z = (m10 - m01) / (4.0 * w)
    q = np.stack([w, x, y, z], axis=-1)
    return q / np.linalg.norm(q, axis=-1, keepdims=True)


def diff_flatness(pos, vel, acc, jerk, yaw=0.0, yaw_dot=0.0, mass=CP.mass, g=CP.gravity):
    """[...,3] arrays -> x[...,10], u[...,4]   (pt_publisher.py:188-248, :129-146)."""
    t_des = acc + np.array([0.0, 0.0, g])
    tn = np.linalg.norm(t_des, axis=-1, keepdims=True)
    z_b = t_des / tn
    u1 = tn[..., 0] * mass
    x_c = np.broadcast_to(np.array([np.cos(yaw), np.sin(yaw), 0.0]), z_b.shape)
    zx = np.cross(z_b, x_c)
    y_b = zx / np.linalg.norm(zx, axis=-1, keepdims=True)
    x_b = np.cross(y_b, z_b)
    R = np.stack([x_b, y_b, z_b], axis=-1)
    h_om = (mass / u1)[..., None] * (jerk - np.sum(z_b * jerk, axis=-1, keepdims=True) * z_b)
    p = -np.sum(h_om * y_b, axis=-1)
    q = np.sum(h_om * x_b, axis=-1)
    r = yaw_dot * z_b[..., 2]
    quat = _quat_from_rot(R)
    x = np.concatenate([pos, vel, quat], axis=-1)
    u = np.stack([p, q, r, u1 / mass], axis=-1)  # collective_force / mass, pt_publisher.py:145
    return x, u


def figure_eight(omega, phi, t):
    """p_r(t) = [2 sin(s), sin(2s), 1 + 0.3 sin(s)], s = omega t + phi, and 3 derivatives."""
    s = omega * t + phi
    sn, cs, s2, c2 = np.sin(s), np.cos(s), np.sin(2 * s), np.cos(2 * s)
    pos = np.stack([2 * sn, s2, 1 + 0.3 * sn], -1)
    vel = np.stack([2 * cs, 2 * c2, 0.3 * cs], -1) * omega[..., None]
    acc = np.stack([-2 * sn, -4 * s2, -0.3 * sn], -1) * (omega ** 2)[..., None]
    jerk = np.stack([-2 * cs, -8 * c2, -0.3 * cs], -1) * (omega ** 3)[..., None]
    return pos, vel, acc, jerk


def figure_eight_traj(B, seed=SEED0, n_seg=40, t_seg=0.5, omega_range=(0.5, 1.5), pairs=False):
    """The figure-eights of make_batch(B, seed=seed) (same omega, phi per instance) as TrajCoefficients.msg arrays: piecewise septic
    polynomials in the reference's normalised segment time (base_pt_publisher.py:100-133: s = (t - time_cum[i]) / time_seg[i],
    derivative d scaled by time_seg^-d) that match position, velocity, acceleration and jerk of the analytic curve at both ends of
    every segment.  Yaw = 0.
    Returns dict coeff_x / coeff_y / coeff_z [B, n_seg*8], coeff_yaw [B, n_seg*4], time_cum [B, n_seg+1], time_seg [B, n_seg],
    final_pt [B, 3] (what BatchedNMPC.ref_set_trajectory takes) + omega, phi.
    pairs: vehicle 2k+1 flies vehicle 2k's curve shifted like make_batch's neighbour windows (phase +-0.2 rad, offset U[-1.5, 1.5]^2 x
    U[0.3, 1.5] m): with other_index = i ^ 1 about a third of the r_horiz gates are open, as in the metric's workload."""
    rng = np.random.Generator(np.random.PCG64(seed))
    omega = rng.uniform(*omega_range, size=B)
    phi = rng.uniform(0.0, 2 * np.pi, size=B)
    offs = np.zeros((B, 3))
    if pairs:
        omega[1::2], phi[1::2] = omega[0:B - B % 2:2], phi[0:B - B % 2:2] + rng.uniform(-0.2, 0.2, size=B // 2)
        offs[1::2] = np.concatenate([rng.uniform(-1.5, 1.5, size=(B // 2, 2)), rng.uniform(0.3, 1.5, size=(B // 2, 1))], axis=1)
    tk = t_seg * np.arange(n_seg + 1)
    pos, vel, acc, jerk = figure_eight(omega[:, None], phi[:, None], tk[None, :])            # [B, n_seg+1, 3]
    pos = pos + offs[:, None, :]
    # rows: p(0), p'(0), p''(0), p'''(0), p(1), p'(1), p''(1), p'''(1) of sum c_i s^i
    A = np.zeros((8, 8))
    for d in range(4):
        for i in range(d, 8):
            fac = float(np.prod(np.arange(i - d + 1, i + 1))) if d else 1.0
            A[d, i] = fac if i == d else 0.0
            A[4 + d, i] = fac
    Ainv = np.linalg.inv(A)
    der = [pos, vel * t_seg, acc * t_seg ** 2, jerk * t_seg ** 3]
    rhs = np.stack([d_[:, :-1] for d_ in der] + [d_[:, 1:] for d_ in der], axis=-1)         # [B, n_seg, 3, 8]
    c = np.einsum("ij,bsaj->bsai", Ainv, rhs)                                                  # [B, n_seg, 3, 8]
    out = {"coeff_" + ax: np.ascontiguousarray(c[:, :, a, :].reshape(B, n_seg * 8)) for a, ax in enumerate("xyz")}
    out["coeff_yaw"] = np.zeros((B, n_seg * 4))
    out["time_cum"] = np.tile(tk, (B, 1))
    out["time_seg"] = np.full((B, n_seg), t_seg)
    out["final_pt"] = np.ascontiguousarray(pos[:, -1, :])
    out["omega"], out["phi"] = omega, phi
    return out


def hover_reference(N=CP.N_node, pos=(0.0, 0.0, 1.0), quirk_b1=False):
    """gen_fix_pt_ref (pt_publisher.py:40-55).  quirk_b1 reproduces u_r[3] = mass*g."""
    xr = np.tile(np.array([*pos, 0, 0, 0, 1, 0, 0, 0], dtype=np.float64), (N + 1, 1))
    c = CP.mass * CP.gravity if quirk_b1 else CP.gravity
    ur = np.tile(np.array([0, 0, 0, c], dtype=np.float64), (N, 1))
    return xr, ur


def make_batch(B, N=CP.N_node, seed=SEED0, downwash=False, dt=CP.th_pred, t0=0.0,
               pos_sigma=0.1, vel_sigma=0.2, quat_sigma=0.03, omega_range=(0.5, 1.5)):
    """Seeded batch of B independent instances.

    Returns dict with x0[B,10], xr[B,N+1,10], ur[B,N,4] (float64) and, when
    downwash is on, other[B,N+1,10] (the neighbour's reference window) and
    ego_xy[B,2] (ego odometry xy used by the r_horiz gate).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    omega = rng.uniform(*omega_range, size=B)
    phi = rng.uniform(0.0, 2 * np.pi, size=B)
    t = t0 + dt * np.arange(N + 1)
    pos, vel, acc, jerk = figure_eight(omega[:, None], phi[:, None], t[None, :])
    xr, ur_full = diff_flatness(pos, vel, acc, jerk)
    ur = np.ascontiguousarray(ur_full[:, :N, :])
    x0 = xr[:, 0, :].copy()
    x0[:, 0:3] += rng.normal(0.0, pos_sigma, size=(B, 3))
    x0[:, 3:6] += rng.normal(0.0, vel_sigma, size=(B, 3))
    x0[:, 6:10] += rng.normal(0.0, quat_sigma, size=(B, 4))
    x0[:, 6:10] /= np.linalg.norm(x0[:, 6:10], axis=1, keepdims=True)
    out = dict(x0=np.ascontiguousarray(x0), xr=np.ascontiguousarray(xr), ur=ur, omega=omega, phi=phi)
    if downwash:
        dphi = rng.uniform(-0.2, 0.2, size=B)
        offs = np.concatenate([rng.uniform(-1.5, 1.5, size=(B, 2)), rng.uniform(0.3, 1.5, size=(B, 1))], axis=1)
        p2, v2, a2, j2 = figure_eight(omega[:, None], (phi + dphi)[:, None], t[None, :])
        other, _ = diff_flatness(p2 + offs[:, None, :], v2, a2, j2)
        out["other"] = np.ascontiguousarray(other)
        out["ego_xy"] = np.ascontiguousarray(x0[:, 0:2])
    return out
