"""BatchedNMPC: B independent quadrotor OCPs solved per call on one MI355X through the C-ABI.

This is the batched form of the reference's controller objects
(nmpc_ctl/nmpc_body_rate_ctl.py:20-112, ndp_nmpc_ctl/ndp_nmpc_body_rate_ctl.py:20-112):
`reset(xr, ur)` seeds the persistent SQP iterate, `update(...)` runs one SQP-RTI iteration per
instance (plus the downwash MLP when neighbour windows are given) and returns u0[B,4].
Host arrays go through ndp_step; torch CUDA tensors go through ndp_step_device (no host copies).
"""
import ctypes as C

import numpy as np

from . import _lib
from .params import downwash_params as DP
from .params import nmpc_params as CP


class NdpError(RuntimeError):
    pass


class BatchedNMPC:
    def __init__(self, batch, N=CP.N_node, disturbance=False, n_rti=1, qp_mode=_lib.QP_AUTO, device=0,
                 dt=CP.th_pred, load_mlp=None, **cfg_overrides):
        self._lib = _lib.load()
        # ndp_create refuses ipm_refine > 0 (the library default: 2) for shapes whose kernels carry no refinement path (N >= 28, the
        # precision studies): such engines are created with ipm_refine = 0 unless the caller says otherwise (and is then told no)
        if "ipm_refine" not in cfg_overrides and (7 * int(N) - 3 > 192 or cfg_overrides.get("qp_precision", 0)):
            cfg_overrides = dict(cfg_overrides, ipm_refine=0)
        self.cfg = _lib.default_cfg(batch=int(batch), N=int(N), n_rti=int(n_rti), use_fd=int(bool(disturbance)),
                                    qp_mode=int(qp_mode), device=int(device), dt=float(dt), r_horiz=DP.r_horiz,
                                    **cfg_overrides)
        self.B, self.N = int(batch), int(N)
        self.disturbance = bool(disturbance)
        self._h = C.c_void_p()
        rc = self._lib.ndp_create(C.byref(self.cfg), C.byref(self._h))
        if rc != 0:
            self._h = None
            raise NdpError(f"ndp_create failed ({rc}): {self._lib.ndp_last_error(None).decode()}")
        if load_mlp is None:
            load_mlp = self.disturbance
        if load_mlp:
            self.set_mlp_weights(_lib.load_weights())

    # ------------------------------------------------------------------ plumbing
    def close(self):
        if getattr(self, "_h", None):
            self._lib.ndp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            raise NdpError(f"{what} failed ({rc}): {self._lib.ndp_last_error(self._h).decode()}")
        return rc

    def set_mlp_weights(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        self._check(self._lib.ndp_set_mlp_weights(self._h, _lib.ptr(blob), blob.size), "ndp_set_mlp_weights")

    # ------------------------------------------------------------------ reference-shaped API (host arrays)
    def reset(self, xr, ur):
        """nmpc_body_rate_ctl.py:86-91 for every instance."""
        xr = _lib.f64(xr, (self.B, self.N + 1, 10))
        ur = _lib.f64(ur, (self.B, self.N, 4))
        self._check(self._lib.ndp_reset(self._h, _lib.ptr(xr), _lib.ptr(ur)), "ndp_reset")

    def update(self, x0, xr, ur, f=None, other=None, ego_xy=None, raise_on_status=True, full=False):
        """update(x0, xr, ur[, f]) for every instance; returns u0[B,4] float64.

        f: [B,N+1,3] disturbance force (NDP).  other/ego_xy: neighbour reference windows and ego
        odometry xy; the force is then predicted on the device (DownwashNN.update + r_horiz gate).
        full=True: returns (u0, X, U, status, ipm_iters) from the same call (ndp_step_ex: one synchronisation) -- what
        the reference's callers read after solve_for_x0 (solver.get / solver.status).
        """
        x0 = _lib.f64(x0, (self.B, 10))
        xr = _lib.f64(xr, (self.B, self.N + 1, 10))
        ur = _lib.f64(ur, (self.B, self.N, 4))
        u0 = np.empty((self.B, 4), dtype=np.float64)
        if f is not None and other is None and np.asarray(f).dtype == np.float64:
            # a float64 force goes to the device as it is (ndp_step_ex_f64): the reference hands acados a float64 p
            # (ndp_nmpc_body_rate_ctl.py:97-99); DownwashNN's float32 output takes the fp32 entry below -- the same numbers
            f64 = _lib.f64(f, (self.B, self.N + 1, 3))
            X = U = st = it = None
            if full:
                X, U = np.empty((self.B, self.N + 1, 10)), np.empty((self.B, self.N, 4))
                st, it = np.empty(self.B, dtype=np.int32), np.empty(self.B, dtype=np.int32)
            rc = self._check(self._lib.ndp_step_ex_f64(self._h, _lib.ptr(x0), _lib.ptr(xr), _lib.ptr(ur), _lib.ptr(f64), _lib.ptr(u0),
                                                       _lib.ptr(X), _lib.ptr(U), _lib.ptr(st), _lib.ptr(it)), "ndp_step_ex_f64")
            if rc != 0 and raise_on_status:
                raise Exception("acados acados_ocp_solver returned status {}. Exiting.".format(rc))
            return (u0, X, U, st, it) if full else u0
        f32 = None if f is None else np.ascontiguousarray(f, dtype=np.float32).reshape(self.B, self.N + 1, 3)
        other = _lib.f64(other, (self.B, self.N + 1, 10))
        ego_xy = _lib.f64(ego_xy, (self.B, 2))
        if full:
            X, U = np.empty((self.B, self.N + 1, 10)), np.empty((self.B, self.N, 4))
            st, it = np.empty(self.B, dtype=np.int32), np.empty(self.B, dtype=np.int32)
            rc = self._check(self._lib.ndp_step_ex(self._h, _lib.ptr(x0), _lib.ptr(xr), _lib.ptr(ur), _lib.ptr(f32),
                                                   _lib.ptr(other), _lib.ptr(ego_xy), _lib.ptr(u0), _lib.ptr(X), _lib.ptr(U),
                                                   _lib.ptr(st), _lib.ptr(it)), "ndp_step_ex")
        else:
            rc = self._check(self._lib.ndp_step(self._h, _lib.ptr(x0), _lib.ptr(xr), _lib.ptr(ur), _lib.ptr(f32),
                                                _lib.ptr(other), _lib.ptr(ego_xy), _lib.ptr(u0)), "ndp_step")
        if rc != 0 and raise_on_status:
            # same text as nmpc_body_rate_ctl.py:109-110
            raise Exception("acados acados_ocp_solver returned status {}. Exiting.".format(rc))
        return (u0, X, U, st, it) if full else u0

    def update_begin(self, x0, xr, ur, f=None, other=None, ego_xy=None, want_iterate=False):
        """First half of update(): packs the host arrays into a page-locked mirror, enqueues ONE launch that reads the mirror over
        PCIe and writes its results into a page-locked block itself (zero-copy: no H2D / D2H copy operation) and returns without
        waiting (ndp_step_begin).  Up to two steps may be in flight: begin tick i+1 before update_end() of tick i and the packing
        of tick i+1 overlaps tick i's kernel.  The arrays may be reused as soon as this returns.  Drain every begun step with
        update_end() before reset() / set_iterate() / close()."""
        x0 = _lib.f64(x0, (self.B, 10))
        xr = _lib.f64(xr, (self.B, self.N + 1, 10))
        ur = _lib.f64(ur, (self.B, self.N, 4))
        f32 = None if f is None else np.ascontiguousarray(f, dtype=np.float32).reshape(self.B, self.N + 1, 3)
        other = _lib.f64(other, (self.B, self.N + 1, 10))
        ego_xy = _lib.f64(ego_xy, (self.B, 2))
        self._check(self._lib.ndp_step_begin(self._h, _lib.ptr(x0), _lib.ptr(xr), _lib.ptr(ur), _lib.ptr(f32), _lib.ptr(other),
                                             _lib.ptr(ego_xy), 1 if want_iterate else 0), "ndp_step_begin")

    def update_end(self, raise_on_status=True, full=False, out=None):
        """Second half: waits for the oldest begun step; returns u0[B,4] (full=True: (u0, X, U, status, ipm_iters); X, U only if
        that step was begun with want_iterate).  `out`: a [B,4] float64 array to write u0 into."""
        u0 = np.empty((self.B, 4), dtype=np.float64) if out is None else out
        st = it = X = U = None
        if full:
            X, U = np.empty((self.B, self.N + 1, 10)), np.empty((self.B, self.N, 4))
            st, it = np.empty(self.B, dtype=np.int32), np.empty(self.B, dtype=np.int32)
        rc = self._check(self._lib.ndp_step_end(self._h, _lib.ptr(u0), _lib.ptr(X), _lib.ptr(U), _lib.ptr(st), _lib.ptr(it)),
                         "ndp_step_end")
        if rc != 0 and raise_on_status:
            raise Exception("acados acados_ocp_solver returned status {}. Exiting.".format(rc))
        return (u0, X, U, st, it) if full else u0

    def update_debug(self, x0, xr, ur, f=None, other=None, ego_xy=None):
        """B = 1 only: one step that also returns the kernel's LDS image after linearisation (tests)."""
        x0, xr, ur = _lib.f64(x0, (1, 10)), _lib.f64(xr, (1, self.N + 1, 10)), _lib.f64(ur, (1, self.N, 4))
        f32 = None if f is None else np.ascontiguousarray(f, dtype=np.float32).reshape(1, self.N + 1, 3)
        other, ego_xy = _lib.f64(other, (1, self.N + 1, 10)), _lib.f64(ego_xy, (1, 2))
        u0 = np.empty((1, 4))
        dump = np.zeros(self._lib.ndp_debug_lds_doubles(self.N))
        self._check(self._lib.ndp_step_debug(self._h, _lib.ptr(x0), _lib.ptr(xr), _lib.ptr(ur), _lib.ptr(f32),
                                             _lib.ptr(other), _lib.ptr(ego_xy), _lib.ptr(u0), _lib.ptr(dump)),
                    "ndp_step_debug")
        return u0, dump

    def downwash(self, other, ego_ref, ego_xy=None):
        """DownwashNN.update for every instance (+ optional gate); returns f[B,N+1,3] float32."""
        other = _lib.f64(other, (self.B, self.N + 1, 10))
        ego_ref = _lib.f64(ego_ref, (self.B, self.N + 1, 10))
        ego_xy = _lib.f64(ego_xy, (self.B, 2))
        f = np.empty((self.B, self.N + 1, 3), dtype=np.float32)
        self._check(self._lib.ndp_downwash(self._h, _lib.ptr(other), _lib.ptr(ego_ref), _lib.ptr(ego_xy), _lib.ptr(f)),
                    "ndp_downwash")
        return f

    # ------------------------------------------------------------------ f3: hover-throttle estimator + actuator command
    def throttle_reset(self):
        self._check(self._lib.ndp_throttle_reset(self._h), "ndp_throttle_reset")

    def throttle_update(self, vz, throttle):
        """HoverThrottleEstimator.update for every instance: returns k_throttle[B]."""
        vz, throttle = _lib.f64(vz, (self.B,)), _lib.f64(throttle, (self.B,))
        k = np.empty(self.B)
        self._check(self._lib.ndp_throttle_update(self._h, _lib.ptr(vz), _lib.ptr(throttle), _lib.ptr(k)), "ndp_throttle_update")
        return k

    def throttle_state(self):
        st = np.empty((self.B, 8))
        self._check(self._lib.ndp_throttle_get_state(self._h, _lib.ptr(st)), "ndp_throttle_get_state")
        return st

    def actuator_cmd(self, u0, k_throttle):
        """nmpc_u_2_att_tgt for every instance: [wx, wy, wz, c] -> [wx, wy, wz, thrust]."""
        u0, k = _lib.f64(u0, (self.B, 4)), _lib.f64(k_throttle, (self.B,))
        cmd = np.empty((self.B, 4))
        self._check(self._lib.ndp_actuator_cmd(self._h, _lib.ptr(u0), _lib.ptr(k), _lib.ptr(cmd)), "ndp_actuator_cmd")
        return cmd

    # ------------------------------------------------------------------ f2: follower reference relay
    def relay_reset(self):
        self._check(self._lib.ndp_relay_reset(self._h), "ndp_relay_reset")

    def relay_formation(self, form):
        """One formation_ref message per instance (nmpc_follower_node.py:44-56): returns the filtered offsets [B,3]."""
        form = _lib.f64(form, (self.B, 3))
        off = np.empty((self.B, 3))
        self._check(self._lib.ndp_relay_formation(self._h, _lib.ptr(form), _lib.ptr(off)), "ndp_relay_formation")
        return off

    def relay_reference(self, xr_lead):
        """Leader windows -> follower references (nmpc_follower_node.py:58-74); ur is the leader's, unchanged."""
        xr_lead = _lib.f64(xr_lead, (self.B, self.N + 1, 10))
        out = np.empty_like(xr_lead)
        self._check(self._lib.ndp_relay_reference(self._h, _lib.ptr(xr_lead), _lib.ptr(out)), "ndp_relay_reference")
        return out

    # ------------------------------------------------------------------ f1: reference window generation
    def ref_set_trajectory(self, coeff_x, coeff_y, coeff_z, coeff_yaw, time_cum, time_seg, final_pt):
        """TrajCoefficients of every instance (NMPCRefPublisher.reset, pt_pub/pt_publisher.py:57-60); all instances
        carry the same number of segments: coeff_x/y/z[B,n_seg*8], coeff_yaw[B,n_seg*4], time_cum[B,n_seg+1], ..."""
        time_seg = _lib.f64(time_seg)
        n_seg = time_seg.shape[1]
        cx, cy, cz = (_lib.f64(np.reshape(c, (self.B, n_seg * 8)), (self.B, n_seg * 8)) for c in (coeff_x, coeff_y, coeff_z))
        cyaw = _lib.f64(np.reshape(coeff_yaw, (self.B, n_seg * 4)), (self.B, n_seg * 4))
        time_cum, final_pt = _lib.f64(time_cum, (self.B, n_seg + 1)), _lib.f64(final_pt, (self.B, 3))
        self._check(self._lib.ndp_ref_set_trajectory(self._h, n_seg, _lib.ptr(cx), _lib.ptr(cy), _lib.ptr(cz), _lib.ptr(cyaw),
                                                     _lib.ptr(time_cum), _lib.ptr(_lib.f64(time_seg, (self.B, n_seg))),
                                                     _lib.ptr(final_pt)), "ndp_ref_set_trajectory")

    def ref_window(self, t):
        """get_nmpc_pts for every instance at trajectory times t[B]: returns xr[B,N+1,10], ur[B,N,4]."""
        t = _lib.f64(t, (self.B,))
        xr, ur = np.empty((self.B, self.N + 1, 10)), np.empty((self.B, self.N, 4))
        self._check(self._lib.ndp_ref_window(self._h, _lib.ptr(t), _lib.ptr(xr), _lib.ptr(ur)), "ndp_ref_window")
        return xr, ur

    # the reference's own sliding list of reference points (NMPCRefPublisher, pt_pub/pt_publisher.py:36-103), on the device
    def ref_list_reset(self):
        """_gen_long_list_w_traj for every vehicle (needs ref_set_trajectory)."""
        self._check(self._lib.ndp_ref_list_reset(self._h), "ndp_ref_list_reset")

    def ref_list_fix_pt(self, x_odom, quirk_b1=True):
        """gen_fix_pt_ref's list: every entry = x_odom[B,10], u = [0, 0, 0, mass*g] (the reference's value)."""
        x_odom = _lib.f64(x_odom, (self.B, 10))
        self._check(self._lib.ndp_ref_list_fix_pt(self._h, _lib.ptr(x_odom), int(bool(quirk_b1))), "ndp_ref_list_fix_pt")

    def ref_list_window(self, t=None):
        """t[B] given: get_nmpc_pts (drop the oldest entry, append the point at t + T_horizon, return the window);
        t None: get_nmpc_ref_from_long_list only."""
        t = _lib.f64(t, (self.B,))
        xr, ur = np.empty((self.B, self.N + 1, 10)), np.empty((self.B, self.N, 4))
        self._check(self._lib.ndp_ref_list_window(self._h, _lib.ptr(t), _lib.ptr(xr), _lib.ptr(ur)), "ndp_ref_list_window")
        return xr, ur

    def ref_list_advance_device(self, t, stream=None):
        import torch
        self._check(self._lib.ndp_ref_list_advance_device(self._h, self._dptr(t, torch.float64, (self.B,)), self._stream(stream)),
                    "ndp_ref_list_advance_device")

    def ref_list_window_device(self, xr_out, ur_out, stream=None):
        import torch
        self._check(self._lib.ndp_ref_list_window_device(
            self._h, self._dptr(xr_out, torch.float64, (self.B, self.N + 1, 10)),
            self._dptr(ur_out, torch.float64, (self.B, self.N, 4)), self._stream(stream)), "ndp_ref_list_window_device")

    def ref_window_device(self, t, xr_out, ur_out, stream=None):
        import torch
        self._check(self._lib.ndp_ref_window_device(
            self._h, self._dptr(t, torch.float64, (self.B,)), self._dptr(xr_out, torch.float64, (self.B, self.N + 1, 10)),
            self._dptr(ur_out, torch.float64, (self.B, self.N, 4)), self._stream(stream)), "ndp_ref_window_device")

    # ------------------------------------------------------------------ the node's control tick (odometry in, command out)
    def tick_config(self, other_index=None, gate=True):
        """Who is whose neighbour (int32[B], < 0 / None: none) and whether the r_horiz gate on the ego odometry applies
        (ndp_nmpc_leader_node.py:40,60-76)."""
        oi = None if other_index is None else np.ascontiguousarray(other_index, dtype=np.int32).reshape(self.B)
        self._check(self._lib.ndp_tick_config(self._h, _lib.ptr(oi), 1 if gate else 0), "ndp_tick_config")

    def tick_reset(self):
        """nmpc_ctl.reset(*ref_pub.get_nmpc_ref_from_long_list()) (nmpc_node.py:92,151-152) on the device."""
        self._check(self._lib.ndp_tick_reset(self._h), "ndp_tick_reset")

    def _tick_in(self, x_odom, t, vz, throttle, estimate, want_u0):
        flags = (_lib.TICK_ESTIMATE if estimate else 0) | (_lib.TICK_WANT_U0 if want_u0 else 0)
        if t is not None and np.ndim(t) == 0:            # one clock for every vehicle: a scalar (NDP_TICK_T_UNIFORM)
            t, flags = np.array([float(t)]), flags | _lib.TICK_T_UNIFORM
        else:
            t = _lib.f64(t, (self.B,))
        return (_lib.f64(x_odom, (self.B, 10)), t, _lib.f64(vz, (self.B,)), _lib.f64(throttle, (self.B,)), flags)

    def tick_begin(self, x_odom, t=None, vz=None, throttle=None, estimate=False, want_u0=False):
        """First half of tick(): ndp_tick_begin -- only x_odom[B,10] (+ t, vz, throttle [B]) cross PCIe; returns without waiting.
        Up to two ticks in flight."""
        x, t, vz, th, flags = self._tick_in(x_odom, t, vz, throttle, estimate, want_u0)
        self._check(self._lib.ndp_tick_begin(self._h, _lib.ptr(x), _lib.ptr(t), _lib.ptr(vz), _lib.ptr(th), flags), "ndp_tick_begin")

    def tick_end(self, raise_on_status=True, full=False, out=None):
        """Second half: cmd[B,4] = [wx, wy, wz, thrust] of the oldest tick (full=True: (cmd, u0, status, ipm_iters); u0 only if the
        tick was begun with want_u0)."""
        cmd = np.empty((self.B, 4)) if out is None else out
        u0 = st = it = None
        if full:
            u0, st, it = np.empty((self.B, 4)), np.empty(self.B, dtype=np.int32), np.empty(self.B, dtype=np.int32)
        rc = self._check(self._lib.ndp_tick_end(self._h, _lib.ptr(cmd), _lib.ptr(u0), _lib.ptr(st), _lib.ptr(it)), "ndp_tick_end")
        if rc != 0 and raise_on_status:
            raise Exception("acados acados_ocp_solver returned status {}. Exiting.".format(rc))
        return (cmd, u0, st, it) if full else cmd

    def tick(self, x_odom, t=None, vz=None, throttle=None, estimate=False, raise_on_status=True, full=False):
        """One control period of ControllerNode (nmpc_node.py:211-231,251-253) for every vehicle: reference list advance ->
        estimator -> control step (downwash from the neighbour's window) -> actuator command.  Returns cmd[B,4]
        (full=True: (cmd, u0, status, ipm_iters))."""
        x, t, vz, th, flags = self._tick_in(x_odom, t, vz, throttle, estimate, full)
        cmd = np.empty((self.B, 4))
        u0 = st = it = None
        if full:
            u0, st, it = np.empty((self.B, 4)), np.empty(self.B, dtype=np.int32), np.empty(self.B, dtype=np.int32)
        rc = self._check(self._lib.ndp_tick(self._h, _lib.ptr(x), _lib.ptr(t), _lib.ptr(vz), _lib.ptr(th), flags, _lib.ptr(cmd),
                                            _lib.ptr(u0), _lib.ptr(st), _lib.ptr(it)), "ndp_tick")
        if rc != 0 and raise_on_status:
            raise Exception("acados acados_ocp_solver returned status {}. Exiting.".format(rc))
        return (cmd, u0, st, it) if full else cmd

    def tick_device(self, x_odom, cmd_out, t=None, vz=None, throttle=None, estimate=False, u0_out=None, stream=None):
        """The tick's launches on CUDA tensors and a caller's stream (ndp_tick_device); no synchronisation."""
        import torch
        B = self.B
        flags = _lib.TICK_ESTIMATE if estimate else 0
        if t is not None and not isinstance(t, torch.Tensor):       # a scalar: one time for every vehicle, read by the call
            t_host = np.array([float(t)])
            tp, flags = _lib.ptr(t_host), flags | _lib.TICK_T_UNIFORM
        else:
            tp = self._dptr(t, torch.float64, (B,))
        self._check(self._lib.ndp_tick_device(
            self._h, self._dptr(x_odom, torch.float64, (B, 10)), tp,
            self._dptr(vz, torch.float64, (B,)), self._dptr(throttle, torch.float64, (B,)), flags,
            self._dptr(cmd_out, torch.float64, (B, 4)), self._dptr(u0_out, torch.float64, (B, 4)), self._stream(stream)),
            "ndp_tick_device")

    # ---- the tick with neighbours on other ranks: advance -> window columns -> (exchange) -> step
    def tick_config_remote(self, windows, other_index, gate=True):
        """Neighbour windows come from `windows` (CUDA tensor [rows, N+1, 6 or 10]: an exchange's gathered / peer-mapped buffer, kept
        alive by the caller); other_index[B] = the row of every vehicle's neighbour (< 0: none).  ndp_tick_config_remote."""
        import torch
        assert isinstance(windows, torch.Tensor) and windows.is_cuda and windows.is_contiguous() and windows.dtype == torch.float64
        assert windows.dim() == 3 and windows.shape[1] == self.N + 1 and windows.shape[2] in (6, 10)
        oi = np.ascontiguousarray(other_index, dtype=np.int32).reshape(self.B)
        self._tick_windows = windows
        self._check(self._lib.ndp_tick_config_remote(self._h, C.c_void_p(windows.data_ptr()), int(windows.shape[2]), C.c_int64(int(windows.shape[0])),
                                                     _lib.ptr(oi), 1 if gate else 0), "ndp_tick_config_remote")

    def tick_advance_device(self, x_odom, t=None, vz=None, throttle=None, estimate=False, stream=None):
        """Stage 1: list advance (t: scalar, CUDA tensor [B] or None) + estimator.  ndp_tick_advance_device."""
        import torch
        flags = _lib.TICK_ESTIMATE if estimate else 0
        if t is not None and not isinstance(t, torch.Tensor):
            t_host = np.array([float(t)])
            tp, flags = _lib.ptr(t_host), flags | _lib.TICK_T_UNIFORM
        else:
            tp = self._dptr(t, torch.float64, (self.B,))
        self._check(self._lib.ndp_tick_advance_device(self._h, self._dptr(x_odom, torch.float64, (self.B, 10)), tp,
                                                      self._dptr(vz, torch.float64, (self.B,)), self._dptr(throttle, torch.float64, (self.B,)),
                                                      flags, self._stream(stream)), "ndp_tick_advance_device")

    def tick_window_pv_device(self, pv_out, stream=None):
        """Stage 2: this tick's window, position / velocity columns -> pv_out [B, N+1, 6] (the exchange's send buffer)."""
        import torch
        self._check(self._lib.ndp_tick_window_pv_device(self._h, self._dptr(pv_out, torch.float64, (self.B, self.N + 1, 6)), self._stream(stream)),
                    "ndp_tick_window_pv_device")

    def tick_step_device(self, x_odom, cmd_out, u0_out=None, stream=None):
        """Stage 3 (behind the exchange): the control step against the exchanged windows + the actuator command."""
        import torch
        self._check(self._lib.ndp_tick_step_device(self._h, self._dptr(x_odom, torch.float64, (self.B, 10)),
                                                   self._dptr(cmd_out, torch.float64, (self.B, 4)), self._dptr(u0_out, torch.float64, (self.B, 4)),
                                                   self._stream(stream)), "ndp_tick_step_device")

    # ------------------------------------------------------------------ f4: plant step (closed-loop rollouts)
    def plant_step(self, x, u, f=None, dt=CP.ts_nmpc, substeps=4):
        x = _lib.f64(x, (self.B, 10)).copy()
        u = _lib.f64(u, (self.B, 4))
        f = _lib.f64(f, (self.B, 3))
        self._check(self._lib.ndp_plant_step(self._h, _lib.ptr(x), _lib.ptr(u), _lib.ptr(f), float(dt), int(substeps)),
                    "ndp_plant_step")
        return x

    def rollout_device(self, ticks, x, log=None, t0=0.0, dt=CP.ts_nmpc, substeps=4, stream=None):
        """Closed loop on the device: reference window -> control step -> plant step, `ticks` times, enqueued on `stream`.
        x[B,10] (CUDA tensor) is the plant state, updated in place; log[ticks,B,10] (optional) receives every state."""
        import torch
        self._check(self._lib.ndp_rollout_device(
            self._h, int(ticks), float(t0), float(dt), int(substeps), self._dptr(x, torch.float64, (self.B, 10)),
            self._dptr(log, torch.float64, (int(ticks), self.B, 10)), self._stream(stream)), "ndp_rollout_device")

    def get_iterate(self):
        X = np.empty((self.B, self.N + 1, 10))
        U = np.empty((self.B, self.N, 4))
        self._check(self._lib.ndp_get_iterate(self._h, _lib.ptr(X), _lib.ptr(U)), "ndp_get_iterate")
        return X, U

    def set_iterate(self, X=None, U=None):
        X = _lib.f64(X, (self.B, self.N + 1, 10))
        U = _lib.f64(U, (self.B, self.N, 4))
        self._check(self._lib.ndp_set_iterate(self._h, _lib.ptr(X), _lib.ptr(U)), "ndp_set_iterate")

    def status(self):
        st = np.zeros(self.B, dtype=np.int32)
        it = np.zeros(self.B, dtype=np.int32)
        self._check(self._lib.ndp_get_status(self._h, _lib.ptr(st), _lib.ptr(it)), "ndp_get_status")
        return st, it

    def host_info(self):
        """{hw_threads, usable_cores, pack_threads} of the host-array path (ndp_debug_host_info)."""
        out = np.zeros(3, dtype=np.int32)
        self._check(self._lib.ndp_debug_host_info(self._h, _lib.ptr(out)), "ndp_debug_host_info")
        return {"hw_threads": int(out[0]), "usable_cores": int(out[1]), "pack_threads": int(out[2])}

    def active_set(self):
        """(sweeps[B], act[B,N,4]) of the last step's QPs (ndp_get_active_set): Riccati sweeps taken by QP_AUTO's active-set
        iterations, and the set kept for the next step (+1 / -1: input on its upper / lower bound)."""
        sw = np.zeros(self.B, dtype=np.int32)
        act = np.zeros((self.B, self.N, 4), dtype=np.int8)
        self._check(self._lib.ndp_get_active_set(self._h, _lib.ptr(sw), _lib.ptr(act)), "ndp_get_active_set")
        return sw & 0xfff, act

    def set_active_set(self, act):
        """The kept sets handed in (act[B,N,4] int8, entries -1 / 0 / +1): the warm start of the next step's QPs (ndp_set_active_set;
        after set_iterate, which empties them)."""
        a = np.ascontiguousarray(act, dtype=np.int8).reshape(self.B, self.N, 4)
        self._check(self._lib.ndp_set_active_set(self._h, _lib.ptr(a)), "ndp_set_active_set")

    def condensed_kept(self):
        """qp_precision 5 / 6 (config 5's condensed study): how many of the last step's QPs per instance kept their condensed solve's
        result (it passed the fp64 inside-the-box test); the others were solved by the fp64 Riccati path."""
        sw = np.zeros(self.B, dtype=np.int32)
        self._check(self._lib.ndp_get_active_set(self._h, _lib.ptr(sw), None), "ndp_get_active_set")
        return (sw >> 12) & 0xf

    # ------------------------------------------------------------------ HBM-resident API (torch CUDA tensors)
    @staticmethod
    def _dptr(t, dtype, shape):
        if t is None:
            return None
        import torch
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous() and t.dtype == dtype
                and tuple(t.shape) == tuple(shape)):
            raise ValueError(f"expected contiguous CUDA tensor {dtype} {tuple(shape)}")
        return C.c_void_p(t.data_ptr())

    def reset_device(self, xr, ur, stream=None):
        import torch
        self._check(self._lib.ndp_reset_device(self._h, self._dptr(xr, torch.float64, (self.B, self.N + 1, 10)),
                                               self._dptr(ur, torch.float64, (self.B, self.N, 4)),
                                               self._stream(stream)), "ndp_reset_device")

    def update_device(self, x0, xr, ur, u0_out, f=None, other=None, ego_xy=None, stream=None, other_index=None):
        """Enqueues one control step on `stream` (default: the library's stream); no synchronisation.

        other: [B,N+1,10] neighbour windows, or -- with other_index (int32[B]: row of `other` holding instance i's
        neighbour, < 0 = none) -- any [rows,N+1,10] or [rows,N+1,6] buffer, e.g. what an all-gather over the GPUs left."""
        self.bind_update_device(x0, xr, ur, u0_out, f=f, other=other, ego_xy=ego_xy, stream=stream, other_index=other_index)()

    def bind_update_device(self, x0, xr, ur, u0_out, f=None, other=None, ego_xy=None, stream=None, other_index=None):
        """update_device with the argument checks done ONCE: returns a callable that enqueues the step on the same buffers (one
        ctypes call, ~3 us of host time instead of ~13) -- for loops that launch from the host at the control step's own pace.
        The tensors must stay alive and in place for as long as the callable is used."""
        import torch
        B, N = self.B, self.N
        stride = 10
        if other is not None and hasattr(other, "dev_ptr"):
            # raw device memory (dist.DevWindows: e.g. another process's window buffer mapped with ndp_peer_open)
            shp = tuple(other.shape)
            if not (len(shp) == 3 and shp[1] == N + 1 and shp[2] in (6, 10) and (other_index is not None or shp[0] == B)):
                raise ValueError("other: expected device windows of shape [rows, N+1, 6 or 10]")
            stride = int(shp[2])
            optr = C.c_void_p(int(other.dev_ptr))
        elif other is not None and (other_index is not None or (other.dim() == 3 and other.shape[2] == 6)):
            # rows picked through other_index (any number of rows), or row i = instance i of a [B, N+1, 6] position / velocity window
            if not (other.is_cuda and other.is_contiguous() and other.dtype == torch.float64 and other.dim() == 3
                    and other.shape[1] == N + 1 and other.shape[2] in (6, 10) and (other_index is not None or other.shape[0] == B)):
                raise ValueError("other: expected a contiguous CUDA float64 [rows, N+1, 6 or 10] tensor")
            stride = int(other.shape[2])
            optr = C.c_void_p(other.data_ptr())
        else:
            optr = self._dptr(other, torch.float64, (B, N + 1, 10))
        args = (self._h, self._dptr(x0, torch.float64, (B, 10)), self._dptr(xr, torch.float64, (B, N + 1, 10)),
                self._dptr(ur, torch.float64, (B, N, 4)), self._dptr(f, torch.float32, (B, N + 1, 3)),
                optr, stride, self._dptr(other_index, torch.int32, (B,)), self._dptr(ego_xy, torch.float64, (B, 2)),
                self._dptr(u0_out, torch.float64, (B, 4)), self._stream(stream))
        fn, check = self._lib.ndp_step_device_ex, self._check

        def launch():
            check(fn(*args), "ndp_step_device")
        return launch

    # ------------------------------------------------------------------ downwash one tick ahead (second stream)
    def downwash_prefetch_device(self, other, ego_ref, ego_xy=None, other_index=None, after_stream=None, on_stream=None):
        """Enqueues gate + MLP for the NEXT control step on the engine's second stream (ndp_downwash_prefetch_device): it runs
        beside the control-step kernel of the current tick.  other / other_index as in update_device; ego_ref = that tick's xr."""
        import torch
        B, N = self.B, self.N
        if hasattr(other, "dev_ptr"):
            stride, optr = int(other.shape[2]), C.c_void_p(int(other.dev_ptr))
        else:
            if not (other.is_cuda and other.is_contiguous() and other.dtype == torch.float64 and other.dim() == 3
                    and other.shape[1] == N + 1 and other.shape[2] in (6, 10) and (other_index is not None or other.shape[0] == B)):
                raise ValueError("other: expected a contiguous CUDA float64 [rows, N+1, 6 or 10] tensor")
            stride, optr = int(other.shape[2]), C.c_void_p(other.data_ptr())
        self._check(self._lib.ndp_downwash_prefetch_device(
            self._h, optr, stride, self._dptr(other_index, torch.int32, (B,)), self._dptr(ego_ref, torch.float64, (B, N + 1, 10)),
            self._dptr(ego_xy, torch.float64, (B, 2)), self._stream(after_stream), self._stream(on_stream)),
            "ndp_downwash_prefetch_device")

    def update_device_prefetched(self, x0, xr, ur, u0_out, stream=None):
        """The control step that consumes the oldest prediction of downwash_prefetch_device (ndp_step_device_prefetched)."""
        import torch
        B, N = self.B, self.N
        self._check(self._lib.ndp_step_device_prefetched(
            self._h, self._dptr(x0, torch.float64, (B, 10)), self._dptr(xr, torch.float64, (B, N + 1, 10)),
            self._dptr(ur, torch.float64, (B, N, 4)), self._dptr(u0_out, torch.float64, (B, 4)), self._stream(stream)),
            "ndp_step_device_prefetched")

    def prefetch_join(self, stream=None):
        self._check(self._lib.ndp_prefetch_join(self._h, self._stream(stream)), "ndp_prefetch_join")

    def prefetch_stats(self):
        out = (C.c_ulonglong * 5)()
        self._check(self._lib.ndp_prefetch_stats(self._h, out), "ndp_prefetch_stats")
        return dict(predictions=int(out[0]), steps=int(out[1]), force_timeouts=int(out[2]), slot_timeouts=int(out[3]),
                    late_waves=int(out[4]))

    def force_slot(self, slot):
        """The force of prediction m (slot m & 1) as a CUDA tensor view [B, N+1, 3] float32 (valid after prefetch_join / stats)."""
        import torch
        from .dist import _DevMem
        ptr = self._lib.ndp_device_force_slot(self._h, int(slot))
        if not ptr:
            raise NdpError("no force slots: downwash_prefetch_device was never called")
        return torch.as_tensor(_DevMem(ptr, (self.B, self.N + 1, 3), "<f4"), device=torch.device("cuda", self.cfg.device))

    def device_force(self):
        """The force the fused downwash (or ndp_downwash_device) of the last step left in HBM: CUDA tensor view [B, N+1, 3] float32
        (ndp_device_force); valid once the step's stream has reached it."""
        import torch
        from .dist import _DevMem
        ptr = self._lib.ndp_device_force(self._h)
        return torch.as_tensor(_DevMem(ptr, (self.B, self.N + 1, 3), "<f4"), device=torch.device("cuda", self.cfg.device))

    def track_steps(self, on=True):
        """Every control step launched from now on marks an event at its completion without a packet of its own (ndp_track_steps)."""
        self._check(self._lib.ndp_track_steps(self._h, 1 if on else 0), "ndp_track_steps")

    def last_step_event(self):
        """The raw HIP event (ctypes.c_void_p) of the control step launched last (track_steps first)."""
        ev = C.c_void_p()
        self._check(self._lib.ndp_last_step_event(self._h, C.byref(ev)), "ndp_last_step_event")
        return ev

    @property
    def refine_active(self):
        """True when cfg.ipm_refine acts on this engine's control steps (three-slot kernels, N <= 27); the five-slot kernels and the
        late-force step ignore it (include/ndp_nmpc.h: ndp_cfg.ipm_refine)."""
        return self._lib.ndp_refine_active(self._h) == 1

    @property
    def work_queue(self):
        """True when this engine's steps send interior-point solves through the in-kernel work queue (cfg.work_queue)."""
        return self._lib.ndp_work_queue_enabled(self._h) == 1

    def downwash_device(self, other, ego_ref, f_out, ego_xy=None, stream=None):
        import torch
        B, N = self.B, self.N
        self._check(self._lib.ndp_downwash_device(
            self._h, self._dptr(other, torch.float64, (B, N + 1, 10)), self._dptr(ego_ref, torch.float64, (B, N + 1, 10)),
            self._dptr(ego_xy, torch.float64, (B, 2)), self._dptr(f_out, torch.float32, (B, N + 1, 3)),
            self._stream(stream)), "ndp_downwash_device")

    @staticmethod
    def _stream(stream):
        if stream is None:
            return None
        return C.c_void_p(stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream))

    def synchronize(self):
        self._check(self._lib.ndp_synchronize(self._h), "ndp_synchronize")

    def debug_stamps(self, enable=True, read=False):
        """Whole-batch phase stamps (profiling hook): returns [B,24] of the last step when read=True."""
        out = np.zeros((self.B, 24)) if read else None
        self._check(self._lib.ndp_debug_stamps(self._h, int(bool(enable)), _lib.ptr(out)), "ndp_debug_stamps")
        return out

    def timing_enable(self, every=1):
        """Bracket every `every`-th launch of each kernel with HIP events (0 / False: off)."""
        self._check(self._lib.ndp_timing_enable(self._h, int(every)), "ndp_timing_enable")

    def timing_read(self, name):
        tot, n = C.c_double(0.0), C.c_int64(0)
        rc = self._lib.ndp_timing_read(self._h, name.encode(), C.byref(tot), C.byref(n))
        if rc < 0:
            return 0.0, 0
        return tot.value, n.value
