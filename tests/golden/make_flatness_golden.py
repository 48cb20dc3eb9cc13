"""Fixtures for SURVEY 8f-1 produced by RUNNING THE REFERENCE'S OWN pt_pub code (build container only).

/root/reference/ndp_nmpc/scripts/pt_pub/{base_pt_publisher,pt_publisher}.py import rospy, tf_conversions, nav_msgs,
geometry_msgs and the catkin-generated ndp_nmpc.msg, none of which exist here.  They are replaced by minimal stand-ins in
sys.modules (plain attribute containers for the messages; rospy.Time / Duration restating genpy.rostime's integer
nanosecond arithmetic; tf_conversions.transformations restating ROS geometry's quaternion_from_matrix /
euler_from_quaternion).  Everything else that runs is the reference's code, unmodified, read from /root/reference:
  * diff_flatness (pt_publisher.py:188-248): t_des, z_b, y_b, x_b, R_wb, h_omega, p/q/r, collective_force.
    R_wb is captured as the matrix the reference hands to quaternion_from_matrix -- it is the reference's own numpy
    result; the quaternion itself comes from the stand-in and is therefore checked in the tests THROUGH R_wb
    (R(q) == R_wb, w > 0), not trusted.
  * NMPCRefPublisher.reset / _gen_long_list_w_traj / get_nmpc_pts / get_nmpc_ref_from_long_list / gen_fix_pt_ref
    (pt_publisher.py:40-103): the 101-entry sliding list, including the start-up duplicate and the reference's own time
    bookkeeping (rospy.Time with nanosecond truncation).
Inputs are the trajectories of tests/golden/ref_golden.npz (coefficients from the reference's PolymOptimizer).

Output (committed): tests/golden/flat_golden.npz
  flat_pvaj[P,12] flat_yaw[P,2]            the trajectory points fed to diff_flatness
  flat_R[P,3,3] flat_rates[P,3] flat_force[P] flat_q[P,4] (x,y,z,w as the stand-in returned it)
  seq_case (int) seq_veh[V'] seq_t[K]      vehicles of ref_golden case `seq_case` and trajectory times (ros_t - start).to_sec() of the K ticks
  seq_xr0[V',21,10] seq_ur0[V',20,4]       get_nmpc_ref_from_long_list() right after reset (nmpc_node.py:151)
  seq_xr[K,V',21,10] seq_ur[K,V',20,4]     get_nmpc_pts(ros_t_k), k = 1..K
  fix_x[V',10] fix_xr[V',21,10] fix_ur[V',20,4]   gen_fix_pt_ref(odom)
"""
import math
import os
import sys

sys.dont_write_bytecode = True    # importing from /root/reference must not leave __pycache__ there (the tree is read-only by contract)
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/ndp_nmpc/scripts"


# ---------------------------------------------------------------- stand-ins for the ROS modules
class _Rec:
    """Message stand-in: attributes created on first access (nested messages)."""
    _leaf = ()

    def __init__(self):
        for k in self._leaf:
            object.__setattr__(self, k, 0.0)

    def __getattr__(self, k):
        v = _Rec()
        object.__setattr__(self, k, v)
        return v


def _msg(name, leaf=()):
    return type(name, (_Rec,), {"_leaf": tuple(leaf)})


Point = _msg("Point", "xyz")
Vector3 = _msg("Vector3", "xyz")
Quaternion = _msg("Quaternion", "xyzw")


class TrajPt(_Rec):
    def __init__(self):
        super().__init__()
        self.position, self.velocity, self.accel, self.jerk = Point(), Vector3(), Vector3(), Vector3()
        self.yaw, self.yaw_dot = 0.0, 0.0


class TrajFullStatePt(_Rec):
    def __init__(self):
        super().__init__()
        self.pose, self.twist = _Rec(), _Rec()
        self.pose.position, self.pose.orientation = Point(), Quaternion()
        self.twist.linear, self.twist.angular = Vector3(), Vector3()
        self.collective_force = 0.0


class TrajCoefficients(_Rec):
    def __init__(self):
        super().__init__()
        self.coeff_x = self.coeff_y = self.coeff_z = self.coeff_yaw = []
        self.traj_time_cum, self.traj_time_seg, self.final_pt = [], [], Point()


class Odometry(_Rec):
    def __init__(self):
        super().__init__()
        self.pose, self.twist = _Rec(), _Rec()
        self.pose.pose, self.twist.twist = _Rec(), _Rec()
        self.pose.pose.position, self.pose.pose.orientation = Point(), Quaternion()
        self.twist.twist.linear, self.twist.twist.angular = Vector3(), Vector3()


class _TVal:
    """genpy.rostime.TVal: integer seconds + nanoseconds, canonical form 0 <= nsecs < 1e9."""

    def __init__(self, secs=0, nsecs=0):
        secs, nsecs = int(secs), int(nsecs)
        secs += nsecs // 1000000000
        nsecs %= 1000000000
        self.secs, self.nsecs = secs, nsecs

    @classmethod
    def from_sec(cls, float_secs):
        secs = int(float_secs)
        nsecs = int((float_secs - secs) * 1000000000)
        return cls(secs, nsecs)

    def to_sec(self):
        return float(self.secs) + float(self.nsecs) / 1e9


class Duration(_TVal):
    pass


class Time(_TVal):
    _now = (1700000000, 123456789)

    @classmethod
    def now(cls):
        return cls(*cls._now)

    def __add__(self, d):
        return Time(self.secs + d.secs, self.nsecs + d.nsecs)

    def __sub__(self, o):
        return Duration(self.secs - o.secs, self.nsecs - o.nsecs)


_captured_R = []


def quaternion_from_matrix(matrix):
    """ROS geometry tf/transformations.py (2009 version shipped with noetic), restated."""
    _captured_R.append(np.array(matrix, dtype=np.float64)[:3, :3].copy())
    q = np.empty((4,), dtype=np.float64)
    M = np.array(matrix, dtype=np.float64, copy=False)[:4, :4]
    t = np.trace(M)
    if t > M[3, 3]:
        q[3] = t
        q[2] = M[1, 0] - M[0, 1]
        q[1] = M[0, 2] - M[2, 0]
        q[0] = M[2, 1] - M[1, 2]
    else:
        i, j, k = 0, 1, 2
        if M[1, 1] > M[0, 0]:
            i, j, k = 1, 2, 0
        if M[2, 2] > M[i, i]:
            i, j, k = 2, 0, 1
        t = M[i, i] - (M[j, j] + M[k, k]) + M[3, 3]
        q[i] = t
        q[j] = M[i, j] + M[j, i]
        q[k] = M[k, i] + M[i, k]
        q[3] = M[k, j] - M[j, k]
    q *= 0.5 / math.sqrt(t * M[3, 3])
    return q


def euler_from_quaternion(q):
    """sxyz Euler angles; only feeds TrajPt.yaw of traj_pt_now (never the window)."""
    x, y, z, w = q
    return (math.atan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y)), math.asin(max(-1.0, min(1.0, 2 * (w * y - z * x)))),
            math.atan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z)))


def _install():
    def mod(name, **kw):
        m = types.ModuleType(name)
        m.__dict__.update(kw)
        sys.modules[name] = m
        return m
    mod("rospy", Time=Time, Duration=Duration, loginfo=lambda *a, **k: None, logwarn=lambda *a, **k: None)
    tfc = mod("tf_conversions")
    tfc.transformations = mod("tf_conversions.transformations", quaternion_from_matrix=quaternion_from_matrix,
                              euler_from_quaternion=euler_from_quaternion)
    mod("geometry_msgs")
    mod("geometry_msgs.msg", Point=Point, Vector3=Vector3, Quaternion=Quaternion)
    mod("nav_msgs")
    mod("nav_msgs.msg", Odometry=Odometry)
    mod("ndp_nmpc")
    mod("ndp_nmpc.msg", TrajPt=TrajPt, TrajFullStatePt=TrajFullStatePt, TrajCoefficients=TrajCoefficients)
    sys.path.insert(0, REF)


def main():
    _install()
    from params import nmpc_params as CP                          # the reference's own constants
    from pt_pub.pt_publisher import NMPCRefPublisher, diff_flatness   # THE REFERENCE'S CODE
    gold = np.load(os.path.join(HERE, "ref_golden.npz"))
    out = {}
    # ---- (1) diff_flatness on the fixture's trajectory points (+ a few aggressive ones that leave the trace branch)
    pv = np.concatenate([gold[f"pvaj_{c}"].reshape(-1, 12) for c in range(4)])[::3]
    yw = np.concatenate([gold[f"yaw_{c}"].reshape(-1, 2) for c in range(4)])[::3]
    extra = np.zeros((4, 12))
    extra[:, 6:9] = [[0, 0, -30.0], [25.0, 0, -9.81], [0, -25.0, -9.81], [3.0, -2.0, 1.0]]
    extra[:, 9:12] = [[1, 2, 3], [-2, 1, 0.5], [0.3, 0.2, -4], [5, 5, 5]]
    pv = np.concatenate([pv, extra])
    yw = np.concatenate([yw, [[0.3, 0.1], [2.0, -0.4], [-1.0, 0.2], [3.1, 1.0]]])
    R, rates, force, quat = [], [], [], []
    for p, y in zip(pv, yw):
        tp = TrajPt()
        tp.position.x, tp.position.y, tp.position.z = p[0:3]
        tp.velocity.x, tp.velocity.y, tp.velocity.z = p[3:6]
        tp.accel.x, tp.accel.y, tp.accel.z = p[6:9]
        tp.jerk.x, tp.jerk.y, tp.jerk.z = p[9:12]
        tp.yaw, tp.yaw_dot = float(y[0]), float(y[1])
        _captured_R.clear()
        fs = diff_flatness(tp)
        R.append(_captured_R[-1])
        rates.append([fs.twist.angular.x, fs.twist.angular.y, fs.twist.angular.z])
        force.append(fs.collective_force)
        o = fs.pose.orientation
        quat.append([o.x, o.y, o.z, o.w])
    out.update(flat_pvaj=pv, flat_yaw=yw, flat_R=np.array(R), flat_rates=np.array(rates), flat_force=np.array(force),
               flat_q=np.array(quat))
    # ---- (2) the sliding list: reset, the window right after it, 60 control ticks 0.02 s apart (the last ones past the
    #      end of the shortest trajectory: hover at final_pt), with a little timer jitter as rospy.Timer has
    case, K = 2, 60
    coeff, tseg, wpts = gold[f"coeff_{case}"], gold[f"tseg_{case}"], gold[f"wpts_{case}"]
    V = coeff.shape[0]
    rng = np.random.Generator(np.random.PCG64(20231213 + 12))
    jitter = rng.uniform(-2e-3, 2e-3, K)
    tseg = tseg.copy()
    tseg[0] *= 0.12                                # vehicle 0: a trajectory of ~1.2 s, finished inside the sequence
    seq_xr0, seq_ur0, seq_xr, seq_ur, seq_t = [], [], np.zeros((K, V, 21, 10)), np.zeros((K, V, 20, 4)), np.zeros(K)
    for v in range(V):
        msg = TrajCoefficients()
        M = tseg.shape[1]
        msg.coeff_x, msg.coeff_y, msg.coeff_z = (list(coeff[v, :, 8 * a:8 * a + 8].reshape(-1)) for a in range(3))
        msg.coeff_yaw = list(coeff[v, :, 24:28].reshape(-1))
        msg.traj_time_seg = list(tseg[v])
        msg.traj_time_cum = list(np.concatenate([[0.0], np.cumsum(tseg[v])]))
        msg.final_pt = Point()
        msg.final_pt.x, msg.final_pt.y, msg.final_pt.z = wpts[v, 0:3, -1]
        # base_pt_publisher.py:93-94 assigns final_pt to traj_pt.position: give it the message's fields
        pub = NMPCRefPublisher()
        pub.reset(msg, Time.now())
        start = pub.start_ros_t
        x0, u0 = pub.get_nmpc_ref_from_long_list()
        seq_xr0.append(x0); seq_ur0.append(u0)
        for k in range(K):
            ros_t = start + Duration.from_sec((k + 1) * CP.ts_nmpc + jitter[k])
            seq_t[k] = (ros_t - start).to_sec()
            xr, ur = pub.get_nmpc_pts(ros_t)
            seq_xr[k, v], seq_ur[k, v] = xr, ur
    out.update(seq_case=np.int64(case), seq_tseg=tseg, seq_t=seq_t, seq_xr0=np.array(seq_xr0), seq_ur0=np.array(seq_ur0),
               seq_xr=seq_xr, seq_ur=seq_ur)
    # ---- (3) gen_fix_pt_ref
    fx = rng.normal(size=(V, 10))
    fx[:, 6:10] /= np.linalg.norm(fx[:, 6:10], axis=1, keepdims=True)
    fxr, fur = [], []
    for v in range(V):
        od = Odometry()
        p, q = od.pose.pose.position, od.pose.pose.orientation
        p.x, p.y, p.z = fx[v, 0:3]
        lv = od.twist.twist.linear
        lv.x, lv.y, lv.z = fx[v, 3:6]
        q.w, q.x, q.y, q.z = fx[v, 6:10]
        pub = NMPCRefPublisher()
        xr, ur = pub.gen_fix_pt_ref(od)
        fxr.append(xr); fur.append(ur)
    out.update(fix_x=fx, fix_xr=np.array(fxr), fix_ur=np.array(fur))
    np.savez_compressed(os.path.join(HERE, "flat_golden.npz"), **out)
    print("ok", out["flat_R"].shape, out["seq_xr"].shape, "seq_t[:3]", seq_t[:3], "u_fix", out["fix_ur"][0, 0])


if __name__ == "__main__":
    main()
