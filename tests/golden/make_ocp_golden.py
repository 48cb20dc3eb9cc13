"""Fixtures for SURVEY 8 rows a1-a6 produced by RUNNING THE REFERENCE'S OWN controller files (build container only).

/root/reference/ndp_nmpc/scripts/nmpc_ctl/nmpc_body_rate_ctl.py and ndp_nmpc_ctl/ndp_nmpc_body_rate_ctl.py import `casadi` and
`acados_template`, neither of which exists here.  What those two packages DO -- code generation, the SQP-RTI / HPIPM
solve -- is unobtainable; what the reference HANDS them is plain Python and runs here, unmodified, under two stand-ins in
sys.modules:
  * `casadi`: SX.sym / vertcat / vcat / Function / types.SimpleNamespace over a minimal expression graph (class SX below: the
    elementary operations exactly as the reference wrote them, evaluated in float64 in that order; derivatives by forward-mode
    dual numbers through the same graph -- exact, no finite differences, no algebraic rewriting);
  * `acados_template`: attribute-recording AcadosOcp / AcadosModel and an AcadosOcpSolver that records every
    set(stage, field, value) and solve_for_x0(x0).
os.chdir (nmpc_body_rate_ctl.py:29) is intercepted (recorded, not performed), safe_mkdir_recursive (:213-227) is replaced by a
no-op after import (nothing may be written under /root/reference), ACADOS_SOURCE_DIR (:32) points nowhere.

Harvested (prefix nmpc_ / ndp_ for the two classes):
  the OCP definition (nmpc_body_rate_ctl.py:36-80): W, W_e, lbu/ubu/idxbu, lbx/ubx/idxbx, dims.N, dims.np, tf, x0, yref, yref_e,
      parameter_values; cost types, every solver option the reference sets (and the list of those it does NOT set) -> ocp_golden.json
  the model (:115-210 / ndp :146-197): f_expl_expr, cost_y_expr, cost_y_expr_e evaluated at K random (x, u, p) with their exact
      Jacobians w.r.t. x, u (and p: the disturbance columns); f_impl_expr == xdot - f_expl_expr checked here
  one ERK4 step of h = tf / N built FROM f_expl_expr (classical tableau, one step per interval: [acados-knowledge]) with
      the exact derivative of that map (dual numbers through the four stages): xn, A_d, B_d at the same points
  the call sequences of reset(xr, ur) and update(x0, xr, ur[, f]) (:86-112 / ndp :84-112) on random inputs: stage, field, value
  one linearisation case (iterate != reference, N = 20): per stage A_k, B_k, b_k = phi(X_k, U_k, p_k) - X_{k+1} and the
      Gauss-Newton pieces J_x, residuals of cost_y_expr at (X_k, U_k; p_k = xr_k[6:10]) -- what the device's LDS image must hold.

Output (committed): tests/golden/ocp_golden.npz (numeric arrays only), tests/golden/ocp_golden.json (option strings / names).
"""
import json
import math
import os
import sys

sys.dont_write_bytecode = True    # importing from /root/reference must not leave __pycache__ there (the tree is read-only by contract)
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/ndp_nmpc/scripts"


# ================================================================== casadi stand-in: expression graph + dual-number evaluation
class SX:
    """One scalar node: symbol, constant, or an elementary operation on nodes (as written by the caller, no simplification)."""
    __slots__ = ("op", "a", "b", "name")

    def __init__(self, op, a=None, b=None, name=None):
        self.op, self.a, self.b, self.name = op, a, b, name

    @staticmethod
    def sym(name, n=None):
        if n is None:
            return SX("sym", name=name)
        return Vec([SX("sym", name=f"{name}_{i}") for i in range(int(n))])

    @staticmethod
    def _lift(v):
        if isinstance(v, SX):
            return v
        if isinstance(v, (int, float, np.integer, np.floating)):
            return SX("const", a=float(v))
        raise TypeError(f"cannot mix SX with {type(v)}")

    def __add__(self, o): return SX("add", self, SX._lift(o))
    def __radd__(self, o): return SX("add", SX._lift(o), self)
    def __sub__(self, o): return SX("sub", self, SX._lift(o))
    def __rsub__(self, o): return SX("sub", SX._lift(o), self)
    def __mul__(self, o): return SX("mul", self, SX._lift(o))
    def __rmul__(self, o): return SX("mul", SX._lift(o), self)
    def __truediv__(self, o): return SX("div", self, SX._lift(o))
    def __rtruediv__(self, o): return SX("div", SX._lift(o), self)
    def __neg__(self): return SX("neg", self)

    def __pow__(self, e):
        if not (isinstance(e, int) and e >= 1):
            raise TypeError("only small positive integer powers occur in the reference")
        return SX("pow", self, SX("const", a=float(e)))

    def size(self):
        return (1, 1)


class Vec:
    """Column vector of SX nodes (what vertcat returns); .size() like casadi's."""

    def __init__(self, items):
        self.items = list(items)

    def size(self):
        return (len(self.items), 1)

    def __len__(self): return len(self.items)
    def __iter__(self): return iter(self.items)
    def __getitem__(self, i): return self.items[i]

    def __sub__(self, o):
        assert isinstance(o, Vec) and len(o) == len(self)
        return Vec([a - b for a, b in zip(self.items, o.items)])

    def __add__(self, o):
        assert isinstance(o, Vec) and len(o) == len(self)
        return Vec([a + b for a, b in zip(self.items, o.items)])


def vertcat(*args):
    out = []
    for a in args:
        if isinstance(a, Vec):
            out.extend(a.items)
        else:
            out.append(SX._lift(a))
    return Vec(out)


def vcat(lst):
    return vertcat(*lst)


def _subs(node, mapping, memo):
    if id(node) in memo:
        return memo[id(node)]
    if node.op == "sym":
        r = mapping.get(id(node), node)
    elif node.op == "const":
        r = node
    elif node.op == "neg":
        r = SX("neg", _subs(node.a, mapping, memo))
    else:
        r = SX(node.op, _subs(node.a, mapping, memo), _subs(node.b, mapping, memo))
    memo[id(node)] = r
    return r


class Function:
    """ca.Function(name, [inputs], [outputs], in_names, out_names[, opts]): calling it substitutes the arguments for the inputs."""

    def __init__(self, name, ins, outs, in_names=None, out_names=None, opts=None):
        self.name, self.ins, self.outs, self.opts = name, [vertcat(i) for i in ins], [vertcat(o) for o in outs], dict(opts or {})
        self.in_names, self.out_names = list(in_names or []), list(out_names or [])

    def __call__(self, *args):
        assert len(args) == len(self.ins)
        mapping = {}
        for formal, actual in zip(self.ins, args):
            actual = vertcat(actual)
            assert len(actual) == len(formal)
            for f_, a_ in zip(formal.items, actual.items):
                mapping[id(f_)] = a_
        memo = {}
        res = [Vec([_subs(n, mapping, memo) for n in o.items]) for o in self.outs]
        return res[0] if len(res) == 1 else res


class Dual:
    """value + gradient w.r.t. a fixed list of inputs (forward mode)."""
    __slots__ = ("v", "d")

    def __init__(self, v, d):
        self.v, self.d = v, d


def _eval(node, env, memo):
    """env: id(symbol node) -> Dual.  Operations in the order the graph holds them."""
    k = id(node)
    if k in memo:
        return memo[k]
    op = node.op
    if op == "sym":
        r = env[k]
    elif op == "const":
        r = Dual(node.a, 0.0)
    elif op == "neg":
        a = _eval(node.a, env, memo)
        r = Dual(-a.v, -a.d)
    else:
        a, b = _eval(node.a, env, memo), _eval(node.b, env, memo)
        if op == "add":
            r = Dual(a.v + b.v, a.d + b.d)
        elif op == "sub":
            r = Dual(a.v - b.v, a.d - b.d)
        elif op == "mul":
            r = Dual(a.v * b.v, a.d * b.v + a.v * b.d)
        elif op == "div":
            if isinstance(b.d, float) and b.d == 0.0:        # division by a constant (disturb_f / CP.mass): d(a / c) = da / c exactly
                r = Dual(a.v / b.v, a.d / b.v)
            else:
                r = Dual(a.v / b.v, (a.d * b.v - a.v * b.d) / (b.v * b.v))
        elif op == "pow":
            e = int(b.v)
            r = Dual(a.v ** e, e * a.v ** (e - 1) * a.d)
        else:
            raise ValueError(op)
    memo[k] = r
    return r


def evaluate(vec, syms, duals):
    """vec: Vec of outputs; syms: list of symbol nodes; duals: list of Dual (same length).  -> values[n], jac[n, nvar]."""
    env = {id(s): d for s, d in zip(syms, duals)}
    memo = {}
    nvar = len(duals[0].d)
    val, jac = np.zeros(len(vec)), np.zeros((len(vec), nvar))
    for i, n in enumerate(vec.items):
        r = _eval(n, env, memo)
        val[i] = r.v
        jac[i] = r.d if isinstance(r.d, np.ndarray) else np.zeros(nvar)
    return val, jac


def symbols_of(vec):
    seen, out = set(), []

    def walk(n):
        if id(n) in seen:
            return
        seen.add(id(n))
        if n.op == "sym":
            out.append(n)
        elif n.op != "const":
            walk(n.a)
            if n.b is not None:
                walk(n.b)
    for n in vec.items:
        walk(n)
    return out


casadi = types.ModuleType("casadi")
casadi.SX, casadi.vertcat, casadi.vcat, casadi.Function, casadi.types = SX, vertcat, vcat, Function, types


# ================================================================== acados_template stand-in: attribute recorders
class _Rec:
    """Attribute container that remembers which attributes were assigned (in order)."""

    def __init__(self):
        object.__setattr__(self, "_set", [])

    def __setattr__(self, k, v):
        self._set.append(k)
        object.__setattr__(self, k, v)


class AcadosModel(_Rec):
    pass


class AcadosOcp(_Rec):
    def __init__(self):
        super().__init__()
        for k in ("dims", "cost", "constraints", "solver_options"):
            object.__setattr__(self, k, _Rec())


class AcadosOcpSolver:
    instances = []

    def __init__(self, ocp, json_file=None, build=True, **kw):
        self.ocp, self.json_file, self.build, self.kw = ocp, json_file, build, kw
        self.N = ocp.dims.N
        self.status = 0
        self.calls = []
        AcadosOcpSolver.instances.append(self)

    def set(self, stage, field, value):
        self.calls.append(("set", int(stage), str(field), np.array(value, dtype=np.float64).ravel().copy()))

    def get(self, stage, field):
        raise RuntimeError("the stand-in solver holds no solution")

    def solve_for_x0(self, x0):
        self.calls.append(("solve_for_x0", -1, "x0", np.array(x0, dtype=np.float64).ravel().copy()))
        return np.zeros(len(self.ocp.model.u))


class AcadosSimSolver:
    pass


acados_template = types.ModuleType("acados_template")
acados_template.AcadosOcp, acados_template.AcadosOcpSolver = AcadosOcp, AcadosOcpSolver
acados_template.AcadosSimSolver, acados_template.AcadosModel = AcadosSimSolver, AcadosModel

FIELD_CODE = {"x": 0, "u": 1, "yref": 2, "p": 3, "x0": 4}
OPTIONS_ACADOS_HAS_BUT_REFERENCE_LEAVES_UNSET = [
    "qp_solver_warm_start", "sim_method_num_stages", "sim_method_num_steps", "nlp_solver_max_iter", "qp_solver_iter_max",
    "qp_solver_tol_stat", "qp_solver_tol_eq", "qp_solver_tol_ineq", "qp_solver_tol_comp", "levenberg_marquardt", "regularize_method",
    "globalization", "nlp_solver_step_length", "cost_discretization",
]


def calls_to_arrays(calls):
    n = len(calls)
    stage, field, length, val = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros((n, 14))
    for i, (_, st, fl, v) in enumerate(calls):
        stage[i], field[i], length[i] = st, FIELD_CODE[fl], v.size
        val[i, :v.size] = v
    return stage, field, length, val


def main():
    sys.modules["casadi"], sys.modules["acados_template"] = casadi, acados_template
    sys.path.insert(0, REF)
    os.environ["ACADOS_SOURCE_DIR"] = "/nonexistent/acados"
    chdirs = []
    real_chdir = os.chdir
    os.chdir = lambda p: chdirs.append(str(p))            # nmpc_body_rate_ctl.py:29 -- recorded, not performed
    try:
        import nmpc_ctl.nmpc_body_rate_ctl as M1          # the reference's files, unmodified
        import ndp_nmpc_ctl.ndp_nmpc_body_rate_ctl as M2
        from params import nmpc_params as CP
        M1.safe_mkdir_recursive = lambda *a, **k: None     # would create ./acados_models
        M2.safe_mkdir_recursive = lambda *a, **k: None
        ctl1 = M1.NMPCBodyRateController()
        ctl2 = M2.NDPNMPCBodyRateController(is_build_acados=False)
    finally:
        os.chdir = real_chdir
    assert len(chdirs) == 2 and chdirs[0].endswith("nmpc_ctl") and chdirs[1].endswith("ndp_nmpc_ctl")

    g, meta = {}, {"generated_by": "tests/golden/make_ocp_golden.py (reference files run unmodified under casadi / acados_template stand-ins)",
                   "field_codes": FIELD_CODE, "chdir_targets": [os.path.relpath(c, REF) for c in chdirs]}
    rng = np.random.Generator(np.random.PCG64(20231213))
    K = 64
    # sample points: iterate-like x, u; reference xr, ur with a unit quaternion; p = [xr[6:10], f] as the reference's update sets it
    pt_x = np.concatenate([rng.normal(0, 2.0, (K, 3)), rng.normal(0, 1.5, (K, 3)), rng.normal(0, 1, (K, 4))], axis=1)
    pt_x[:, 6:10] /= np.linalg.norm(pt_x[:, 6:10], axis=1, keepdims=True)
    pt_x[:, 6:10] *= 1.0 + rng.normal(0, 0.02, (K, 1))                       # the OCP never renormalises
    pt_u = np.concatenate([rng.uniform(-6, 6, (K, 3)), rng.uniform(0, 27, (K, 1))], axis=1)
    pt_xr = np.concatenate([rng.normal(0, 2.0, (K, 3)), rng.normal(0, 1.5, (K, 3)), rng.normal(0, 1, (K, 4))], axis=1)
    pt_xr[:, 6:10] /= np.linalg.norm(pt_xr[:, 6:10], axis=1, keepdims=True)
    pt_ur = np.concatenate([rng.uniform(-3, 3, (K, 3)), rng.uniform(5, 15, (K, 1))], axis=1)
    pt_f = rng.normal(0, 2.0, (K, 3))
    pt_f[:4] = 0.0
    g.update(pt_x=pt_x, pt_u=pt_u, pt_xr=pt_xr, pt_ur=pt_ur, pt_f=pt_f)

    for pre, ctl in (("nmpc_", ctl1), ("ndp_", ctl2)):
        sol = ctl.solver
        ocp, model = sol.ocp, sol.ocp.model
        nx, nu, npar = model.x.size()[0], model.u.size()[0], model.p.size()[0]
        assert (nx, nu) == (10, 4) and npar == (4 if pre == "nmpc_" else 7)
        names = dict(x=[s.name for s in model.x], u=[s.name for s in model.u], p=[s.name for s in model.p])
        assert names["x"] == ["x", "y", "z", "vx", "vy", "vz", "qw", "qx", "qy", "qz"] and names["u"] == ["wx", "wy", "wz", "c"]
        # ---- OCP definition
        for k in ("W", "W_e", "yref", "yref_e"):
            g[pre + k] = np.array(getattr(ocp.cost, k), dtype=np.float64)
        for k in ("lbu", "ubu", "lbx", "ubx", "x0"):
            g[pre + k] = np.array(getattr(ocp.constraints, k), dtype=np.float64)
        for k in ("idxbu", "idxbx"):
            g[pre + k] = np.array(getattr(ocp.constraints, k), dtype=np.int64)
        g[pre + "N"] = np.int64(ocp.dims.N)
        g[pre + "np"] = np.int64(ocp.dims.np)
        g[pre + "tf"] = np.float64(ocp.solver_options.tf)
        g[pre + "parameter_values"] = np.array(ocp.parameter_values, dtype=np.float64)
        assert sol.N == CP.N_node
        so = ocp.solver_options
        meta[pre + "ocp"] = {
            "model_name": model.name, "json_file": sol.json_file, "build": bool(sol.build), "state_names": names["x"], "control_names": names["u"],
            "param_names": names["p"], "cost_type": ocp.cost.cost_type, "cost_type_e": ocp.cost.cost_type_e,
            "solver_options_set": {k: (getattr(so, k) if not isinstance(getattr(so, k), (np.integer, np.floating)) else float(getattr(so, k)))
                                   for k in dict.fromkeys(so._set)},
            "solver_options_not_set_by_the_reference": [k for k in OPTIONS_ACADOS_HAS_BUT_REFERENCE_LEAVES_UNSET if k not in so._set],
            "constraints_set": list(dict.fromkeys(ocp.constraints._set)), "cost_set": list(dict.fromkeys(ocp.cost._set)),
            "acados_include_path": ocp.acados_include_path, "acados_lib_path": ocp.acados_lib_path,
            "f_function_opts": {},
        }
        # ---- model expressions at the sample points
        xs, us, ps, xd = list(model.x), list(model.u), list(model.p), list(model.xdot)
        nvar = nx + nu + npar
        f_val, f_jac = np.zeros((K, nx)), np.zeros((K, nx, nvar))
        y_val, y_jac = np.zeros((K, nx + nu)), np.zeros((K, nx + nu, nvar))
        ye_val, ye_jac = np.zeros((K, nx)), np.zeros((K, nx, nvar))
        rk_xn, rk_A, rk_B = np.zeros((K, nx)), np.zeros((K, nx, nx)), np.zeros((K, nx, nu))
        h = float(so.tf) / int(ocp.dims.N)
        pvals = np.concatenate([pt_xr[:, 6:10], pt_f], axis=1)[:, :npar]
        # every symbol the expressions use is a state, a control or a parameter (allow_free in the NDP Function notwithstanding)
        used = {id(s) for s in symbols_of(model.f_expl_expr) + symbols_of(model.cost_y_expr) + symbols_of(model.cost_y_expr_e)}
        assert used <= {id(s) for s in xs + us + ps}

        def duals(xv, uv, pv):
            z = np.concatenate([xv, uv, pv])
            return [Dual(float(z[i]), np.eye(nvar)[i]) for i in range(nvar)]

        def f_at(xd_, ud_, pd_):
            """f_expl_expr on Dual inputs -> list of Dual (one evaluation of the reference's graph)."""
            env = {id(s): d for s, d in zip(xs + us + ps, xd_ + ud_ + pd_)}
            memo = {}
            return [_eval(n, env, memo) for n in model.f_expl_expr.items]

        for i in range(K):
            d = duals(pt_x[i], pt_u[i], pvals[i])
            f_val[i], f_jac[i] = evaluate(model.f_expl_expr, xs + us + ps, d)
            y_val[i], y_jac[i] = evaluate(model.cost_y_expr, xs + us + ps, d)
            ye_val[i], ye_jac[i] = evaluate(model.cost_y_expr_e, xs + us + ps, d)
            # f_impl_expr = xdot - f_expl (:185 / ndp :189)
            xdv = rng.normal(0, 1, nx)
            zero = np.zeros(nvar + nx)
            dd = [Dual(dv.v, zero) for dv in d] + [Dual(float(v), zero) for v in xdv]
            fi, _ = evaluate(model.f_impl_expr, xs + us + ps + xd, dd)
            assert np.array_equal(fi, xdv - f_val[i])
            # classical RK4, ONE step of h per shooting interval, through the reference's own f (dual numbers: exact derivative)
            xd_, ud_, pd_ = d[:nx], d[nx:nx + nu], d[nx + nu:]

            def axpy(a, k, x_):
                return [Dual(x_[j].v + a * k[j].v, x_[j].d + a * k[j].d) for j in range(nx)]
            k1 = f_at(xd_, ud_, pd_)
            k2 = f_at(axpy(0.5 * h, k1, xd_), ud_, pd_)
            k3 = f_at(axpy(0.5 * h, k2, xd_), ud_, pd_)
            k4 = f_at(axpy(h, k3, xd_), ud_, pd_)
            for j in range(nx):
                s_v = k1[j].v + 2.0 * k2[j].v + 2.0 * k3[j].v + k4[j].v
                s_d = k1[j].d + 2.0 * k2[j].d + 2.0 * k3[j].d + k4[j].d
                rk_xn[i, j] = xd_[j].v + h / 6.0 * s_v
                full = xd_[j].d + h / 6.0 * s_d
                rk_A[i, j], rk_B[i, j] = full[:nx], full[nx:nx + nu]
        g.update({pre + "f": f_val, pre + "dfdx": f_jac[:, :, :nx], pre + "dfdu": f_jac[:, :, nx:nx + nu], pre + "dfdp": f_jac[:, :, nx + nu:],
                  pre + "y": y_val, pre + "dydx": y_jac[:, :, :nx], pre + "dydu": y_jac[:, :, nx:nx + nu], pre + "dydp": y_jac[:, :, nx + nu:],
                  pre + "ye": ye_val, pre + "dyedx": ye_jac[:, :, :nx], pre + "dyedu": ye_jac[:, :, nx:nx + nu],
                  pre + "rk4_h": np.float64(h), pre + "rk4_xn": rk_xn, pre + "rk4_A": rk_A, pre + "rk4_B": rk_B})
        # ---- reset / update call sequences on random inputs
        N = sol.N
        xr = rng.normal(0, 1, (N + 1, nx))
        ur = rng.normal(0, 1, (N, nu))
        x0 = rng.normal(0, 1, nx)
        f32 = rng.normal(0, 1, (N + 1, 3)).astype(np.float32)              # DownwashNN.update returns float32 (downwash_nn.py:28)
        sol.calls.clear()
        ctl.reset(xr, ur)
        st, fl, ln, vl = calls_to_arrays(sol.calls)
        g.update({pre + "reset_stage": st, pre + "reset_field": fl, pre + "reset_len": ln, pre + "reset_val": vl})
        sol.calls.clear()
        u0 = ctl.update(x0, xr, ur) if pre == "nmpc_" else ctl.update(x0, xr, ur, f32)
        assert u0.shape == (nu,)
        st, fl, ln, vl = calls_to_arrays(sol.calls)
        g.update({pre + "update_stage": st, pre + "update_field": fl, pre + "update_len": ln, pre + "update_val": vl,
                  pre + "call_xr": xr, pre + "call_ur": ur, pre + "call_x0": x0, pre + "call_f": f32})
        # status != 0 -> the reference's exception text (:109-110)
        sol.status = 4
        try:
            ctl.update(x0, xr, ur) if pre == "nmpc_" else ctl.update(x0, xr, ur, f32)
            raise AssertionError("no exception on status 4")
        except Exception as e:                                       # noqa: BLE001
            meta[pre + "ocp"]["status_exception_text"] = str(e)
        sol.status = 0
        if pre == "ndp_":
            # ---- one linearisation case at an iterate away from the reference (what the device's LDS image must hold)
            synth_rng = np.random.Generator(np.random.PCG64(77))
            t = np.arange(N + 1) * h
            w = 0.9
            p_r = np.stack([2 * np.sin(w * t), np.sin(2 * w * t), 1 + 0.3 * np.sin(w * t)], axis=1)
            v_r = np.stack([2 * w * np.cos(w * t), 2 * w * np.cos(2 * w * t), 0.3 * w * np.cos(w * t)], axis=1)
            q_r = synth_rng.normal(0, 0.15, (N + 1, 4)) + np.array([1.0, 0, 0, 0])
            q_r /= np.linalg.norm(q_r, axis=1, keepdims=True)
            lin_xr = np.concatenate([p_r, v_r, q_r], axis=1)
            lin_ur = np.concatenate([synth_rng.normal(0, 0.3, (N, 3)), 9.81 + synth_rng.normal(0, 0.5, (N, 1))], axis=1)
            lin_X = lin_xr + synth_rng.normal(0, 0.05, (N + 1, nx))
            lin_U = lin_ur + synth_rng.normal(0, 0.2, (N, nu))
            lin_x0 = lin_xr[0] + synth_rng.normal(0, 0.1, nx)
            lin_f = synth_rng.normal(0, 1.5, (N + 1, 3)).astype(np.float32)
            for tag, c, use_f in (("lin_nmpc_", ctl1, False), ("lin_ndp_", ctl2, True)):
                m = c.solver.ocp.model
                xs_, us_, ps_ = list(m.x), list(m.u), list(m.p)
                nv = nx + nu + len(ps_)
                A, Bm, bb = np.zeros((N, nx, nx)), np.zeros((N, nx, nu)), np.zeros((N, nx))
                Jy, res = np.zeros((N + 1, nx + nu, nx + nu)), np.zeros((N + 1, nx + nu))
                for k in range(N + 1):
                    pk = np.concatenate([lin_xr[k, 6:10], lin_f[k].astype(np.float64)])[:len(ps_)]
                    uk = lin_U[k] if k < N else np.zeros(nu)
                    z = np.concatenate([lin_X[k], uk, pk])
                    d = [Dual(float(z[i]), np.eye(nv)[i]) for i in range(nv)]
                    if k < N:
                        yv, yj = evaluate(m.cost_y_expr, xs_ + us_ + ps_, d)
                        Jy[k] = yj[:, :nx + nu]
                        res[k] = yv - np.concatenate([lin_xr[k], lin_ur[k]])
                        env_syms = xs_ + us_ + ps_

                        def f_at2(xd_):
                            env = {id(s): dd_ for s, dd_ in zip(env_syms, xd_ + d[nx:])}
                            memo = {}
                            return [_eval(n, env, memo) for n in m.f_expl_expr.items]

                        def axpy2(a, kk, x_):
                            return [Dual(x_[j].v + a * kk[j].v, x_[j].d + a * kk[j].d) for j in range(nx)]
                        xd0 = d[:nx]
                        k1 = f_at2(xd0)
                        k2 = f_at2(axpy2(0.5 * h, k1, xd0))
                        k3 = f_at2(axpy2(0.5 * h, k2, xd0))
                        k4 = f_at2(axpy2(h, k3, xd0))
                        for j in range(nx):
                            s_v = k1[j].v + 2.0 * k2[j].v + 2.0 * k3[j].v + k4[j].v
                            s_d = k1[j].d + 2.0 * k2[j].d + 2.0 * k3[j].d + k4[j].d
                            full = xd0[j].d + h / 6.0 * s_d
                            A[k, j], Bm[k, j] = full[:nx], full[nx:nx + nu]
                            bb[k, j] = xd0[j].v + h / 6.0 * s_v - lin_X[k + 1, j]
                    else:
                        yv, yj = evaluate(m.cost_y_expr_e, xs_ + us_ + ps_, d)
                        Jy[k, :nx, :nx] = yj[:, :nx]
                        res[k, :nx] = yv - lin_xr[k]
                g.update({tag + "A": A, tag + "B": Bm, tag + "b": bb, tag + "Jy": Jy, tag + "res": res})
            g.update(lin_x0=lin_x0, lin_xr=lin_xr, lin_ur=lin_ur, lin_X=lin_X, lin_U=lin_U, lin_f=lin_f)
    meta["nmpc_ocp"]["f_function_opts"] = {}
    np.savez_compressed(os.path.join(HERE, "ocp_golden.npz"), **g)
    with open(os.path.join(HERE, "ocp_golden.json"), "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True, default=str)
    print("wrote ocp_golden.npz (%d arrays) and ocp_golden.json" % len(g))
    for k in ("nmpc_ocp", "ndp_ocp"):
        print(k, json.dumps(meta[k]["solver_options_set"]))


if __name__ == "__main__":
    main()
