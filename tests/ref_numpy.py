"""Independent numpy restatements used ONLY to pin the C oracle (tests).

Everything here is written straight from the reference's symbolic expressions
(nmpc_body_rate_ctl.py:147-180) and from textbook definitions (dense KKT
systems, finite differences), sharing no code with oracle/ndp_oracle.c.
"""
import numpy as np

NX, NU = 10, 4
G = 9.81
MASS = 1.4844


def f_dyn(x, u, fd=None):
    """nmpc_body_rate_ctl.py:147-158 (+ ndp_nmpc_body_rate_ctl.py:155-157)."""
    vx, vy, vz, qw, qx, qy, qz = x[3:10]
    wx, wy, wz, c = u
    ds = np.array([
        vx, vy, vz,
        2 * (qx * qz + qw * qy) * c,
        2 * (qy * qz - qw * qx) * c,
        (1 - 2 * qx ** 2 - 2 * qy ** 2) * c - G,
        (-wx * qx - wy * qy - wz * qz) * 0.5,
        (wx * qw + wz * qy - wy * qz) * 0.5,
        (wy * qw - wz * qx + wx * qz) * 0.5,
        (wz * qw + wy * qx - wx * qy) * 0.5,
    ])
    if fd is not None:
        ds[3:6] += np.asarray(fd) / MASS
    return ds


def rk4(x, u, fd=None, h=0.1):
    k1 = f_dyn(x, u, fd)
    k2 = f_dyn(x + 0.5 * h * k1, u, fd)
    k3 = f_dyn(x + 0.5 * h * k2, u, fd)
    k4 = f_dyn(x + h * k3, u, fd)
    return x + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)


def cost_y(x, u, qr):
    """cost_y_expr, nmpc_body_rate_ctl.py:164-180,194."""
    qw, qx, qy, qz = x[6:10]
    qwr, qxr, qyr, qzr = qr
    qe_x = qwr * qx - qw * qxr + qyr * qz - qy * qzr
    qe_y = qwr * qy - qw * qyr - qxr * qz + qx * qzr
    qe_z = qxr * qy - qx * qyr + qwr * qz - qw * qzr
    sy = np.array([x[0], x[1], x[2], x[3], x[4], x[5], qwr, qe_x + qxr, qe_y + qyr, qe_z + qzr])
    return sy if u is None else np.concatenate([sy, u])


def fd_jac(fun, z, eps=1e-6):
    z = np.asarray(z, dtype=np.float64)
    f0 = fun(z)
    J = np.zeros((f0.size, z.size))
    for i in range(z.size):
        d = np.zeros_like(z)
        d[i] = eps
        J[:, i] = (fun(z + d) - fun(z - d)) / (2 * eps)
    return J


def gn_blocks(x, u, xr, ur, W, scale):
    """Gauss-Newton H = s J'WJ, g = s J'W(y - yref) with J by finite differences."""
    qr = xr[6:10]
    if u is None:
        fun = lambda z: cost_y(z, None, qr)  # noqa: E731
        z0, yref = x, xr
    else:
        fun = lambda z: cost_y(z[:NX], z[NX:], qr)  # noqa: E731
        z0, yref = np.concatenate([x, u]), np.concatenate([xr, ur])
    J = fd_jac(fun, z0)
    res = fun(z0) - yref
    return scale * J.T @ W @ J, scale * J.T @ W @ res


def kkt_solve(qp, fixed=None):
    """Dense KKT solve of the equality-constrained QP (+ optional variables pinned to values).

    Variable order: dx_0..dx_N (10 each), du_0..du_{N-1} (4 each).
    fixed: list of (var_index, value).  Returns dx, du, multipliers of the pinned variables
    (sign: lam > 0 means the pin pushes the variable up, i.e. it is a LOWER bound multiplier).
    """
    A, B, b, Q, q, Rd, r, dx0 = (qp[k] for k in ("A", "B", "b", "Q", "q", "Rd", "r", "dx0"))
    N = A.shape[0]
    nz = (N + 1) * NX + N * NU
    xo = lambda k: k * NX  # noqa: E731
    uo = lambda k: (N + 1) * NX + k * NU  # noqa: E731
    H = np.zeros((nz, nz))
    g = np.zeros(nz)
    for k in range(N + 1):
        H[xo(k):xo(k) + NX, xo(k):xo(k) + NX] = Q[k]
        g[xo(k):xo(k) + NX] = q[k]
    for k in range(N):
        H[uo(k):uo(k) + NU, uo(k):uo(k) + NU] = np.diag(Rd[k])
        g[uo(k):uo(k) + NU] = r[k]
    fixed = fixed or []
    ne = (N + 1) * NX + len(fixed)
    E = np.zeros((ne, nz))
    e = np.zeros(ne)
    E[0:NX, xo(0):xo(0) + NX] = np.eye(NX)
    e[0:NX] = dx0
    for k in range(N):
        rows = slice((k + 1) * NX, (k + 2) * NX)
        E[rows, xo(k + 1):xo(k + 1) + NX] = np.eye(NX)
        E[rows, xo(k):xo(k) + NX] = -A[k]
        E[rows, uo(k):uo(k) + NU] = -B[k]
        e[rows] = b[k]
    for i, (vi, val) in enumerate(fixed):
        E[(N + 1) * NX + i, vi] = 1.0
        e[(N + 1) * NX + i] = val
    KKT = np.block([[H, E.T], [E, np.zeros((ne, ne))]])
    sol = np.linalg.solve(KKT, np.concatenate([-g, e]))
    z = sol[:nz]
    mult = -sol[nz + (N + 1) * NX:]  # H z + g + E' nu = 0 -> pin multiplier = -nu
    dx = z[:(N + 1) * NX].reshape(N + 1, NX)
    du = z[(N + 1) * NX:].reshape(N, NU)
    return dx, du, mult


def bound_table(qp):
    """[(var_index, lo, hi)] of the QP's box constraints (du all stages; dv stages 1..N-1)."""
    N = qp["A"].shape[0]
    out = []
    for k in range(N):
        for i in range(NU):
            out.append(((N + 1) * NX + k * NU + i, qp["lu"][k, i], qp["uu"][k, i]))
    for k in range(1, N):
        for i in range(3):
            out.append((k * NX + 3 + i, qp["lv"][k, i], qp["uv"][k, i]))
    return out


def active_set_solve(qp, max_iter=200):
    """Textbook primal-feasible-agnostic active-set iteration on the dense KKT system:
    pin violated variables to the violated bound, release pins whose multiplier has the
    wrong sign, until the KKT conditions of the box-constrained QP hold exactly."""
    table = bound_table(qp)
    active = {}  # var -> ('lo'|'hi')
    for _ in range(max_iter):
        fixed = [(v, (lo if active[v] == "lo" else hi)) for (v, lo, hi) in table if v in active]
        dx, du, mult = kkt_solve(qp, fixed)
        z = np.concatenate([dx.ravel(), du.ravel()])
        changed = False
        # release wrong-signed pins (most negative first)
        worst, worst_v = -1e-10, None
        i = 0
        for (v, lo, hi) in table:
            if v in active:
                lam = mult[i] if active[v] == "lo" else -mult[i]
                if lam < worst:
                    worst, worst_v = lam, v
                i += 1
        if worst_v is not None:
            del active[worst_v]
            changed = True
        else:
            viol, vv, side = 1e-10, None, None
            for (v, lo, hi) in table:
                if v in active:
                    continue
                if lo - z[v] > viol:
                    viol, vv, side = lo - z[v], v, "lo"
                if z[v] - hi > viol:
                    viol, vv, side = z[v] - hi, v, "hi"
            if vv is not None:
                active[vv] = side
                changed = True
        if not changed:
            return dx, du, active
    raise RuntimeError("active set did not converge")


def pdas_solve(qp, max_iter=60):
    """Primal-dual active-set iteration on the dense KKT system: ALL violated bounds are pinned and ALL wrong-signed pins are
    released at once (Hintermueller-Ito-Kunisch); a handful of dense solves instead of one per constraint, which is what makes
    hundreds of N = 20 and dozens of N = 40 problems affordable as an independent check.  Terminates at a point that satisfies
    the KKT conditions of the box-constrained QP exactly (same stopping rule as active_set_solve); if the sets start to cycle
    it hands over to the one-at-a-time method."""
    table = bound_table(qp)
    active, seen = {}, set()
    for _ in range(max_iter):
        key = frozenset(active.items())
        if key in seen:
            return active_set_solve(qp)
        seen.add(key)
        fixed = [(v, (lo if active[v] == "lo" else hi)) for (v, lo, hi) in table if v in active]
        try:
            dx, du, mult = kkt_solve(qp, fixed)
        except np.linalg.LinAlgError:           # pinning everything that is violated at once over-determined a stage
            return active_set_solve(qp)
        z = np.concatenate([dx.ravel(), du.ravel()])
        new, i = {}, 0
        for (v, lo, hi) in table:
            if v in active:
                lam = mult[i] if active[v] == "lo" else -mult[i]
                if lam >= -1e-10:
                    new[v] = active[v]
                i += 1
            elif lo - z[v] > 1e-10:
                new[v] = "lo"
            elif z[v] - hi > 1e-10:
                new[v] = "hi"
        if new == active:
            return dx, du, active
        active = new
    return active_set_solve(qp)


def separation(qp, dx, du, active):
    """How clearly the exact solution (dx, du, active set) decides each bound: min over the inactive bounds of the distance to
    the bound and over the active ones of the multiplier's magnitude.  An interior-point answer stopped at complementarity mu
    is off by ~ mu / separation (times a modest constant): a bound 0.01 from active leaves a multiplier mu / 0.01 on it."""
    table = bound_table(qp)
    fixed = [(v, (lo if active[v] == "lo" else hi)) for (v, lo, hi) in table if v in active]
    _, _, mult = kkt_solve(qp, fixed)
    z = np.concatenate([dx.ravel(), du.ravel()])
    margin = min(min(z[v] - lo, hi - z[v]) for (v, lo, hi) in table if v not in active)
    return min(margin, np.abs(mult).min() if len(mult) else np.inf)


# ------------------------------------------------------------------------------------------- interior point with a pluggable Riccati
# The oracle's interior-point loop (absolute form, Mehrotra predictor-corrector, oracle/ndp_oracle.c: qp_solve_ws) restated in numpy
# with the Newton-system solver as a parameter, to compare formulations of the Riccati recursion on problems with active STATE
# bounds (tests/test_oracle_pins.py): the explicit recursion, a square-root form (QR factor of P propagated), and either of them
# followed by refinement solves.
def _rollout(qp, K, kf):
    A, B, b, dx0 = qp["A"], qp["B"], qp["b"], qp["dx0"]
    N = A.shape[0]
    dx, du = np.zeros((N + 1, NX)), np.zeros((N, NU))
    dx[0] = dx0
    for k in range(N):
        du[k] = kf[k] + K[k] @ dx[k]
        dx[k + 1] = A[k] @ dx[k] + B[k] @ du[k] + b[k]
    return dx, du


def riccati_explicit(qp, Qe, qe, Re, re):
    """P = Hxx - Hxu Lam^-1 Hux, the textbook recursion (what oracle and device run)."""
    A, B, b = qp["A"], qp["B"], qp["b"]
    N = A.shape[0]
    P, p = Qe[N].copy(), qe[N].copy()
    K, kf = [None] * N, [None] * N
    for k in range(N - 1, -1, -1):
        Pb, PA, PB = p + P @ b[k], P @ A[k], P @ B[k]
        Lam, Hux, hu = np.diag(Re[k]) + B[k].T @ PB, B[k].T @ PA, re[k] + B[k].T @ Pb
        Hxx, hx = Qe[k] + A[k].T @ PA, qe[k] + A[k].T @ Pb
        L = np.linalg.cholesky(Lam)
        K[k] = -np.linalg.solve(L.T, np.linalg.solve(L, Hux))
        kf[k] = -np.linalg.solve(L.T, np.linalg.solve(L, hu))
        P = Hxx + Hux.T @ K[k]
        P, p = 0.5 * (P + P.T), hx + Hux.T @ kf[k]
    return _rollout(qp, K, kf)


def _psd_factor(Q):
    w, V = np.linalg.eigh(Q)
    return (V * np.sqrt(np.clip(w, 0.0, None))).T          # F with F'F = Q


def riccati_sqrt(qp, Qe, qe, Re, re):
    """Square-root form: an upper-triangular S with P = S'S is propagated through the QR factorisation of
    [S+ [B A]; sqrt(R) 0; 0 F_Q] -- P is never formed as a difference."""
    A, B, b = qp["A"], qp["B"], qp["b"]
    N = A.shape[0]
    S, p = _psd_factor(Qe[N]), qe[N].copy()
    K, kf = [None] * N, [None] * N
    for k in range(N - 1, -1, -1):
        Z = np.vstack([np.hstack([S @ B[k], S @ A[k]]), np.hstack([np.diag(np.sqrt(Re[k])), np.zeros((NU, NX))]),
                       np.hstack([np.zeros((NX, NU)), _psd_factor(Qe[k])])])
        Rf = np.linalg.qr(Z, mode="r")
        Ruu, Rux, Rxx = Rf[:NU, :NU], Rf[:NU, NU:], Rf[NU:NU + NX, NU:]
        Pb = p + S.T @ (S @ b[k])
        hu, hx = re[k] + B[k].T @ Pb, qe[k] + A[k].T @ Pb
        K[k] = -np.linalg.solve(Ruu, Rux)
        y = np.linalg.solve(Ruu.T, hu)
        kf[k] = -np.linalg.solve(Ruu, y)
        p, S = hx - Rux.T @ y, Rxx
    return _rollout(qp, K, kf)


def refined(base, n):
    """base followed by n refinement solves: gradient g + H z, zero defects, zero initial state (oracle: refine_solution)."""
    def ric(qp, Qe, qe, Re, re):
        dx, du = base(qp, Qe, qe, Re, re)
        qp0 = dict(qp)
        qp0["b"], qp0["dx0"] = np.zeros_like(qp["b"]), np.zeros(NX)
        for _ in range(n):
            ddx, ddu = base(qp0, Qe, np.einsum("kij,kj->ki", Qe, dx) + qe, Re, Re * du + re)
            dx, du = dx + ddx, du + ddu
        return dx, du
    return ric


def ipm_absolute(qp, tol, ric, mu0=10.0, thr0=0.1, tau=0.995, mu_floor=0.1, iter_max=60):
    N = qp["A"].shape[0]
    st = np.array([k for k in range(N) for _ in range(NU)] + [k for k in range(1, N) for _ in range(3)])
    ix = np.array([i for _ in range(N) for i in range(NU)] + [NU + i for _ in range(1, N) for i in range(3)])
    lo = np.concatenate([qp["lu"].ravel(), qp["lv"][1:N].ravel()])
    hi = np.concatenate([qp["uu"].ravel(), qp["uv"][1:N].ravel()])
    m, isu = len(st), ix < NU

    def val(dx, du):
        return np.where(isu, du[st, np.minimum(ix, NU - 1)], dx[st, 3 + np.maximum(ix - NU, 0)])
    tl, tu = np.maximum(-lo, thr0), np.maximum(hi, thr0)
    ll, lu = mu0 / tl, mu0 / tu
    mu = (ll * tl + lu * tu).sum() / (2 * m)
    norm0 = max(1.0, ll.max(), lu.max(), np.abs(-lo - tl).max(), np.abs(hi - tu).max(), np.abs(qp["q"]).max(), np.abs(qp["r"]).max(),
                np.abs(qp["b"]).max(), np.abs(qp["dx0"]).max())
    rho, it = 1.0, 0
    zx, zu = np.zeros((N + 1, NX)), np.zeros((N, NU))
    dll = dlu = dtl = dtu = np.zeros(m)
    while not (mu <= tol and rho * norm0 <= tol) and it < iter_max:
        it += 1
        sig = 0.0
        for ps in range(2):
            sl = sig - dll * dtl if ps else np.zeros(m)
            su = sig - dlu * dtu if ps else np.zeros(m)
            gl, gu = ll / tl, lu / tu
            Gam, gam = gl + gu, -sl / tl - ll - gl * lo + su / tu + lu - gu * hi
            Qe, qe, Re, re = qp["Q"].copy(), qp["q"].copy(), qp["Rd"].copy(), qp["r"].copy()
            for i in range(m):
                if isu[i]:
                    Re[st[i], ix[i]] += Gam[i]
                    re[st[i], ix[i]] += gam[i]
                else:
                    j = 3 + ix[i] - NU
                    Qe[st[i], j, j] += Gam[i]
                    qe[st[i], j] += gam[i]
            nx_, nu_ = ric(qp, Qe, qe, Re, re)
            zn = val(nx_, nu_)
            dtl, dtu = zn - lo - tl, hi - zn - tu
            dll, dlu = sl / tl - ll - ll / tl * dtl, su / tu - lu - lu / tu * dtu
            al = 1.0
            for d, v in ((dtl, tl), (dtu, tu), (dll, ll), (dlu, lu)):
                neg = d < 0
                if neg.any():
                    al = min(al, (-v[neg] / d[neg]).min())
            if ps == 0:
                mua = ((ll + al * dll) * (tl + al * dtl) + (lu + al * dlu) * (tu + al * dtu)).sum() / (2 * m)
                sig = max((mua / mu) ** 3 * mu, mu_floor * tol)
            else:
                if al < 1:
                    al *= tau
                zx, zu = zx + al * (nx_ - zx), zu + al * (nu_ - zu)
                tl, tu, ll, lu = tl + al * dtl, tu + al * dtu, ll + al * dll, lu + al * dlu
                mu, rho = (ll * tl + lu * tu).sum() / (2 * m), rho * (1 - al)
    return zx, zu, it
