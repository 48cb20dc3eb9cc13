#!/usr/bin/env python3
"""Pins the one thing this repository cannot pin itself: u0 after a REAL acados / HPIPM solve ("parity unpinned").

Runs ONLY on a machine that has `acados_template` + `casadi` (and an acados build, ACADOS_SOURCE_DIR set) and a checkout of the
reference, given by --reference; never on the GPU box, never in this repository's CI.  It copies no reference file: the reference's
two controller classes are imported from where they lie and driven through their own API --

    ctl = NMPCBodyRateController(is_build_acados=True)          # nmpc_ctl/nmpc_body_rate_ctl.py:21-84
    ctl.reset(xr, ur); u0 = ctl.update(x0, xr, ur)              # :86-112
    ctl = NDPNMPCBodyRateController(True); ctl.update(x0, xr, ur, f)   # ndp_nmpc_ctl/ndp_nmpc_body_rate_ctl.py:91-112

-- on the committed inputs tests/golden/acados_inputs.npz (tests/golden/make_acados_inputs.py: 576 instances x 3 ticks, nominal,
perturbed with active input bounds, with the downwash force).  It writes tests/golden/acados_golden.npz:
    <case>_u0 [3, B, 4], <case>_X [3, B, 21, 10], <case>_U [3, B, 20, 4], <case>_status [3, B], <case>_qp_iter [3, B]
    (+ acados / casadi versions as strings).
Commit that file: tests/test_acados_golden.py (skipped while it is absent) then holds the CPU oracle AND -- with -m gpu -- the
device to 1e-5 against it, and `parity` is pinned.  If the check misses the bar, scripts/acados_sensitivity.py's table (DESIGN
section 2) says which [acados-knowledge] assumption a miss of that size points at.

    python3 scripts/acados_crosscheck.py --reference /path/to/ndp_nmpc_qd [--cases nmpc_nominal,ndp_nominal] [--limit 32]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", required=True, help="checkout of Li-Jinjie/ndp_nmpc_qd (the directory that holds ndp_nmpc/)")
    ap.add_argument("--cases", default="", help="comma-separated subset of the cases in acados_inputs.npz")
    ap.add_argument("--limit", type=int, default=0, help="first N instances of every case only (a quick look)")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "acados_golden.npz"))
    a = ap.parse_args()
    try:
        import acados_template
        import casadi
    except ImportError as e:
        raise SystemExit(f"acados_crosscheck.py needs acados_template and casadi ({e}): run it where the reference itself runs")
    scripts = os.path.join(os.path.abspath(a.reference), "ndp_nmpc", "scripts")
    if not os.path.isdir(scripts):
        raise SystemExit(f"{scripts} does not exist: --reference must be a checkout of the reference repository")
    sys.path.insert(0, scripts)
    cwd = os.getcwd()
    from nmpc_ctl import NMPCBodyRateController                  # (the constructors chdir into their own directory, nmpc_body_rate_ctl.py:29)
    from ndp_nmpc_ctl import NDPNMPCBodyRateController
    G = np.load(os.path.join(ROOT, "tests", "golden", "acados_inputs.npz"))
    cases = sorted({k.rsplit("_", 1)[0] for k in G.files if k.endswith("_x0")})
    if a.cases:
        cases = [c for c in cases if c in a.cases.split(",")]
    ctl = {False: NMPCBodyRateController(True), True: NDPNMPCBodyRateController(True)}
    os.chdir(cwd)
    out = {"acados_template_version": np.array(str(getattr(acados_template, "__version__", "unknown"))),
           "casadi_version": np.array(str(casadi.__version__))}
    for c in cases:
        x0, xr, ur = G[c + "_x0"], G[c + "_xr"], G[c + "_ur"]
        f = G[c + "_f"] if c + "_f" in G.files else None
        T, B = x0.shape[0], x0.shape[1] if not a.limit else min(a.limit, x0.shape[1])
        N = xr.shape[2] - 1
        u0, X, U = np.zeros((T, B, 4)), np.zeros((T, B, N + 1, 10)), np.zeros((T, B, N, 4))
        st, qi = np.zeros((T, B), dtype=np.int32), np.full((T, B), -1, dtype=np.int32)
        k = ctl[f is not None]
        for b in range(B):
            k.reset(xr[0, b], ur[0, b])                            # one solver object, re-seeded per instance: the iterate persists over the 3 ticks
            for t in range(T):
                try:
                    u0[t, b] = k.update(x0[t, b], xr[t, b], ur[t, b], f[t, b]) if f is not None else k.update(x0[t, b], xr[t, b], ur[t, b])
                except Exception:                                  # "acados acados_ocp_solver returned status {}. Exiting." (:109-110)
                    u0[t, b] = np.nan
                st[t, b] = int(k.solver.status)
                try:
                    qi[t, b] = int(np.sum(k.solver.get_stats("qp_iter")))
                except Exception:
                    pass
                for i in range(N + 1):
                    X[t, b, i] = k.solver.get(i, "x")
                for i in range(N):
                    U[t, b, i] = k.solver.get(i, "u")
            if b % 32 == 0:
                print(f"{c}: instance {b} / {B}", flush=True)
        out.update({c + "_u0": u0, c + "_X": X, c + "_U": U, c + "_status": st, c + "_qp_iter": qi})
    np.savez_compressed(a.out, **out)
    print("wrote", a.out, "-- now: python -m pytest tests/test_acados_golden.py   (and -m gpu on an MI355X)")


if __name__ == "__main__":
    main()
