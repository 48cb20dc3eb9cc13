#!/usr/bin/env python3
"""What a mismatch against real acados would MEAN: the u0 sensitivity of each [acados-knowledge] assumption of SURVEY A.4 on the
committed cross-check inputs (tests/golden/acados_inputs.npz) -- the CPU restatement with the assumption flipped against the
restatement as pinned.  CPU only (oracle).  -> profiles/r05_acados_assumption_sensitivity.txt, DESIGN section 2.
    python3 scripts/acados_sensitivity.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

VARIANTS = (("terminal cost also scaled by the interval (W_e dt)", O.VAR_TERMINAL_TIMES_DT),
            ("velocity box also on the terminal state (lbx_e / ubx_e)", O.VAR_BOUNDS_STAGE_N),
            ("ERK with sim_method_num_steps = 2", O.VAR_ERK_2_STEPS),
            ("stage costs NOT scaled by the interval", O.VAR_NO_DT_SCALING))


def run(G, case, bits):
    O.set_variant(bits)
    try:
        x0, xr, ur, f = G[case + "_x0"], G[case + "_xr"], G[case + "_ur"], (G[case + "_f"] if case + "_f" in G else None)
        cfg = O.default_cfg(use_fd=f is not None)
        X, U = xr[0].copy(), ur[0].copy()
        out, sts = [], []
        for t in range(x0.shape[0]):
            u, st, it = O.step_batch(cfg, x0[t], xr[t], ur[t], None if f is None else f[t], X, U)
            out.append(u)
            sts.append(st)
        return np.array(out), np.array(sts)
    finally:
        O.set_variant(0)


def main():
    O.build()
    G = np.load(os.path.join(ROOT, "tests", "golden", "acados_inputs.npz"))
    cases = sorted({k.rsplit("_", 1)[0] for k in G.files if k.endswith("_x0")})
    print("u0 sensitivity of the [acados-knowledge] assumptions (SURVEY A.4): max / median over instances and ticks of |u0(variant) - u0(pinned)| / max(1, |u0|)")
    print("a cross-check against real acados (scripts/acados_crosscheck.py) that misses the 1e-5 bar by one of these sizes names its cause\n")
    print("%-58s " % "assumption flipped" + " ".join("%-24s" % c for c in cases))
    base = {c: run(G, c, 0) for c in cases}
    for name, bits in VARIANTS:
        cells = []
        for c in cases:
            u, st = run(G, c, bits)
            ok = (st == 0) & (base[c][1] == 0)
            e = (np.abs(u - base[c][0]) / np.maximum(1.0, np.abs(base[c][0]))).max(axis=2)[ok]
            cells.append("%.1e / %.1e" % (e.max(), np.median(e)))
        print("%-58s " % name + " ".join("%-24s" % x for x in cells))
    print("\nfor scale: two correct interior-point codes that stop one iteration apart differ by up to ~4e-6 on near-active instances; the bar is 1e-5")
    for c in cases:
        print(f"  {c}: {G[c + '_x0'].shape[1]} instances x {G[c + '_x0'].shape[0]} ticks, pinned restatement: "
              f"{int((base[c][1] != 0).sum())} solves with status != 0")


if __name__ == "__main__":
    main()
