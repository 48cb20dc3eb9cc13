"""Fixture for SURVEY 8f-1 (reference window generation), generated with the reference's own PolymOptimizer
(/root/reference/ndp_nmpc/scripts/pt_pub/polym_optimizer.py, pure numpy).  Run in the build container only.

What comes from the imported reference class:
  * coeff_*     PolymOptimizer(MinMethod.SNAP / ACCEL).get_coeff(waypoints)        polym_optimizer.py:38-102
  * pvaj, yaw   positions / velocities / accelerations / jerks / yaw / yaw rate at sample times, evaluated with the
                reference's get_poly_params (polym_optimizer.py:104-139) through the two-line formula of
                _get_output_value (pt_pub/base_pt_publisher.py:136-143), which itself cannot be imported (rospy).
What is NOT from the reference: the flatness map (pt_publisher.py:188-248) needs rospy/tf_conversions, absent here --
its outputs are pinned in tests/test_ref_window_row.py by physical identities instead.

Output (committed): tests/golden/ref_golden.npz, per case c in 0..3:
  wpts_c[V,4,M+1] (x,y,z,yaw waypoints)  tseg_c[V,M]  coeff_c[V,M,28] (x8 y8 z8 yaw4 per segment)
  tq_c[V,S] sample times                 pvaj_c[V,S,12]  yaw_c[V,S,2]
"""
import os
import sys

sys.dont_write_bytecode = True    # importing from /root/reference must not leave __pycache__ there (the tree is read-only by contract)

import numpy as np

sys.path.insert(0, "/root/reference/ndp_nmpc/scripts/pt_pub")
from polym_optimizer import MinMethod, PolymOptimizer  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.Generator(np.random.PCG64(20231213 + 11))
out = {}
for case, M in enumerate((1, 2, 3, 6)):
    V, S = 5, 24
    wpts = np.zeros((V, 4, M + 1))
    wpts[:, 0:3] = np.cumsum(rng.uniform(-1.5, 1.5, (V, 3, M + 1)), axis=2)
    wpts[:, 2] = 1.0 + 0.3 * rng.uniform(-1, 1, (V, M + 1))
    wpts[:, 3] = np.cumsum(rng.uniform(-0.6, 0.6, (V, M + 1)), axis=1)
    tseg = rng.uniform(2.0, 5.0, (V, M))
    coeff = np.zeros((V, M, 28))
    tq = np.zeros((V, S))
    pvaj = np.zeros((V, S, 12))
    yaw = np.zeros((V, S, 2))
    for v in range(V):
        ox, oyaw = PolymOptimizer(MinMethod.SNAP), PolymOptimizer(MinMethod.ACCEL)
        cs = [ox.get_coeff(wpts[v, a]) for a in range(3)] + [oyaw.get_coeff(wpts[v, 3])]
        for a in range(3):
            coeff[v, :, 8 * a:8 * a + 8] = cs[a].reshape(M, 8)
        coeff[v, :, 24:28] = cs[3].reshape(M, 4)
        tcum = np.concatenate([[0.0], np.cumsum(tseg[v])])
        tq[v] = np.sort(np.concatenate([rng.uniform(0, tcum[-1], S - 3), [0.0], tcum[-1] * np.array([0.5, 0.999999])]))
        for s in range(S):
            t = tq[v, s]
            idx = int(np.argwhere(tcum > t)[0].item()) - 1                     # base_pt_publisher.py:100
            ts = (t - tcum[idx]) / tseg[v, idx]                                # :102-103
            for d in range(4):
                for a in range(3):
                    c = cs[a][idx * 8:(idx + 1) * 8, :]
                    pvaj[v, s, 3 * d + a] = (ox.get_poly_params(d, ts) / np.power(tseg[v, idx], d) @ c).item()
            c = cs[3][idx * 4:(idx + 1) * 4, :]
            for d in range(2):
                yaw[v, s, d] = (oyaw.get_poly_params(d, ts) / np.power(tseg[v, idx], d) @ c).item()
    out.update({f"wpts_{case}": wpts, f"tseg_{case}": tseg, f"coeff_{case}": coeff, f"tq_{case}": tq,
                f"pvaj_{case}": pvaj, f"yaw_{case}": yaw})
np.savez_compressed(os.path.join(HERE, "ref_golden.npz"), **out)
print("ok", out["coeff_1"][0, 0, :4], out["pvaj_3"][0, 5, :3])
