"""Fixture for SURVEY 8f-2 (follower reference relay), generated with the reference's own AlphaFilter
(/root/reference/ndp_nmpc/scripts/hv_throttle_est/alpha_filter.py, pure python).  Run in the build container only.

FollowerNode (nmpc_follower_node.py:44-77): every formation_ref message passes through three AlphaFilters
(alpha = 0.8, y0 = first message, :32,44-56); every leader PredXU message becomes the follower's reference:
x[i][0:3] += filtered offset, u copied (:58-74).  The relay arithmetic (an add per position component) is
restated here with numpy; the filter outputs come from the imported class.

Output (committed): tests/golden/relay_golden.npz
  form[T,V,3]     raw formation_ref messages         off[T,V,3]   filter outputs after each message
  xr_lead[V,21,10], ur_lead[V,20,4]  leader windows   xr_fol[T,V,21,10]  follower references after tick T
"""
import os
import sys

sys.dont_write_bytecode = True    # importing from /root/reference must not leave __pycache__ there (the tree is read-only by contract)

import numpy as np

sys.path.insert(0, "/root/reference/ndp_nmpc/scripts/hv_throttle_est")
from alpha_filter import AlphaFilter  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.Generator(np.random.PCG64(20231213 + 12))
T, V = 40, 6
form = np.zeros((T, V, 3))
form[:] = rng.uniform(-1.5, 1.5, (1, V, 3))
form[10:] += rng.uniform(-1, 1, (1, V, 3))          # step change (nmpc_leader_node.py:48-58 switches the offsets)
form[25:] = rng.uniform(-1.5, 1.5, (1, V, 3))
xr_lead = rng.normal(0, 1, (V, 21, 10))
ur_lead = rng.normal(0, 1, (V, 20, 4))
filt = [[None] * 3 for _ in range(V)]
off = np.zeros((T, V, 3))
xr_fol = np.zeros((T, V, 21, 10))
for t in range(T):
    for v in range(V):
        for a in range(3):
            if filt[v][a] is None:
                filt[v][a] = AlphaFilter(alpha=0.8, y0=form[t, v, a])        # nmpc_follower_node.py:45-52
            off[t, v, a] = filt[v][a].update(form[t, v, a])                  # :54-56
        x = xr_lead[v].copy()
        x[:, 0] += off[t, v, 0]; x[:, 1] += off[t, v, 1]; x[:, 2] += off[t, v, 2]   # :66-68
        xr_fol[t, v] = x
np.savez_compressed(os.path.join(HERE, "relay_golden.npz"), form=form, off=off, xr_lead=xr_lead, ur_lead=ur_lead, xr_fol=xr_fol)
print("ok", off[0, 0], off[11, 0], off[-1, 0])
