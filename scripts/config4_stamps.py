import sys; sys.path.insert(0, '.')
import numpy as np, torch
import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import dist as nd
F = 1024
s = nd.make_config4_shard(0, 1, F, "formation")
B = len(s["gids"])
dev = torch.device("cuda", 0)
d = {k: torch.from_numpy(s[k]).to(dev) for k in ("x0", "xr", "ur", "ego_xy", "other_index")}
eng = ndp.BatchedNMPC(B, N=20, disturbance=True, device=0, work_queue=2)
u0 = torch.empty(B, 4, dtype=torch.float64, device=dev)
eng.reset_device(d["xr"], d["ur"])
kw = dict(other=d["xr"], ego_xy=d["ego_xy"], other_index=d["other_index"])
for _ in range(5):
    eng.update_device(d["x0"], d["xr"], d["ur"], u0, **kw)
eng.synchronize()
eng.debug_stamps(True)
for _ in range(3):
    eng.update_device(d["x0"], d["xr"], d["ur"], u0, **kw)
eng.synchronize()
t = eng.debug_stamps(False, read=True)
tile = t[:, 10] - t[:, 11]
whole = t[:, 8] - t[:, 9]
print("tile time percentiles", np.percentile(tile, [0, 10, 50, 90, 100]))
print("whole percentiles", np.percentile(whole, [0, 10, 50, 90, 100]))
wg = whole.reshape(-1, 4).max(1)
print("WG max mean", wg.mean(), "frac WG short", (wg < 40000).mean())
print("kernel span (cycles)", (t[:, 8].max() - t[:, 9].min()), "sum WG / 256 CUs", wg.sum() / 256)
