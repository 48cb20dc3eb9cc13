import sys; sys.path.insert(0, '.')
import numpy as np, torch
import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import dist as ndist
B = 1024
dev = torch.device("cuda", 0)
b = ndist.make_formation_shard(B, 0, 1, N=20, t0=0.0)
d = {k: torch.from_numpy(b[k]).to(dev) for k in ("x0", "xr", "ur", "other", "ego_xy")}
eng = ndp.BatchedNMPC(B, N=20, disturbance=True, device=0)
u0 = torch.empty(B, 4, dtype=torch.float64, device=dev)
eng.reset_device(d["xr"], d["ur"])
kw = dict(other=d["other"], ego_xy=d["ego_xy"])
for _ in range(20):
    eng.update_device(d["x0"], d["xr"], d["ur"], u0, **kw)
eng.synchronize()
eng.debug_stamps(True)
for _ in range(3):
    eng.update_device(d["x0"], d["xr"], d["ur"], u0, **kw)
eng.synchronize()
t = eng.debug_stamps(False, read=True)
ent, ext = t[:, 12], t[:, 13]          # 100 MHz real time at wave entry / exit
print("entry spread (us): p50 %.2f p90 %.2f max %.2f" % tuple((np.percentile(ent, [50, 90, 100]) - ent.min()) / 100))
print("wave duration (us): min %.2f p50 %.2f max %.2f" % tuple(np.percentile(ext - ent, [0, 50, 100]) / 100))
print("first entry -> last exit (us): %.2f" % ((ext.max() - ent.min()) / 100))
wg = ent.reshape(-1, 4)
print("entry by workgroup index (us since first): ", np.round((wg[::32, 0] - ent.min()) / 100, 2))
cyc = t[:, 15] - t[:, 14]
print("shader cycles per wave p50 %.0f; clock %.3f GHz" % (np.median(cyc), np.median(cyc / ((ext - ent) * 10e-9)) / 1e9))
print("entry -> stamp 9 (cycles): p50 %.0f; stamp 8 -> exit: p50 %.0f" % (np.median(t[:, 9] - t[:, 14]), np.median(t[:, 15] - t[:, 8])))
