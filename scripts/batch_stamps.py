"""Whole-batch phase stamps (shader clock) of the fused control step: every instance of a B = 1024 launch reports
where its wave was at each phase boundary.  Run on the GPU box:  python scripts/batch_stamps.py [B] [workload]"""
import sys; sys.path.insert(0, '.')
import numpy as np
import torch
import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import dist as ndist

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
downwash = (sys.argv[2] if len(sys.argv) > 2 else "ndp_downwash") == "ndp_downwash"
dev = torch.device("cuda", 0)
b = ndist.make_formation_shard(B, 0, 1, N=20, t0=0.0)
d = {k: torch.from_numpy(b[k]).to(dev) for k in ("x0", "xr", "ur", "other", "ego_xy")}
eng = ndp.BatchedNMPC(B, N=20, disturbance=downwash, device=0)
u0 = torch.empty(B, 4, dtype=torch.float64, device=dev)
eng.reset_device(d["xr"], d["ur"])
kw = dict(other=d["other"], ego_xy=d["ego_xy"]) if downwash else {}
for _ in range(20):
    eng.update_device(d["x0"], d["xr"], d["ur"], u0, **kw)
eng.synchronize()
eng.debug_stamps(True)
for _ in range(3):
    eng.update_device(d["x0"], d["xr"], d["ur"], u0, **kw)
eng.synchronize()
t = eng.debug_stamps(False, read=True)
names = ['start', 'tables', 'stage_in', 'cost', 'linearize', 'pre_sweep', 'backward', 'forward', 'end']
t0 = t[:, 9].min() if downwash else t[:, 0].min()
print(f"B={B} downwash={downwash}: all times in shader-clock ticks relative to the first wave's first stamp")
if downwash:
    print("kernel entry (stamp 9)    : min %d median %d max %d" % tuple(np.percentile(t[:, 9] - t0, [0, 50, 100])))
    print("fragments staged (11)-(9) : min %d median %d max %d" % tuple(np.percentile(t[:, 11] - t[:, 9], [0, 50, 100])))
    print("mlp tile (10)-(11)        : min %d median %d max %d" % tuple(np.percentile(t[:, 10] - t[:, 11], [0, 50, 100])))
    print("mlp end -> rti start      : median %d" % np.median(t[:, 0] - t[:, 10]))
for i in range(1, 9):
    dphase = t[:, i] - t[:, i - 1]
    print("%-26s: min %d median %d max %d" % ((names[i],) + tuple(np.percentile(dphase, [0, 50, 100]))))
print("last stamp (end)          : min %d median %d max %d" % tuple(np.percentile(t[:, 8] - t0, [0, 50, 100])))
