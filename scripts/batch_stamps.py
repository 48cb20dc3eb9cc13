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
# stamps: 9 kernel entry (fused), 0 program start, 1 tables requested, 2 inputs committed, 3 cost blocks, 11 / 10 downwash tile
# start / end (fused), 4 linearised, 5 sweep start, 6 backward done, 7 forward done, 8 end.  With a compile-time horizon the fused
# kernel computes the tile between the cost and the linearisation phase (the weight fragments are staged under the first phases);
# otherwise before the program starts -- the phases are printed in the order the stamps were taken.
label = {9: 'kernel entry', 0: 'program start', 1: 'tables', 2: 'stage_in', 3: 'cost', 11: 'fragments staged / barrier', 10: 'mlp tile',
         4: 'linearize', 5: 'pre_sweep', 6: 'backward', 7: 'forward', 8: 'end'}
ids = [i for i in (([9, 11, 10] if downwash else []) + list(range(9)))]
ids.sort(key=lambda i: np.median(t[:, i] - t[:, 0]))
t0 = t[:, ids[0]].min()
print(f"B={B} downwash={downwash}: shader-clock ticks; each line = time since the previous stamp, stamps in the order they were taken")
print("%-28s: min %d median %d max %d   (relative to the first wave's first stamp)" % ((label[ids[0]],) + tuple(np.percentile(t[:, ids[0]] - t0, [0, 50, 100]))))
for a_, b_ in zip(ids[:-1], ids[1:]):
    dphase = t[:, b_] - t[:, a_]
    print("%-28s: min %d median %d max %d" % ((label[b_],) + tuple(np.percentile(dphase, [0, 50, 100]))))
print("%-28s: min %d median %d max %d" % (("whole wave program",) + tuple(np.percentile(t[:, 8] - t[:, ids[0]], [0, 50, 100]))))
