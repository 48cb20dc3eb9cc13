"""Entry-to-exit time of every wave of ONE fused control-step launch (stamps 14 / 15: shader clock at kernel entry / exit; 12 / 13: the
100 MHz real-time counter there) and the launch's span from the first wave's entry to the last wave's exit -- what the phase stamps of
scripts/batch_stamps.py do not cover (the kernel's first instructions and its last stores).  python scripts/wave_span.py [B]"""
import sys; sys.path.insert(0, '.')
import numpy as np
import torch
import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import dist as ndist

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
b = ndist.make_formation_shard(B, 0, 1, N=20, t0=0.0)
d = {k: torch.from_numpy(b[k]).to(dev) for k in ("x0", "xr", "ur", "other", "ego_xy")}
eng = ndp.BatchedNMPC(B, N=20, disturbance=True, device=0)
u0 = torch.empty(B, 4, dtype=torch.float64, device=dev)
eng.reset_device(d["xr"], d["ur"])
kw = dict(other=d["other"], ego_xy=d["ego_xy"])
for _ in range(50):
    eng.update_device(d["x0"], d["xr"], d["ur"], u0, **kw)
eng.synchronize()
res = []
for rep in range(15):
    eng.debug_stamps(True)
    eng.update_device(d["x0"], d["xr"], d["ur"], u0, **kw)
    eng.synchronize()
    t = eng.debug_stamps(False, read=True)
    wave = t[:, 15] - t[:, 14]
    span = t[:, 15].max() - t[:, 14].min()
    span_rt = (t[:, 13].max() - t[:, 12].min()) * 10.0      # ns
    pro = t[:, 9] - t[:, 14]                                   # entry -> the "kernel entry" phase stamp
    epi = t[:, 15] - t[:, 8]                                   # last phase stamp -> exit
    res.append((np.median(wave), wave.max(), span, span_rt, np.median(pro), pro.max(), np.median(epi), epi.max(), np.ptp(t[:, 12]) * 10.0,
                np.ptp(t[:, 13]) * 10.0, np.median(t[:, 13] - t[:, 12]) * 10.0, (t[:, 13] - t[:, 12]).max() * 10.0))
r = np.median(np.array(res), axis=0)
print("wave entry->exit median %d max %d cycles | launch span (first entry -> last exit) %.0f ns | prologue median %d max %d | epilogue median %d max %d cycles | "
      "entry times spread %.0f ns, exit times spread %.0f ns | wave entry->exit real time median %.0f max %.0f ns"
      % (r[0], r[1], r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10], r[11]))
