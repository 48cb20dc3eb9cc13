"""Profiling driver of config 5's condensed study kernels (rocprofv3 -- python3 scripts/cond_driver.py <prec> [B] [N] [steps]):
device-resident steps of ONE engine with qp_precision <prec> (5 = fp32 instruction, 6 = bf16 instruction, 0 = the fp64 product path)."""
import sys
sys.path.insert(0, '.')
sys.path.insert(0, __file__.rsplit('/', 2)[0])
import torch
import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import synth

prec = int(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
N = int(sys.argv[3]) if len(sys.argv) > 3 else 40
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
dev = torch.device("cuda", 0)
b = synth.make_batch(B, N=N, seed=synth.SEED0 + 5)
d = {k: torch.from_numpy(b[k]).to(dev) for k in ("x0", "xr", "ur")}
eng = ndp.BatchedNMPC(B, N=N, n_rti=2, qp_precision=prec, work_queue=2 if prec == 0 else 0)
u = torch.empty(B, 4, dtype=torch.float64, device=dev)
eng.reset_device(d["xr"], d["ur"])
for _ in range(steps):
    eng.update_device(d["x0"], d["xr"], d["ur"], u)
torch.cuda.synchronize()
print("done", prec, B, N, steps)
