"""BASELINE config 5: N = 40 horizon, 2 RTI iterations per step, batch 4096, one MI355X -- the fp64 product path
against the CPU oracle, and the precision study (qp_precision 1 = fp32, 2 = bf16 operands of the sweeps' matrix
instructions, fp32 accumulators; numerics only, see include/ndp_nmpc.h).  Run on the GPU box; prints one JSON line."""
import json
import sys
import time

sys.path.insert(0, '.')
import numpy as np
import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import synth
from oracle import oracle as O

B, N, NS = 4096, 40, 256
b = synth.make_batch(B, N=N, seed=20231213 + 5)
cfgo = O.default_cfg(N=N, n_rti=2)
Xo, Uo = b["xr"][:NS].copy(), b["ur"][:NS].copy()
uo, sto, ito = O.step_batch(cfgo, b["x0"][:NS], b["xr"][:NS], b["ur"][:NS], None, Xo, Uo)
out = {"config": "N=40, 2 RTI iterations, batch=4096, no downwash", "oracle_sample": NS,
       }
ref, free = None, None
for prec, name in ((0, "fp64"), (1, "fp32_study"), (2, "bf16_study")):
    eng = ndp.BatchedNMPC(B, N=N, n_rti=2, qp_precision=prec)
    eng.reset(b["xr"], b["ur"])
    u0 = eng.update(b["x0"], b["xr"], b["ur"], raise_on_status=False)
    st, it = eng.status()
    rel = np.abs(u0[:NS] - uo) / np.maximum(1.0, np.abs(uo))
    if prec == 0:
        free = it[:NS] == 0                      # fp64 AUTO path left early: no bound active in that instance's QPs
        out["instances_with_active_bounds_in_sample"] = int((~free).sum())
    d = {"max_rel_err_vs_oracle": float(rel.max()), "max_rel_err_no_active_bounds": float(rel[free].max()),
         "status_nonzero": int((st != 0).sum())}
    if prec == 0:
        ref = u0
        import torch
        dev = torch.device("cuda", 0)
        t = {k: torch.from_numpy(b[k]).to(dev) for k in ("x0", "xr", "ur")}
        u0d = torch.empty(B, 4, dtype=torch.float64, device=dev)
        eng.reset_device(t["xr"], t["ur"])
        for _ in range(10):
            eng.update_device(t["x0"], t["xr"], t["ur"], u0d)
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            eng.update_device(t["x0"], t["xr"], t["ur"], u0d)
        eng.synchronize()
        el = (time.perf_counter() - t0) / 50
        d["ms_per_step"] = el * 1e3
        d["solves_per_s"] = B / el
    else:
        d["max_rel_err_vs_fp64_device_all_4096"] = float((np.abs(u0 - ref) / np.maximum(1.0, np.abs(ref))).max())
    out[name] = d
print(json.dumps(out))
