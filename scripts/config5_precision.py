"""BASELINE config 5: N = 40 horizon, 2 RTI iterations per step, batch 4096, one MI355X -- "fp32 vs bf16 MFMA on the QP".
For each QP precision: u0 error against the CPU oracle (256 instances) and against the fp64 device result (all 4096), and
the throughput of the device-resident step.  qp_precision 0 = the product path (v_mfma_f64_16x16x4_f64), 3 = the sweeps on
v_mfma_f32_16x16x4_f32, 4 = on v_mfma_f32_16x16x16_bf16 (fp32 accumulate); 1 / 2 = the first round's operand-rounding studies on
the fp64 kernel (numerics only).  A second batch with perturbed starts shows what the interior-point loop does to the
fp32 sweeps.  Run on the GPU box; prints one JSON line."""
import json
import sys
import time

sys.path.insert(0, '.')
import numpy as np
import torch
import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import synth
from oracle import oracle as O

B, N, NS = 4096, 40, 256
dev = torch.device("cuda", 0)
out = {"config": "N=40, 2 RTI iterations, batch=4096, no downwash", "oracle_sample": NS}
for label, kw in (("nominal", {}), ("perturbed", dict(pos_sigma=0.5, vel_sigma=1.0, quat_sigma=0.15))):
    b = synth.make_batch(B, N=N, seed=20231213 + 5, **kw)
    cfgo = O.default_cfg(N=N, n_rti=2)
    Xo, Uo = b["xr"][:NS].copy(), b["ur"][:NS].copy()
    uo, sto, ito = O.step_batch(cfgo, b["x0"][:NS], b["xr"][:NS], b["ur"][:NS], None, Xo, Uo)
    t = {k: torch.from_numpy(b[k]).to(dev) for k in ("x0", "xr", "ur")}
    res, ref, free = {}, None, None
    for prec, name in ((0, "fp64"), (0, "fp64_work_list"), (3, "fp32_mfma"), (4, "bf16_mfma"), (1, "fp32_rounding_study"), (2, "bf16_rounding_study")):
        # "fp64" and the fp32 / bf16 kernels solve in place (one launch per step); "fp64_work_list" is the product default at this
        # batch size: producer + consumer launch (the producer carries no interior-point code and does not spill at N = 40)
        eng = ndp.BatchedNMPC(B, N=N, n_rti=2, qp_precision=prec, work_queue=1 if name == "fp64_work_list" else 2)
        eng.reset(b["xr"], b["ur"])
        u0 = eng.update(b["x0"], b["xr"], b["ur"], raise_on_status=False)
        st, it = eng.status()
        ok = (st[:NS] == 0) & (sto == 0)
        rel = np.abs(u0[:NS] - uo) / np.maximum(1.0, np.abs(uo))
        if name == "fp64":
            ref, free = u0, it == 0                  # fp64 AUTO path left early: no bound active in that instance's QPs
            out.setdefault(label, {})["frac_interior_point"] = float((~free).mean())
        d = {"max_rel_err_vs_oracle": float(rel[ok].max()), "status_nonzero": int((st != 0).sum())}
        if name != "fp64":
            relall = np.abs(u0 - ref) / np.maximum(1.0, np.abs(ref))
            good = st == 0
            d["max_rel_err_vs_fp64_device_no_active_bounds"] = float(relall[free & good].max())
            if (~free & good).any():
                d["max_rel_err_vs_fp64_device_interior_point_instances"] = float(relall[~free & good].max())
        if prec in (0, 3, 4):
            u0d = torch.empty(B, 4, dtype=torch.float64, device=dev)
            eng.reset_device(t["xr"], t["ur"])
            for _ in range(10):
                eng.update_device(t["x0"], t["xr"], t["ur"], u0d)
            eng.synchronize()
            t0 = time.perf_counter()
            for _ in range(40):
                eng.update_device(t["x0"], t["xr"], t["ur"], u0d)
            eng.synchronize()
            el = (time.perf_counter() - t0) / 40
            d["ms_per_step"] = el * 1e3
            d["solves_per_s"] = B / el
        res[name] = d
    out[label].update(res)
print(json.dumps(out))
