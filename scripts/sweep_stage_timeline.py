"""Timeline of ONE stage of the backward Riccati sweep (stage N / 2 of the headline kernel, B = 1, debug path): the shader clock at
which each of the stage's results is available.  Needs a library built with -DNDP_FINE_STAMPS:
    bash scripts/dev_kernel.sh -DNDP_FINE_STAMPS
    NDP_NMPC_LIB=$PWD/ndp_nmpc_qd_amd/libndp_nmpc_hip_dev.so python3 scripts/sweep_stage_timeline.py     -> profiles/r05_sweep_stage_timeline.txt
Floor of the stage's dependent matrix chain: 7 x 64 (v_mfma_f64_16x16x4) + 2 x 44 (v_mfma_f64_4x4x4_4b) = 536 cycles; the product
kernel's stages take 895 (profiles/r04_phase_stamps_b1024.txt: 17.9 k cycles / 20)."""
import sys; sys.path.insert(0, '.')
import numpy as np
import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import synth, _lib
b = synth.make_batch(1, seed=3, downwash=True)
KT = _lib.lds_layout(20)["stamps"]
names = ["Lam gathered (LDS / DPP)", "W = [Hxx;Hux] M' ready (3 x 16x16x4)", "bracket C' + M'' W ready (3 x 16x16x4, cofactor VALU in between)",
         "adj(Lam) T ready (1 x 4x4x4_4b)", "Lam^-1 T scaled: det, 1 / det (VALU chain)", "H_k ready (1 x 16x16x4 on bracket)", "K' ready (1 x 4x4x4_4b)"]
for fused in (False, True):
    eng = ndp.BatchedNMPC(1, disturbance=fused)
    rows = []
    for rep in range(5):
        eng.reset(b['xr'], b['ur'])
        kw = dict(other=b['other'], ego_xy=b['ego_xy']) if fused else {}
        u0, d = eng.update_debug(b['x0'], b['xr'], b['ur'], **kw)
        t, ft = d[KT:KT + 16], d[KT + 16:KT + 32]
        idx = [0, 1, 2, 3, 4, 5, 13, 14]
        rows.append(([ft[i] - ft[0] for i in idx], t[6] - t[5]))
    r = np.median(np.array([x[0] for x in rows]), axis=0)
    print(f"{'fused' if fused else 'plain'} kernel, stage N/2 of the backward sweep (median of 5 runs; cycles since the stage's H operand was ready); "
          f"whole backward sweep {np.median([x[1] for x in rows]):.0f} cycles (with the stamped stage's waits)")
    for n, a, c in zip(names, r[1:], np.diff(r)):
        print(f"   +{c:6.0f}  = {a:6.0f}   {n}")
