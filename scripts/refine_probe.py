import sys; sys.path.insert(0,'.')
import numpy as np
import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import synth
from oracle import oracle as O
from tests import ref_numpy as R
for seed in (46, 47, 50, 54, 57, 58, 64, 69, 77, 78):
    b = synth.make_batch(1, seed=seed, pos_sigma=1.5, vel_sigma=3.0, quat_sigma=0.2)
    x0, xr, ur = b["x0"][0], b["xr"][0], b["ur"][0]
    cfgo = O.default_cfg(); cfgo.qp_mode = 1
    qp = O.linearize(cfgo, x0, xr, ur, None, xr.copy(), ur.copy())
    dxf,_,_ = O.qp_solve(cfgo, qp)
    box = 0.8*np.abs((xr+dxf)[4:20,3:6]).max()
    out=[seed]
    for tol in (1e-8,1e-10):
        for refine in (2,0):
            eng = ndp.BatchedNMPC(1, qp_mode=1, tol=tol, ipm_refine=refine, lbv=[-box]*3, ubv=[box]*3)
            eng.reset(b["xr"], b["ur"])
            u0, X, U, st, it = eng.update(b["x0"], b["xr"], b["ur"], raise_on_status=False, full=True)
            cfgo = O.default_cfg(); cfgo.qp_mode, cfgo.tol, cfgo.refine = 1, tol, refine
            for i in range(3): cfgo.lbv[i], cfgo.ubv[i] = -box, box
            qpb = O.linearize(cfgo, x0, xr, ur, None, xr.copy(), ur.copy()); dxa,dua,act = R.pdas_solve(qpb)
            Xo,Uo = xr.copy(), ur.copy(); u0o, sto = O.step(cfgo, x0, xr, ur, None, Xo, Uo)
            out.append("tol %g r%d st %d/%d it %d/%d err %.0e d %.0e |"%(tol,refine,st[0],sto.status,it[0],sto.ipm_iters,max(np.abs(X[0]-xr-dxa).max(),np.abs(U[0]-ur-dua).max()),max(np.abs(X[0]-Xo).max(),np.abs(U[0]-Uo).max())))
            eng.close()
    print(*out)
