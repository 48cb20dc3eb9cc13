"""In-kernel phase stamps (s_memtime) of one control step, B = 1, debug path.  Run on the GPU box."""
import sys; sys.path.insert(0, '.')
import numpy as np
import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import synth
b = synth.make_batch(1, seed=3, downwash=True)
names = ['start', 'tables', 'stage_in', 'cost', 'linearize', 'pre_sweep(dump)', 'backward', 'forward', 'end']
from ndp_nmpc_qd_amd import _lib
KT = _lib.lds_layout(20)["stamps"]      # the stamps follow the LDS image in the debug dump
for fused in (False, True):
    eng = ndp.BatchedNMPC(1, disturbance=fused)
    for rep in range(3):
        eng.reset(b['xr'], b['ur'])
        kw = dict(other=b['other'], ego_xy=b['ego_xy']) if fused else {}
        u0, d = eng.update_debug(b['x0'], b['xr'], b['ur'], **kw)
        t = d[KT:KT + 11]
        ft = d[KT + 16:KT + 22]
        line = ' '.join(f'{n}:{int(t[i] - t[i - 1])}' for i, n in enumerate(names) if i > 0)
        extra = f' | mlp_tile:{int(t[10] - t[9])} mlp_end->start:{int(t[0] - t[10])}' if fused else ''
        print('fused' if fused else 'plain', 'rep', rep, line, 'total', int(t[8] - t[0]), extra)
        if rep == 2:
            print('   one backward stage: issue Wf+bracket %d | adjugate(+1/det) %d | G,Kt issue %d | Lam^-1 T ready %d | H ready %d' % tuple(int(ft[i + 1] - ft[i]) for i in range(5)))
