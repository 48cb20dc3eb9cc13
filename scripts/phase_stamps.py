import sys; sys.path.insert(0,'.')
import numpy as np
import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import synth
b=synth.make_batch(1,seed=3)
eng=ndp.BatchedNMPC(1)
eng.reset(b['xr'],b['ur'])
names=['start','tables','stage_in','cost','linearize','pre_sweep','backward','forward','end']
for rep in range(3):
    eng.reset(b['xr'],b['ur'])
    u0,d=eng.update_debug(b['x0'],b['xr'],b['ur'])
    KT=48+3*(21*10+20*4)+20*86+21*47
    t=d[KT:KT+9]
    print('rep',rep,' '.join(f'{n}:{int(t[i]-t[i-1])}' for i,n in enumerate(names) if i>0),'total',int(t[8]-t[0]))
