import sys; sys.path.insert(0,'.')
import numpy as np
import ndp_nmpc_qd_amd as ndp
g=np.load('tests/golden/relay_golden.npz')
T,V,_=g['form'].shape
eng=ndp.BatchedNMPC(V,load_mlp=False)
for t in range(T):
    off=eng.relay_formation(g['form'][t])
    n=(off!=g['off'][t]).sum()
    if n: print(t,n,np.abs(off-g['off'][t]).max(), off[off!=g['off'][t]][:2], g['off'][t][off!=g['off'][t]][:2])
print('done')
