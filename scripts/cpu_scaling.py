import os, time, numpy as np, sys
sys.path.insert(0,'.')
print('nproc', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for p in ('/sys/fs/cgroup/cpu.max','/sys/fs/cgroup/cpu/cpu.cfs_quota_us','/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
    try: print(p, open(p).read().strip())
    except Exception as e: print(p, 'n/a')
from oracle import oracle as O
from ndp_nmpc_qd_amd import synth
cfg=O.default_cfg()
b=synth.make_batch(1024,seed=5)
for nt in (1,2,4,8,16,32,64,128):
    X,U=b['xr'].copy(),b['ur'].copy()
    O.step_batch(cfg,b['x0'],b['xr'],b['ur'],None,X,U,nthreads=nt)
    t=time.time()
    for _ in range(4): O.step_batch(cfg,b['x0'],b['xr'],b['ur'],None,X,U,nthreads=nt)
    dt=time.time()-t
    print(nt,'threads',int(4*1024/dt),'solves/s')
