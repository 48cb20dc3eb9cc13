"""Per-step cost of the RCCL form of the neighbour exchange on ONE GPU with a one-rank communicator (what a rank of an N > 1 run pays
per tick apart from the wire): batch 1024, N = 20, host-launched.  (a) the library's own all-gather (ndp_xchg_*: pack + ncclAllGather
on its own stream) + a bound step, (b) torch.distributed.all_gather_into_tensor + update_device, (c) the control step alone."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29543")
import numpy as np
import torch
import torch.distributed as dist

import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import dist as ndist

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
B, N, T = 1024, 20, 8
ticks = []
for t in range(T):
    b = ndist.make_formation_shard(B, 0, 1, N=N, t0=0.02 * t)
    ticks.append({k: torch.from_numpy(b[k]).to(dev) for k in ("x0", "xr", "ur", "ego_xy", "other")})
eng = ndp.BatchedNMPC(B, disturbance=True)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
u0 = torch.empty(B, 4, dtype=torch.float64, device=dev)
gathered = [torch.empty(B, N + 1, 6, dtype=torch.float64, device=dev) for _ in range(2)]
pv = torch.empty(B, N + 1, 6, dtype=torch.float64, device=dev)
ex = ndist.RcclExchange(B, N, 0)
eng.reset_device(ticks[0]["xr"], ticks[0]["ur"], stream=stream)
K = 400


def timed(body, prime):
    best = 1e9
    for rep in range(3):
        prime()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            body(i)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / K)
    return best * 1e6


bound = {}


def lib_body(i):
    ex.end(stream)
    ex.begin(ticks[(i + 1) % T]["other"], gathered[(i + 1) % 2], stream)
    key = (i % T, i % 2)
    if key not in bound:
        d = ticks[i % T]
        bound[key] = eng.bind_update_device(d["x0"], d["xr"], d["ur"], u0, other=gathered[i % 2], ego_xy=d["ego_xy"], stream=stream)
    bound[key]()


pend = [None]


def torch_body(i):
    ndist.exchange_pv_end(pend[0])
    pend[0] = ndist.exchange_pv_begin(ticks[(i + 1) % T]["other"], pv, gathered[(i + 1) % 2], force_collective=True)
    d = ticks[i % T]
    eng.update_device(d["x0"], d["xr"], d["ur"], u0, other=gathered[i % 2], ego_xy=d["ego_xy"], stream=stream)


alone = {}


def alone_body(i):
    if i % T not in alone:
        d = ticks[i % T]
        alone[i % T] = eng.bind_update_device(d["x0"], d["xr"], d["ur"], u0, other=d["other"], ego_xy=d["ego_xy"], stream=stream)
    alone[i % T]()


t_lib = timed(lib_body, lambda: ex.begin(ticks[0]["other"], gathered[0], stream))


def lib_body_nodep(i):
    ex.end(stream)
    ex.begin(ticks[(i + 1) % T]["other"], gathered[(i + 1) % 2], None)      # the windows are in place: no ordering behind the compute stream
    bound[(i % T, i % 2)]()


t_lib2 = timed(lib_body_nodep, lambda: ex.begin(ticks[0]["other"], gathered[0], None))

eng.track_steps(True)
bound.clear()


def lib_body_event(i):
    ex.end(stream)
    # gathered[(i + 1) % 2] was read last by the control step launched last (tick i - 1): the gather waits for ITS completion event
    ex.begin(ticks[(i + 1) % T]["other"], gathered[(i + 1) % 2], None, after_event=eng.last_step_event() if i else None)
    key = (i % T, i % 2)
    if key not in bound:
        d = ticks[i % T]
        bound[key] = eng.bind_update_device(d["x0"], d["xr"], d["ur"], u0, other=gathered[i % 2], ego_xy=d["ego_xy"], stream=stream)
    bound[key]()


t_lib3 = timed(lib_body_event, lambda: ex.begin(ticks[0]["other"], gathered[0], None))
eng.track_steps(False)


def prime_torch():
    pend[0] = ndist.exchange_pv_begin(ticks[0]["other"], pv, gathered[0], force_collective=True)


t_torch = timed(torch_body, prime_torch)
t_alone = timed(alone_body, lambda: None)
print("batch 1024, one-rank communicator, launched from the host, us per tick:")
print("  control step alone (bound launch)                                   %6.1f" % t_alone)
print("  + the library's all-gather of the tick's windows (ndp_xchg_*)       %6.1f" % t_lib)
print("    the same, NO ordering behind the compute stream (unsafe: the gather may overwrite a buffer still being read) %5.1f" % t_lib2)
print("    the same, ordered behind the last reader's completion event (ndp_track_steps: no packet on the compute stream) %5.1f" % t_lib3)
print("  + torch.distributed.all_gather_into_tensor (pack copy + c10d call)  %6.1f" % t_torch)
ex.close()
dist.destroy_process_group()
