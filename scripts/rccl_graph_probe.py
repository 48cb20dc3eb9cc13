"""Can one control step's neighbour exchange (an RCCL all-gather started async, waited on the compute stream) be captured
into a hipGraph together with the step's kernel and replayed?  One rank is enough to exercise the mechanics on a
single-GPU box: python scripts/rccl_graph_probe.py (sets up a 1-rank nccl process group itself)."""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, ".")
import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import dist as ndist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
B, N, T = 1024, 20, 4
ticks = []
for t in range(T):
    b = ndist.make_formation_shard(B, 0, 1, N=N, t0=0.02 * t)
    ticks.append({k: torch.from_numpy(b[k]).to(dev) for k in ("x0", "xr", "ur", "ego_xy")})
eng = ndp.BatchedNMPC(B, disturbance=True)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
u0 = torch.empty(B, 4, dtype=torch.float64, device=dev)
gathered = [torch.empty(B, N + 1, ndist.PV_COLS, dtype=torch.float64, device=dev) for _ in range(2)]
pv_local = torch.empty(B, N + 1, ndist.PV_COLS, dtype=torch.float64, device=dev)
pending = {}


def prefetch(i):
    # force the collective even with one rank (exchange_pv_begin would shortcut it): this probe is about capturing RCCL
    pv_local.copy_(ticks[i % T]["xr"][:, :, :ndist.PV_COLS])
    pending[i] = dist.all_gather_into_tensor(gathered[i % 2].view(-1), pv_local.view(-1), async_op=True)


def step(i):
    d = ticks[i % T]
    if i not in pending:
        prefetch(i)
    pending.pop(i).wait()
    other = gathered[i % 2]
    prefetch(i + 1)
    eng.update_device(d["x0"], d["xr"], d["ur"], u0, other=other, ego_xy=d["ego_xy"], stream=stream)


eng.reset_device(ticks[0]["xr"], ticks[0]["ur"], stream=stream)
for i in range(8):
    step(i)
for w in pending.values():
    w.wait()
pending.clear()
torch.cuda.synchronize()
ref = u0.cpu().numpy().copy()
t0 = time.perf_counter()
for i in range(64):
    step(8 + i)
for w in pending.values():
    w.wait()
pending.clear()
torch.cuda.synchronize()
host_us = (time.perf_counter() - t0) / 64 * 1e6
G = 64
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=stream, capture_error_mode="relaxed"):
        for i in range(G):
            step(i)
        for w in pending.values():
            w.wait()
        pending.clear()
    torch.cuda.set_stream(stream)
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        g.replay()
    torch.cuda.synchronize()
    graph_us = (time.perf_counter() - t0) / (4 * G) * 1e6
    st, _ = eng.status()
    print(f"captured: host-launched step {host_us:.1f} us, replayed step {graph_us:.1f} us, status ok {(st == 0).all()}, "
          f"u0 finite {bool(np.isfinite(u0.cpu().numpy()).all())}")
except Exception as e:  # noqa: BLE001
    print("capture failed:", type(e).__name__, str(e)[:400])
dist.destroy_process_group()
