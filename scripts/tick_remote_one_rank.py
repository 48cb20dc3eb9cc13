"""bench.tick_remote_ranks with ONE rank and a real RCCL communicator (torch.distributed, backend nccl, world size 1): the leg the
N > 1 bench lines carry as `tick_remote`, on the one-GPU box -- the all-gather is a real ncclAllGather call (no xGMI traffic), the
neighbour "rank" is the rank itself.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
import torch
import torch.distributed as dist

import bench
import ndp_nmpc_qd_amd as ndp
from ndp_nmpc_qd_amd import synth

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
from ndp_nmpc_qd_amd import dist as ndist
out = {"torch_collective": bench.tick_remote_ranks(ndp, synth, dist, torch, 1024, 20, 0, 1, 0, dev, dev, stream, False, n_ticks=300)}
x = ndist.RcclExchange(1024, 20, 0)
out["library_collective"] = bench.tick_remote_ranks(ndp, synth, dist, torch, 1024, 20, 0, 1, 0, dev, dev, stream, False, n_ticks=300, xchg=x)
out["library_collective_one_period_ahead"] = bench.tick_remote_ranks(ndp, synth, dist, torch, 1024, 20, 0, 1, 0, dev, dev, stream, False, n_ticks=300, xchg=x, ahead=True)
x.close()
print(json.dumps(bench.compact(out)))
dist.destroy_process_group()
