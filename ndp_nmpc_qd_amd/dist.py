"""Multi-GPU sharding of the batched control step: one process per GPU, torch.distributed (RCCL on ROCm).

Every OCP instance is independent, so instances shard across ranks with no data-path collective except
one: the downwash predictor of instance i needs its neighbour's reference window, which lives on another
rank when a formation is spread over GPUs.  That is the reference's PredXU exchange
(nmpc_node.py:116-133 -> ndp_nmpc_leader_node.py:40,60-76: the neighbour publishes its 21x10 float64
`nmpc_x_ref`, the leader subtracts its own) -- here ONE all-gather of the ranks' xr windows per control
step, after which each rank reads the slice that holds its neighbours.

Placement (vehicle-major): instance i of rank r and instance i of rank (r+1) % W belong to the same
formation; rank r's downwash input is the window of rank (r+1) % W.  With W = 1 the neighbour windows
are given directly.
"""
import numpy as np

from . import synth


def neighbour_rank(rank, world):
    return (rank + 1) % world


def exchange_neighbours(xr_local, gathered=None, group=None):
    """All-gathers the ranks' reference windows and returns (other_local, gathered).

    xr_local: [B_local, N+1, 10] float64 tensor (CUDA for RCCL, CPU for gloo).
    other_local is a VIEW into the gathered buffer (no copy): the windows of rank (r+1) % W.
    """
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if gathered is None:
        gathered = torch.empty((world,) + tuple(xr_local.shape), dtype=xr_local.dtype, device=xr_local.device)
    dist.all_gather_into_tensor(gathered.view(-1), xr_local.reshape(-1), group=group)
    return gathered[neighbour_rank(rank, world)], gathered


def exchange_neighbours_begin(xr_local, gathered, group=None):
    """Starts the all-gather of exchange_neighbours without waiting for it (async_op): with RCCL it runs on the
    process group's own stream, ordered after everything already enqueued on the current stream, so a control-step
    kernel launched next on the current stream overlaps it.  The reference windows are functions of time only
    (trajectory generator output), so the windows of tick i+1 can be gathered while tick i is being solved.
    Returns the work handle for exchange_neighbours_end."""
    import torch.distributed as dist
    return dist.all_gather_into_tensor(gathered.view(-1), xr_local.reshape(-1), group=group, async_op=True)


def exchange_neighbours_end(work, gathered, group=None):
    """Makes the current stream (RCCL) / the caller (gloo) wait for the gather started by exchange_neighbours_begin and
    returns the neighbour windows: a view of `gathered`, the slice of rank (r+1) % W."""
    import torch.distributed as dist
    work.wait()
    return gathered[neighbour_rank(dist.get_rank(group), dist.get_world_size(group))]


def make_formation_shard(B_local, rank, world, N=20, seed=synth.SEED0 + 4, t0=0.0):
    """Synthetic formation data for one rank (vehicle-major placement).

    Formation i flies one figure-eight (same on every rank: the seed does not depend on the rank);
    the vehicle on rank r is displaced by a rank-specific offset.  Returns the same dict as
    synth.make_batch plus 'offset'; 'other' is what the exchange must deliver (used by tests as the
    expected value and by the single-GPU path directly).
    """
    base = synth.make_batch(B_local, N=N, seed=seed, downwash=False, t0=t0)

    def offs(r):
        rng = np.random.Generator(np.random.PCG64([seed, 1000 + r]))
        return np.concatenate([rng.uniform(-0.75, 0.75, (B_local, 2)), rng.uniform(0.0, 0.75, (B_local, 1)) * (r % 2 + 1)], axis=1)

    mine, theirs = offs(rank), offs(neighbour_rank(rank, world))
    out = dict(base)
    out["xr"] = base["xr"].copy()
    out["xr"][:, :, 0:3] += mine[:, None, :]
    out["x0"] = base["x0"].copy()
    out["x0"][:, 0:3] += mine
    other = base["xr"].copy()
    other[:, :, 0:3] += theirs[:, None, :]
    out["other"] = other
    out["ego_xy"] = np.ascontiguousarray(out["x0"][:, 0:2])
    out["offset"] = mine
    return out
