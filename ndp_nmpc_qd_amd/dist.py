"""Multi-GPU sharding of the batched control step: one process per GPU, torch.distributed (RCCL on ROCm).

Every OCP instance is independent, so instances shard across ranks with no data-path collective except
one: the downwash predictor of instance i needs its neighbour's reference window, which lives on another
rank when a formation is spread over GPUs.  That is the reference's PredXU exchange
(nmpc_node.py:116-133 -> ndp_nmpc_leader_node.py:40,60-76: the neighbour publishes its 21x10 float64
`nmpc_x_ref`, the leader subtracts its own) -- here ONE all-gather of the ranks' xr windows per control
step, after which each rank reads the slice that holds its neighbours.

Placement (vehicle-major): instance i of rank r and instance i of rank (r+1) % W belong to the same
formation; rank r's downwash input is the window of rank (r+1) % W.  With W = 1 the neighbour windows
are given directly.  The gate and the MLP read only the position / velocity columns of a window
(downwash_nn.py:22), so the exchange that bench.py runs moves just those: exchange_pv_begin / _end below
(1 008 B per instance instead of 1 680); exchange_neighbours (full windows) is the plain form of the same thing.
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


# ------------------------------------------------------------------------------------------- BASELINE config 4
# 4096 three-vehicle formations (12 288 instances) over the GPUs of one node.  Reference semantics: vehicle 0
# of a formation runs the NDP controller and reads vehicle 1's published reference (ndp_nmpc_leader_node.py:40,60-76);
# vehicles 1 and 2 are plain NMPC followers tracking the leader's window shifted by (0, +-1, 0)
# (nmpc_leader_node.py:42-43, nmpc_follower_node.py:44-77; three_qd_ndp_nmpc.launch:4-12).
# Global instance g = 3 * formation + vehicle.  Placements:
#   "vehicle"   (vehicle-major round-robin): rank = g % W, local index g // W -- a leader's neighbour g + 1 lives on the
#               next rank: one all-gather of the [B_local, N+1, 6] position/velocity columns per control step (all that
#               the gate and the MLP read, downwash_nn.py:22; fp64 so that (other - ego) is the reference's subtraction);
#   "formation" (formation-major): rank = g // B_local -- a formation stays on one GPU, no exchange; the kernel reads the
#               neighbour's window straight out of the local xr through other_index.
PV_COLS = 6


def config4_gids(rank, world, n_form, placement):
    """Global ids of this rank's instances in local order, and B_local."""
    total = 3 * n_form
    if total % world or (placement == "formation" and (total // world) % 3):
        raise ValueError(f"{total} instances do not split over {world} ranks")
    bl = total // world
    g = np.arange(bl) * world + rank if placement == "vehicle" else rank * bl + np.arange(bl)
    return g.astype(np.int64), bl


def config4_other_index(rank, world, n_form, placement, peer_rows=False):
    """int32[B_local]: row of the neighbour buffer holding each local instance's neighbour (-1 = none).  The buffer is the
    all-gathered [W * B_local, N+1, 6] array (vehicle-major), the local xr [B_local, N+1, 10] (formation-major), or -- vehicle-major
    with peer_rows -- the window buffer of rank (r+1) % W (PeerWindows.neighbour: a leader's neighbour g + 1 always lives there)."""
    g, bl = config4_gids(rank, world, n_form, placement)
    nb = np.where(g % 3 == 0, g + 1, -1)                   # leaders read vehicle 1 of their formation; followers nobody
    if placement == "vehicle" and peer_rows:
        row = nb // world
    elif placement == "vehicle":
        row = (nb % world) * bl + nb // world
    else:
        row = nb - rank * bl
    return np.where(nb >= 0, row, -1).astype(np.int32)


def make_config4_all(n_form, N=20, seed=synth.SEED0 + 4, t0=0.0):
    """The whole synthetic formation set, global order [3 * n_form]: x0, xr, ur, ego_xy (every rank can build it: the
    seed does not depend on the rank).  Leaders fly 0.4 m above follower 1's track, displaced sideways by 0..2.5 m, so that
    about 40 % of them sit inside the r_horiz gate."""
    base = synth.make_batch(n_form, N=N, seed=seed, downwash=False, t0=t0)
    rng = np.random.Generator(np.random.PCG64([seed, 77]))
    xr = np.repeat(base["xr"], 3, axis=0)                  # [3F, N+1, 10], formation-major global order
    side = rng.uniform(0.0, 2.5, n_form)
    xr[0::3, :, 1] += 1.0 + side[:, None]
    xr[0::3, :, 2] += 0.4
    xr[1::3, :, 1] += 1.0
    xr[2::3, :, 1] -= 1.0
    ur = np.repeat(base["ur"], 3, axis=0)
    x0 = xr[:, 0].copy()
    x0[:, 0:3] += rng.normal(0.0, 0.05, (3 * n_form, 3))
    return dict(x0=x0, xr=xr, ur=ur, ego_xy=np.ascontiguousarray(x0[:, 0:2]))


def make_config4_shard(rank, world, n_form, placement, N=20, seed=synth.SEED0 + 4, t0=0.0):
    """This rank's instances of make_config4_all plus `other_index` and `gids`."""
    allv = make_config4_all(n_form, N=N, seed=seed, t0=t0)
    g, _ = config4_gids(rank, world, n_form, placement)
    out = {k: np.ascontiguousarray(v[g]) for k, v in allv.items()}
    out["other_index"] = config4_other_index(rank, world, n_form, placement)
    out["gids"] = g
    return out


def exchange_pv_begin(xr_local, pv_local, gathered, group=None):
    """Vehicle-major exchange, started without waiting: packs the position/velocity columns of this rank's windows
    (pv_local[B_local, N+1, 6]) and all-gathers them into gathered[W * B_local, N+1, 6].  Returns the work handle (None for
    a single rank, where the pack IS the gathered buffer)."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        gathered.view(pv_local.shape).copy_(xr_local[:, :, :PV_COLS])
        return None
    pv_local.copy_(xr_local[:, :, :PV_COLS])
    return dist.all_gather_into_tensor(gathered.view(-1), pv_local.view(-1), group=group, async_op=True)


def exchange_pv_end(work):
    if work is not None:
        work.wait()


# ------------------------------------------------------------------------------------------- peer windows (pull model)
class _DevMem:
    """Raw device memory as an object torch.as_tensor can alias (CUDA array interface, no ownership)."""

    def __init__(self, ptr, shape, typestr="<f8"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class DevWindows:
    """float64 windows [rows, N+1, 6 or 10] at a raw device address (no torch tensor: the memory may belong to another
    process and another GPU).  BatchedNMPC.update_device accepts it as `other`."""

    def __init__(self, dev_ptr, shape):
        self.dev_ptr, self.shape = int(dev_ptr), tuple(int(x) for x in shape)


class PeerWindows:
    """The neighbour exchange without a per-step collective: every rank keeps the reference windows it publishes
    (`ticks` x [B_local, N+1, 10] float64: what nmpc_node.py:116-133 publishes per tick) in a buffer allocated with
    ndp_peer_alloc, the 64-byte handles are exchanged ONCE (all_gather_object), and each rank maps the buffer of the
    rank that holds its neighbours (ndp_peer_open).  `local[t]` is this rank's window tensor of tick slot t (fill it, then
    publish()); `neighbour[t]` (DevWindows: a raw device address) is rank (r+1) % W's -- pass it as `other` of BatchedNMPC.update_device: the control-step
    kernel reads it out of the neighbour GPU's HBM over xGMI.  With one rank the neighbour is the local buffer.

    publish() = device synchronisation + barrier: the windows of the bench are written once; a deployment that rewrites a
    slot every tick orders writer and readers with its own events, as a ROS publisher / subscriber pair does."""

    def __init__(self, B_local, N, ticks, device, group=None, same_process_ok=True):
        import ctypes as C
        import torch
        import torch.distributed as dist
        from . import _lib
        self._lib = _lib.load()
        self.device = int(device)
        self.shape = (int(ticks), int(B_local), int(N) + 1, 10)
        nbytes = 8 * int(np.prod(self.shape))
        ptr, handle = C.c_void_p(), (C.c_ubyte * 64)()
        rc = self._lib.ndp_peer_alloc(self.device, nbytes, C.byref(ptr), handle)
        self._own = ptr.value if rc == 0 else None
        self._mapped = None
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if self.world > 1:
            # every rank takes part in every collective below whatever happened locally, and all ranks reach the same verdict
            allocs = [None] * self.world
            dist.all_gather_object(allocs, int(rc), group=group)
            if any(allocs):
                self.close()
                raise RuntimeError(f"ndp_peer_alloc failed on some rank: {allocs}")
        elif rc:
            raise RuntimeError(f"ndp_peer_alloc failed ({rc})")
        with torch.cuda.device(self.device):
            self._local = torch.as_tensor(_DevMem(self._own, self.shape), device=torch.device("cuda", self.device))
        self.local = [self._local[t] for t in range(self.shape[0])]
        slot = 8 * int(np.prod(self.shape[1:]))
        if self.world == 1:       # the neighbours are this rank's own vehicles: the same raw-address form over the own buffer
            self.neighbour = [DevWindows(self._own + t * slot, self.shape[1:]) for t in range(self.shape[0])]
            return
        handles = [None] * self.world
        dist.all_gather_object(handles, bytes(handle), group=group)
        nb = handles[neighbour_rank(self.rank, self.world)]
        buf = (C.c_ubyte * 64).from_buffer_copy(nb)
        mp = C.c_void_p()
        rc = self._lib.ndp_peer_open(self.device, buf, C.byref(mp))
        ok = [None] * self.world
        dist.all_gather_object(ok, int(rc), group=group)         # every rank learns whether every mapping worked
        if any(ok):
            if rc == 0:
                self._lib.ndp_peer_close(self.device, mp)
            self.close()
            raise RuntimeError(f"ndp_peer_open failed on some rank: {ok}")
        self._mapped = mp.value
        self.neighbour = [DevWindows(self._mapped + t * slot, self.shape[1:]) for t in range(self.shape[0])]

    def publish(self):
        import torch
        import torch.distributed as dist
        torch.cuda.synchronize(self.device)
        if self.world > 1:
            dist.barrier(group=self.group)

    def close(self):
        if self._mapped:
            self._lib.ndp_peer_close(self.device, self._mapped)
            self._mapped = None
        if self._own:
            self.local, self.neighbour, self._local = None, None, None
            self._lib.ndp_peer_free(self.device, self._own)
            self._own = None
