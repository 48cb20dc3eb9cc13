"""Multi-GPU sharding of the batched control step: one process per GPU, torch.distributed (RCCL on ROCm).

Every OCP instance is independent, so instances shard across ranks with no data-path collective except
one: the downwash predictor of instance i needs its neighbour's reference window, which lives on another
rank when a formation is spread over GPUs.  That is the reference's PredXU exchange
(nmpc_node.py:116-133,229-230 -> ndp_nmpc_leader_node.py:40,60-76: every tick the neighbour publishes its 21x10 float64
`nmpc_x_ref`, the leader subtracts its own) -- here one exchange per control step, in one of two forms:
  * ONE all-gather of the ranks' windows (exchange_pv_begin / _end: only the position / velocity columns travel, all that the
    gate and the MLP read, downwash_nn.py:22 -- 1 008 B per instance instead of 1 680; exchange_neighbours is the plain full-window
    form), after which each rank reads the slice that holds its neighbours;
  * publish / subscribe through peer-mapped window slots over xGMI with device-side epoch ordering (PeerWindows below,
    csrc/peer_epoch.hpp): no collective, no host round trip.

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


def config4_gids(rank, world, n_form, placement, order="leaders_first"):
    """Global ids of this rank's instances in LOCAL order, and B_local.

    order: "interleaved" -- ascending global id (every workgroup of four consecutive instances holds one or two leaders);
    "leaders_first" (default) -- the rank's leaders (NDP controller: gate + downwash network, ~22 us per workgroup) in front of its
    followers (plain NMPC, ~17.6 us).  Instances are independent, so the local order is free; workgroups are dispatched in order, so
    with more workgroups than CUs (1 536 instances per GPU at eight GPUs = 384 workgroups on 256 CUs) the long ones start first and
    the second round consists of short ones only."""
    total = 3 * n_form
    if total % world or (placement == "formation" and (total // world) % 3):
        raise ValueError(f"{total} instances do not split over {world} ranks")
    bl = total // world
    g = np.arange(bl) * world + rank if placement == "vehicle" else rank * bl + np.arange(bl)
    if order == "leaders_first":
        g = np.concatenate([g[g % 3 == 0], g[g % 3 != 0]])
    elif order != "interleaved":
        raise ValueError(f"unknown instance order {order!r}")
    return g.astype(np.int64), bl


_CFG4_MAPS = {}


def _config4_maps(world, n_form, placement, order):
    """owner rank and local row of every global instance."""
    key = (world, n_form, placement, order)
    if key not in _CFG4_MAPS:
        total = 3 * n_form
        owner, loc = np.empty(total, dtype=np.int64), np.empty(total, dtype=np.int64)
        for q in range(world):
            gq, bl = config4_gids(q, world, n_form, placement, order)
            owner[gq], loc[gq] = q, np.arange(bl)
        _CFG4_MAPS.clear()
        _CFG4_MAPS[key] = (owner, loc)
    return _CFG4_MAPS[key]


def config4_other_index(rank, world, n_form, placement, peer_rows=False, order="leaders_first"):
    """int32[B_local]: row of the neighbour buffer holding each local instance's neighbour (-1 = none).  The buffer is the
    all-gathered [W * B_local, N+1, 6] array (vehicle-major), the local xr [B_local, N+1, 10] (formation-major), or -- vehicle-major
    with peer_rows -- the window buffer of rank (r+1) % W (PeerWindows.neighbour: a leader's neighbour g + 1 always lives there)."""
    g, bl = config4_gids(rank, world, n_form, placement, order)
    owner, loc = _config4_maps(world, n_form, placement, order)
    lead = g % 3 == 0                                      # leaders read vehicle 1 of their formation; followers nobody
    nb = np.where(lead, g + 1, 0)
    if placement == "vehicle" and peer_rows:
        assert (owner[nb[lead]] == neighbour_rank(rank, world)).all()
        row = loc[nb]
    elif placement == "vehicle":
        row = owner[nb] * bl + loc[nb]
    else:
        assert (owner[nb[lead]] == rank).all()
        row = loc[nb]
    return np.where(lead, row, -1).astype(np.int32)


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


def make_config4_shard(rank, world, n_form, placement, N=20, seed=synth.SEED0 + 4, t0=0.0, order="leaders_first"):
    """This rank's instances of make_config4_all (in the local order of config4_gids) plus `other_index` and `gids`."""
    allv = make_config4_all(n_form, N=N, seed=seed, t0=t0)
    g, _ = config4_gids(rank, world, n_form, placement, order)
    out = {k: np.ascontiguousarray(v[g]) for k, v in allv.items()}
    out["other_index"] = config4_other_index(rank, world, n_form, placement, order=order)
    out["gids"] = g
    return out


def exchange_pv_begin(xr_local, pv_local, gathered, group=None, force_collective=False):
    """Vehicle-major exchange, started without waiting: packs the position/velocity columns of this rank's windows
    (pv_local[B_local, N+1, 6]) and all-gathers them into gathered[W * B_local, N+1, 6].  Returns the work handle (None for
    a single rank, where the pack IS the gathered buffer -- unless force_collective asks for the collective call all the same:
    a one-rank RCCL group then exercises the call, its stream hand-over and its graph capture on a one-GPU box)."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (force_collective and dist.is_initialized()):
        gathered.view(pv_local.shape).copy_(xr_local[:, :, :PV_COLS])
        return None
    pv_local.copy_(xr_local[:, :, :PV_COLS])
    return dist.all_gather_into_tensor(gathered.view(-1), pv_local.view(-1), group=group, async_op=True)


def exchange_pv_end(work):
    if work is not None:
        work.wait()


# ------------------------------------------------------------------------------------------- RCCL all-gather issued by the library
def loaded_rccl_path():
    """The librccl this process already holds (torch's), so that the C-ABI library binds the same one -- or "" (loader's search path)."""
    try:
        with open("/proc/self/maps") as fh:
            for ln in fh:
                if "librccl" in ln:
                    return ln.split()[-1]
    except OSError:
        pass
    return ""


class RcclExchange:
    """The per-tick neighbour exchange as ONE RCCL all-gather of the ranks' position / velocity windows, issued by the C-ABI library on
    a HIP stream of its own (include/ndp_nmpc.h: ndp_xchg_*): pack launch + ncclAllGather + two event operations per tick, no
    torch.distributed call on the step's path (c10d's all_gather_into_tensor costs ~25 us of host time per call -- more than a control
    step lasts).  The communicator's id travels once through torch.distributed (any backend), or is local with one rank.

        ex = RcclExchange(B_local, N, device, group)
        ex.begin(xr_next, gathered[(i + 1) % 2], stream)      # tick i+1's windows: runs beside tick i's control step
        ex.end(stream)                                        # `stream` waits (on the device) for the gather started last
    """

    def __init__(self, B_local, N, device, group=None):
        import ctypes as C
        import torch.distributed as dist
        from . import _lib
        self._lib = _lib.load()
        self.device, self.rows = int(device), int(B_local) * (int(N) + 1)
        multi = dist.is_initialized() and dist.get_world_size(group) > 1
        self.world = dist.get_world_size(group) if multi else 1
        self.rank = dist.get_rank(group) if multi else 0
        path = loaded_rccl_path().encode()
        uid = (C.c_ubyte * 128)()
        rc = 0
        if self.rank == 0:
            rc = self._lib.ndp_xchg_unique_id(path, uid)
        payload = [(rc, bytes(uid))]
        if multi:
            dist.broadcast_object_list(payload, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        rc, raw = payload[0]
        if rc:
            raise RuntimeError(f"ndp_xchg_unique_id failed ({rc}): RCCL could not be bound")
        uid = (C.c_ubyte * 128).from_buffer_copy(raw)
        h = C.c_void_p()
        rc = self._lib.ndp_xchg_create(self.device, self.rank, self.world, uid, path, C.byref(h))
        if rc:
            raise RuntimeError(f"ndp_xchg_create failed ({rc})")
        self._h = h

    def begin(self, xr, gathered, stream=None, after_event=None):
        """xr: this rank's [B_local, N+1, 10] float64 windows of the tick; gathered: [world * B_local, N+1, 6].  stream: whatever
        produces xr runs there (the gather is ordered behind everything it holds); after_event: a raw HIP event the gather waits for
        (BatchedNMPC.last_step_event(): the control step that read `gathered` last).  One of the two must cover the last reader of
        `gathered`."""
        import ctypes as C
        if not (xr.is_contiguous() and gathered.is_contiguous() and xr.numel() == self.rows * 10
                and gathered.numel() == self.world * self.rows * PV_COLS):
            raise ValueError(f"RcclExchange.begin: expected contiguous xr [{self.rows // 1} rows x 10] and gathered "
                             f"[{self.world} x {self.rows} rows x {PV_COLS}], got {tuple(xr.shape)} / {tuple(gathered.shape)}")
        rc = self._lib.ndp_xchg_begin(self._h, C.c_void_p(xr.data_ptr()), self.rows, C.c_void_p(gathered.data_ptr()),
                                      C.c_void_p(stream.cuda_stream) if stream is not None else None, after_event)
        if rc:
            raise RuntimeError(f"ndp_xchg_begin failed ({rc}): {self._lib.ndp_xchg_last_error(self._h).decode()}")

    def end(self, stream):
        import ctypes as C
        rc = self._lib.ndp_xchg_end(self._h, C.c_void_p(stream.cuda_stream))
        if rc:
            raise RuntimeError(f"ndp_xchg_end failed ({rc})")

    def tick(self, eng, xr_next, gathered_next, stream):
        """end(stream) for the gather begun last + begin() of the next tick's behind the last reader of `gathered_next` (the engine's
        last tracked control step, else `stream`): one call per control tick (ndp_xchg_tick)."""
        import ctypes as C
        if not (xr_next.is_contiguous() and gathered_next.is_contiguous() and xr_next.numel() == self.rows * 10
                and gathered_next.numel() == self.world * self.rows * PV_COLS):
            raise ValueError("RcclExchange.tick: xr_next / gathered_next of the wrong size or not contiguous")
        rc = self._lib.ndp_xchg_tick(self._h, eng._h if eng is not None else None, C.c_void_p(stream.cuda_stream),
                                     C.c_void_p(xr_next.data_ptr()), self.rows, C.c_void_p(gathered_next.data_ptr()))
        if rc:
            raise RuntimeError(f"ndp_xchg_tick failed ({rc}): {self._lib.ndp_xchg_last_error(self._h).decode()}")

    def tick_windows(self, eng, gathered, stream=None):
        """ndp_tick with neighbours on other ranks: the engine's window of this tick (its reference list, behind tick_advance_device)
        packed and all-gathered into `gathered` [world * B_local, N+1, 6] on `stream` -- between tick_advance_device and tick_step_device
        (ndp_xchg_tick_windows)."""
        import ctypes as C
        if not (gathered.is_contiguous() and gathered.numel() == self.world * self.rows * PV_COLS):
            raise ValueError("RcclExchange.tick_windows: gathered of the wrong size or not contiguous")
        rc = self._lib.ndp_xchg_tick_windows(self._h, eng._h, C.c_void_p(gathered.data_ptr()),
                                             C.c_void_p(stream.cuda_stream) if stream is not None else None)
        if rc:
            raise RuntimeError(f"ndp_xchg_tick_windows failed ({rc}): {self._lib.ndp_xchg_last_error(self._h).decode()}")

    def tick_async(self, on=True):
        """Begins launched by a thread of the exchange's own (ndp_xchg_tick_async): the caller's thread then only launches the steps."""
        rc = self._lib.ndp_xchg_tick_async(self._h, 1 if on else 0)
        if rc:
            raise RuntimeError(f"ndp_xchg_tick_async failed ({rc}): {self._lib.ndp_xchg_last_error(self._h).decode()}")

    def tick_begin(self, eng, gathered_next, t=None):
        """The remote tick one control period ahead: list advance to the NEXT period's trajectory time t (scalar, CUDA tensor [B] or None:
        no advance), window columns, all-gather into `gathered_next` -- on the exchange's own stream, beside the control step of the
        current tick (ndp_xchg_tick_begin)."""
        import ctypes as C
        import numpy as np
        import torch
        from . import _lib
        flags = 0
        if t is None:
            tp = None
        elif isinstance(t, torch.Tensor):
            tp = C.c_void_p(t.data_ptr())
        else:
            self._t_host = np.array([float(t)])
            tp, flags = _lib.ptr(self._t_host), _lib.TICK_T_UNIFORM
        rc = self._lib.ndp_xchg_tick_begin(self._h, eng._h, tp, flags, C.c_void_p(gathered_next.data_ptr()))
        if rc:
            raise RuntimeError(f"ndp_xchg_tick_begin failed ({rc}): {eng._lib.ndp_last_error(eng._h).decode()}")

    def tick_step(self, eng, x_odom, cmd_out, gathered, stream=None, estimate=False, u0_out=None):
        """... and the tick itself: estimator (optional), device-side wait for `gathered`'s all-gather, control step + actuator command on
        `stream` (ndp_xchg_tick_step)."""
        import ctypes as C
        from . import _lib
        rc = self._lib.ndp_xchg_tick_step(self._h, eng._h, C.c_void_p(x_odom.data_ptr()), None, None, _lib.TICK_ESTIMATE if estimate else 0,
                                          C.c_void_p(cmd_out.data_ptr()), C.c_void_p(u0_out.data_ptr()) if u0_out is not None else None,
                                          C.c_void_p(gathered.data_ptr()), C.c_void_p(stream.cuda_stream) if stream is not None else None)
        if rc:
            raise RuntimeError(f"ndp_xchg_tick_step failed ({rc}): {eng._lib.ndp_last_error(eng._h).decode()}")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ndp_xchg_destroy(self._h)
            self._h = None


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
    """The neighbour exchange as publish / subscribe over xGMI, one exchange PER CONTROL TICK like the reference's PredXU topic
    (nmpc_node.py:116-133,229-230 publishes a new window every tick; ndp_nmpc_leader_node.py:40,60-76 consumes the latest).

    Every rank owns a buffer with two window slots ([B_local, N+1, 10] float64 each) allocated with ndp_peer_alloc; the 64-byte
    IPC handles are exchanged ONCE (all_gather_object) and each rank maps the buffer of rank (r+1) % W, which holds its
    neighbours.  Per tick, `other = peer.publish_device(xr_tick, stream)` enqueues a copy launch and a one-wave launch that put this
    rank's windows into its own slot, publish the tick number and wait for the neighbour's (csrc/peer_epoch.hpp: epoch words, reader
    acknowledgement before a slot is reused, bounded waits); `other` (DevWindows: a raw device address) is the neighbour's slot
    of that tick -- pass it to BatchedNMPC.update_device on the same stream: the control-step kernel reads it out of the
    neighbour GPU's HBM.  No collective, no host round trip; capturable into a hipGraph holding an even number of ticks.
    With one rank the neighbour is the rank's own buffer (same code path)."""

    def __init__(self, B_local, N, device, group=None, timeout_us=200000):
        import ctypes as C
        import torch
        import torch.distributed as dist
        from . import _lib
        self._lib = _lib.load()
        self.device = int(device)
        self.shape = (int(B_local), int(N) + 1, 10)
        self.n = int(np.prod(self.shape))
        self.timeout_us = int(timeout_us)
        nbytes, off0, stride = C.c_size_t(), C.c_size_t(), C.c_size_t()
        self._lib.ndp_peer_layout(self.n, C.byref(nbytes), C.byref(off0), C.byref(stride))
        self._off = (off0.value, off0.value + stride.value)
        ptr, handle = C.c_void_p(), (C.c_ubyte * 64)()
        rc = self._lib.ndp_peer_alloc(self.device, nbytes.value, C.byref(ptr), handle)
        self._own = ptr.value if rc == 0 else None
        self._mapped = None
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.tick = 0                      # host mirror of the device-side tick count (slot parity of the next launches)
        if self.world > 1:
            # every rank takes part in every collective below whatever happened locally, and all ranks reach the same verdict
            allocs = [None] * self.world
            dist.all_gather_object(allocs, int(rc), group=group)
            if any(allocs):
                self.close(collective=False)
                raise RuntimeError(f"ndp_peer_alloc failed on some rank: {allocs}")
        elif rc:
            raise RuntimeError(f"ndp_peer_alloc failed ({rc})")
        if self.world == 1:       # the neighbours are this rank's own vehicles: the same protocol over the own buffer
            self._nb = self._own
        else:
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
                self.close(collective=False)
                raise RuntimeError(f"ndp_peer_open failed on some rank: {ok}")
            self._mapped = mp.value
            self._nb = self._mapped
        self.neighbour = [DevWindows(self._nb + self._off[s], self.shape) for s in (0, 1)]
        with torch.cuda.device(self.device):       # this rank's own slots as tensors (tests read them back)
            self.local = [torch.as_tensor(_DevMem(self._own + self._off[s], self.shape), device=torch.device("cuda", self.device))
                          for s in (0, 1)]

    def publish_device(self, xr, stream=None):
        """Enqueues this tick's publish launch on `stream` (xr: this rank's [B_local, N+1, 10] float64 CUDA windows of the tick)
        and returns the neighbour's windows of the same tick (DevWindows) for the control step enqueued next on that stream."""
        import ctypes as C
        import torch
        if not (isinstance(xr, torch.Tensor) and xr.is_cuda and xr.is_contiguous() and xr.dtype == torch.float64
                and tuple(xr.shape) == self.shape):
            raise ValueError(f"expected a contiguous CUDA float64 tensor {self.shape}")
        self.tick += 1
        slot = self.tick & 1
        sp = None if stream is None else C.c_void_p(stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream))
        rc = self._lib.ndp_peer_publish_device(self.device, C.c_void_p(xr.data_ptr()), self.n, C.c_void_p(self._own),
                                               C.c_void_p(self._nb), slot, self.timeout_us, sp)
        if rc:
            raise RuntimeError(f"ndp_peer_publish_device failed ({rc})")
        return self.neighbour[slot]

    def stats(self):
        """Synchronises the device; counters of this rank's publish launches."""
        import ctypes as C
        out = (C.c_ulonglong * 4)()
        rc = self._lib.ndp_peer_stats(self.device, C.c_void_p(self._own), out)
        if rc:
            raise RuntimeError(f"ndp_peer_stats failed ({rc})")
        return dict(ticks=int(out[0]), ack_timeouts=int(out[1]), epoch_timeouts=int(out[2]), slot_mismatches=int(out[3]))

    def close(self, collective=True):
        """Synchronise, unmap the neighbour's buffer, wait for every rank to have done so, then free the own buffer: the rank
        that reads it may still have kernels in flight.  collective=False (error paths, or a rank leaving on its own) skips the
        barrier -- the readers' mapping keeps the memory alive either way."""
        import torch
        import torch.distributed as dist
        if self._own or self._mapped:
            torch.cuda.synchronize(self.device)
        if self._mapped:
            self._lib.ndp_peer_close(self.device, self._mapped)
            self._mapped = None
        if collective and self.world > 1 and dist.is_initialized():
            dist.barrier(group=self.group)
        if self._own:
            self.local, self.neighbour = None, None
            self._lib.ndp_peer_free(self.device, self._own)
            self._own = None
