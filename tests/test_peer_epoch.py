"""The per-tick neighbour exchange's ordering protocol (csrc/peer_epoch.hpp) between two CPU processes.

The product runs PeerProto inside peer_publish_kernel (ndp_hip.hip) on buffers mapped across GPUs; tests/emu/peer_emu.cpp runs
the SAME protocol text on a CPU memory backend over POSIX shared memory, so that gloo ranks can check -- without a GPU --
what the reference gets from its ROS topic (nmpc_node.py:116-133,229-230 -> ndp_nmpc_leader_node.py:40,60-76): the reader of
tick t sees exactly the windows the neighbour published for tick t, slots are not overwritten while they are being read, and
a rank whose peer has gone away keeps stepping on the last windows it got (counted, not silent)."""
import ctypes as C
import os
import socket
import subprocess
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _emu():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "emu"), "libpeer_emu.so"], stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(HERE, "emu", "libpeer_emu.so"))
    lib.peer_emu_buffer_bytes.argtypes = [C.c_size_t]
    lib.peer_emu_buffer_bytes.restype = C.c_size_t
    lib.peer_emu_slot_offset.argtypes = [C.c_size_t, C.c_int]
    lib.peer_emu_slot_offset.restype = C.c_size_t
    lib.peer_emu_publish.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_uint]
    lib.peer_emu_publish.restype = C.c_ulonglong
    lib.peer_emu_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    return lib


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _window(rank, t, n):
    """What rank publishes at tick t: every element carries (rank, t), so a torn or stale read is visible anywhere."""
    return np.full(n, 1000.0 * rank + t) + np.arange(n) * 1e-6


def _addr(shm):
    return C.addressof(C.c_char.from_buffer(shm.buf))


def _stats(lib, shm):
    out = (C.c_ulonglong * 4)()
    lib.peer_emu_stats(_addr(shm), out)
    return dict(ticks=out[0], ack_timeouts=out[1], epoch_timeouts=out[2], slot_mismatches=out[3])


def _worker(rank, world, port, q, ticks, leave_after):
    """Every rank publishes its window per tick and reads rank (r+1) % W's; random delays on both sides.  leave_after: rank 1
    stops publishing after that many ticks and closes its buffer (rank 0 keeps stepping)."""
    from multiprocessing import shared_memory
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = _emu()
    n = 210                                       # one vehicle's [21][10] window
    own = shared_memory.SharedMemory(create=True, size=lib.peer_emu_buffer_bytes(n))
    own.buf[:] = bytes(len(own.buf))
    nb = None
    try:
        names = [None] * world
        dist.all_gather_object(names, own.name)   # the IPC handle exchange, once
        nb = shared_memory.SharedMemory(name=names[(rank + 1) % world])
        rng = np.random.default_rng(100 + rank)
        off = [lib.peer_emu_slot_offset(n, s) for s in (0, 1)]
        bad, stale = 0, 0
        timeout_us = 30000 if leave_after else 5000000
        my_ticks = leave_after if (leave_after and rank == 1) else ticks
        for t in range(1, my_ticks + 1):
            if rng.random() < 0.3:
                time.sleep(rng.random() * 2e-3)   # the ranks drift apart: sometimes the writer is late, sometimes the reader
            src = _window(rank, t, n)
            got_t = lib.peer_emu_publish(src.ctypes.data_as(C.c_void_p), n, _addr(own), _addr(nb), t & 1, timeout_us)
            assert got_t == t
            if rng.random() < 0.3:
                time.sleep(rng.random() * 2e-3)   # a slow reader: the slot must survive until it has been read (the acknowledgement)
            # the control step's read of the neighbour's slot of tick t (in the product: the kernel launched next)
            seen = np.frombuffer(nb.buf, dtype=np.float64, count=n, offset=off[t & 1]).copy()
            want = _window((rank + 1) % world, t, n)
            if not np.array_equal(seen, want):
                whole = np.unique(np.round(seen - np.arange(n) * 1e-6)).size == 1      # one consistent (older) window, not a torn one
                if leave_after and rank == 0 and t > leave_after and whole:
                    stale += 1                    # the publisher has left: its last windows are what there is
                else:
                    bad += 1
        st = _stats(lib, own)
        q.put((rank, bad, stale, st))
        dist.barrier()
    finally:
        if nb is not None:
            nb.close()
        own.close()
        own.unlink()
        dist.destroy_process_group()


def _run(world, ticks, leave_after=0):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, ticks, leave_after)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_reader_sees_the_writers_tick_world_size_2():
    """400 ticks, both ranks publishing and reading with random delays: every read of tick t returns the neighbour's windows
    of tick t -- never the slot's previous content (tick t-2), never a half-overwritten one; no wait timed out."""
    res = _run(2, 400)
    for rank, bad, stale, st in res:
        assert bad == 0 and stale == 0, (rank, bad, stale)
        assert st["ticks"] == 400 and st["ack_timeouts"] == 0 and st["epoch_timeouts"] == 0 and st["slot_mismatches"] == 0, st


def test_ring_of_three_ranks():
    res = _run(3, 150)
    for rank, bad, stale, st in res:
        assert bad == 0 and st["ticks"] == 150 and st["epoch_timeouts"] == 0 and st["ack_timeouts"] == 0, (rank, bad, st)


def test_reader_survives_the_publisher_leaving():
    """Rank 1 publishes 20 ticks and stops; rank 0 goes on for 40: its waits for rank 1's epoch time out (counted), it reads
    rank 1's last published windows (whole, never torn) -- what the reference's subscriber does when no new PredXU arrives
    (ndp_nmpc_leader_node.py:60-76 keeps the last message) -- and its own publishing is not blocked by the missing reader
    acknowledgements beyond the bounded wait."""
    res = _run(2, 40, leave_after=20)
    (r0, bad0, stale0, st0), (r1, bad1, stale1, st1) = res
    assert bad0 == 0 and bad1 == 0
    assert st1["ticks"] == 20
    assert st0["ticks"] == 40
    assert stale0 >= 17 and st0["epoch_timeouts"] >= 17, (stale0, st0)      # ticks 21..40 (rank 1 may have been a tick ahead)
    assert st0["ack_timeouts"] >= 15, st0                                    # nobody acknowledges rank 0's slots any more


def test_layout_matches_the_c_abi():
    """The emulation and the product agree on where the slots are (same header: peer_epoch.hpp)."""
    from ndp_nmpc_qd_amd import _lib
    lib, emu = _lib.load(), _emu()
    for n in (210, 1024 * 210, 12345):
        nbytes, off0, stride = C.c_size_t(), C.c_size_t(), C.c_size_t()
        lib.ndp_peer_layout(n, C.byref(nbytes), C.byref(off0), C.byref(stride))
        assert nbytes.value == emu.peer_emu_buffer_bytes(n)
        assert off0.value == emu.peer_emu_slot_offset(n, 0) == 512
        assert off0.value + stride.value == emu.peer_emu_slot_offset(n, 1)
        assert stride.value >= 8 * n and stride.value % 256 == 0
