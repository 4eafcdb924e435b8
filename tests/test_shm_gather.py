"""The shared-memory host_all_gather of include/humid_hip.h (humid_shm_*): ranks = processes of one node.
No GPU is touched: runs in the CPU suite."""
import ctypes as C
import multiprocessing as mp
import os

import numpy as np


def _rank_main(name, rank, world, rounds, q):
    try:
        from humid_amd import _lib
        lib = _lib.load(import_torch=False)
        h = C.c_void_p()
        rc = lib.humid_shm_open(C.byref(h), name.encode(), rank, world, 1 << 16)
        assert rc == 0, rc
        ok = True
        for k in range(rounds):
            n = 1 + (k * 37) % 4000                                  # sizes vary from call to call
            mine = (np.arange(n, dtype=np.uint32) * 7 + rank * 1000003 + k).astype(np.uint32)
            out = np.zeros(world * n, dtype=np.uint32)
            rc = lib.humid_shm_all_gather(h, mine.ctypes.data_as(C.c_void_p), mine.nbytes, out.ctypes.data_as(C.c_void_p))
            ok = ok and rc == 0
            for r in range(world):
                exp = (np.arange(n, dtype=np.uint32) * 7 + r * 1000003 + k).astype(np.uint32)
                ok = ok and np.array_equal(out[r * n:(r + 1) * n], exp)
        big = np.zeros((1 << 16) + 8, dtype=np.uint8)                 # more than a slot holds: refused, nobody waits
        rc_big = lib.humid_shm_all_gather(h, big.ctypes.data_as(C.c_void_p), big.nbytes, big.ctypes.data_as(C.c_void_p))
        lib.humid_shm_close(h)
        q.put((rank, ok and rc_big < 0))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))


import pytest  # noqa: E402


@pytest.mark.parametrize("world,rounds", [(3, 400), (8, 150), (16, 60)])
def test_shared_memory_gather_between_processes(world, rounds):
    """3 ranks; 8 (one node's GPUs: what bench.py --gpus 8 opens); 16 (the largest group of the exchange pass, here
    more processes than cores: waiting ranks give their core up)"""
    ctx = mp.get_context("spawn")
    name = "/humid_test_%d" % os.getpid()
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(name, r, world, rounds, q)) for r in list(range(1, world)) + [0]]   # rank 0 comes last
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert got == {r: True for r in range(world)}, got
    assert not os.path.exists("/dev/shm" + name)                      # rank 0 unlinked it on close


def _abort_main(name, rank, world, q):
    try:
        from humid_amd import _lib
        lib = _lib.load(import_torch=False)
        h = C.c_void_p()
        assert lib.humid_shm_open(C.byref(h), name.encode(), rank, world, 4096) == 0
        x = np.full(4, rank, dtype=np.uint32)
        out = np.zeros(4 * world, dtype=np.uint32)
        rc1 = lib.humid_shm_all_gather(h, x.ctypes.data_as(C.c_void_p), x.nbytes, out.ctypes.data_as(C.c_void_p))
        if rank == 1:
            lib.humid_shm_abort(h)                                    # leaves the group instead of joining the second gather
            rc2 = -1
        else:
            rc2 = lib.humid_shm_all_gather(h, x.ctypes.data_as(C.c_void_p), x.nbytes, out.ctypes.data_as(C.c_void_p))
        lib.humid_shm_close(h)
        q.put((rank, (rc1, rc2)))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))


def test_stale_segment_is_not_attached_to_and_abort_wakes_the_waiting_ranks():
    """(1) a segment of the same name left by a crashed run -- its arrival counters far ahead --
    must not be taken for rank 0's: the other ranks start FIRST here, map the stale file, get no echo and map again
    once rank 0 has replaced it; (2) a rank that gives the group up (humid_shm_abort) ends the others' wait at once
    instead of after the 120 s spin limit."""
    import time
    ctx = mp.get_context("spawn")
    world = 3
    name = "/humid_stale_%d" % os.getpid()
    size = 64 * (world + 1) + 2 * world * 4096
    with open("/dev/shm" + name, "wb") as f:                          # the stale segment: right size, counters at 10^6
        stale = np.zeros(size // 8, dtype=np.uint64)
        stale[0:8 * world:8] = 1_000_000
        f.write(stale.tobytes())
    q = ctx.Queue()
    others = [ctx.Process(target=_abort_main, args=(name, r, world, q)) for r in (1, 2)]
    for p in others:
        p.start()
    time.sleep(1.0)                                                   # they have mapped the stale file by now
    p0 = ctx.Process(target=_abort_main, args=(name, 0, world, q))
    p0.start()
    t0 = time.time()
    got = dict(q.get(timeout=60) for _ in range(world))
    for p in others + [p0]:
        p.join(timeout=30)
    assert got[1] == (0, -1) and got[0] == (0, -1) and got[2] == (0, -1), got
    assert time.time() - t0 < 30                                      # (not the 120 s spin limit)
