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


def test_shared_memory_gather_between_processes():
    ctx = mp.get_context("spawn")
    world, rounds = 3, 400
    name = "/humid_test_%d" % os.getpid()
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(name, r, world, rounds, q)) for r in (1, 2, 0)]   # rank 0 comes last
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert got == {0: True, 1: True, 2: True}, got
    assert not os.path.exists("/dev/shm" + name)                      # rank 0 unlinked it on close
