"""The N>1 path on CPU: world_size 2 and 3 over gloo, orchestration from humid_amd/sharded.py,
stage compute from tests/cpu_stage_ops.py (numpy + oracle).  Every rank's shard results must be
bit-identical to a single-process oracle run over the whole read set."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cpu_stage_ops import CpuStageOps
        from humid_amd.sharded import ShardedDedup
        from humid_amd.synth import synth_words
        from oracle import pyoracle as orc
        n_reads, n, d, method, sizes, mode, p_sub, dense = case
        edit = dense == "edit"
        shard_mode = "exchange" if dense == "exchange" else "allgather"
        dense = bool(dense)
        words, filt = synth_words(n_reads, 4242, n, p_sub=p_sub, p_n=2e-3, mode=mode, genome_bp=3000)
        ocid, okeep, osum, _ = orc.dedup_run(words, filt, n, d, method, edit=edit)
        if sizes is None:
            base = n_reads // world
            sizes = [base] * (world - 1) + [n_reads - base * (world - 1)]
        off = sum(sizes[:rank])
        w = torch.from_numpy(words[off:off + sizes[rank]].view(np.int64).copy())
        f = torch.from_numpy(filt[off:off + sizes[rank]].copy())
        cid = torch.zeros(sizes[rank], dtype=torch.int32)
        keep = torch.zeros(sizes[rank], dtype=torch.uint8)
        sd = ShardedDedup(word_nt=n, distance=d, method=method, ops=CpuStageOps(), dense_return=dense,
                          partition_search=dense,   # old pair: replicated search + reduce-scatter
                          mode=shard_mode, edit=edit)
        for _ in range(2):          # second pass re-uses the instance (cached shard sizes)
            s = sd.run(w, f, cid, keep)
        ok = (np.array_equal(cid.numpy().view(np.uint32), ocid[off:off + sizes[rank]]) and
              np.array_equal(keep.numpy(), okeep[off:off + sizes[rank]]))
        ok = ok and all(s[k] == osum[k] for k in ("total", "usable", "unique", "clusters"))
        ok = ok and (sd.mode_used == shard_mode or (shard_mode == "exchange" and d >= n))
        q.put((rank, bool(ok), {k: s[k] for k in ("total", "usable", "unique", "clusters")}, osum))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc(), str(e)))
    finally:
        dist.destroy_process_group()


CASES = [
    # reads, word_nt, d, method, shard sizes, mode, p_sub
    (6000, 24, 1, 0, None, "umi", 5e-3),
    (5000, 12, 2, 0, None, "umi", 1e-2),
    (4000, 24, 2, 1, None, "genome", 5e-3),
    (3001, 8, 1, 0, "uneven", "umi", 1e-2),
    (300, 32, 1, 0, None, "umi", 1e-2),
    (40, 3, 1, 0, None, "umi", 0.0),       # tiny word space: most ranges empty
    (500, 2, 2, 0, None, "umi", 0.0),      # d >= n: no prefix, exchange mode falls back to the all-gather
    (3000, 16, 3, 1, None, "umi", 2e-2),   # four combinations
]


def _matrix():
    """world 2: every case in every mode; world 3: every case in exchange mode, a few in the
    all-gather mode with the dense return (keeps the CPU suite within a couple of minutes)"""
    names = {"exchange": "exchange", True: "allgather_dense_return", False: "allgather_reduce_scatter",
             "edit": "edit_distance"}
    out = []
    for ci, case in enumerate(CASES):
        for dense in ("exchange", True, False):
            out.append(pytest.param(2, case, dense, id="case%d-%s-2" % (ci, names[dense])))
        out.append(pytest.param(3, case, "exchange", id="case%d-exchange-3" % ci))
        if ci in (1, 7):                                      # d = 2 and d = 3 cases: Levenshtein neighbours
            out.append(pytest.param(2, case, "edit", id="case%d-edit_distance-2" % ci))
        if ci in (0, 1, 3, 5):
            out.append(pytest.param(3, case, True, id="case%d-allgather_dense_return-3" % ci))
    # one node's worth of ranks as real processes (what `bench.py --gpus 8` starts, minus the GPUs), and an odd count
    out.append(pytest.param(8, CASES[0], "exchange", id="case0-exchange-8"))
    out.append(pytest.param(8, CASES[1], True, id="case1-allgather_dense_return-8"))
    out.append(pytest.param(5, CASES[3], "exchange", id="case3-exchange-5"))
    return out


@pytest.mark.parametrize("world,case,dense", _matrix())
def test_sharded_matches_single_process(world, case, dense):
    case = list(case) + [dense]
    if case[4] == "uneven":
        n = case[0]
        case[4] = [n // 5] + [n - n // 5 - 7 * (world - 2)] + [7] * (world - 2)
        assert sum(case[4]) == n
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, tuple(case), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, got, want in sorted(res):
        assert ok, (rank, got, want)


def test_splitters_cover_and_balance():
    from humid_amd.sharded import splitters_from_hist
    rng = np.random.default_rng(0)
    for world in (1, 2, 3, 8):
        for bits, n in ((12, 24), (12, 32), (6, 3), (2, 1)):
            hist = rng.integers(0, 100, size=1 << bits)
            hist[rng.random(1 << bits) < 0.5] = 0
            rs = splitters_from_hist(hist, world, n, bits)
            assert len(rs) == world
            assert sum(e for _, _, e in rs) == int(hist.sum())
            nxt = 0
            for lo, hi, e in rs:
                if lo > hi:
                    assert e == 0
                    continue
                assert lo == nxt or e == 0 or lo >= nxt
                assert hi >= lo
                nxt = hi + 1
            live = [x for x in rs if x[0] <= x[1]]
            assert live[0][0] == 0 and live[-1][1] >= 4 ** n - 1   # every valid word is covered
            for a, b in zip(live[:-1], live[1:]):
                assert a[1] + 1 == b[0]
    # skew: everything in one bin -> one rank owns it all, still a cover
    hist = np.zeros(4096, dtype=np.int64)
    hist[77] = 1000
    rs = splitters_from_hist(hist, 8, 24, 12)
    assert sum(e for _, _, e in rs) == 1000 and max(e for _, _, e in rs) == 1000
