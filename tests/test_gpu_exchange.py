"""-m gpu: the multi-GPU orchestration (humid_amd/sharded.py) with the REAL HIP stage ops for
P = 1..8 ranks on one GPU -- the ranks are threads, collectives come from tests/fake_dist.py.
Every rank's shard must be bit-identical to a single-process oracle run over the whole read set."""
import threading

import numpy as np
import pytest

from humid_amd.synth import synth_words
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu


def run_ranks(P, words, filt, n, d, method, mode, sizes=None, plan_segments=0, passes=1, edit=False, bucket_walk=None):
    import torch
    from fake_dist import FakeDist, FakeWorld
    from humid_amd.sharded import HipStageOps, ShardedDedup
    dev = torch.device("cuda:0")
    N = len(words)
    if sizes is None:
        sizes = [N // P] * (P - 1) + [N - (N // P) * (P - 1)]
    offs = [sum(sizes[:q]) for q in range(P + 1)]
    world = FakeWorld(P)
    out, errs = [None] * P, []

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            ops = HipStageOps(0)
            if plan_segments:
                ops.set_option("plan_segments", plan_segments)
            if bucket_walk is not None:
                ops.set_option("bucket_walk", bucket_walk)
            sd = ShardedDedup(device=0, word_nt=n, distance=d, method=method, ops=ops,
                              dist=FakeDist(world, r), mode=mode, edit=edit)
            w = torch.from_numpy(words[offs[r]:offs[r + 1]].view(np.int64).copy()).to(dev)
            f = torch.from_numpy(filt[offs[r]:offs[r + 1]].copy()).to(dev)
            c = torch.zeros(sizes[r], dtype=torch.int32, device=dev)
            k = torch.zeros(sizes[r], dtype=torch.uint8, device=dev)
            for _ in range(passes):
                s = sd.run(w, f, c, k)
            torch.cuda.synchronize()
            if sd.mode_used == "exchange" and s["usable"]:
                s = dict(s, **ops.kernel_ms())
            out[r] = (c.cpu().numpy().view(np.uint32), k.cpu().numpy(), s, sd.mode_used)
            ops.close()
        except Exception:  # pragma: no cover
            import traceback
            errs.append((r, traceback.format_exc()))
            world.barrier.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not errs, errs[0][1]
    return out, offs


def check(P, words, filt, n, d, method, mode="exchange", expect_mode=None, **kw):
    ocid, okeep, osum, _ = orc.dedup_run(words, filt, n, d, method)
    p = orc.Pipeline(n)
    p.read_data(words, filt)
    edges = p.find_hamming_neighbours(d)
    out, offs = run_ranks(P, words, filt, n, d, method, mode, **kw)
    for r in range(P):
        cid, keep, s, used = out[r]
        assert used == (expect_mode or mode)
        assert np.array_equal(cid, ocid[offs[r]:offs[r + 1]]), ("cluster_id", r)
        assert np.array_equal(keep, okeep[offs[r]:offs[r + 1]]), ("keep", r)
        for k in ("total", "usable", "unique", "clusters"):
            assert s[k] == osum[k], (k, s[k], osum[k])
        assert s["edges"] == p.n_edges


@pytest.mark.parametrize("P", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("cfg", [(200_000, 24, 1, "umi", 0), (60_000, 12, 2, "umi", 1),
                                 (50_000, 32, 1, "umi", 0), (3000, 4, 1, "umi", 0),
                                 (80_000, 24, 2, "genome", 0), (40_000, 16, 3, "umi", 0)])
def test_exchange_mode_virtual_ranks(P, cfg):
    n_reads, n, d, mode, method = cfg
    words, filt = synth_words(n_reads, 177 + P, n, p_sub=5e-3, p_n=1e-3, mode=mode, genome_bp=20000)
    check(P, words, filt, n, d, method)


def test_exchange_ranges_use_word_ordered_buckets():
    """inside its value range a rank still gets word-ordered LDS buckets (keys = (word - lo) * scale):
    count_mode_used 2, no unique sort -- and an amplicon-like prefix falls back to hashed buckets"""
    words, filt = synth_words(1_200_000, 61, 24, p_sub=2e-3, p_n=1e-3)
    ocid, okeep, osum, _ = orc.dedup_run(words, filt, 24, 1, 0)
    out, offs = run_ranks(4, words, filt, 24, 1, 0, "exchange")
    for r in range(4):
        cid, keep, s, used = out[r]
        assert np.array_equal(cid, ocid[offs[r]:offs[r + 1]]) and np.array_equal(keep, okeep[offs[r]:offs[r + 1]])
        assert s["count_mode_used"] == 2, s
    skew = (words & np.uint64(0xffffff)) | (np.uint64(0x5a5a5a) << np.uint64(24))      # constant 12-nt prefix
    ocid, okeep, osum, _ = orc.dedup_run(skew, filt, 24, 1, 0)
    out, offs = run_ranks(3, skew, filt, 24, 1, 0, "exchange")
    for r in range(3):
        cid, keep, s, used = out[r]
        assert np.array_equal(cid, ocid[offs[r]:offs[r + 1]]) and np.array_equal(keep, okeep[offs[r]:offs[r + 1]])
        assert s["count_mode_used"] in (0, 1) or s["ms_k_insert"] == 0.0     # ranks without reads: nothing ran


@pytest.mark.parametrize("P", [16, 11])
@pytest.mark.parametrize("cfg", [(240_000, 24, 1, 0), (120_000, 24, 2, 1), (90_000, 40, 1, 0)])
def test_exchange_sixteen_ranks(cfg, P):
    """the largest group the pass takes (MAX_RANKS = 16 thread ranks on the one GPU) and an odd one between 8 and 16:
    sixteen value ranges, pairs that cross them in the first two nucleotides, owner-local clustering with fifteen
    foreign ranges -- one- and two-word words, both methods, every shard against the oracle.  (Round 3 ran the whole
    pass with at most 8 ranks until the end; at exactly 16 the cursor of the records that go to EVERY rank and the
    count of the flagged interior records shared a word of a small counter block: the pass failed its own check of
    the record indices.  Found by this test.)"""
    n_reads, n, d, method = cfg
    if n > 32:
        from humid_amd.synth import synth_wide_words
        words, filt = synth_wide_words(n_reads, 1600 + n, n, p_sub=5e-3, p_n=1e-3)
        ocid, okeep, osum, _ = orc.dedup_run(words, filt, n, d, method)
        out, offs = run_ranks(P, words, filt, n, d, method, "exchange")
        for r in range(P):
            cid, keep, s, used = out[r]
            assert used == "exchange"
            assert np.array_equal(cid, ocid[offs[r]:offs[r + 1]]) and np.array_equal(keep, okeep[offs[r]:offs[r + 1]])
            assert s["edges"] == osum["edges"] and s["clusters"] == osum["clusters"] and s["unique"] == osum["unique"]
        return
    words, filt = synth_words(n_reads, 1600 + d, n, p_sub=5e-3, p_n=1e-3)
    check(P, words, filt, n, d, method)


def test_exchange_uneven_shards_and_reuse():
    words, filt = synth_words(100_003, 5, 24, p_sub=5e-3, p_n=1e-3)
    check(4, words, filt, 24, 1, 0, sizes=[50_000, 3, 0, 50_000], passes=2)
    # sixteen ranks, among them empty shards, one-read shards and one that holds most of the reads
    sizes = [0, 1, 0, 20_000, 3, 0, 0, 61_000, 2, 5000, 0, 7, 10_000, 1, 0, 0]
    sizes[7] += 100_003 - sum(sizes)
    check(16, words, filt, 24, 1, 0, sizes=sizes, passes=2)


def test_exchange_skewed_words():
    """nearly all words in one value range / one combination bucket: owners are unbalanced, results
    must not change"""
    rng = np.random.default_rng(3)
    base = np.uint64(0x00ab_cdef_0123)
    w = np.full(30_000, base, dtype=np.uint64)
    for pos in (0, 5, 13, 22):
        sh = np.uint64(2 * pos)
        w = (w & ~(np.uint64(3) << sh)) | (rng.integers(0, 4, size=len(w)).astype(np.uint64) << sh)
    w[::50] = rng.integers(0, 1 << 48, size=len(w[::50]), dtype=np.uint64)
    f = (rng.random(len(w)) < 0.01).astype(np.uint8)
    check(4, w, f, 24, 1, 0)
    check(3, w, f, 24, 2, 1)
    check(16, w, f, 24, 1, 0)        # most of the sixteen value ranges hold next to nothing
    check(13, w, f, 24, 2, 1)


def test_exchange_all_filtered_and_tiny():
    w = np.zeros(40, dtype=np.uint64)
    check(2, w, np.ones(40, np.uint8), 24, 1, 0)
    check(4, np.arange(3, dtype=np.uint64), np.zeros(3, np.uint8), 24, 1, 0)
    check(16, np.arange(5, dtype=np.uint64), np.zeros(5, np.uint8), 24, 1, 0)      # fewer reads than ranks
    check(16, w, np.ones(40, np.uint8), 24, 1, 0)
    check(2, w, np.zeros(40, np.uint8), 24, 0, 0)        # d = 0: no pairs at all


def test_exchange_falls_back_without_prefix():
    words, filt = synth_words(2000, 9, 2, p_sub=0.0)
    check(2, words, filt, 2, 3, 0, expect_mode="allgather")


@pytest.mark.parametrize("segs", [3, 4, 6])
def test_exchange_forced_plans(segs):
    words, filt = synth_words(60_000, 21 + segs, 24, p_sub=1e-2, p_n=1e-3)
    check(3, words, filt, 24, 2, 0, plan_segments=segs)
    check(16, words, filt, 24, 2, 0, plan_segments=segs)


def test_allgather_mode_virtual_ranks():
    words, filt = synth_words(120_000, 8, 24, p_sub=5e-3, p_n=1e-3)
    check(3, words, filt, 24, 1, 0, mode="allgather")


@pytest.mark.parametrize("P,mode,expect", [(16, "allgather", "allgather"), (17, "exchange", "allgather"),
                                           (24, "allgather", "allgather")])
@pytest.mark.parametrize("cfg", [(150_000, 24, 1, 0), (60_000, 40, 2, 1)])
def test_more_ranks_than_the_exchange_pass_takes(P, mode, expect, cfg):
    """groups at and beyond MAX_RANKS = 16: a request for the exchange pass with 17 ranks runs the all-gather stages
    (DESIGN 4b), and those stages with 16 and 24 ranks -- one- and two-word words, every shard against the oracle"""
    n_reads, n, d, method = cfg
    if n > 32 and P > 16:
        from fake_dist import FakeDist, FakeWorld
        from humid_amd.sharded import HipStageOps, ShardedDedup
        ops = HipStageOps(0)
        with pytest.raises(NotImplementedError, match="at most 16 ranks"):      # a stated limit, refused up front
            ShardedDedup(device=0, word_nt=n, distance=d, method=method, ops=ops, dist=FakeDist(FakeWorld(P), 3), mode=mode)
        ops.close()
        return
    if n > 32:
        from humid_amd.synth import synth_wide_words
        words, filt = synth_wide_words(n_reads, 1700 + P + d, n, p_sub=5e-3, p_n=1e-3)
        ocid, okeep, osum, _ = orc.dedup_run(words, filt, n, d, method)
        out, offs = run_ranks(P, words, filt, n, d, method, mode)
        for r in range(P):
            cid, keep, s, used = out[r]
            assert used == expect
            assert np.array_equal(cid, ocid[offs[r]:offs[r + 1]]) and np.array_equal(keep, okeep[offs[r]:offs[r + 1]])
            assert s["edges"] == osum["edges"] and s["clusters"] == osum["clusters"] and s["unique"] == osum["unique"]
        return
    words, filt = synth_words(n_reads, 1700 + P + d, n, p_sub=5e-3, p_n=1e-3)
    check(P, words, filt, n, d, method, mode=mode, expect_mode=expect)


@pytest.mark.parametrize("P,mode,expect", [(16, "exchange", "exchange"), (16, "allgather", "allgather"), (17, None, "allgather")])
def test_edit_distance_sixteen_ranks(P, mode, expect):
    """-e with the join dealt out over 16 ranks, and over 17 (no mode named: the all-gather stages)"""
    from test_oracle_vs_bruteforce import indel_words
    rng = np.random.default_rng(1616)
    words = indel_words(rng, 30_000, 20, p_indel=0.4)
    filt = (rng.random(len(words)) < 0.01).astype(np.uint8)
    ocid, okeep, osum, _ = orc.dedup_run(words, filt, 20, 2, 0, edit=True)
    out, offs = run_ranks(P, words, filt, 20, 2, 0, mode, edit=True)
    for r in range(P):
        cid, keep, s, used = out[r]
        assert used == expect
        assert np.array_equal(cid, ocid[offs[r]:offs[r + 1]]) and np.array_equal(keep, okeep[offs[r]:offs[r + 1]])
        assert s["clusters"] == osum["clusters"] and s["edges"] == osum["edges"]


def test_exchange_entry_points_reject_bad_arguments():
    """the new stage entry points fail with a code and a message, never with a fault"""
    import torch
    import humid_amd
    from humid_amd.sharded import HipStageOps
    ops = HipStageOps(0)
    dev = torch.device("cuda:0")
    w = torch.arange(100, dtype=torch.int64, device=dev)
    c = torch.ones(100, dtype=torch.int32, device=dev)
    nc, pb = ops.plan_info(24, 1, 1000)
    assert nc == 2 and pb == 24
    assert ops.plan_info(2, 3, 10) == (1, 0)                       # d >= n: one empty-mask combination
    with pytest.raises(humid_amd.HumidError) as e:
        ops.combo_route(w, c, 0, 24, 1, 1000, 5, 4)                # combination out of range
    assert e.value.code == -1
    with pytest.raises(humid_amd.HumidError) as e:
        ops.combo_route(w, c, (1 << 32) - 50, 24, 1, 1000, 1, 4)   # global index beyond 32 bits
    assert e.value.code == -5
    with pytest.raises(humid_amd.HumidError) as e:
        ops.pairs_keyed(w, False, 0, c, 24, 1, 1000, 1)            # plain array with a sorted combination
    assert e.value.code == -1
    with pytest.raises(humid_amd.HumidError) as e:
        ops.pairs_keyed(w, False, 0, c, 33, 1, 1000, 0)            # wide words: single-GPU only
    assert e.value.code == -2
    with pytest.raises(humid_amd.HumidError) as e:
        ops.kernel_ms()                                            # nothing counted yet
    assert e.value.code == -6
    # an all-owned count with a word outside the promised range is refused
    with pytest.raises(humid_amd.HumidError) as e:
        ops.count_dense(w, None, 24, 10, 50, [0, 100])
    assert e.value.code == -1
    # words 0..99 are pairwise neighbours in the low nucleotides: a real answer comes back
    u, usable, _ = ops.count_dense(w, None, 24, 0, 99, [0, 100])
    assert (u, usable) == (100, 100)
    rec = ops.pairs_keyed(w, False, 7, c, 24, 1, 100, 0)
    e_ids = rec[:, 0].cpu().numpy()
    a, b = e_ids >> 32, e_ids & 0xffffffff
    assert len(a) and (a < b).all() and a.min() >= 7 and b.max() <= 106
    assert (rec[:, 1].cpu().numpy() == (1 | (1 << 32))).all()
    nodes, cedges, cnt = ops.compact_nodes(rec)
    assert nodes.numel() == 100 and (cnt.cpu().numpy() == 1).all()
    assert (cedges.cpu().numpy() >> 32).max() < 100
    ops.close()


@pytest.mark.parametrize("P,d,mode", [(2, 2, "exchange"), (3, 3, "exchange"), (5, 2, "exchange"), (5, 3, "exchange"), (3, 6, "exchange"),
                                      (2, 3, "allgather"), (3, 2, "allgather"), (5, 6, "allgather")])
def test_edit_distance_virtual_ranks(P, d, mode):
    """-e on several ranks: the joins of the Levenshtein search are dealt out over the ranks, shares gathered and
    made unique -- inside the library's exchange pass (round 3: it all-gathers the unique words) and stage by
    stage in the all-gather mode; every shard bit-identical to the oracle's -e run"""
    from test_oracle_vs_bruteforce import indel_words
    rng = np.random.default_rng(40 + P + d)
    words = indel_words(rng, 20_000 if d < 6 else 6000, 20, p_indel=0.4)
    filt = (rng.random(len(words)) < 0.01).astype(np.uint8)
    ocid, okeep, osum, _ = orc.dedup_run(words, filt, 20, d, 0, edit=True)
    out, offs = run_ranks(P, words, filt, 20, d, 0, mode, edit=True)
    for r in range(P):
        cid, keep, s, used = out[r]
        assert used == mode
        assert np.array_equal(cid, ocid[offs[r]:offs[r + 1]]) and np.array_equal(keep, okeep[offs[r]:offs[r + 1]])
        assert s["clusters"] == osum["clusters"] and s["edges"] == osum["edges"]


@pytest.mark.parametrize("P", [1, 2, 5, 16])
@pytest.mark.parametrize("aligned", [True, False])
def test_route_is_the_stable_owner_major_order(P, aligned):
    """humid_stage_route alone: routed words = the usable words in owner-major order, input order inside
    every owner's block, perm = their read indices.  aligned: value ranges cut at the bins of a 12-bit
    prefix histogram (owners from the LDS table); otherwise arbitrary cuts (the compare loop).  A wrong
    count must be reported by humid_stage_route_check."""
    import torch
    import humid_amd
    from humid_amd.sharded import HipStageOps
    rng = np.random.default_rng(P * 2 + aligned)
    n_reads, n = 300_001, 24
    words, filt = synth_words(n_reads, 3 + P, n, p_sub=2e-3, p_n=5e-3)
    top = (1 << 64) - 1
    cuts = sorted(int(x) for x in rng.integers(1, 1 << 12, size=P - 1))          # bins
    if aligned:
        edges = [0] + [c << (2 * n - 12) for c in cuts] + [None]
    else:
        edges = [0] + [(c << (2 * n - 12)) + int(rng.integers(1, 1 << 20)) for c in cuts] + [None]
    ranges = []
    for q in range(P):
        lo, hi = edges[q], (top if q == P - 1 else edges[q + 1] - 1)
        ranges.append((lo, hi, 0) if lo <= hi else (1, 0, 0))                      # equal cuts: an empty range
    owner = np.full(n_reads, P, np.int64)
    for q, (lo, hi, _) in enumerate(ranges):
        if lo <= hi:
            owner[(words >= np.uint64(lo)) & (words <= np.uint64(hi))] = q
    owner[filt != 0] = P
    order = np.argsort(owner, kind="stable")
    n_send = int((owner < P).sum())
    send_counts = [int((owner == q).sum()) for q in range(P)]
    ops = HipStageOps(0)
    dev = torch.device("cuda:0")
    d_w = torch.from_numpy(words.view(np.int64).copy()).to(dev)
    d_f = torch.from_numpy(filt.copy()).to(dev)
    routed, perm = ops.route(d_w, d_f, ranges, send_counts)
    ops.route_check()
    assert np.array_equal(perm.cpu().numpy()[:n_send].view(np.uint32), order[:n_send].astype(np.uint32))
    assert np.array_equal(routed.cpu().numpy().view(np.uint64), words[order[:n_send]])
    if P > 1 and send_counts[0] > 0:
        bad = list(send_counts)
        bad[0] -= 1
        bad[1] += 1
        ops.route(d_w, d_f, ranges, bad)
        with pytest.raises(humid_amd.HumidError):
            ops.route_check()
    ops.close()


@pytest.mark.parametrize("P", [1, 3])
def test_stage_by_stage_python_form_still_matches(P, monkeypatch):
    """the exchange mode exists twice over the same entry points: the library's single call
    (humid_dedup_run_exchange, the default with the HIP ops) and the stage-by-stage Python form
    (HUMID_PY_ORCHESTRATION=1, also what the oracle-backed gloo tests drive).  Same results."""
    words, filt = synth_words(150_000, 23, 24, p_sub=4e-3, p_n=2e-3)
    ocid, okeep, osum, _ = orc.dedup_run(words, filt, 24, 2, 0)
    monkeypatch.setenv("HUMID_PY_ORCHESTRATION", "1")
    out, offs = run_ranks(P, words, filt, 24, 2, 0, "exchange")
    for r in range(P):
        cid, keep, s, used = out[r]
        assert used == "exchange"
        assert np.array_equal(cid, ocid[offs[r]:offs[r + 1]]) and np.array_equal(keep, okeep[offs[r]:offs[r + 1]])
        assert s["edges"] == osum["edges"] and s["clusters"] == osum["clusters"]


@pytest.mark.parametrize("P,walk", [(1, 3), (3, 3), (2, 200), (4, None), (16, 3), (16, None)])
def test_exchange_large_buckets_go_through_the_tiles(P, walk):
    """the pair search of the exchange mode (emit_pairs: count, scan, fill into an edge list) with the
    bounded walk: buckets longer than the walk are finished by k_pairs_tiles in its emitting modes.
    Small walks on ordinary words (nearly every bucket takes that road) and the default walk on words
    that share one 12-nt prefix (a 30 000-word bucket), d = 1 and 2, against the oracle."""
    rng = np.random.default_rng(17 + P)
    if walk is None:
        tails = rng.choice(1 << 24, size=30_000, replace=False).astype(np.uint64)
        uniq = (np.uint64(0x3c5a96) << np.uint64(24)) | tails
        words = np.repeat(uniq, rng.poisson(0.4, size=len(uniq)) + 1)
        rng.shuffle(words)
        extra, ef = synth_words(40_000, 5, 24, p_sub=4e-3, p_n=1e-3)
        words = np.concatenate([words, extra])
        filt = np.concatenate([np.zeros(len(words) - len(extra), np.uint8), ef])
        perm = rng.permutation(len(words))
        words, filt = words[perm], filt[perm]
    else:
        words, filt = synth_words(90_000, 40 + P, 24, p_sub=6e-3, p_n=1e-3)
    for d in (1, 2):
        ocid, okeep, osum, _ = orc.dedup_run(words, filt, 24, d, 0)
        out, offs = run_ranks(P, words, filt, 24, d, 0, "exchange", bucket_walk=walk)
        for r in range(P):
            cid, keep, s, used = out[r]
            assert used == "exchange"
            assert np.array_equal(cid, ocid[offs[r]:offs[r + 1]]) and np.array_equal(keep, okeep[offs[r]:offs[r + 1]])
            assert s["edges"] == osum["edges"] and s["clusters"] == osum["clusters"]


@pytest.mark.parametrize("walk", [None, 50])
def test_allgather_mode_large_buckets_go_through_the_tiles(walk):
    """round 3: the position shares of the all-gather mode (humid_stage_pairs) walk
    bounded too; a 30 000-word bucket is finished by k_pairs_tiles restricted to the rank's own first
    positions.  2 virtual ranks, d = 1 and 2, against the oracle."""
    rng = np.random.default_rng(23)
    tails = rng.choice(1 << 24, size=30_000, replace=False).astype(np.uint64)
    uniq = (np.uint64(0x3c5a96) << np.uint64(24)) | tails
    words = np.repeat(uniq, rng.poisson(0.4, size=len(uniq)) + 1)
    extra, ef = synth_words(40_000, 5, 24, p_sub=4e-3, p_n=1e-3)
    words = np.concatenate([words, extra])
    filt = np.concatenate([np.zeros(len(words) - len(extra), np.uint8), ef])
    perm = rng.permutation(len(words))
    words, filt = words[perm], filt[perm]
    for d in (1, 2):
        ocid, okeep, osum, _ = orc.dedup_run(words, filt, 24, d, 0)
        out, offs = run_ranks(2, words, filt, 24, d, 0, "allgather", bucket_walk=walk)
        for r in range(2):
            cid, keep, s, used = out[r]
            assert used == "allgather"
            assert np.array_equal(cid, ocid[offs[r]:offs[r + 1]]) and np.array_equal(keep, okeep[offs[r]:offs[r + 1]])
            assert s["edges"] == osum["edges"] and s["clusters"] == osum["clusters"]


@pytest.mark.parametrize("P,n", [(2, 40), (3, 64)])
def test_exchange_wide_words_in_lds_tables(P, n):
    """enough two-word words per rank for the LDS-table count (k_dedup_lds_wide) over a rank's own range of
    top-64-bit values (key_map within the range, as for one-word words): every shard against the oracle"""
    from humid_amd.synth import synth_wide_words
    # ~385 k usable reads per rank = 2^11 buckets of ~190: well inside the LDS tables (with fewer reads the
    # ranges' histogram looks uneven and the sort is kept; just below a power of two times 350 the decision
    # is marginal)
    words, filt = synth_wide_words(400_000 * P, 31 * P + n, n, p_sub=2e-3, p_n=1e-3)
    ocid, okeep, osum, _ = orc.dedup_run(words, filt, n, 1, 0)
    out, offs = run_ranks(P, words, filt, n, 1, 0, "exchange")
    for r in range(P):
        cid, keep, s, used = out[r]
        assert used == "exchange" and s["count_mode_used"] == 2, s
        assert np.array_equal(cid, ocid[offs[r]:offs[r + 1]]) and np.array_equal(keep, okeep[offs[r]:offs[r + 1]])
        assert s["edges"] == osum["edges"] and s["clusters"] == osum["clusters"] and s["unique"] == osum["unique"]


@pytest.mark.parametrize("P", [1, 2, 3])
@pytest.mark.parametrize("n,d", [(33, 1), (40, 2), (48, 1), (64, 2)])
def test_exchange_wide_words(P, n, d):
    """33 <= word_nt <= 64 on several ranks (round 2): value ranges from the words' top 64 bits, two-word
    words routed and counted by sorting at their owners, 24-byte items for the other combinations --
    every rank's shard against one oracle run over the whole read set"""
    from humid_amd.synth import synth_wide_words
    words, filt = synth_wide_words(30_000, 100 * P + n + d, n, p_sub=6e-3, p_n=2e-3)
    ocid, okeep, osum, _ = orc.dedup_run(words, filt, n, d, 0)
    out, offs = run_ranks(P, words, filt, n, d, 0, "exchange")
    for r in range(P):
        cid, keep, s, used = out[r]
        assert used == "exchange"
        assert np.array_equal(cid, ocid[offs[r]:offs[r + 1]]) and np.array_equal(keep, okeep[offs[r]:offs[r + 1]])
        assert s["edges"] == osum["edges"] and s["clusters"] == osum["clusters"] and s["unique"] == osum["unique"]


@pytest.mark.parametrize("P", [1, 2, 3])
@pytest.mark.parametrize("n,d,method", [(33, 1, 0), (40, 2, 0), (48, 1, 1), (64, 2, 0)])
def test_allgather_mode_wide_words(P, n, d, method):
    """33 <= word_nt <= 64 in the ALL-GATHER mode, stage by stage (round 3: humid_stage_histogram / _count_dense /
    _unique / _graph / _owner_perm_wide take two-word words; value ranges are ranges of the words' top 64 bits):
    every rank's shard, of uneven sizes, against one oracle run over the whole read set"""
    from humid_amd.synth import synth_wide_words
    words, filt = synth_wide_words(30_000 + 7 * P, 300 * P + n + d, n, p_sub=6e-3, p_n=2e-3)
    ocid, okeep, osum, _ = orc.dedup_run(words, filt, n, d, method)
    out, offs = run_ranks(P, words, filt, n, d, method, "allgather")
    for r in range(P):
        cid, keep, s, used = out[r]
        assert used == "allgather"
        assert np.array_equal(cid, ocid[offs[r]:offs[r + 1]]) and np.array_equal(keep, okeep[offs[r]:offs[r + 1]])
        assert s["edges"] == osum["edges"] and s["clusters"] == osum["clusters"] and s["unique"] == osum["unique"]


@pytest.mark.parametrize("P,method", [(4, 0), (4, 1), (8, 0), (3, 0), (16, 0), (16, 1), (12, 0)])
def test_exchange_components_spanning_three_and_more_ranges(P, method):
    """owner-local clustering (round 3): clusters whose leaves lie in three and four value ranges -- families whose
    members differ in the FIRST nucleotides, the ones that decide the owner -- with count ladders the
    directional method climbs and floods along (64 > 24 > 9 > 3), flat ones it does not cross, and chains
    through the second nucleotide; ids, keep flags and all summary counts against the oracle"""
    rng = np.random.default_rng(100 + P)
    n_fam = 3000
    base = rng.integers(0, 1 << 48, size=n_fam, dtype=np.uint64)
    words = []
    for f in range(n_fam):
        ladder = (64, 24, 9, 3) if f % 3 else (10, 10, 10, 10)
        for k in range(4):                                   # the first nucleotide: A C G T -> four quarters of the walk
            w = (int(base[f]) & ~(3 << 46)) | (k << 46)
            words += [w] * ladder[k]
            if f % 5 == 0 and k == 3:                        # a tail through the second nucleotide
                w2 = w ^ (1 << 44)
                words += [w2]
    words = np.array(words, dtype=np.uint64)
    words = words[rng.permutation(len(words))]
    filt = (rng.random(len(words)) < 0.002).astype(np.uint8)
    # the oracle's clusters really do span the ranges
    p = orc.Pipeline(24)
    p.read_data(words, filt)
    p.find_hamming_neighbours(1)
    p.find_clusters(bool(method))
    lv = p.leaves()
    first_nt = (lv["word"] >> np.uint64(46)).astype(np.int64)
    spans = {}
    for cid_, nt in zip(lv["cluster_id"].tolist(), first_nt.tolist()):
        spans.setdefault(cid_, set()).add(nt)
    assert sum(1 for v in spans.values() if len(v) >= 3) >= 500
    check(P, words, filt, 24, 1, method)


@pytest.mark.parametrize("fail_after", [1, 2, 3, 4])
def test_exchange_one_rank_failing_ends_the_pass_on_every_rank(fail_after):
    """failure is collective.  Rank 1 leaves the pass with an error right after its k-th host
    gather (test hook "test_fail_before_gather"); it still joins the NEXT gather with its error code in the status
    word, so ranks 0 and 2 return an error from that gather instead of waiting for ever -- and nobody hangs."""
    import threading
    import torch
    import humid_amd
    from fake_dist import FakeDist, FakeWorld
    from humid_amd.sharded import HipStageOps, ShardedDedup
    P = 3
    words, filt = synth_words(90_000, 31, 24, p_sub=5e-3)
    world = FakeWorld(P)
    dev = torch.device("cuda:0")
    result = [None] * P

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            ops = HipStageOps(0)
            if r == 1:
                ops.set_option("test_fail_before_gather", fail_after)
            sd = ShardedDedup(device=0, word_nt=24, distance=1, ops=ops, dist=FakeDist(world, r), mode="exchange")
            n = 30_000
            w = torch.from_numpy(words[r * n:(r + 1) * n].view(np.int64)).to(dev)
            f = torch.from_numpy(filt[r * n:(r + 1) * n]).to(dev)
            c = torch.zeros(n, dtype=torch.int32, device=dev)
            k = torch.zeros(n, dtype=torch.uint8, device=dev)
            try:
                sd.run(w, f, c, k)
                result[r] = "ok"
            except humid_amd.HumidError as e:
                result[r] = str(e)
            ops.close()
        except Exception as e:  # pragma: no cover
            result[r] = "crash: %r" % (e,)
            world.barrier.abort()

    th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(P)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in th), "a rank is still waiting for a peer that left"
    assert "test: this rank fails" in result[1], result
    for r in (0, 2):
        assert "rank 1 left the pass" in result[r], result
