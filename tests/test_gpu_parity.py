"""-m gpu parity tests: the HIP path (through the C ABI) against the CPU oracle, bit-exact."""
import json
import os

import numpy as np
import pytest

import humid_amd
from humid_amd.synth import synth_words
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[(0, 0, 1), (0, 1, 1), (1, 0, 1), (0, 1, 0)],
                ids=["lds_hashed_buckets", "lds_ordered_buckets", "global_table", "lds_ordered_library_radix"])
def dd(request):
    """every parity test runs with all exact-count variants (humid_ctx_set_option count_mode /
    count_order; the ordered variant falls back to hashed buckets by itself on skewed words) and with
    both partition forms (tile_partition: hand-written LDS-staged passes / library radix passes)"""
    d = humid_amd.Dedup()
    d.set_option("count_mode", request.param[0])
    d.set_option("count_order", request.param[1])
    d.set_option("tile_partition", request.param[2])
    d.count_mode = request.param[0]
    d.count_order = request.param[1]
    yield d
    d.close()


def oracle_full(words, filt, n, d, maximum):
    p = orc.Pipeline(n)
    p.read_data(words, filt)
    p.find_hamming_neighbours(d)
    p.find_clusters(maximum)
    cid, keep = p.map_reads()
    return p, cid, keep


def check_against_oracle(dd, words, filt, n, d, maximum, deep=True):
    cid, keep, s = dd.run(words, filt, word_nt=n, distance=d, method=int(maximum))
    p, ocid, okeep = oracle_full(words, filt, n, d, maximum)
    os_ = p.summary()
    for k in ("total", "usable", "unique", "clusters", "edges"):
        assert s[k] == os_[k], (k, s[k], os_[k])
    assert np.array_equal(cid, ocid)
    assert np.array_equal(keep, okeep)
    if deep and s["unique"]:
        lv, olv = dd.leaves(), p.leaves()
        assert np.array_equal(lv["word"], olv["word"])
        assert np.array_equal(lv["count"].astype(np.uint64), olv["count"])
        assert np.array_equal(lv["degree"], olv["degree"])
        assert np.array_equal(lv["cluster_id"], olv["cluster_id"])
        assert np.array_equal(lv["is_max_leaf"], olv["is_max_leaf"])
        off, idx = dd.adjacency()
        ooff, oidx = p.adjacency()
        assert np.array_equal(off.astype(np.uint64), ooff)
        assert np.array_equal(idx, oidx)
        cl, ocl = dd.clusters(), p.clusters()
        assert np.array_equal(cl["size"], ocl["size"])
        assert np.array_equal(cl["max_count"].astype(np.uint64), ocl["max_count"])
        assert np.array_equal(cl["max_leaf"], ocl["max_leaf"])
        assert dd.histograms() == orc.histograms(p)
    return s


def dense_words(rng, n_reads, n, k):
    base = rng.integers(0, 4 ** min(n, 31), dtype=np.uint64)
    w = np.full(n_reads, base, dtype=np.uint64)
    for _ in range(k):
        pos = int(rng.integers(0, n))
        sh = np.uint64(2 * pos)
        v = rng.integers(0, 4, size=n_reads).astype(np.uint64)
        w = (w & ~(np.uint64(3) << sh)) | (v << sh)
    return w


def test_at_least_double_device(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "ref_test_cluster.json")))
    ctx = humid_amd.Context()
    for c in g["at_least_double"]:
        assert humid_amd.at_least_double(c["a"], c["b"], ctx) == c["expect"], c["ref"]
    ctx.close()


def test_reference_cluster_scenario_on_gpu(golden_dir):
    """tests/test_cluster.cc:73-137 driven through humid_cluster_graph"""
    g = json.load(open(os.path.join(golden_dir, "ref_test_cluster.json")))["assign_directional"]
    cg = humid_amd.ClusterGraph(g["counts"])
    for a, b in g["links"]:
        cg.link(a, b)
    r = cg.find_clusters(False)
    assert r["n_clusters"] == 2
    assert r["leaf_cluster"].tolist() == g["calls"][-1]["expect_leaf_cluster"]
    assert r["size"].tolist() == g["expect_size"]
    assert r["max_leaf"].tolist() == g["expect_max_leaf"]
    assert r["max_count"].tolist() == g["expect_max_count"]
    cg.close()


def test_reference_max_neighbour_chains_on_gpu(golden_dir):
    """tests/test_cluster.cc:45-71: the climb ends on the 4-count leaf, so cluster 1's maxLeaf
    is that leaf when the findClusters loop starts from leaf 0"""
    g = json.load(open(os.path.join(golden_dir, "ref_test_cluster.json")))
    for case in g["max_neighbour"]:
        if case["preassigned"] or case["counts"] == [0]:
            continue
        cg = humid_amd.ClusterGraph(case["counts"])
        for a, b in case["links"]:
            cg.link(a, b)
        r = cg.find_clusters(False)
        assert int(r["max_leaf"][0]) == case["queries"][0]["expect"], case["name"]
        cg.close()


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("maximum", [False, True])
def test_random_graphs_arbitrary_list_order(seed, maximum):
    """explicit graphs whose neighbour lists are in link() order, not ascending"""
    rng = np.random.default_rng(100 + seed)
    u = int(rng.integers(2, 300))
    counts = rng.integers(1, 20, size=u)
    cg = humid_amd.ClusterGraph(counts)
    og = orc.Graph(counts)
    seen = set()
    for _ in range(int(rng.integers(1, 3 * u))):
        a, b = int(rng.integers(0, u)), int(rng.integers(0, u))
        if a == b or (a, b) in seen or (b, a) in seen:
            continue
        seen.add((a, b))
        cg.link(a, b)
        og.link(a, b)
    r = cg.find_clusters(maximum)
    nc = og.find_clusters(maximum)
    lc, size, mc, ml = og.export(nc)
    assert r["n_clusters"] == nc
    assert np.array_equal(r["leaf_cluster"], lc)
    assert np.array_equal(r["size"], size)
    assert np.array_equal(r["max_count"].astype(np.uint64), mc)
    assert np.array_equal(r["max_leaf"].astype(np.int64), ml)
    cg.close()


@pytest.mark.parametrize("seed", range(10))
@pytest.mark.parametrize("d", [1, 2, 3])
@pytest.mark.parametrize("maximum", [False, True])
def test_dense_small(dd, seed, d, maximum):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(4, 33))
    n_reads = int(rng.integers(1, 3000))
    words = dense_words(rng, n_reads, n, int(rng.integers(1, 6)))
    filt = (rng.random(n_reads) < 0.05).astype(np.uint8)
    check_against_oracle(dd, words, filt, n, d, maximum)


@pytest.mark.parametrize("cfg", [
    # (reads, word_nt, d, mode, p_sub) -- shapes of BASELINE.json configs at oracle-friendly size
    (100_000, 24, 1, "umi", 1e-3),     # config 1: 100k SE UMI=8
    (300_000, 24, 1, "umi", 1e-3),     # config 2/metric shape, reduced
    (200_000, 24, 2, "genome", 1e-3),  # config 5 shape: no UMI, d=2
    (50_000, 32, 1, "umi", 5e-3),      # widest single-word
    (50_000, 12, 2, "umi", 1e-2),      # short words: heavy neighbourhoods
    (20_000, 5, 1, "umi", 1e-2),       # saturated word space (every word has neighbours)
    (5_000, 1, 1, "umi", 1e-2),        # d >= n: one zero-width segment
])
@pytest.mark.parametrize("maximum", [False, True])
def test_synthetic_configs(dd, cfg, maximum):
    n_reads, n, d, mode, p_sub = cfg
    words, filt = synth_words(n_reads, 1000 + n + d, n, p_sub=p_sub, mode=mode,
                              genome_bp=200_000 if mode == "genome" else 0)
    check_against_oracle(dd, words, filt, n, d, maximum)


def test_edge_cases(dd):
    # empty
    cid, keep, s = dd.run(np.zeros(0, np.uint64), np.zeros(0, np.uint8))
    assert len(cid) == 0 and s["unique"] == 0
    # everything filtered
    cid, keep, s = dd.run(np.arange(7, dtype=np.uint64), np.ones(7, np.uint8))
    assert cid.tolist() == [0] * 7 and keep.tolist() == [0] * 7 and s["usable"] == 0
    # one read
    check_against_oracle(dd, np.array([5], np.uint64), np.zeros(1, np.uint8), 24, 1, False)
    # all identical
    check_against_oracle(dd, np.full(1000, 12345, np.uint64), np.zeros(1000, np.uint8), 24, 1, False)
    # n = 32 with the all-T word (equals the table's EMPTY sentinel)
    w = np.array([2 ** 64 - 1, 2 ** 64 - 1, 2 ** 64 - 2, 2 ** 64 - 5, 7, 2 ** 64 - 1], dtype=np.uint64)
    check_against_oracle(dd, w, np.zeros(6, np.uint8), 32, 1, False)
    check_against_oracle(dd, w, np.zeros(6, np.uint8), 32, 1, True)
    # n = 32 with the word whose mix64() equals the LDS table's EMPTY sentinel (unmix64(~0))
    sp = 0xcf9a04affa6badc0
    w = np.array([sp, sp, sp ^ 1, sp ^ 3, 9, sp, 2 ** 64 - 1], dtype=np.uint64)
    check_against_oracle(dd, w, np.zeros(len(w), np.uint8), 32, 1, False)
    big = np.concatenate([np.full(5000, sp, np.uint64), np.arange(3000, dtype=np.uint64) * np.uint64(977)])
    check_against_oracle(dd, big, np.zeros(len(big), np.uint8), 32, 1, True)
    # distance 0: exact duplicates only
    words, filt = synth_words(20000, 3, 24, p_sub=1e-2)
    check_against_oracle(dd, words, filt, 24, 0, False)


def np_mix64(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(30); x *= np.uint64(0xbf58476d1ce4e5b9)
    x ^= x >> np.uint64(27); x *= np.uint64(0x94d049bb133111eb)
    x ^= x >> np.uint64(31)
    return x


def test_bucket_overflow_falls_back_to_global_table(dd):
    """> 2048 distinct words in ONE hash bucket: the LDS table overflows, the run is redone with
    the global table, results stay exact"""
    rng = np.random.default_rng(3)
    cand = rng.integers(0, 4 ** 24, size=1_500_000, dtype=np.uint64)
    with np.errstate(over="ignore"):
        top = np_mix64(cand) >> np.uint64(59)          # N = 20000 -> 2^5 buckets
    words = np.unique(cand[top == 7])[:20000]
    assert len(words) == 20000
    s = check_against_oracle(dd, words, np.zeros(len(words), np.uint8), 24, 1, False)
    if dd.count_order == 1:
        assert s["count_mode_used"] == 2               # word-ordered buckets spread these words out
        # ... but 20000 distinct words sharing their top 5 word bits overflow an ORDERED bucket:
        # the run is redone with hashed buckets
        same_prefix = (words & np.uint64((1 << 43) - 1)) | (np.uint64(13) << np.uint64(43))
        same_prefix = np.unique(same_prefix)
        s2 = check_against_oracle(dd, same_prefix, np.zeros(len(same_prefix), np.uint8), 24, 1, False)
        assert s2["count_mode_used"] == 0
    else:
        assert s["count_mode_used"] == 1               # fallback (or the forced global mode)
    # the same amount of reads, but duplicates of few words: no overflow, LDS path is used
    few = np.repeat(words[:100], 200)
    s = check_against_oracle(dd, few, np.zeros(len(few), np.uint8), 24, 1, False)
    assert s["count_mode_used"] in ((1,) if dd.count_mode == 1 else (0, 2))


@pytest.mark.parametrize("d,s", [(1, 2), (1, 3), (1, 4), (2, 3), (2, 4), (2, 5), (3, 4), (3, 6)])
def test_forced_pigeonhole_plans(dd, d, s):
    """every plan (s segments, combos of s-d) finds the same neighbour graph: the automatic choice
    only picks the larger s at millions of unique words, so the test forces it"""
    dd.set_option("plan_segments", s)
    try:
        for n, mode in ((24, "umi"), (24, "genome"), (13, "umi")):
            words, filt = synth_words(60_000, 500 + d + s, n, p_sub=8e-3, mode=mode, genome_bp=30_000)
            check_against_oracle(dd, words, filt, n, d, False)
    finally:
        dd.set_option("plan_segments", 0)


@pytest.mark.parametrize("coop", [1, 0])
def test_big_components_both_kernels(dd, coop):
    """dense components far above 32 leaves: workgroup-cooperative flood vs the one-lane loop"""
    dd.set_option("coop_big", coop)
    try:
        rng = np.random.default_rng(17)
        # saturated 6-nt space: 4096 words, one component, geometric counts (deep climbs, wide floods)
        w = rng.integers(0, 4 ** 6, size=40000, dtype=np.uint64)
        w = np.concatenate([w, np.repeat(rng.integers(0, 4 ** 6, size=200, dtype=np.uint64), 50)])
        check_against_oracle(dd, w, np.zeros(len(w), np.uint8), 6, 1, False)
        check_against_oracle(dd, w, np.zeros(len(w), np.uint8), 6, 2, False)
        # many mid-sized components (40-400 leaves): d=2 over a few hundred seeds of 12 nt
        seeds = rng.integers(0, 4 ** 12, size=300, dtype=np.uint64)
        ws = np.repeat(seeds, 300)
        for _ in range(2):
            pos = rng.integers(0, 12, size=len(ws)).astype(np.uint64) * np.uint64(2)
            v = rng.integers(0, 4, size=len(ws)).astype(np.uint64)
            ws = (ws & ~(np.uint64(3) << pos)) | (v << pos)
        check_against_oracle(dd, ws, np.zeros(len(ws), np.uint8), 12, 2, False)
    finally:
        dd.set_option("coop_big", 1)


def test_ordered_buckets_are_chosen_for_uniform_prefixes_only():
    d = humid_amd.Dedup()
    words, filt = synth_words(400_000, 9, 24)                      # UMI first: uniform top bits
    cid, keep, s = d.run(words, filt)
    assert s["count_mode_used"] == 2
    ocid, okeep, _, _ = orc.dedup_run(words, filt, 24, 1, 0)
    assert np.array_equal(cid, ocid) and np.array_equal(keep, okeep)
    skew = (words & np.uint64((1 << 30) - 1)) | (np.uint64(0x2aaaa) << np.uint64(30))   # one read prefix
    cid, keep, s = d.run(skew, filt)
    assert s["count_mode_used"] == 0
    ocid, okeep, _, _ = orc.dedup_run(skew, filt, 24, 1, 0)
    assert np.array_equal(cid, ocid) and np.array_equal(keep, okeep)
    d.close()


def test_repeatable_at_scale(dd):
    """same input, same answer, run after run (the edge loss of round 1, DESIGN.md section 3a, showed
    as run-to-run differences: single waves computed wrong bucket keys)"""
    words, filt = synth_words(3_000_000, 1002, 24)
    ref = None
    for _ in range(4):
        cid, keep, s = dd.run(words, filt)
        cur = (s["edges"], s["clusters"], s["unique"], int(cid.astype(np.uint64).sum()), int(keep.sum()))
        ref = ref or cur
        assert cur == ref


def test_oracle_parity_at_scale(dd):
    """3 M reads of the metric workload, every array bit for bit against the oracle, for each
    exact-count variant: a loss of a few edges in 10^5 (the rate once seen at 10 M reads, DESIGN.md
    section 3a) cannot hide at this size.  bench.py repeats the comparison on the full 10 M reads."""
    words, filt = synth_words(3_000_000, 1002, 24)
    check_against_oracle(dd, words, filt, 24, 1, False, deep=True)


def test_oracle_parity_at_scale_genome_d2(dd1):
    """3 M reads of BASELINE config 5's shape (no UMI: 12 + 12 nucleotides of a fragment's two ends drawn from a
    random genome, d = 2: six combinations of 12 nt, dense neighbourhoods), every array against the oracle.  The
    six-combination key kernels and the first-combination rule at d = 2 were only oracle-compared up to 200 k
    reads before round 3."""
    words, filt = synth_words(3_000_000, 1005, 24, mode="genome")
    check_against_oracle(dd1, words, filt, 24, 2, False, deep=True)


def test_record_path_with_ten_bit_first_level():
    """8-byte partition records hold the key bits below the coarse bin + the read index: 2 n - d1 + ceil(log2 N) <= 64.
    When the balanced split of the bucket bits does not fit, the FIRST level takes up to 10 bits (1024 coarse bins) --
    how 24-nt words stay on the record path between 33 M and 67 M reads (BASELINE configs 3 and 5).  The same
    arithmetic at a size the oracle finishes: 26-nt words (52 key bits), 3 M reads (22 index bits, 2^14 buckets):
    52 - 7 + 22 = 67 with the balanced split, 64 with d1 = 10.  Every array against the oracle; the run must have
    stayed on the record path."""
    words, filt = synth_words(3_000_000, 1026, 26)
    dq = humid_amd.Dedup()
    s = check_against_oracle(dq, words, filt, 26, 1, False, deep=True)
    assert s["records8"] and s["count_mode_used"] == 2, (s["count_mode_used"], s["records8"])
    dq.close()


SHAPE_SIZES = sorted({(350 << k) + e for k in range(5, 12) for e in (0, 1)} |        # the bucket bits change (PART_TARGET 350)
                     {(1 << k) + e for k in range(14, 20) for e in (0, 1)} |           # the index bits of a record change
                     {(1 << 15) * 3 + e for e in (-1, 0, 1)} | {8192 * 37 + e for e in (-1, 0, 1)})   # tile edges


@pytest.mark.parametrize("order", [1, -1], ids=["ordered_buckets_forced", "default"])
@pytest.mark.parametrize("n", [24, 13, 31, 32])
def test_sizes_where_the_partition_changes_shape(n, order):
    """read counts at which the record path changes its shape -- one bucket bit more (350 x 2^k reads), one index bit
    more in a record (2^k reads), a last tile of one read -- each at the last size of the old shape and the first
    of the new one.  With word-ordered buckets forced (the record path from 64 buckets on: asserted for 24-nt words;
    the default decides for them from 65 536 reads on only) and with the default context; reads, clusters and
    summary against the oracle."""
    dq = humid_amd.Dedup()
    dq.set_option("count_order", order)
    # (the oracle runs on one core of the GPU box: all sizes for 24-nt words with the buckets forced, a third of them else)
    for N in SHAPE_SIZES if (n == 24 and order == 1) else SHAPE_SIZES[(n + order) % 3::3]:
        words, filt = synth_words(N, 3000 + n, n, p_sub=3e-3, p_n=1e-3)
        s = check_against_oracle(dq, words, filt, n, 1, False, deep=False)
        if n == 24 and order == 1 and N > 350 << 5:       # (the default's sampler keeps a 1.5 x margin and says no at the
            assert s["records8"] and s["count_mode_used"] == 2, (N, s["count_mode_used"], s["records8"])   # fullest small shapes)
    dq.close()


def test_oracle_parity_metric_words_d2(dd1):
    """1 M reads of the metric workload at d = 2, both methods"""
    words, filt = synth_words(1_000_000, 1002, 24)
    check_against_oracle(dd1, words, filt, 24, 2, False, deep=True)
    check_against_oracle(dd1, words, filt, 24, 2, True, deep=False)


@pytest.fixture(scope="module")
def dd1():
    """one context with the default settings, for tests that do not depend on the count variant"""
    d = humid_amd.Dedup()
    yield d
    d.close()


@pytest.mark.parametrize("d", [19, 20, 23, 24, 40])
def test_distances_beyond_every_pigeonhole_plan(dd1, d):
    """-m 20 and above at n = 24 have no plan within the 20-combination tables: all pairs are
    compared (the tables were overrun and neighbours silently wrong)"""
    dd = dd1
    words, filt = synth_words(1500, 40 + d, 24, p_sub=5e-2)
    check_against_oracle(dd, words, filt, 24, d, False)
    check_against_oracle(dd, words[:300], filt[:300], 24, d, True)


def test_all_pairs_plan_refuses_huge_inputs(dd1):
    dd = dd1
    rng = np.random.default_rng(1)
    w = rng.integers(0, 4 ** 24, size=400_000, dtype=np.uint64)
    with pytest.raises(humid_amd.HumidError) as e:
        dd.run(w, np.zeros(len(w), np.uint8), word_nt=24, distance=22)
    assert e.value.code == -5


def test_pair_count_overflow_is_reported(dd1):
    """2E >= 2^32: the 32-bit adjacency offsets would wrap; the 64-bit degree sum sees it"""
    dd = dd1
    w = np.arange(70_000, dtype=np.uint64)          # 9-nt words, d = 9: all 2.4e9 pairs are neighbours
    with pytest.raises(humid_amd.HumidError) as e:
        dd.run(w, np.zeros(len(w), np.uint8), word_nt=9, distance=9)
    assert e.value.code == -5


def test_unsupported_and_invalid(dd):
    w = np.zeros(4, np.uint64)
    f = np.zeros(4, np.uint8)
    with pytest.raises(humid_amd.HumidError) as e:
        dd.run(np.zeros((4, 2), np.uint64), f, word_nt=65)     # wide words stop at 64 nt
    assert e.value.code == -2
    with pytest.raises(humid_amd.HumidError) as e:
        dd.run(w, f, word_nt=0)
    assert e.value.code == -1
    with pytest.raises(humid_amd.HumidError) as e:
        dd.run(w, f, method=7)
    assert e.value.code == -1


def test_count_ties_and_steals(dd):
    """shared low-count leaf between two maxima, equal counts (no edge), ratio exactly 2"""
    A = orc.pack_word([0, 0, 0, 0, 0, 0])
    B = orc.pack_word([0, 0, 0, 0, 0, 1])   # neighbour of A and C
    Cw = orc.pack_word([0, 0, 0, 0, 1, 1])  # neighbour of B only
    D = orc.pack_word([3, 3, 3, 3, 3, 3])
    E = orc.pack_word([3, 3, 3, 3, 3, 2])
    reads = [A] * 4 + [B] * 2 + [Cw] * 4 + [D] * 3 + [E] * 3
    rng = np.random.default_rng(5)
    for _ in range(5):
        w = np.array(reads, dtype=np.uint64)
        rng.shuffle(w)
        for mx in (False, True):
            check_against_oracle(dd, w, np.zeros(len(w), np.uint8), 6, 1, mx)


def test_large_bucket(dd):
    """amplicon-like: one read prefix shared by thousands of UMIs -> one big segment bucket"""
    rng = np.random.default_rng(11)
    n_reads = 60000
    umi = rng.integers(0, 4 ** 6, size=n_reads, dtype=np.uint64)   # 4096 distinct UMIs
    words = (umi << np.uint64(36)) | np.uint64(0x123456789)        # same 18-nt read part
    check_against_oracle(dd, words, np.zeros(n_reads, np.uint8), 24, 1, False)
    check_against_oracle(dd, words, np.zeros(n_reads, np.uint8), 24, 1, True)


@pytest.mark.parametrize("walk", [1, 7, 300])
@pytest.mark.parametrize("cfg", [(120_000, 24, 1), (60_000, 12, 2), (40_000, 16, 3), (20_000, 40, 2)])
def test_tiles_take_over_beyond_the_bucket_walk(dd1, walk, cfg):
    """k_pairs bounded to `walk` followers: every bucket longer than that is completed by k_big_runs +
    k_pairs_tiles (squares of the upper triangle, the partly-near squares filtered pair by pair).  With a
    tiny walk nearly every bucket of every combination takes that road: all arrays against the oracle."""
    n_reads, n, d = cfg
    dd1.set_option("bucket_walk", walk)
    try:
        if n <= 32:
            words, filt = synth_words(n_reads, 5 + walk, n, p_sub=8e-3, p_n=1e-3)
            check_against_oracle(dd1, words, filt, n, d, False)
            check_against_oracle(dd1, words, filt, n, d, True, deep=False)
        else:
            from humid_amd.synth import synth_wide_words
            words, filt = synth_wide_words(n_reads, 5 + walk, n, p_sub=8e-3, p_n=1e-3)
            check_against_oracle(dd1, words, filt, n, d, False)
    finally:
        dd1.set_option("bucket_walk", 1024)


def one_prefix_words(rng, n_words, n, prefix_nt, reads_per_word=1.3):
    """n_words DISTINCT n-nt words that all start with the same prefix_nt nucleotides, each read once
    or a few times; read order shuffled"""
    tail_bits = 2 * (n - prefix_nt)
    tails = rng.choice(1 << tail_bits, size=n_words, replace=False).astype(np.uint64)
    prefix = np.uint64(int(rng.integers(0, 4 ** prefix_nt))) << np.uint64(tail_bits)
    uniq = prefix | tails
    reps = rng.poisson(reads_per_word - 1, size=n_words) + 1
    words = np.repeat(uniq, reps)
    rng.shuffle(words)
    return words


@pytest.mark.parametrize("d", [1, 2])
def test_one_prefix_shared_by_200k_words(dd, d):
    """skew the pigeonhole cannot split: 200 000 distinct 24-nt words with the same first 12
    nucleotides, so the bucket of the first-half key (d = 1) / of the leading segments (d = 2) holds
    every one of them -- 2 * 10^10 pairs in one bucket, taken by k_pairs_tiles; the exact-count
    partition sees one 9-nt prefix only and falls back by itself.  Everything against the oracle."""
    rng = np.random.default_rng(40 + d)
    words = one_prefix_words(rng, 200_000, 24, 12)
    filt = (rng.random(len(words)) < 1e-3).astype(np.uint8)
    s = check_against_oracle(dd, words, filt, 24, d, False)
    assert s["unique"] >= 199_000 and s["edges"] > 0
    check_against_oracle(dd, words, filt, 24, d, True, deep=False)


@pytest.mark.parametrize("block_words", [20_000, 60_000])
def test_grouping_of_a_huge_key_bucket(dd1, block_words):
    """bucket order of the second-half combination comes from a two-level grouping (k_group_fine); 20 000 /
    60 000 distinct words with the SAME last 12 nucleotides put that many words into one coarse bin -- the
    roads for bins of 8161 .. 32768 words and beyond -- and into one key bucket (tiles); plus ordinary
    words around them"""
    rng = np.random.default_rng(5)
    heads = rng.choice(1 << 24, size=block_words, replace=False).astype(np.uint64)
    block = (heads << np.uint64(24)) | np.uint64(0x6b1e57)
    words, filt = synth_words(80_000, 21, 24, p_sub=4e-3, p_n=1e-3)
    words = np.concatenate([words, np.repeat(block, rng.poisson(0.3, size=len(block)) + 1)])
    filt = np.concatenate([filt, np.zeros(len(words) - len(filt), np.uint8)])
    perm = rng.permutation(len(words))
    words, filt = words[perm], filt[perm]
    for d in (1, 2):
        check_against_oracle(dd1, words, filt, 24, d, False, deep=(d == 1))
    dd1.set_option("group_buckets", 0)                      # and the library sort gives the same
    check_against_oracle(dd1, words, filt, 24, 1, False, deep=False)
    dd1.set_option("group_buckets", 1)


def test_padded_partition_outgrown_by_duplicated_words(dd):
    """the first partition level gives every coarse bin a fixed room (mean + 25 % + 1024 reads) and skips
    its histogram pass; all reads of ONE word share a bin, so a word that makes up a third of the reads
    outgrows that room: the overflow must be noticed, the run repeated with the histogram pass, and the
    results stay those of the oracle -- also for the runs after it (the context remembers)"""
    rng = np.random.default_rng(77)
    words, filt = synth_words(600_000, 9, 24, p_sub=3e-3, p_n=1e-3)
    hot = rng.random(len(words)) < 0.35
    words = words.copy()
    words[hot] = np.uint64(0x0123456789ab)
    dd.set_option("padded_partition", 1)
    for rep in range(2):
        s = check_against_oracle(dd, words, filt, 24, 1, False, deep=(rep == 0))
        assert s["unique"] > 100_000
    words2, filt2 = synth_words(200_000, 10, 24, p_sub=3e-3)
    check_against_oracle(dd, words2, filt2, 24, 1, False, deep=False)
    dd.set_option("padded_partition", 1)


@pytest.mark.parametrize("top30,n_special,expect_records", [(0x15555555, 40, True), (0x2aaaaaaa, 200, True), (0x3fffffff, 40, True),
                                                            (0x3fffffff, 90, False)])
def test_count_table_runs_and_spill(top30, n_special, expect_records):
    """k_dedup_rec ranks the unique words of a bucket by their place in a table whose home function is monotone in
    the word and which does not wrap: words that share bucket AND home bits form one run of
    neighbouring entries and are ordered by comparing inside the run; a run that passes the table's spill entries
    is an overflow and the read set is counted by the kernels of round 2.  n_special distinct words with the same
    first 15 nucleotides (every bucket and home bit, whatever the bucket count) among 300 000 ordinary reads: runs
    of 40 and 200 entries in the middle of the table and 40 words whose home is its LAST entry (they fit the 64 spill
    entries) stay on the record path, 90 such words do not -- all against the oracle, unique words and their order
    included.  Word-ordered buckets are FORCED (count_order = 1): left to itself the pipeline samples the reads first
    and would not try ordered buckets on the fuller of these read sets at all."""
    rng = np.random.default_rng(top30 & 0xffff)
    words, filt = synth_words(300_000, 31, 24, p_sub=3e-3, p_n=1e-3)
    low = rng.choice(1 << 18, size=n_special, replace=False).astype(np.uint64)
    special = (np.uint64(top30) << np.uint64(18)) | low
    special = np.repeat(special, rng.integers(1, 5, size=n_special))
    words = np.concatenate([words, special])
    filt = np.concatenate([filt, np.zeros(len(special), np.uint8)])
    perm = rng.permutation(len(words))
    words, filt = words[perm], filt[perm]
    dq = humid_amd.Dedup()
    dq.set_option("count_order", 1)
    for rep in range(2):                                    # (the second run: what the context remembers of the first)
        s = check_against_oracle(dq, words, filt, 24, 1, False, deep=(rep == 0))
        assert bool(s.get("records8")) == expect_records, (s["count_mode_used"], s.get("records8"))
    dq.close()


def test_long_chain_component(dd):
    """a path-shaped component thousands of leaves deep (the reference recursion overflows here)"""
    # words 0..L-1 in unary-like Gray walk: consecutive words differ in one nucleotide
    L = 5000
    words = []
    w = 0
    rng = np.random.default_rng(2)
    seen = {0}
    path = [0]
    n = 24
    while len(path) < L:
        pos = int(rng.integers(0, n))
        v = int(rng.integers(1, 4))
        old = (w >> (2 * pos)) & 3
        nw = (w & ~(3 << (2 * pos))) | (((old + v) & 3) << (2 * pos))
        if nw in seen:
            continue
        seen.add(nw)
        path.append(nw)
        w = nw
    reads = []
    for i, x in enumerate(path):
        reads += [x] * (1 + (i % 3))
    words = np.array(reads, dtype=np.uint64)
    check_against_oracle(dd, words, np.zeros(len(words), np.uint8), n, 1, False)
    check_against_oracle(dd, words, np.zeros(len(words), np.uint8), n, 1, True)


def test_full_size_properties(dd):
    """BASELINE metric size (10 M reads): size-independent properties"""
    n_reads = 10_000_000
    words, filt = synth_words(n_reads, 1002, 24)
    cid, keep, s = dd.run(words, filt)
    assert s["total"] == n_reads and s["usable"] == int((filt == 0).sum())
    # filtered <=> cluster 0, never kept
    assert np.array_equal(cid == 0, filt == 1)
    assert not keep[filt == 1].any()
    # exactly one kept read per cluster, ids are 1..C
    assert int(keep.sum()) == s["clusters"] == int(cid.max())
    kept_ids = np.sort(cid[keep == 1])
    assert np.array_equal(kept_ids, np.arange(1, s["clusters"] + 1, dtype=np.uint32))
    # equal words share a cluster
    order = np.argsort(words, kind="stable")
    sw, sc, sf = words[order], cid[order], filt[order]
    same = (sw[1:] == sw[:-1]) & (sf[1:] == 0) & (sf[:-1] == 0)
    assert np.array_equal(sc[1:][same], sc[:-1][same])
    # counts histogram sums to unique; sizes sum to usable
    h = dd.histograms()
    assert sum(v for _, v in h["counts"]) == s["unique"]
    assert sum(k * v for k, v in h["counts"]) == s["usable"]
    assert sum(k * v for k, v in h["clusters"]) == s["usable"]
    assert sum(v for _, v in h["clusters"]) == s["clusters"]
    assert sum(k * v for k, v in h["neigh"]) == 2 * s["edges"]
    # idempotence: deduplicating the kept reads keeps all of them (no two kept words are
    # directional duplicates of each other once counts are 1 -- a 1-count never absorbs a 1-count)
    kw = words[keep == 1]
    cid2, keep2, s2 = dd.run(kw, np.zeros(len(kw), np.uint8))
    assert int(keep2.sum()) == len(kw) == s2["clusters"]
    # cluster ids ascend with the smallest word... of the creating leaf: id 1 belongs to the
    # cluster holding the smallest usable word
    smallest = sw[sf == 0][0]
    assert cid[(words == smallest) & (filt == 0)][0] >= 1


# ------------------------------------------------------------------------------------------
# multi-GPU stage entry points, exercised on ONE GPU: P virtual ranks, one context each
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("P", [1, 2, 4, 8])
@pytest.mark.parametrize("cfg", [(200_000, 24, 1, "umi", 0), (60_000, 12, 2, "umi", 1),
                                 (50_000, 32, 1, "umi", 0), (3000, 4, 1, "umi", 0),
                                 (80_000, 24, 2, "genome", 0), (2000, 2, 3, "umi", 0)])
def test_stage_functions_with_virtual_ranks(P, cfg):
    import torch
    from humid_amd.sharded import HipStageOps, splitters_from_hist
    n_reads, n, d, mode, method = cfg
    words, filt = synth_words(n_reads, 77 + P, n, p_sub=5e-3, p_n=1e-3, mode=mode)
    ocid, okeep, osum, _ = orc.dedup_run(words, filt, n, d, method)
    dev = torch.device("cuda:0")
    g_w = torch.from_numpy(words.view(np.int64)).to(dev)
    g_f = torch.from_numpy(filt).to(dev)
    ops = [HipStageOps(0) for _ in range(P)]
    bits = min(12, 2 * n)
    hist = ops[0].histogram(g_w, g_f, n, bits).cpu().numpy()
    assert int(hist.sum()) == osum["usable"]
    ranges = splitters_from_hist(hist, P, n, bits)
    u_all, uw, uc = [], [], []
    usable = 0
    for r in range(P):
        lo, hi, exp = ranges[r]
        u, us = ops[r].count(g_w, g_f, n, lo, hi, max(exp, 1))
        usable += us
        u_all.append(u)
        w, c = ops[r].unique()
        uw.append(w.clone())
        uc.append(c.clone())
    assert usable == osum["usable"] and sum(u_all) == osum["unique"]
    gw, gc = torch.cat(uw), torch.cat(uc)
    assert bool((gw.cpu().numpy().view(np.uint64)[1:] > gw.cpu().numpy().view(np.uint64)[:-1]).all())
    cid_g, ismax_g, gs = ops[0].graph(gw, gc, n, d, method)
    assert gs["clusters"] == osum["clusters"]
    # partitioned search: the shares are disjoint and cover every pair; the graph built from the
    # gathered edge list equals the searched one
    shares = [ops[r].pairs(gw, n, d, r, P).clone() for r in range(P)]
    e_all = torch.cat(shares) if shares else torch.zeros(0, dtype=torch.int64, device=dev)
    assert e_all.numel() == gs["edges"]
    assert torch.unique(e_all).numel() == e_all.numel()
    cid_e, ismax_e, ges = ops[P - 1].graph_edges(gw, gc, e_all, n, d, method)
    assert ges["clusters"] == gs["clusters"] and ges["edges"] == gs["edges"]
    assert torch.equal(cid_e, cid_g) and torch.equal(ismax_e, ismax_g)
    cid_g, ismax_g = cid_e.clone(), ismax_e.clone()
    tot_cid = torch.zeros(n_reads, dtype=torch.int32, device=dev)
    tot_keep = torch.zeros(n_reads, dtype=torch.int32, device=dev)
    off = 0
    for r in range(P):
        o_c = torch.empty(n_reads, dtype=torch.int32, device=dev)
        o_k = torch.empty(n_reads, dtype=torch.uint8, device=dev)
        l_c = cid_g[off:off + u_all[r]].clone()
        l_m = ismax_g[off:off + u_all[r]].clone()
        ops[r].map(l_c, l_m, o_c, o_k)
        tot_cid += o_c
        tot_keep += o_k.to(torch.int32)
        off += u_all[r]
    assert np.array_equal(tot_cid.cpu().numpy().view(np.uint32), ocid)
    assert np.array_equal(tot_keep.cpu().numpy().astype(np.uint8), okeep)
    # dense result return: owners emit packed per-shard streams, home shards scatter them
    bounds = [n_reads * q // P for q in range(P + 1)]
    packed, send = [], []
    off = 0
    for r in range(P):
        pk, cnts = ops[r].owned_results(cid_g[off:off + u_all[r]].clone(), ismax_g[off:off + u_all[r]].clone(), bounds)
        packed.append(pk.clone())
        send.append(cnts)
        off += u_all[r]
    for q in range(P):
        lw, lf = g_w[bounds[q]:bounds[q + 1]].clone(), g_f[bounds[q]:bounds[q + 1]].clone()
        perm, recv_counts = ops[q].owner_perm(lw, lf, ranges)
        assert recv_counts == [send[o][q] for o in range(P)]
        parts = [packed[o][sum(send[o][:q]):sum(send[o][:q + 1])] for o in range(P)]
        recv = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int32, device=dev)
        oc = torch.empty(bounds[q + 1] - bounds[q], dtype=torch.int32, device=dev)
        ok = torch.empty(bounds[q + 1] - bounds[q], dtype=torch.uint8, device=dev)
        ops[q].scatter(perm, recv, oc, ok)
        assert np.array_equal(oc.cpu().numpy().view(np.uint32), ocid[bounds[q]:bounds[q + 1]])
        assert np.array_equal(ok.cpu().numpy(), okeep[bounds[q]:bounds[q + 1]])
    # dense count variant (owned reads compacted, LDS tables): same unique arrays, same streams
    off = 0
    for r in range(P):
        lo, hi, exp = ranges[r]
        u, us, cnts = ops[r].count_dense(g_w, g_f, n, lo, hi, bounds)
        assert u == u_all[r] and cnts == send[r]
        w, c = ops[r].unique()
        assert torch.equal(w, uw[r]) and torch.equal(c, uc[r])
        pk = ops[r].map_dense(cid_g[off:off + u].clone(), ismax_g[off:off + u].clone())
        assert torch.equal(pk, packed[r])
        off += u
    for o in ops:
        o.close()


def test_sharded_world1_nccl():
    """ShardedDedup end to end over RCCL with a single rank (all collectives degenerate)."""
    import socket
    import torch
    import torch.distributed as dist
    from humid_amd.sharded import ShardedDedup
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=dev)
    try:
        words, filt = synth_words(300_000, 31, 24, p_sub=2e-3)
        ocid, okeep, osum, _ = orc.dedup_run(words, filt, 24, 1, 0)
        d_w = torch.from_numpy(words.view(np.int64)).to(dev)
        d_f = torch.from_numpy(filt).to(dev)
        d_c = torch.zeros(len(words), dtype=torch.int32, device=dev)
        d_k = torch.zeros(len(words), dtype=torch.uint8, device=dev)
        sd = ShardedDedup(device=0, word_nt=24, distance=1)
        for _ in range(2):
            s = sd.run(d_w, d_f, d_c, d_k)
        assert np.array_equal(d_c.cpu().numpy().view(np.uint32), ocid)
        assert np.array_equal(d_k.cpu().numpy(), okeep)
        for k in ("total", "usable", "unique", "clusters"):
            assert s[k] == osum[k]
        # the same through the humid_comm callbacks (all_to_all_single / all_gather_into_tensor of RCCL on
        # views of the library's buffers): with one rank the library would not call them by itself
        d_c.zero_()
        d_k.zero_()
        sd.ops.set_option("force_comm", 1)
        for _ in range(2):
            s = sd.run(d_w, d_f, d_c, d_k)
        sd.ops.set_option("force_comm", 0)
        assert np.array_equal(d_c.cpu().numpy().view(np.uint32), ocid)
        assert np.array_equal(d_k.cpu().numpy(), okeep)
        assert s["edges"] == osum["edges"] and s["clusters"] == osum["clusters"]
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------
# device-side word packing (makeWord, src/fastq.cc:146-161, on the GPU: humid_dedup_run_bases)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 5, 23, 24, 32, 33, 47, 64])
def test_device_packing_matches_make_word(dd1, n):
    """raw symbols in (ACGT, N, lower case, other bytes), packed words + filtered flags out: equal to
    the oracle's makeWord restatement symbol by symbol, and the run on them equal to the run on words"""
    rng = np.random.default_rng(n)
    for n_reads in (0, 1, 255, 256, 257, 5000):
        alphabet = np.frombuffer(b"ACGTACGTACGTACGTACGTACGTACGTNacgt.*", dtype=np.uint8)
        mol = rng.integers(0, max(n_reads // 3, 1), size=n_reads)
        clean = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(max(n_reads // 3, 1), n))]
        bases = clean[mol].copy() if n_reads else np.zeros((0, n), np.uint8)
        noise = rng.random(bases.shape) < 0.01
        bases[noise] = alphabet[rng.integers(0, len(alphabet), size=int(noise.sum()))]
        cid, keep, s = dd1.run_bases(bases, word_nt=n, distance=1)
        w, f = dd1.packed_words()
        # the oracle's makeWord on every row
        ew = np.zeros((n_reads, 2) if n > 32 else n_reads, np.uint64)
        ef = np.zeros(n_reads, np.uint8)
        for i in range(n_reads):
            data, fl = orc.make_word(bases[i].tobytes().decode("latin-1"))
            if n > 32:
                ew[i, 0], ew[i, 1] = orc.pack_word(data[:n - 32]), orc.pack_word(data[n - 32:])
            else:
                ew[i] = orc.pack_word(data)
            ef[i] = fl
        assert np.array_equal(f, ef)
        assert np.array_equal(w, ew)
        if n_reads:
            cid2, keep2, s2 = dd1.run(ew, ef, word_nt=n, distance=1)
            assert np.array_equal(cid, cid2) and np.array_equal(keep, keep2)
            assert all(s[k] == s2[k] for k in ("total", "usable", "unique", "clusters", "edges"))
