"""-m gpu parity tests of the edit-distance mode (-e, src/humid.cc:140-158): the HIP path through the
C ABI against the CPU oracle's trie Levenshtein search, bit-exact (cluster ids, keep flags, leaves,
adjacency, clusters)."""
import numpy as np
import pytest

import humid_amd
from humid_amd.synth import synth_words
from oracle import pyoracle as orc
from test_oracle_vs_bruteforce import indel_words

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dd():
    d = humid_amd.Dedup()
    yield d
    d.close()


def check_edit(dd, words, filt, n, d, maximum, deep=True):
    cid, keep, s = dd.run(words, filt, word_nt=n, distance=d, method=int(maximum), edit=True)
    p = orc.Pipeline(n)
    p.read_data(words, filt)
    p.find_edit_neighbours(d)
    p.find_clusters(maximum)
    ocid, okeep = p.map_reads()
    os_ = p.summary()
    for k in ("total", "usable", "unique", "clusters", "edges"):
        assert s[k] == os_[k], (k, s[k], os_[k])
    assert np.array_equal(cid, ocid) and np.array_equal(keep, okeep)
    if deep and s["unique"]:
        lv, olv = dd.leaves(), p.leaves()
        assert np.array_equal(lv["degree"], olv["degree"])
        assert np.array_equal(lv["cluster_id"], olv["cluster_id"])
        off, idx = dd.adjacency()
        ooff, oidx = p.adjacency()
        assert np.array_equal(off.astype(np.uint64), ooff) and np.array_equal(idx, oidx)
        cl, ocl = dd.clusters(), p.clusters()
        assert np.array_equal(cl["size"], ocl["size"]) and np.array_equal(cl["max_leaf"], ocl["max_leaf"])
    return s


@pytest.mark.parametrize("seed", range(5))
@pytest.mark.parametrize("d", [2, 3])
@pytest.mark.parametrize("maximum", [False, True])
def test_edit_indel_families(dd, seed, d, maximum):
    rng = np.random.default_rng(300 + seed)
    n = int(rng.integers(6, 33))
    words = indel_words(rng, int(rng.integers(500, 6000)), n)
    filt = (rng.random(len(words)) < 0.02).astype(np.uint8)
    check_edit(dd, words, filt, n, d, maximum)


@pytest.mark.parametrize("seed", range(4))
@pytest.mark.parametrize("d", [4, 5])
@pytest.mark.parametrize("maximum", [False, True])
def test_edit_two_indel_pairs(dd, seed, d, maximum):
    """-e -m 4 / 5: up to two insertions and two deletions (offset vectors in [-2, 2], five-diagonal
    dynamic programme).  Families whose members carry several indels, so pairs at distance 4 and 5 with
    text shifted by two positions exist; every array against the oracle's trie search."""
    rng = np.random.default_rng(900 + 10 * d + seed)
    n = int(rng.integers(8, 25))
    n_reads = int(rng.integers(300, 2500))
    bases = rng.integers(0, 4, size=(max(2, n_reads // 15), n))
    words = np.zeros(n_reads, dtype=np.uint64)
    for r in range(n_reads):
        sq = bases[rng.integers(0, len(bases))].tolist()
        for _ in range(int(rng.integers(0, 3))):            # up to two deletion + insertion events
            del sq[int(rng.integers(0, n))]
            sq.insert(int(rng.integers(0, n)), int(rng.integers(0, 4)))
        if rng.random() < 0.3:
            sq[int(rng.integers(0, n))] = int(rng.integers(0, 4))
        words[r] = orc.pack_word(sq)
    filt = (rng.random(n_reads) < 0.02).astype(np.uint8)
    check_edit(dd, words, filt, n, d, maximum)


def test_edit_two_shifted_stretches_are_found(dd):
    """a hand-made pair at edit distance 4 whose middle is shifted by TWO positions: found with -m 4,
    not with -m 3"""
    n = 20
    a = [0, 1, 2, 3, 0, 1, 2, 3, 1, 1, 2, 2, 3, 3, 0, 0, 1, 2, 3, 0]
    b = a[2:] + [2, 1]                      # two deletions in front, two insertions at the end
    c = a[:3] + a[5:] + [3, 3]              # two deletions inside, two insertions at the end
    words = np.array([orc.pack_word(x) for x in (a, b, c)] * 3, dtype=np.uint64)
    filt = np.zeros(len(words), np.uint8)
    s4 = check_edit(dd, words, filt, n, 4, False)
    s3 = check_edit(dd, words, filt, n, 3, False)
    assert s4["edges"] > s3["edges"]


@pytest.mark.parametrize("n", [4, 9, 24, 32])
@pytest.mark.parametrize("d", [2, 3])
def test_edit_finds_more_than_hamming(dd, n, d):
    """shifted variants are edit neighbours but far apart in Hamming distance"""
    rng = np.random.default_rng(n * 7 + d)
    words = indel_words(rng, 4000, n, p_indel=0.6)
    filt = np.zeros(len(words), np.uint8)
    se = check_edit(dd, words, filt, n, d, False)
    _, _, sh = dd.run(words, filt, word_nt=n, distance=d)          # Hamming run on the same words
    assert se["edges"] >= sh["edges"]
    if n >= 9:
        assert se["edges"] > sh["edges"]


def test_edit_distance_one_and_zero_equal_hamming(dd):
    words, filt = synth_words(50_000, 3, 24, p_sub=5e-3)
    for d in (0, 1):
        a = dd.run(words, filt, word_nt=24, distance=d, edit=True)
        b = dd.run(words, filt, word_nt=24, distance=d, edit=False)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2]["edges"] == b[2]["edges"]


def test_edit_synthetic_umi_at_size(dd):
    """the metric's data shape (substitution errors only) at a size the oracle's trie search finishes
    in seconds: edit distance 2 still has to agree pair for pair"""
    words, filt = synth_words(120_000, 17, 24, p_sub=5e-3, p_n=1e-3)
    check_edit(dd, words, filt, 24, 2, False, deep=False)


@pytest.mark.parametrize("segs", [3, 4, 5, 6])
def test_edit_forced_plans(segs):
    dq = humid_amd.Dedup()
    dq.set_option("plan_segments", segs)
    rng = np.random.default_rng(segs)
    words = indel_words(rng, 3000, 24, p_indel=0.5)
    check_edit(dq, words, np.zeros(len(words), np.uint8), 24, 2, False)
    dq.close()


def many_indel_words(rng, n_reads, n, events):
    bases = rng.integers(0, 4, size=(max(2, n_reads // 15), n))
    words = np.zeros(n_reads, dtype=np.uint64)
    for r in range(n_reads):
        sq = bases[rng.integers(0, len(bases))].tolist()
        for _ in range(int(rng.integers(0, events + 1))):     # deletion + insertion events
            del sq[int(rng.integers(0, n))]
            sq.insert(int(rng.integers(0, n)), int(rng.integers(0, 4)))
        for _ in range(int(rng.integers(0, 3))):
            sq[int(rng.integers(0, n))] = int(rng.integers(0, 4))
        words[r] = orc.pack_word(sq)
    return words


@pytest.mark.parametrize("seed", range(3))
@pytest.mark.parametrize("d", [6, 7, 8, 11])
@pytest.mark.parametrize("maximum", [False, True])
def test_edit_any_distance(dd, seed, d, maximum):
    """-e beyond 5 edits (round 3; the reference's trie search has no limit, src/humid.cc:140-158): three and more
    insertion/deletion pairs -- offsets to +-d/2 in the joins, candidates verified by the whole dynamic programme
    (bit vectors: LevX<0>).  Families with up to five indel events per member; every array against the oracle."""
    rng = np.random.default_rng(7000 + 10 * d + seed)
    n = int(rng.integers(max(12, d + 2), 33))
    n_reads = int(rng.integers(300, 2000))
    words = many_indel_words(rng, n_reads, n, 5)
    filt = (rng.random(n_reads) < 0.02).astype(np.uint8)
    check_edit(dd, words, filt, n, d, maximum)


@pytest.mark.parametrize("n,d", [(5, 6), (8, 8), (12, 30), (24, 21), (7, 6)])
def test_edit_distances_that_compare_all_pairs(dd, n, d):
    """d >= n (every pair is a neighbour pair) and d beyond every plan (one join of everything with everything,
    verified)"""
    rng = np.random.default_rng(100 * n + d)
    words = many_indel_words(rng, 700, n, 3)
    filt = (rng.random(len(words)) < 0.02).astype(np.uint8)
    check_edit(dd, words, filt, n, d, False)
    check_edit(dd, words, filt, n, d, True, deep=False)


def test_edit_option_does_not_stick(dd):
    w = np.zeros(4, np.uint64)
    f = np.zeros(4, np.uint8)
    dd.run(w, f, word_nt=24, distance=6, edit=True)
    dd.run(w, f, word_nt=24, distance=1, edit=False)


def wide_indel_words(rng, n_reads, n):
    """indel_words for 33 <= n <= 64: u64[N, 2] ([:, 0] = first n-32 nucleotides, [:, 1] = last 32)"""
    bases = rng.integers(0, 4, size=(max(2, n_reads // 12), n))
    out = np.zeros((n_reads, 2), dtype=np.uint64)
    for r in range(n_reads):
        s = bases[rng.integers(0, len(bases))].tolist()
        if rng.random() < 0.4:
            del s[int(rng.integers(0, n))]
            s.insert(int(rng.integers(0, n)), int(rng.integers(0, 4)))
        if rng.random() < 0.4:
            s[int(rng.integers(0, n))] = int(rng.integers(0, 4))
        hi = lo = 0
        for x in s[:n - 32]:
            hi = (hi << 2) | x
        for x in s[n - 32:]:
            lo = (lo << 2) | x
        out[r] = (hi, lo)
    return out


@pytest.mark.parametrize("n", [33, 40, 48, 64])
@pytest.mark.parametrize("d", [2, 3, 6])
def test_edit_wide_words(dd, n, d):
    """-e with two-word (wide) words: shifts across the word boundary, keys cut to 64 bits"""
    rng = np.random.default_rng(n + d)
    words = wide_indel_words(rng, 2500, n)
    filt = (rng.random(len(words)) < 0.02).astype(np.uint8)
    check_edit(dd, words, filt, n, d, False)
    check_edit(dd, words, filt, n, d, True, deep=False)


def test_edit_joins_cut_long_runs_into_pieces(dd):
    """round 3: 14 000 words that share their last 12 nucleotides -- one run of 14 000
    equal keys in the joins of the combinations made of the last segments.  A lane verifies at most bucket_walk
    candidates; the COUNT pass notices the long run and the join is redone in pieces (k_edit_chunks /
    k_edit_join_chunks).  -e -m 2, ids / flags / degrees / adjacency against the oracle."""
    rng = np.random.default_rng(77)
    heads = rng.choice(1 << 24, size=14_000, replace=False).astype(np.uint64)      # (30 000: 34 s of oracle on the GPU box)
    uniq = (heads << np.uint64(24)) | np.uint64(0x96a53c)
    words = np.repeat(uniq, rng.poisson(0.3, size=len(uniq)) + 1)
    extra, ef = synth_words(20_000, 6, 24, p_sub=4e-3, p_n=1e-3)
    words = np.concatenate([words, extra])
    filt = np.concatenate([np.zeros(len(words) - len(extra), np.uint8), ef])
    perm = rng.permutation(len(words))
    words, filt = words[perm], filt[perm]
    check_edit(dd, words, filt, 24, 2, False, deep=True)
    dd.set_option("bucket_walk", 37)                     # nearly every run in pieces
    try:
        check_edit(dd, words[:20_000], filt[:20_000], 24, 3, True, deep=False)
    finally:
        dd.set_option("bucket_walk", 1024)
