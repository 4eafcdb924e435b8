"""C oracle (trie + explicit stacks) against the independent numpy/Python checker."""
import numpy as np
import pytest

import bruteforce as bf
from humid_amd.synth import synth_words
from oracle import pyoracle as orc


def run_oracle(words, filt, n, d, maximum):
    p = orc.Pipeline(n)
    p.read_data(words, filt)
    p.find_hamming_neighbours(d)
    p.find_clusters(maximum)
    cid, keep = p.map_reads()
    return p, cid, keep


def dense_words(rng, n_reads, n, alphabet_positions=4):
    """words confined to a small sub-space so that neighbours and ties are common"""
    base = rng.integers(0, 4 ** n, dtype=np.uint64)
    w = np.full(n_reads, base, dtype=np.uint64)
    for _ in range(alphabet_positions):
        pos = int(rng.integers(0, n))
        sh = np.uint64(2 * pos)
        v = rng.integers(0, 4, size=n_reads).astype(np.uint64)
        w = (w & ~(np.uint64(3) << sh)) | (v << sh)
    return w


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("d", [1, 2])
@pytest.mark.parametrize("maximum", [False, True])
def test_dense_small(seed, d, maximum):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(4, 13))
    n_reads = int(rng.integers(1, 400))
    words = dense_words(rng, n_reads, n, alphabet_positions=int(rng.integers(1, 5)))
    filt = (rng.random(n_reads) < 0.05).astype(np.uint8)
    p, cid, keep = run_oracle(words, filt, n, d, maximum)
    bcid, bkeep, det = bf.dedup(words, filt, d, maximum)
    lv = p.leaves()
    assert np.array_equal(lv["word"], det["unique"])
    assert np.array_equal(lv["count"], det["count"].astype(np.uint64))
    off, idx = p.adjacency()
    for u in range(p.unique):
        assert idx[int(off[u]):int(off[u + 1])].tolist() == det["nbrs"][u]
    assert lv["cluster_id"].tolist() == det["leaf_cluster"]
    assert np.array_equal(cid, bcid)
    assert np.array_equal(keep, bkeep)


@pytest.mark.parametrize("cfg", [(2000, 24, 1, "umi"), (1500, 24, 2, "genome"), (800, 32, 1, "umi"),
                                 (500, 1, 1, "umi"), (700, 3, 2, "umi")])
def test_synth(cfg):
    n_reads, n, d, mode = cfg
    words, filt = synth_words(n_reads, 7, n, p_sub=0.01, p_n=0.002, mode=mode,
                              genome_bp=5000 if mode == "genome" else 0)
    p, cid, keep = run_oracle(words, filt, n, d, False)
    bcid, bkeep, det = bf.dedup(words, filt, d, False)
    assert np.array_equal(cid, bcid)
    assert np.array_equal(keep, bkeep)
    assert p.unique == len(det["unique"])


def test_empty_and_all_filtered():
    p, cid, keep = run_oracle(np.zeros(0, np.uint64), np.zeros(0, np.uint8), 24, 1, False)
    assert p.unique == 0 and len(cid) == 0
    w = np.arange(5, dtype=np.uint64)
    p, cid, keep = run_oracle(w, np.ones(5, np.uint8), 24, 1, False)
    assert p.unique == 0 and cid.tolist() == [0] * 5 and keep.tolist() == [0] * 5


def test_keep_is_first_read_of_max_leaf():
    # family: centre x5 (reads 1,3,4,5,6), satellite x1 first in file
    c = orc.pack_word([0, 1, 2, 3])
    s = orc.pack_word([0, 1, 2, 0])
    words = np.array([s, c, s ^ s ^ s, c, c, c, c], dtype=np.uint64)
    words[2] = s
    filt = np.zeros(7, np.uint8)
    p, cid, keep = run_oracle(words, filt, 4, 1, False)
    assert cid.tolist() == [1] * 7
    assert keep.tolist() == [0, 1, 0, 0, 0, 0, 0]


def test_deep_chain_no_stack_overflow():
    # counts halve along a long path graph: the reference recursion would be U deep
    n = 200000
    g = orc.Graph([1] * n)
    for i in range(n - 1):
        g.link(i, i + 1)
    assert g.find_clusters(True) == 1


@pytest.mark.parametrize("n", [33, 40, 48, 63, 64])
@pytest.mark.parametrize("d", [0, 1, 2])
@pytest.mark.parametrize("maximum", [False, True])
def test_wide_words(n, d, maximum):
    """33 <= n <= 64: two uint64 per word ([hi, lo], oracle/humid_oracle.h)"""
    from humid_amd.synth import synth_wide_words
    words, filt = synth_wide_words(500, 7 + n, n, p_sub=0.02, p_n=0.01)
    p, cid, keep = run_oracle(words, filt, n, d, maximum)
    bcid, bkeep, det = bf.dedup(words, filt, d, maximum)
    assert np.array_equal(cid, bcid) and np.array_equal(keep, bkeep)
    lv = p.leaves()
    assert np.array_equal(lv["word"], det["unique"])
    assert np.array_equal(lv["count"], det["count"].astype(np.uint64))
    off, idx = p.adjacency()
    flat = [x for row in det["nbrs"] for x in row]
    assert idx.tolist() == flat


def test_wide_matches_one_word_embedding():
    """a constant prefix in front of 32-nt words changes nothing"""
    lo, filt = synth_words(2000, 3, word_nt=32, p_sub=0.01)
    wide = np.stack([np.full(len(lo), 0x2d, dtype=np.uint64), lo], axis=1)
    _, c1, k1 = run_oracle(lo, filt, 32, 1, False)
    _, c2, k2 = run_oracle(wide, filt, 36, 1, False)
    assert np.array_equal(c1, c2) and np.array_equal(k1, k2)


def indel_words(rng, n_reads, n, p_indel=0.3):
    """families of a few base words; variants by a deletion + an insertion (one shifted stretch),
    substitutions, or both -- the pairs edit distance finds and Hamming distance does not"""
    bases = rng.integers(0, 4, size=(max(2, n_reads // 12), n))
    out = np.zeros(n_reads, dtype=np.uint64)
    for r in range(n_reads):
        s = bases[rng.integers(0, len(bases))].tolist()
        if rng.random() < p_indel:
            i = int(rng.integers(0, n))
            del s[i]
            s.insert(int(rng.integers(0, n)), int(rng.integers(0, 4)))
        for _ in range(int(rng.integers(0, 3))):
            if rng.random() < 0.4:
                s[int(rng.integers(0, n))] = int(rng.integers(0, 4))
        w = 0
        for x in s:
            w = (w << 2) | x
        out[r] = w
    return out


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("d", [1, 2, 3])
@pytest.mark.parametrize("maximum", [False, True])
def test_edit_distance_neighbours(seed, d, maximum):
    """-e: the trie Levenshtein search of the oracle against an all-pairs dynamic programme"""
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(5, 17))
    words = indel_words(rng, int(rng.integers(20, 220)), n)
    filt = (rng.random(len(words)) < 0.03).astype(np.uint8)
    cid, keep, summ, _ = orc.dedup_run(words, filt, n, d, int(maximum), edit=True)
    bcid, bkeep, det = bf.dedup(words, filt, d, maximum, edit_nt=n)
    assert np.array_equal(cid, bcid) and np.array_equal(keep, bkeep)
    p = orc.Pipeline(n)
    p.read_data(words, filt)
    p.find_edit_neighbours(d)
    off, idx = p.adjacency()
    assert idx.tolist() == [x for row in det["nbrs"] for x in row]


@pytest.mark.parametrize("seed", range(3))
@pytest.mark.parametrize("d", [4, 5, 6, 7, 9, 12])
def test_edit_distance_neighbours_beyond_two_indel_pairs(seed, d):
    """-e at larger distances (the HIP path has no limit since round 3 and is checked against this oracle at d = 6 .. 30,
    tests/test_gpu_edit.py): families whose members carry up to five deletion + insertion events and a few
    substitutions; the trie search against the all-pairs dynamic programme, both methods; d >= n included"""
    rng = np.random.default_rng(900 + 17 * d + seed)
    n = int(rng.integers(6, 15))
    n_reads = int(rng.integers(20, 120))
    bases = rng.integers(0, 4, size=(max(2, n_reads // 10), n))
    words = np.zeros(n_reads, dtype=np.uint64)
    for r in range(n_reads):
        sq = bases[rng.integers(0, len(bases))].tolist()
        for _ in range(int(rng.integers(0, 6))):
            del sq[int(rng.integers(0, n))]
            sq.insert(int(rng.integers(0, n)), int(rng.integers(0, 4)))
        for _ in range(int(rng.integers(0, 3))):
            sq[int(rng.integers(0, n))] = int(rng.integers(0, 4))
        words[r] = orc.pack_word(sq)
    filt = (rng.random(n_reads) < 0.03).astype(np.uint8)
    for maximum in (False, True):
        cid, keep, summ, _ = orc.dedup_run(words, filt, n, d, int(maximum), edit=True)
        bcid, bkeep, det = bf.dedup(words, filt, d, maximum, edit_nt=n)
        assert np.array_equal(cid, bcid) and np.array_equal(keep, bkeep)
    p = orc.Pipeline(n)
    p.read_data(words, filt)
    p.find_edit_neighbours(d)
    off, idx = p.adjacency()
    assert idx.tolist() == [x for row in det["nbrs"] for x in row]


@pytest.mark.parametrize("n,d", [(33, 2), (40, 6), (48, 7), (64, 3)])
def test_edit_distance_neighbours_of_two_word_words(n, d):
    """-e on words of 33 .. 64 nucleotides (insertions and deletions that cross the word boundary), the oracle's trie
    search against the all-pairs dynamic programme"""
    rng = np.random.default_rng(3300 + n + d)
    n_reads = 48
    bases = rng.integers(0, 4, size=(5, n))
    words = np.zeros((n_reads, 2), dtype=np.uint64)
    for r in range(n_reads):
        sq = bases[rng.integers(0, len(bases))].tolist()
        for _ in range(int(rng.integers(0, 4))):
            del sq[int(rng.integers(0, n))]
            sq.insert(int(rng.integers(0, n)), int(rng.integers(0, 4)))
        for _ in range(int(rng.integers(0, 3))):
            sq[int(rng.integers(0, n))] = int(rng.integers(0, 4))
        hi = lo = 0
        for x in sq[:n - 32]:
            hi = (hi << 2) | x
        for x in sq[n - 32:]:
            lo = (lo << 2) | x
        words[r] = (hi, lo)
    filt = (rng.random(n_reads) < 0.03).astype(np.uint8)
    for maximum in (False, True):
        cid, keep, summ, _ = orc.dedup_run(words, filt, n, d, int(maximum), edit=True)
        bcid, bkeep, det = bf.dedup(words, filt, d, maximum, edit_nt=n)
        assert np.array_equal(cid, bcid) and np.array_equal(keep, bkeep)
    assert summ["edges"] == sum(len(row) for row in det["nbrs"]) // 2 and summ["edges"] > 0


def test_edit_distance_one_is_hamming_distance_one():
    """equal-length words: one edit can only be a substitution"""
    words, filt = synth_words(3000, 9, 12, p_sub=0.03)
    a = orc.dedup_run(words, filt, 12, 1, 0, edit=True)
    b = orc.dedup_run(words, filt, 12, 1, 0, edit=False)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
