"""-m gpu parity tests for wide words (33 <= word_nt <= 64, two uint64 per read): the HIP path
through the C ABI against the CPU oracle, bit-exact (include/humid_hip.h, kernels_wide.hip.h)."""
import numpy as np
import pytest

import humid_amd
from humid_amd.synth import synth_wide_words, synth_words
from oracle import pyoracle as orc
from test_gpu_parity import check_against_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dd():
    d = humid_amd.Dedup()
    yield d
    d.close()


def dense_wide(rng, n_reads, n, k):
    """reads around one base word: k random positions (both halves) re-drawn per read"""
    nh = n - 32
    w = np.zeros((n_reads, 2), dtype=np.uint64)
    w[:, 0] = rng.integers(0, 1 << min(2 * nh, 62), dtype=np.uint64)
    w[:, 1] = rng.integers(0, 1 << 62, dtype=np.uint64)
    for _ in range(k):
        pos = int(rng.integers(0, n))
        col, sh = (0, 2 * (nh - 1 - pos)) if pos < nh else (1, 2 * (n - 1 - pos))
        v = rng.integers(0, 4, size=n_reads).astype(np.uint64)
        w[:, col] = (w[:, col] & ~(np.uint64(3) << np.uint64(sh))) | (v << np.uint64(sh))
    return w


@pytest.mark.parametrize("n", [33, 40, 48, 63, 64])
@pytest.mark.parametrize("d", [0, 1, 2])
@pytest.mark.parametrize("maximum", [False, True])
def test_wide_synth_parity(dd, n, d, maximum):
    w, f = synth_wide_words(3000, 100 + n + d, n, p_sub=0.01, p_n=0.01)
    s = check_against_oracle(dd, w, f, n, d, maximum)
    assert s["count_mode_used"] == 3


@pytest.mark.parametrize("n_reads", [0, 1, 2, 5, 257])
def test_wide_tiny(dd, n_reads):
    w, f = synth_wide_words(n_reads, 5 + n_reads, 48, p_sub=0.05, p_n=0.05)
    check_against_oracle(dd, w, f, 48, 1, False)


@pytest.mark.parametrize("n,k,d", [(40, 4, 1), (64, 5, 2), (33, 6, 1), (48, 3, 3)])
def test_wide_dense_neighbourhoods(dd, n, k, d):
    rng = np.random.default_rng(n * 10 + k)
    w = dense_wide(rng, 6000, n, k)
    f = (rng.random(6000) < 0.02).astype(np.uint8)
    check_against_oracle(dd, w, f, n, d, False)
    check_against_oracle(dd, w, f, n, d, True)


def test_wide_all_filtered_and_all_equal(dd):
    w = np.zeros((100, 2), dtype=np.uint64)
    check_against_oracle(dd, w, np.ones(100, np.uint8), 48, 1, False)
    check_against_oracle(dd, w, np.zeros(100, np.uint8), 48, 1, False)


def test_wide_n64_extremes(dd):
    """n = 64: no spare key bit for the filtered flag (third sort pass); the all-T word equals the
    all-ones pattern, filtered reads carry it too"""
    rng = np.random.default_rng(64)
    w = np.full((500, 2), np.uint64(0xffffffffffffffff))
    w[::3, 1] ^= np.uint64(1)            # last nucleotide T -> G: a distance-1 neighbour
    w[::7, 0] = np.uint64(0)
    f = (rng.random(500) < 0.3).astype(np.uint8)
    check_against_oracle(dd, w, f, 64, 1, False)
    check_against_oracle(dd, w, f, 64, 2, True)


def test_wide_larger(dd):
    w, f = synth_wide_words(200_000, 77, 50)
    s = check_against_oracle(dd, w, f, 50, 1, False, deep=False)
    assert s["count_mode_used"] == 2          # evenly spread heads: LDS tables over head-ordered buckets
    check_against_oracle(dd, w, f, 50, 2, False, deep=False)


def test_full_umi_variant_n36(dd):
    """SURVEY 8(d), config 3's full-UMI variant: `-n 36` = 12 + 12 + 12 nucleotides (72 bits, two uint64 per
    word).  500 k reads, d = 1, both methods, every array against the oracle; counted in LDS tables."""
    w, f = synth_wide_words(500_000, 1003, 36)
    s = check_against_oracle(dd, w, f, 36, 1, False, deep=True)
    assert s["count_mode_used"] == 2
    check_against_oracle(dd, w, f, 36, 1, True, deep=False)


@pytest.mark.parametrize("n", [33, 40, 48, 63, 64])
def test_wide_lds_buckets_against_the_sort(n):
    """round 2: wide words counted in LDS tables (buckets cut by the words' top 64 bits, entries claimed by
    position, k_dedup_lds_wide) -- against the oracle and, array for array, against the sorting count"""
    w, f = synth_wide_words(150_000, 900 + n, n, p_sub=4e-3, p_n=1e-3)
    a = humid_amd.Dedup()
    s = check_against_oracle(a, w, f, n, 1, False, deep=False)
    assert s["count_mode_used"] == 2
    cid_a, keep_a, sa = a.run(w, f, word_nt=n, distance=2)
    b = humid_amd.Dedup()
    b.set_option("count_order", 0)            # keeps the sort
    cid_b, keep_b, sb = b.run(w, f, word_nt=n, distance=2)
    assert sa["count_mode_used"] == 2 and sb["count_mode_used"] == 3
    assert np.array_equal(cid_a, cid_b) and np.array_equal(keep_a, keep_b)
    for k in ("usable", "unique", "clusters", "edges"):
        assert sa[k] == sb[k]
    la, lb = a.leaves(), b.leaves()
    for k in la:
        assert np.array_equal(la[k], lb[k]), k
    a.close()
    b.close()


@pytest.mark.parametrize("n", [33, 41, 64])
@pytest.mark.parametrize("n_reads", [300, 3000, 40_000])
def test_wide_lds_buckets_forced_on_small_inputs(n, n_reads):
    dq = humid_amd.Dedup()
    dq.set_option("count_order", 1)
    w, f = synth_wide_words(n_reads, 7 * n + n_reads, n, p_sub=0.01, p_n=0.01)
    for d in (0, 1, 2):
        s = check_against_oracle(dq, w, f, n, d, d == 2)
        assert s["count_mode_used"] == 2
    dq.close()


def test_wide_lds_buckets_overflow_goes_back_to_the_sort():
    """all heads in one bucket (a constant 8-nt prefix in front of words that share 20 more nucleotides):
    forced head-ordered buckets overflow, the run is counted by sorting and says so"""
    rng = np.random.default_rng(5)
    n_reads = 20_000
    w = np.empty((n_reads, 2), dtype=np.uint64)
    w[:, 0] = np.uint64(0x1b1b)
    w[:, 1] = (np.uint64(0x123456789a) << np.uint64(24)) | rng.integers(0, 1 << 24, size=n_reads, dtype=np.uint64)
    f = (rng.random(n_reads) < 0.01).astype(np.uint8)
    dq = humid_amd.Dedup()
    dq.set_option("count_order", 1)
    s = check_against_oracle(dq, w, f, 40, 1, False, deep=False)
    assert s["count_mode_used"] == 3
    dq.close()


@pytest.mark.parametrize("n,order", [(33, 1), (64, 1), (48, -1)], ids=["33-ordered_buckets_forced", "64-ordered_buckets_forced", "48-default"])
def test_wide_sizes_where_the_partition_changes_shape(n, order):
    """two-word words at the read counts where the partition changes shape (tests/test_gpu_parity.py::
    test_sizes_where_the_partition_changes_shape: bucket bits, index bits, tile edges), a third of them per length"""
    from test_gpu_parity import SHAPE_SIZES
    dq = humid_amd.Dedup()
    dq.set_option("count_order", order)
    for N in SHAPE_SIZES[(n % 3)::3]:
        words, filt = synth_wide_words(N, 4000 + n, n, p_sub=3e-3, p_n=1e-3)
        check_against_oracle(dq, words, filt, n, 1, False, deep=False)
    dq.close()


@pytest.mark.parametrize("n,d,segs", [(64, 1, 3), (64, 1, 4), (48, 2, 5), (40, 1, 6), (64, 2, 3)])
def test_wide_forced_plans(n, d, segs):
    """plans whose combination keys exceed 64 bits are cut to 64 (make_plan): still every pair"""
    dq = humid_amd.Dedup()
    dq.set_option("plan_segments", segs)
    rng = np.random.default_rng(segs)
    w = np.concatenate([dense_wide(rng, 4000, n, 4), synth_wide_words(4000, segs, n, p_sub=0.02)[0]])
    f = np.zeros(len(w), np.uint8)
    check_against_oracle(dq, w, f, n, d, False)
    dq.close()


def test_wide_agrees_with_one_word_path(dd):
    """a constant 8-nt prefix in front of 32-nt words changes nothing: the n = 40 wide run must give
    exactly the clusters of the n = 32 one-word run"""
    lo, f = synth_words(50_000, 11, word_nt=32, p_sub=5e-3)
    w = np.stack([np.full(len(lo), 0x1b1b, dtype=np.uint64), lo], axis=1)
    cid_w, keep_w, sw = dd.run(w, f, word_nt=40, distance=1)
    d1 = humid_amd.Dedup()
    cid_n, keep_n, sn = d1.run(lo, f, word_nt=32, distance=1)
    d1.close()
    assert np.array_equal(cid_w, cid_n) and np.array_equal(keep_w, keep_n)
    for k in ("usable", "unique", "clusters", "edges"):
        assert sw[k] == sn[k]


def test_wide_bad_arguments(dd):
    w, f = synth_wide_words(10, 1, 48)
    with pytest.raises(ValueError):
        dd.run(w[:, 0], f, word_nt=48)
    with pytest.raises(humid_amd.HumidError) as e:
        dd.run(np.zeros((10, 2), np.uint64), f, word_nt=65)
    assert e.value.code == -2
