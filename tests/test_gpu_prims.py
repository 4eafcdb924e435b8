"""-m gpu: the device-wide exclusive scan and the LSD radix sort of humid_amd/csrc/prims.hip.h, driven directly
(tests/csrc/prims_harness.hip) at the sizes where their forms change and on key distributions the pipeline's own
inputs rarely produce -- against numpy (cumsum / stable argsort).  Integer work: bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K = 64 * 1024                      # items of the chained scan at one per thread (64 workgroups x 1024 threads)
SCAN_SIZES = sorted({1, 2, 63, 64, 65, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097, 8191, 8192, 8193, 16383, 16384, 16385,
                     K - 1, K, K + 1, 2 * K - 1, 2 * K, 2 * K + 1, 4 * K - 1, 4 * K, 4 * K + 1, 8 * K - 1, 8 * K, 8 * K + 1,
                     16 * K - 1, 16 * K, 16 * K + 1, 16 * K + 4096, 100_000, 1_000_003, 5_000_011})


@pytest.fixture(scope="module")
def ph():
    import prims_harness
    lib = prims_harness.load()
    yield lib
    lib.ph_release()


def dev(a):
    import torch
    return torch.from_numpy(a.view(np.int64 if a.dtype == np.uint64 else np.int32).copy()).to("cuda:0")


def host(t, dtype):
    return t.cpu().numpy().view(dtype)


@pytest.mark.parametrize("chain", [1, 0])
@pytest.mark.parametrize("width", [32, 64])
def test_exclusive_scan_at_every_form_boundary(ph, width, chain):
    """tiny / one workgroup / chained single launch with 1, 2, 4, 8 (16: 4-byte items) items per thread / reduce-top-down,
    each at its first and last size, out of place and in place (the sort scans its histograms in place)"""
    import torch
    rng = np.random.default_rng(7 + width + chain)
    dt = np.uint32 if width == 32 else np.uint64
    fn = ph.ph_exscan_u32 if width == 32 else ph.ph_exscan_u64
    for n in SCAN_SIZES:
        x = rng.integers(0, 1 << (12 if width == 32 else 40), size=n, dtype=np.uint64).astype(dt)
        if n > 10:
            x[rng.integers(0, n, size=3)] = 0
        want = np.concatenate([np.zeros(1, dt), np.cumsum(x, dtype=dt)[:-1]])
        d_in = dev(x)
        d_out = torch.full((n + 1,), -1, dtype=d_in.dtype, device="cuda:0")       # one guard item behind the output
        assert fn(d_in.data_ptr(), d_out.data_ptr(), n, chain) == 0
        got = host(d_out, dt)
        assert np.array_equal(got[:n], want), ("out of place", n, int(np.flatnonzero(got[:n] != want)[0]))
        assert got[n] == dt(-1 & ((1 << width) - 1)), ("wrote behind the output", n)
        assert fn(d_in.data_ptr(), d_in.data_ptr(), n, chain) == 0
        assert np.array_equal(host(d_in, dt), want), ("in place", n)


def test_scan_wraps_its_sums_like_the_type(ph):
    """sums beyond 2^32 in 4-byte items wrap (the callers scan counts that fit, the sort's histograms among them)"""
    import torch
    x = np.full(300_000, 0xfff0_0000 >> 4, dtype=np.uint32)
    want = np.concatenate([np.zeros(1, np.uint32), np.cumsum(x, dtype=np.uint32)[:-1]])
    for chain in (1, 0):
        d_in = dev(x)
        d_out = torch.empty_like(d_in)
        assert ph.ph_exscan_u32(d_in.data_ptr(), d_out.data_ptr(), len(x), chain) == 0
        assert np.array_equal(host(d_out, np.uint32), want)


def test_chained_scan_epoch_wraps_around(ph):
    """the chain's flags hold the epoch of the last launch that used a tile and are never cleared -- except when the
    32-bit epoch wraps: a flag left by launch 1 must not pass for launch 2^32 + 1's"""
    import torch
    rng = np.random.default_rng(99)
    n = 40 * 1024                                             # 40 tiles at one item per thread
    a = rng.integers(0, 1000, size=n, dtype=np.uint64).astype(np.uint32)
    b = rng.integers(0, 1000, size=n, dtype=np.uint64).astype(np.uint32)
    ph.ph_set_epoch(0)
    d_a, d_b = dev(a), dev(b)
    out = torch.empty_like(d_a)
    assert ph.ph_exscan_u32(d_a.data_ptr(), out.data_ptr(), n, 1) == 0 and ph.ph_epoch() == 1      # flags of 40 tiles: 1
    ph.ph_set_epoch(0xffffffff)
    assert ph.ph_exscan_u32(d_b.data_ptr(), out.data_ptr(), n, 1) == 0 and ph.ph_epoch() == 1      # 0 is skipped
    want = np.concatenate([np.zeros(1, np.uint32), np.cumsum(b, dtype=np.uint32)[:-1]])
    assert np.array_equal(host(out, np.uint32), want)


def sort_reference(keys, b0, b1):
    digit = (keys >> np.uint64(b0)) & np.uint64((1 << (b1 - b0)) - 1) if b1 - b0 < 64 else keys
    return np.argsort(digit, kind="stable")


SORT_SIZES = [1, 2, 63, 64, 65, 511, 512, 513, 4095, 4096, 4097, 8192 + 5, 70_001, 1_000_003]
BIT_RANGES = {32: [(0, 32), (0, 1), (0, 8), (0, 9), (0, 13), (3, 29), (24, 32), (5, 6)],
              64: [(0, 64), (0, 1), (0, 33), (32, 64), (7, 50), (0, 56), (60, 64)]}


@pytest.mark.parametrize("width", [32, 64])
@pytest.mark.parametrize("dist", ["uniform", "two_values", "one_value", "sorted", "reversed", "few_bits"])
def test_radix_sort_is_the_stable_sort_of_the_bit_range(ph, width, dist):
    """keys with values (given / the running index) and keys alone; every size where the tile count or the last
    pass changes; distributions that put a whole tile into one digit"""
    import torch
    rng = np.random.default_rng(len(dist) * 100 + width)
    dt = np.uint32 if width == 32 else np.uint64
    fn = ph.ph_sort_u32 if width == 32 else ph.ph_sort_u64
    for n in SORT_SIZES:
        full = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)
        if dist == "two_values":
            full = np.where(rng.random(n) < 0.5, full[0], full[-1])
        elif dist == "one_value":
            full = np.full(n, full[0])
        elif dist == "sorted":
            full = np.sort(full)
        elif dist == "reversed":
            full = np.sort(full)[::-1].copy()
        elif dist == "few_bits":
            full = full & np.uint64(0x0101_0101_0101_0101)
        keys = (full >> np.uint64(64 - width)).astype(dt)
        vals = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
        ranges = BIT_RANGES[width] if n in (65, 4097, 70_001) or dist == "uniform" else BIT_RANGES[width][:3]
        for b0, b1 in ranges:
            order = sort_reference(keys.astype(np.uint64), b0, b1)
            d_k, d_v = dev(keys), dev(vals)
            o_k = torch.full((n + 1,), -1, dtype=d_k.dtype, device="cuda:0")
            o_v = torch.full((n + 1,), -1, dtype=torch.int32, device="cuda:0")
            assert fn(d_k.data_ptr(), o_k.data_ptr(), d_v.data_ptr(), o_v.data_ptr(), n, b0, b1, 1, 0) == 0
            assert np.array_equal(host(o_k, dt)[:n], keys[order]), ("keys", n, b0, b1)
            assert np.array_equal(host(o_v, np.uint32)[:n], vals[order]), ("values", n, b0, b1)
            assert host(o_k, dt)[n] == dt(-1 & ((1 << width) - 1)) and host(o_v, np.uint32)[n] == 0xffffffff
            assert np.array_equal(host(d_k, dt), keys), "the input was written"
            assert fn(d_k.data_ptr(), o_k.data_ptr(), None, o_v.data_ptr(), n, b0, b1, 1, 1) == 0      # values: 0, 1, 2 ...
            assert np.array_equal(host(o_v, np.uint32)[:n], order.astype(np.uint32)), ("iota", n, b0, b1)
            o_k.fill_(-1)
            assert fn(d_k.data_ptr(), o_k.data_ptr(), None, None, n, b0, b1, 0, 0) == 0              # keys alone
            assert np.array_equal(host(o_k, dt)[:n], keys[order]), ("keys alone", n, b0, b1)


def test_radix_sort_large(ph):
    """20 M pairs, 64-bit keys, all eight passes (the size class of the count's fallback sort)"""
    import torch
    rng = np.random.default_rng(5)
    n = 20_000_003
    keys = rng.integers(0, 1 << 63, size=n, dtype=np.uint64)
    keys[::7] = keys[0]
    d_k = dev(keys)
    o_k = torch.empty_like(d_k)
    o_v = torch.empty(n, dtype=torch.int32, device="cuda:0")
    assert ph.ph_sort_u64(d_k.data_ptr(), o_k.data_ptr(), None, o_v.data_ptr(), n, 0, 64, 1, 1) == 0
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(host(o_v, np.uint32), order.astype(np.uint32))
    assert np.array_equal(host(o_k, np.uint64), keys[order])
