"""The host's own DEFLATE/gzip decoder (humid_amd/csrc/host/fast_inflate.cpp) against Python's zlib
on every block type and awkward input, through the CLI's --gunzip development switch (no GPU)."""
import gzip
import os
import subprocess
import zlib

import numpy as np
import pytest

from cli_util import HUMID
from humid_amd.synth import synth_fastq


def gunzip(tmp, blob, name="x"):
    src, dst = os.path.join(tmp, name + ".gz"), os.path.join(tmp, name + ".out")
    open(src, "wb").write(blob)
    rc = subprocess.call([HUMID, "--gunzip", src, dst], stderr=subprocess.DEVNULL)
    return rc, (open(dst, "rb").read() if rc == 0 else None)


def gz_member(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=31, memlevel=8):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, memlevel, strategy)
    return c.compress(data) + c.flush()


def payloads():
    rng = np.random.default_rng(0)
    text = (b"the quick brown fox jumps over the lazy dog. " * 3000)
    fastq = b"".join(b"@r%d_ACGTACGT\n%s\n+\n%s\n" % (i, bytes(rng.choice(list(b"ACGT"), 100).astype(np.uint8)), b"I" * 100)
                     for i in range(4000))
    return {
        "empty": b"",
        "one_byte": b"A",
        "short": b"hello, hello, hello",
        "text": text,
        "fastq": fastq,
        "random": rng.integers(0, 256, 300_000, dtype=np.uint8).tobytes(),           # incompressible: stored blocks
        "runs": b"".join(bytes([int(b)]) * int(n) for b, n in zip(rng.integers(0, 256, 500), rng.integers(1, 3000, 500))),
        "far_matches": (rng.integers(0, 256, 40_000, dtype=np.uint8).tobytes()) * 6,  # distances near 32 K
        "skewed": bytes(rng.choice([65, 66, 67, 200], p=[0.9, 0.05, 0.04, 0.01], size=200_000).astype(np.uint8)),
    }


@pytest.mark.parametrize("name", list(payloads().keys()))
def test_matches_zlib_on_all_block_types(name, tmp_path):
    data = payloads()[name]
    variants = {
        "l1": gz_member(data, 1), "l6": gz_member(data, 6), "l9": gz_member(data, 9),
        "stored": gz_member(data, 0),
        "fixed": gz_member(data, 6, zlib.Z_FIXED),
        "huffman_only": gz_member(data, 6, zlib.Z_HUFFMAN_ONLY),
        "rle": gz_member(data, 6, zlib.Z_RLE),
        "small_window": gz_member(data, 6, wbits=16 + 9, memlevel=1),
        "python_gzip": gzip.compress(data, 4),
    }
    for vname, blob in variants.items():
        rc, out = gunzip(str(tmp_path), blob, name + "_" + vname)
        assert rc == 0 and out == data, (name, vname, rc)


def test_multi_member_header_fields_and_flush_points(tmp_path):
    p = payloads()
    # members back to back, one of them empty
    blob = gz_member(p["text"]) + gz_member(b"") + gz_member(p["fastq"], 1) + gz_member(p["random"], 0)
    rc, out = gunzip(str(tmp_path), blob)
    assert rc == 0 and out == p["text"] + p["fastq"] + p["random"]
    # FNAME / FCOMMENT / FEXTRA / FHCRC in the header
    body = gz_member(p["short"])[10:]
    hdr = b"\x1f\x8b\x08" + bytes([4 | 8 | 16]) + b"\x00\x00\x00\x00\x00\x03" + b"\x05\x00HELLO" + b"name.fastq\x00" + b"a comment\x00"
    rc, out = gunzip(str(tmp_path), hdr + body, "hdr")
    assert rc == 0 and out == p["short"]
    # sync flushes create empty stored blocks and many small dynamic blocks
    c = zlib.compressobj(6, zlib.DEFLATED, 31)
    parts = []
    for k in range(0, len(p["fastq"]), 7001):
        parts.append(c.compress(p["fastq"][k:k + 7001]))
        parts.append(c.flush(zlib.Z_SYNC_FLUSH if k % 2 else zlib.Z_FULL_FLUSH))
    parts.append(c.flush())
    rc, out = gunzip(str(tmp_path), b"".join(parts), "flush")
    assert rc == 0 and out == p["fastq"]


def test_damaged_streams_are_declined(tmp_path):
    data = payloads()["fastq"]
    good = gz_member(data)
    assert gunzip(str(tmp_path), good)[0] == 0
    assert gunzip(str(tmp_path), good[:len(good) // 2], "cut")[0] == 3            # truncated
    assert gunzip(str(tmp_path), good[:-3], "cut_trailer")[0] == 3
    bad_crc = bytearray(good)
    bad_crc[-6] ^= 0x40
    assert gunzip(str(tmp_path), bytes(bad_crc), "crc")[0] == 3
    flipped = bytearray(good)
    flipped[len(good) // 3] ^= 0x10                                               # corrupt the bit stream
    rc, out = gunzip(str(tmp_path), bytes(flipped), "flip")
    assert rc == 3 or out != data                                                 # never a silent pass
    assert rc == 3
    assert gunzip(str(tmp_path), b"not a gzip file at all", "junk")[0] == 3
    assert gunzip(str(tmp_path), good + b"\x00\x00\x00\x00", "trailing")[0] == 3  # trailing garbage: zlib decides


def test_large_fastq_roundtrip(tmp_path):
    files = synth_fastq(str(tmp_path), 60000, 3, n_files=1, read_len=150)
    raw = open(files[0], "rb").read()
    for level in (1, 4, 9):
        rc, out = gunzip(str(tmp_path), gzip.compress(raw, level), "big%d" % level)
        assert rc == 0 and out == raw


@pytest.mark.parametrize("chunk", ["20000", "70000", "300000"])
@pytest.mark.parametrize("level", [1, 4, 9])
def test_parallel_two_pass_inflate(chunk, level, tmp_path):
    """the two-pass parallel decoder (block starts searched at the chunk cuts, 16-bit symbols with an
    unknown window, windows resolved afterwards) gives the bytes of the serial one; HUMID_TIMING shows
    which path ran"""
    files = synth_fastq(str(tmp_path), 40000, 5, n_files=1, read_len=150, p_sub=2e-3)
    raw = open(files[0], "rb").read()
    blob = gzip.compress(raw, level)
    src, dst = str(tmp_path / "p.gz"), str(tmp_path / "p.out")
    open(src, "wb").write(blob)
    env = dict(os.environ, HUMID_PAR_INFLATE_CHUNK=chunk, HUMID_THREADS="6", HUMID_TIMING="1")
    p = subprocess.run([HUMID, "--gunzip", src, dst], env=env, stderr=subprocess.PIPE)
    assert p.returncode == 0 and open(dst, "rb").read() == raw
    assert b"parallel" in p.stderr, p.stderr


def test_parallel_inflate_members_and_fallbacks(tmp_path):
    """several members (two large ones; bgzip-like 64 KB members whose only block is final) are
    decoded in parallel across the member boundaries; stored-only data and damaged input make the
    parallel attempt give up and the serial decoder (or its refusal) decides"""
    files = synth_fastq(str(tmp_path), 20000, 6, n_files=1, read_len=100)
    raw = open(files[0], "rb").read()
    env = dict(os.environ, HUMID_PAR_INFLATE_CHUNK="30000", HUMID_THREADS="4", HUMID_TIMING="1")
    cases = {
        "two_members": (gzip.compress(raw[:len(raw) // 2]) + gzip.compress(raw[len(raw) // 2:]), True),
        "bgzip_like": (b"".join(gzip.compress(raw[k:k + 65280], 6) for k in range(0, len(raw), 65280)), True),
        "stored": (gz_member(np.random.default_rng(1).integers(0, 256, 400_000, dtype=np.uint8).tobytes(), 0), False),
    }
    for name, (blob, parallel) in cases.items():
        src, dst = str(tmp_path / (name + ".gz")), str(tmp_path / (name + ".out"))
        open(src, "wb").write(blob)
        p = subprocess.run([HUMID, "--gunzip", src, dst], env=env, stderr=subprocess.PIPE)
        assert p.returncode == 0 and open(dst, "rb").read() == gzip.decompress(blob), name
        assert (b"parallel" in p.stderr) == parallel, (name, p.stderr)
    good = gzip.compress(raw, 6)
    for where in (len(good) // 2, len(good) - 6):              # the bit stream / the CRC
        bad = bytearray(good)
        bad[where] ^= 0x21
        src = str(tmp_path / "bad.gz")
        open(src, "wb").write(bytes(bad))
        assert subprocess.call([HUMID, "--gunzip", src, str(tmp_path / "bad.out")], env=env) == 3
    members = gzip.compress(raw[:300000]) + gzip.compress(raw[300000:])
    bad = bytearray(members)
    bad[len(gzip.compress(raw[:300000])) - 7] ^= 0x08            # CRC of the FIRST member
    open(str(tmp_path / "bad2.gz"), "wb").write(bytes(bad))
    assert subprocess.call([HUMID, "--gunzip", str(tmp_path / "bad2.gz"), str(tmp_path / "bad2.out")], env=env) == 3


def test_fuzzed_streams_never_crash_or_pass_silently(tmp_path):
    """bit flips, truncations, insertions, deletions and overwritten ranges: the decoders (serial and
    parallel) either decline or return exactly what zlib returns -- no crash, no hang"""
    import random
    files = synth_fastq(str(tmp_path), 6000, 5, n_files=1, read_len=120)
    raw = open(files[0], "rb").read()
    rnd = random.Random(7)
    base = [gzip.compress(raw, 6), gzip.compress(raw[:200000], 1) + gzip.compress(raw[200000:], 9),
            b"".join(gzip.compress(raw[k:k + 65280]) for k in range(0, len(raw), 65280))]
    for it in range(60):
        b = bytearray(rnd.choice(base))
        kind = it % 5
        if kind == 0:
            b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
        elif kind == 1:
            del b[rnd.randrange(len(b)):]
        elif kind == 2:
            p = rnd.randrange(len(b))
            b[p:p] = bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 50)))
        elif kind == 3:
            p = rnd.randrange(len(b))
            del b[p:p + rnd.randrange(1, 2000)]
        else:
            p = rnd.randrange(len(b))
            q = min(len(b), p + rnd.randrange(1, 5000))
            b[p:q] = bytes(rnd.randrange(256) for _ in range(q - p))
        src, dst = str(tmp_path / "f.gz"), str(tmp_path / "f.out")
        open(src, "wb").write(bytes(b))
        env = dict(os.environ, HUMID_THREADS="4")
        if it % 2 == 0:
            env["HUMID_PAR_INFLATE_CHUNK"] = str(rnd.choice([5000, 20000, 100000]))
        rc = subprocess.call([HUMID, "--gunzip", src, dst], env=env, timeout=60, stderr=subprocess.DEVNULL)
        assert rc in (0, 3), (it, kind, rc)
        if rc == 0:
            assert open(dst, "rb").read() == gzip.decompress(bytes(b)), (it, kind)
