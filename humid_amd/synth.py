"""Deterministic synthetic inputs of the BASELINE.json shapes.

Model: molecules = ceil(N/4); family size 1 + Geometric(mean 3); every emitted
base substituted with p_sub (1e-3) and called 'N' with p_n (1e-4); records
globally shuffled.  `synth_words` produces the packed words the hot path
consumes (the first `word_nt` bases that src/fastq.cc:116-161 would take);
`synth_fastq` writes small full FastQ files for the CLI end-to-end tests.

No external data: numpy PCG64 seeded with 1000 + config number.
"""
from __future__ import annotations

import os

import numpy as np

__all__ = ["synth_words", "synth_wide_words", "synth_fastq", "fast_fastq", "CONFIG_SEEDS", "pack_bases"]

CONFIG_SEEDS = {1: 1001, 2: 1002, 3: 1003, 4: 1004, 5: 1005}

_COMP = np.array([3, 2, 1, 0], dtype=np.uint8)


def _family_of_read(rng, n_reads):
    """molecule index per read, families 1+Geom(mean 3), truncated so sum == n_reads."""
    m = max(1, -(-n_reads // 4))
    sizes = rng.geometric(0.25, size=m).astype(np.int64)
    tot = int(sizes.sum())
    while tot < n_reads:  # top up with more molecules
        extra = rng.geometric(0.25, size=max(16, (n_reads - tot) // 4 + 16)).astype(np.int64)
        sizes = np.concatenate([sizes, extra])
        tot = int(sizes.sum())
    mol = np.repeat(np.arange(len(sizes), dtype=np.int64), sizes)[:n_reads]
    n_mol = int(mol[-1]) + 1 if n_reads else 0
    return mol, n_mol


def pack_bases(bases: np.ndarray) -> np.ndarray:
    """[N, n] codes 0..3 -> uint64 words, first nucleotide most significant."""
    n = bases.shape[1]
    w = np.zeros(bases.shape[0], dtype=np.uint64)
    for i in range(n):
        w = (w << np.uint64(2)) | bases[:, i].astype(np.uint64)
    return w


def synth_words(n_reads: int, seed: int, word_nt: int = 24, p_sub: float = 1e-3,
                p_n: float = 1e-4, mode: str = "umi", genome_bp: int = 4_000_000,
                shuffle: bool = True):
    """Packed words + filtered flags for `n_reads` reads.

    mode "umi": every word base iid uniform per molecule (configs 1-4: UMI + read
    prefixes of random inserts).  mode "genome": no UMI, word = first word_nt/2
    bases of each mate of a fragment drawn from a random `genome_bp` genome
    (config 5: prefixes collide, d=2 satellites fan out).
    """
    if not (1 <= word_nt <= 32):
        raise ValueError("word_nt must be 1..32")
    rng = np.random.Generator(np.random.PCG64(seed))
    mol, n_mol = _family_of_read(rng, n_reads)
    if mode == "umi":
        hi = 1 << (2 * word_nt)
        if word_nt == 32:
            mw = rng.integers(0, 1 << 63, size=n_mol, dtype=np.uint64) * np.uint64(2) + \
                rng.integers(0, 2, size=n_mol, dtype=np.uint64)
        else:
            mw = rng.integers(0, hi, size=n_mol, dtype=np.uint64)
    elif mode == "genome":
        h1 = word_nt // 2
        h2 = word_nt - h1
        genome = rng.integers(0, 4, size=genome_bp, dtype=np.uint8)
        start = rng.integers(0, genome_bp - 600, size=n_mol)
        flen = rng.integers(200, 500, size=n_mol)
        a = np.stack([genome[start + i] for i in range(h1)], axis=1)
        end = start + flen
        b = np.stack([_COMP[genome[end - 1 - i]] for i in range(h2)], axis=1)
        mw = pack_bases(np.concatenate([a, b], axis=1))
    else:
        raise ValueError("mode")
    words = mw[mol]
    total_bases = n_reads * word_nt
    # substitutions
    k = int(rng.binomial(total_bases, p_sub)) if total_bases else 0
    if k:
        idx = rng.integers(0, total_bases, size=k)
        rd = idx // word_nt
        pos = idx % word_nt
        delta = rng.integers(1, 4, size=k).astype(np.uint64)
        shift = (2 * (word_nt - 1 - pos)).astype(np.uint64)
        old = (words[rd] >> shift) & np.uint64(3)
        new = (old + delta) & np.uint64(3)
        np.bitwise_xor.at(words, rd, (old ^ new) << shift)
    filtered = np.zeros(n_reads, dtype=np.uint8)
    k2 = int(rng.binomial(total_bases, p_n)) if total_bases else 0
    if k2:
        idx = rng.integers(0, total_bases, size=k2)
        rd = idx // word_nt
        pos = idx % word_nt
        filtered[rd] = 1
        # the reference pushes the code of 'G' for an unknown base (src/fastq.cc:156)
        shift = (2 * (word_nt - 1 - pos)).astype(np.uint64)
        for r, s in zip(rd.tolist(), shift.tolist()):
            words[r] = (int(words[r]) & ~(3 << s)) | (2 << s)
    if shuffle and n_reads:
        perm = rng.permutation(n_reads)
        words = words[perm]
        filtered = filtered[perm]
    return np.ascontiguousarray(words), np.ascontiguousarray(filtered)


def _rand_bits(rng, bits, size):
    """uniform integers of `bits` (<= 64) bits as uint64"""
    if bits == 0:
        return np.zeros(size, dtype=np.uint64)
    if bits < 64:
        return rng.integers(0, 1 << bits, size=size, dtype=np.uint64)
    return rng.integers(0, 1 << 63, size=size, dtype=np.uint64) * np.uint64(2) + \
        rng.integers(0, 2, size=size, dtype=np.uint64)


def synth_wide_words(n_reads: int, seed: int, word_nt: int = 48, p_sub: float = 1e-3,
                     p_n: float = 1e-4, shuffle: bool = True):
    """`synth_words` mode "umi" for 33 <= word_nt <= 64: u64[N, 2] words ([:, 0] = the first
    word_nt-32 nucleotides, [:, 1] = the last 32) + filtered flags."""
    if not (33 <= word_nt <= 64):
        raise ValueError("word_nt must be 33..64")
    rng = np.random.Generator(np.random.PCG64(seed))
    mol, n_mol = _family_of_read(rng, n_reads)
    nh = word_nt - 32
    words = np.stack([_rand_bits(rng, 2 * nh, n_mol)[mol], _rand_bits(rng, 64, n_mol)[mol]], axis=1) \
        if n_reads else np.zeros((0, 2), dtype=np.uint64)
    total_bases = n_reads * word_nt

    def place(pos):   # nucleotide position -> (column, bit shift)
        col = (pos >= nh).astype(np.int64)
        shift = np.where(pos < nh, 2 * (nh - 1 - pos), 2 * (word_nt - 1 - pos)).astype(np.uint64)
        return col, shift

    k = int(rng.binomial(total_bases, p_sub)) if total_bases else 0
    if k:
        idx = rng.integers(0, total_bases, size=k)
        rd, (col, shift) = idx // word_nt, place(idx % word_nt)
        delta = rng.integers(1, 4, size=k).astype(np.uint64)
        for r, c, sh, dl in zip(rd.tolist(), col.tolist(), shift.tolist(), delta.tolist()):
            old = (int(words[r, c]) >> sh) & 3
            words[r, c] = int(words[r, c]) ^ ((old ^ ((old + dl) & 3)) << sh)
    filtered = np.zeros(n_reads, dtype=np.uint8)
    k2 = int(rng.binomial(total_bases, p_n)) if total_bases else 0
    if k2:
        idx = rng.integers(0, total_bases, size=k2)
        rd, (col, shift) = idx // word_nt, place(idx % word_nt)
        filtered[rd] = 1
        for r, c, sh in zip(rd.tolist(), col.tolist(), shift.tolist()):   # code of 'G', src/fastq.cc:156
            words[r, c] = (int(words[r, c]) & ~(3 << sh)) | (2 << sh)
    if shuffle and n_reads:
        perm = rng.permutation(n_reads)
        words = words[perm]
        filtered = filtered[perm]
    return np.ascontiguousarray(words), np.ascontiguousarray(filtered)


_ALPHA = np.frombuffer(b"ACGTN", dtype=np.uint8)


def synth_fastq(out_dir: str, n_reads: int, seed: int, n_files: int = 1, umi_len: int = 8,
                umi_in_header: bool = True, umi_file: bool = False, read_len: int = 150,
                p_sub: float = 1e-3, p_n: float = 1e-4, short_frac: float = 0.0,
                header_style: str = "_", prefix: str = "syn"):
    """Write small synthetic FastQ files; returns the list of file names.

    n_files counts the read files (1 = SE, 2 = PE); umi_file adds a further file of
    `umi_len`-nt reads (config 3).  header_style "_" -> "@r<idx>_<UMI>",
    ":" -> "@r<idx>:<UMI>" (BCL Convert style).  short_frac: fraction of reads cut
    to < 8 nt (exercises 'N' padding, src/fastq.cc:135-136).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    mol, n_mol = _family_of_read(rng, n_reads)
    os.makedirs(out_dir, exist_ok=True)

    def noisy(clean):  # clean: [n_mol, L] codes -> per read with errors
        x = clean[mol].copy()
        sub = rng.random(x.shape) < p_sub
        x[sub] = (x[sub] + rng.integers(1, 4, size=int(sub.sum()))) & 3
        x[rng.random(x.shape) < p_n] = 4
        return x

    umi = noisy(rng.integers(0, 4, size=(n_mol, umi_len), dtype=np.uint8)) if umi_len else None
    mates = [noisy(rng.integers(0, 4, size=(n_mol, read_len), dtype=np.uint8))
             for _ in range(n_files)]
    perm = rng.permutation(n_reads)
    lens = np.full(n_reads, read_len)
    if short_frac > 0:
        short = rng.random(n_reads) < short_frac
        lens[short] = rng.integers(0, 8, size=int(short.sum()))
    names = []
    qual_full = "I" * read_len
    for f in range(n_files):
        name = os.path.join(out_dir, "%s_R%d.fastq" % (prefix, f + 1))
        names.append(name)
        with open(name, "w") as fh:
            for j, r in enumerate(perm.tolist()):
                hdr = "@r%d" % j
                if f == 0 and umi_in_header and umi is not None:
                    hdr += header_style + _ALPHA[umi[r]].tobytes().decode()
                    hdr += " 1:N:0"
                L = int(lens[j])
                seq = _ALPHA[mates[f][r][:L]].tobytes().decode()
                fh.write("%s\n%s\n+\n%s\n" % (hdr, seq, qual_full[:L]))
    if umi_file and umi is not None:
        name = os.path.join(out_dir, "%s_UMI.fastq" % prefix)
        names.append(name)
        with open(name, "w") as fh:
            for j, r in enumerate(perm.tolist()):
                seq = _ALPHA[umi[r]].tobytes().decode()
                fh.write("@r%d\n%s\n+\n%s\n" % (j, seq, "I" * umi_len))
    return names


def fast_fastq(path: str, n_reads: int, seed: int, read_len: int = 150, umi_len: int = 8, mate: int = 0,
               p_sub: float = 1e-3, chunk: int = 1_000_000):
    """Large FastQ files for end-to-end timing (bench.py's e2e leg, tools/bench_cli.py), written with
    numpy in chunks: fixed-width records `@r<9 digits>[_<UMI>]\n<read>\n+\n<quality>\n`.  The molecule
    of read i (n_reads/4 molecules, uniformly drawn) and its UMI depend on `seed` only, the read
    sequence on (seed, mate): files written with mate = 0, 1 form a pair; the UMI goes into the header
    of mate 0 when umi_len > 0.  Returns the file size."""
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    n_mol = n_reads // 4 + 1
    base = np.random.Generator(np.random.PCG64(seed))
    mol_umi = alpha[base.integers(0, 4, size=(n_mol, max(umi_len, 1)), dtype=np.uint8)]
    mrng = np.random.Generator(np.random.PCG64([seed, 1000 + mate]))
    mol_seq = alpha[mrng.integers(0, 4, size=(n_mol, read_len), dtype=np.uint8)]
    with_umi = umi_len > 0 and mate == 0
    rec_len = 2 + 9 + (1 + umi_len if with_umi else 0) + 1 + read_len + 3 + read_len + 1
    full = np.empty((min(chunk, max(n_reads, 1)), rec_len), dtype=np.uint8)   # one buffer, reused by every chunk
    with open(path, "wb") as fh:
        for c0 in range(0, n_reads, chunk):
            c1 = min(n_reads, c0 + chunk)
            m = c1 - c0
            crng = np.random.Generator(np.random.PCG64([seed, 7, c0]))       # same molecules for both mates
            mol = crng.integers(0, n_mol, size=m)
            erng = np.random.Generator(np.random.PCG64([seed, 9 + mate, c0]))
            seq = mol_seq[mol]
            k = int(erng.binomial(m * read_len, p_sub))            # sparse substitutions
            if k:
                pos = erng.integers(0, m * read_len, size=k)
                seq.reshape(-1)[pos] = alpha[erng.integers(0, 4, size=k, dtype=np.uint8)]
            num = np.arange(c0, c1, dtype=np.int64)
            idx = np.empty((m, 9), dtype=np.uint8)
            for d in range(9):
                idx[:, 8 - d] = (num // 10 ** d) % 10 + 48
            buf = full[:m]
            p = 0
            buf[:, p:p + 2] = np.frombuffer(b"@r", dtype=np.uint8); p += 2
            buf[:, p:p + 9] = idx; p += 9
            if with_umi:
                buf[:, p] = ord("_"); p += 1
                buf[:, p:p + umi_len] = mol_umi[mol][:, :umi_len]; p += umi_len
            buf[:, p] = ord("\n"); p += 1
            buf[:, p:p + read_len] = seq; p += read_len
            buf[:, p:p + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8); p += 3
            buf[:, p:p + read_len] = ord("I"); p += read_len
            buf[:, p] = ord("\n")
            buf.tofile(fh)
    return os.path.getsize(path)
