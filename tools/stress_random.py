#!/usr/bin/env python3
"""Randomised differential campaign on one GPU: random word length / distance / method / size / skew,
single-GPU path and the multi-GPU exchange orchestration (thread ranks) against the CPU oracle,
bit-exact.  usage: stress_random.py [n_cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401

import humid_amd  # noqa: E402
from humid_amd.synth import synth_wide_words, synth_words  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402
from test_gpu_exchange import run_ranks  # noqa: E402


def make_case(rng):
    wide = rng.random() < float(os.environ.get("STRESS_WIDE", "0.15"))
    n = int(rng.integers(33, 65)) if wide else int(rng.integers(2, 33))
    d = int(rng.choice([0, 1, 1, 1, 2, 2, 3]))
    method = int(rng.random() < 0.3)
    n_reads = int(rng.choice([1, 7, 300, 5000, 70_000, 300_000, 5000, 70_000, 300_000, 1_300_000]))
    kind = rng.choice(["umi", "genome", "dense", "dup"]) if not wide else "umi"
    if n_reads > 300_000 and (d >= 2 and n < 20 or kind == "dense"):   # (dense neighbourhoods at this size: minutes of oracle)
        d = min(d, 1)
        kind = "umi" if kind == "dense" else kind
    p_sub = float(rng.choice([0, 1e-3, 1e-2, 5e-2]))
    seed = int(rng.integers(1, 1 << 30))
    if wide:
        w, f = synth_wide_words(n_reads, seed, n, p_sub=p_sub, p_n=1e-3)
    elif kind == "dense":
        r2 = np.random.default_rng(seed)
        base = r2.integers(0, 1 << min(2 * n, 62), dtype=np.uint64)
        w = np.full(n_reads, base, dtype=np.uint64)
        for _ in range(int(r2.integers(1, 6))):
            sh = np.uint64(2 * int(r2.integers(0, n)))
            w = (w & ~(np.uint64(3) << sh)) | (r2.integers(0, 4, size=n_reads).astype(np.uint64) << sh)
        f = (r2.random(n_reads) < 0.01).astype(np.uint8)
    elif kind == "dup":
        r2 = np.random.default_rng(seed)
        pool = r2.integers(0, 1 << min(2 * n, 62), size=max(1, n_reads // 50), dtype=np.uint64)
        w = pool[r2.integers(0, len(pool), size=n_reads)]
        f = (r2.random(n_reads) < 0.02).astype(np.uint8)
    else:
        w, f = synth_words(n_reads, seed, n, p_sub=p_sub, p_n=1e-3, mode=kind,
                           genome_bp=int(rng.choice([2000, 50_000, 4_000_000])))
    edit = (not wide) and d in (2, 3) and n_reads <= 70_000 and rng.random() < 0.5
    if (not wide) and n_reads <= 5000 and rng.random() < 0.15:   # two insertion/deletion pairs (round 2)
        edit, d = True, int(rng.choice([4, 5, 4, 5, 6, 7, 9]))   # (6 and beyond: the whole dynamic programme, end of round 3)
    if edit and rng.random() < 0.5:                       # families with deletions + insertions
        from test_oracle_vs_bruteforce import indel_words
        w = indel_words(np.random.default_rng(seed), n_reads, n, p_indel=0.4)
        f = (np.random.default_rng(seed + 1).random(n_reads) < 0.01).astype(np.uint8)
        kind = "indel"
    return dict(n=n, d=d, method=method, kind=str(kind), wide=wide, reads=n_reads, p_sub=p_sub, seed=seed,
                edit=bool(edit)), w, f


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    dd = humid_amd.Dedup()
    t0 = time.time()
    bad = 0
    for ci in range(n_cases):
        desc, w, f = make_case(rng)
        n, d, method = desc["n"], desc["d"], desc["method"]
        if desc["kind"] == "dense" and d >= 2 and desc["reads"] > 70_000:
            d = desc["d"] = 1                                     # the single-thread oracle would take minutes
        edit = desc["edit"]
        # round 2: the bounded bucket walk (tiles beyond it) with random bounds, single GPU and exchange
        walk = int(rng.choice([1024, 1024, 1, 6, 80]))
        desc["walk"] = walk
        dd.set_option("bucket_walk", walk)
        # the count variants: automatic, word-ordered LDS buckets forced (small inputs and two-word words
        # included; an overflow falls back), hashed buckets / the sort
        order = int(rng.choice([-1, -1, 1, 1, 0]))
        desc["order"] = order
        dd.set_option("count_order", order)
        ocid, okeep, osum, _ = orc.dedup_run(w, f, n, d, method, edit=edit)
        cid, keep, s = dd.run(w, f, word_nt=n, distance=d, method=method, edit=edit)
        ok = np.array_equal(cid, ocid) and np.array_equal(keep, okeep) and \
            all(s[k] == osum[k] for k in ("usable", "unique", "clusters"))
        ok_x = True
        if desc["reads"] > 1:                                       # (two-word words since round 2; -e and up to 16 ranks: end of round 3)
            P = int(rng.choice([2, 3, 4, 5, 2, 3, 4, 5, 8, 11, 16]))
            out, offs = run_ranks(P, w, f, n, d, method, "exchange", bucket_walk=walk, edit=edit)
            for r in range(P):
                c2, k2, s2, used = out[r]
                ok_x = ok_x and np.array_equal(c2, ocid[offs[r]:offs[r + 1]]) and \
                    np.array_equal(k2, okeep[offs[r]:offs[r + 1]]) and s2["clusters"] == osum["clusters"]
            desc["P"] = P
        if not (ok and ok_x):
            bad += 1
            print("MISMATCH", desc, "single ok" if ok else "single BAD", "exchange ok" if ok_x else "exchange BAD", flush=True)
        if ci % 20 == 19:
            print("%d cases, %d mismatches, %.0f s" % (ci + 1, bad, time.time() - t0), flush=True)
    print("done: %d cases, %d mismatches, %.0f s" % (n_cases, bad, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
