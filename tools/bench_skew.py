"""Skewed buckets: a 10M-read run in which ONE pigeonhole bucket holds `--big`
distinct words (they share their first 12 nucleotides), d = 1, timed with the tile kernel
(bucket_walk 1024, the default) and -- at sizes where that still ends -- with the unbounded
thread-per-position walk (bucket_walk 0).  The result is checked against the oracle.

  python tools/bench_skew.py [--reads 10000000] [--big 1000000] [--old-max 200000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import humid_amd                      # noqa: E402
from humid_amd.synth import synth_words   # noqa: E402


def skew_words(n_reads, big, seed):
    rng = np.random.default_rng(seed)
    tails = rng.choice(1 << 24, size=big, replace=False).astype(np.uint64)
    bigw = (np.uint64(0x9c3a71) << np.uint64(24)) | tails
    words, filt = synth_words(n_reads - big, seed, 24, p_sub=1e-3, p_n=1e-4)
    w = np.concatenate([words, bigw])
    f = np.concatenate([filt, np.zeros(big, np.uint8)])
    p = rng.permutation(len(w))
    return w[p], f[p]


def timed(dd, w, f, reps):
    best = None
    for _ in range(reps):
        t = time.perf_counter()
        cid, keep, s = dd.run(w, f, word_nt=24, distance=1, method=0)
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    return best, cid, keep, s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--big", type=int, default=1_000_000)
    ap.add_argument("--old-max", type=int, default=200_000, help="largest bucket the unbounded walk is timed on")
    ap.add_argument("--no-oracle", action="store_true")
    a = ap.parse_args()
    dd = humid_amd.Dedup()
    out = []
    for big in sorted({50_000, 200_000, a.big}):
        w, f = skew_words(a.reads, big, 7)
        dd.set_option("bucket_walk", 1024)
        dd.run(w, f, word_nt=24, distance=1, method=0)               # warm-up (allocations)
        t_new, cid, keep, s = timed(dd, w, f, 3)
        row = {"reads": a.reads, "bucket_words": big, "tiles_s": round(t_new, 4),
               "device_ms": {k: round(v, 3) for k, v in s.items() if k.startswith("ms_")},
               "unique": s["unique"], "edges": s["edges"]}
        if big <= a.old_max:
            dd.set_option("bucket_walk", 0)
            t_old, cid2, keep2, s2 = timed(dd, w, f, 1)
            row["thread_per_position_s"] = round(t_old, 4)
            row["same_result"] = bool(np.array_equal(cid, cid2) and np.array_equal(keep, keep2))
        if not a.no_oracle and big == a.big:
            from oracle import pyoracle as orc
            t = time.perf_counter()
            ocid, okeep, os_, _ = orc.dedup_run(w, f, 24, 1, 0)
            row["oracle_s"] = round(time.perf_counter() - t, 2)
            row["verified_vs_oracle"] = bool(np.array_equal(cid, ocid) and np.array_equal(keep, okeep)
                                             and os_["edges"] == s["edges"])
        print(json.dumps(row), flush=True)
        out.append(row)
    dd.close()


if __name__ == "__main__":
    main()
