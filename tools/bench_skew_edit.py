#!/usr/bin/env python3
"""Skewed buckets in the edit-distance search and the all-gather mode's share search:
one run of `--big` words that share their LAST 12 nucleotides (-e -m 2: one run of equal keys in the joins of the
last segments), timed with the bounded walk (default: a lane verifies at most 1024 candidates, long runs go in
pieces) and without (bucket_walk 0: a lane walks the whole run).  Results compared between the two.

  python tools/bench_skew_edit.py [--reads 2000000] [--big 30000,100000]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import humid_amd                      # noqa: E402
from humid_amd.synth import synth_words   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=2_000_000)
    ap.add_argument("--big", default="30000,100000")
    a = ap.parse_args()
    dd = humid_amd.Dedup()
    for big in [int(x) for x in a.big.split(",")]:
        rng = np.random.default_rng(big)
        heads = rng.choice(1 << 24, size=big, replace=False).astype(np.uint64)
        bigw = (heads << np.uint64(24)) | np.uint64(0x96a53c)
        words, filt = synth_words(a.reads - big, 7, 24, p_sub=1e-3, p_n=1e-4)
        w = np.concatenate([words, bigw]); f = np.concatenate([filt, np.zeros(big, np.uint8)])
        p = rng.permutation(len(w)); w, f = w[p], f[p]
        res = {}
        for walk in (1024, 0):
            dd.set_option("bucket_walk", walk)
            best = None
            for _ in range(2):
                t = time.perf_counter()
                cid, keep, s = dd.run(w, f, word_nt=24, distance=2, method=0, edit=True)
                dt = time.perf_counter() - t
                best = dt if best is None else min(best, dt)
            res[walk] = (best, cid, keep, s)
            print("-e -m 2, %d reads, one run of %d equal keys, bucket_walk %4d: %.4f s (edges %d, clusters %d)" %
                  (a.reads, big, walk, best, s["edges"], s["clusters"]), flush=True)
        same = np.array_equal(res[1024][1], res[0][1]) and np.array_equal(res[1024][2], res[0][2])
        print("   bounded == unbounded: %s" % same, flush=True)
    dd.set_option("bucket_walk", 1024)


if __name__ == "__main__":
    main()
