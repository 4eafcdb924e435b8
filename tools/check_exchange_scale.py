#!/usr/bin/env python3
"""Exchange-mode orchestration at scale on ONE GPU: P thread-ranks (tests/fake_dist.py) with the real
HIP stage ops against the single-GPU direct path over the concatenated read set (bit-exact)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import humid_amd  # noqa: E402
from humid_amd.synth import synth_words  # noqa: E402
from test_gpu_exchange import run_ranks  # noqa: E402


def main():
    cases = [(4, 10_000_000, 24, 1, "umi"), (8, 16_000_000, 24, 1, "umi"), (4, 8_000_000, 24, 2, "genome"),
             (3, 6_000_000, 32, 2, "umi")]
    for P, n_reads, n, d, mode in cases:
        words, filt = synth_words(n_reads, 99 + P, n, mode=mode)
        dd = humid_amd.Dedup()
        cid, keep, s = dd.run(words, filt, word_nt=n, distance=d)
        dd.close()
        t0 = time.time()
        out, offs = run_ranks(P, words, filt, n, d, 0, "exchange", passes=2)
        dt = time.time() - t0
        ok = True
        for r in range(P):
            c, k, sr, used = out[r]
            ok = ok and used == "exchange" and np.array_equal(c, cid[offs[r]:offs[r + 1]]) and \
                np.array_equal(k, keep[offs[r]:offs[r + 1]])
            ok = ok and all(sr[x] == s[x] for x in ("total", "usable", "unique", "clusters", "edges"))
        print("P=%d reads=%d n=%d d=%d %s: bit-exact vs single GPU: %s  (unique %d, edges %d, clusters %d; %.1f s)"
              % (P, n_reads, n, d, mode, ok, s["unique"], s["edges"], s["clusters"], dt), flush=True)
        if not ok:
            sys.exit(1)


if __name__ == "__main__":
    main()
