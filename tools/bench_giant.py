#!/usr/bin/env python3
"""One dense giant component: the saturated 9-nt word space (262 144 words, every word has
27 (d=1) / 351 (d=2) neighbours).  Workgroup-cooperative big-component kernel vs one lane."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import humid_amd

rng = np.random.default_rng(5)
n = 9
w = rng.integers(0, 4 ** n, size=3_000_000, dtype=np.uint64)
w = np.concatenate([w, np.repeat(rng.integers(0, 4 ** n, size=3000, dtype=np.uint64), 300)])
f = np.zeros(len(w), np.uint8)
dd = humid_amd.Dedup()
for d in (1, 2):
    ref = None
    for coop in (1, 0):
        dd.set_option("coop_big", coop)
        dd.run(w, f, word_nt=n, distance=d)
        t0 = time.perf_counter()
        cid, keep, s = dd.run(w, f, word_nt=n, distance=d)
        dt = time.perf_counter() - t0
        sig = (s["clusters"], s["edges"], int(cid.astype(np.uint64).sum()), int(keep.sum()))
        ref = ref or sig
        print("d=%d coop_big=%d: unique %d edges %d clusters %d  ms_cluster %.2f  wall %.1f ms  same=%s"
              % (d, coop, s["unique"], s["edges"], s["clusters"], s["ms_cluster"], 1e3 * dt, sig == ref), flush=True)
