#!/usr/bin/env python3
"""PCIe-inclusive rate of the boundary that hands over HOST buffers (humid_dedup_run):
H2D + device path + D2H, 10 M reads.  Reported in DESIGN.md; never bench.py's `value`."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import humid_amd  # noqa: E402
from humid_amd.synth import synth_words  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
words, filt = synth_words(n, 1002, 24)
dd = humid_amd.Dedup()
for _ in range(3):
    dd.run(words, filt)
t = []
for _ in range(10):
    t0 = time.perf_counter()
    cid, keep, s = dd.run(words, filt)
    t.append(time.perf_counter() - t0)
t = float(np.median(t))
print("humid_dedup_run (pageable host buffers in/out), %d reads: %.3f ms wall per call = %.1f M reads/s"
      % (n, 1e3 * t, n / t / 1e6))
print("  device part %.3f ms, H2D %.3f ms, D2H %.3f ms (hipEvents)" % (s["ms_total"], s["ms_h2d"], s["ms_d2h"]))
