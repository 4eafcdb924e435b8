#!/usr/bin/env python3
"""-e beyond 5 edits (the whole dynamic programme per candidate, keys of one segment): time per pass at sizes where
the search is still a search and where it approaches all pairs.  24-nt synthetic UMI words."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import humid_amd  # noqa: E402
from humid_amd.synth import synth_words  # noqa: E402

dev = torch.device("cuda:0")
for n_reads in (100_000, 1_000_000):
    w, f = synth_words(n_reads, 1002, 24)
    d_w = torch.from_numpy(w.view(np.int64)).to(dev)
    d_f = torch.from_numpy(f).to(dev)
    d_c = torch.zeros(n_reads, dtype=torch.int32, device=dev)
    d_k = torch.zeros(n_reads, dtype=torch.uint8, device=dev)
    dd = humid_amd.Dedup(device=0)
    dd.set_option("edit_distance", 1)
    for d in (5, 6, 7, 8):
        if n_reads > 100_000 and d > 6:
            continue
        s = dd.run_device(d_w.data_ptr(), d_f.data_ptr(), d_c.data_ptr(), d_k.data_ptr(), n_reads, 24, d, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s = dd.run_device(d_w.data_ptr(), d_f.data_ptr(), d_c.data_ptr(), d_k.data_ptr(), n_reads, 24, d, 0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%d reads (%d unique words) -e -m %d: %.1f ms/pass; edges %d clusters %d" % (
            n_reads, s["unique"], d, 1e3 * dt, s["edges"], s["clusters"]), flush=True)
    dd.close()
