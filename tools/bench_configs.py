#!/usr/bin/env python3
"""Throughput of the single-GPU hot path on the BASELINE.json config shapes (words resident in
HBM), with the size-independent sanity properties checked on each.  Not the bench.py contract --
a table for BASELINE.md / DESIGN.md."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import humid_amd  # noqa: E402
from humid_amd.synth import synth_wide_words, synth_words  # noqa: E402

CONFIGS = [
    # name, reads, word_nt, distance, mode
    ("C1 100k SE UMI8 d1", 100_000, 24, 1, "umi"),
    ("C2/metric 10M UMI8 d1", 10_000_000, 24, 1, "umi"),
    ("C3 50M PE+UMI file d1", 50_000_000, 24, 1, "umi"),
    ("C4 shard 25M UMI8 d1", 25_000_000, 24, 1, "umi"),
    ("C5 50M PE no-UMI d2", 50_000_000, 24, 2, "genome"),
    ("C3b 50M PE+UMI file -n 36 (12+12+12: the full UMI) d1", 50_000_000, 36, 1, "wide"),   # the full-UMI variant of config 3
    ("W6 10M wide 48 nt d1", 10_000_000, 48, 1, "wide"),      # two uint64 per word (sorted count stage)
    ("W7 10M wide 64 nt d2", 10_000_000, 64, 2, "wide"),
    ("W8 50M wide 48 nt d1", 50_000_000, 48, 1, "wide"),      # 2^18 buckets: the largest LDS-table case of two-word words
    ("W9 100M wide 48 nt d1", 100_000_000, 48, 1, "wide"),    # beyond it: counted by sorting
]


def main():
    only = sys.argv[1:] or None
    dev = torch.device("cuda:0")
    dd = humid_amd.Dedup(device=0)
    for ci, (name, n, nt, d, mode) in enumerate(CONFIGS):
        if only and str(ci + 1) not in only:
            continue
        t0 = time.time()
        if mode == "wide":
            words, filt = synth_wide_words(n, 1001 + ci, nt)
        else:
            words, filt = synth_words(n, 1001 + ci, nt, mode=mode)
        tg = time.time() - t0
        d_w = torch.from_numpy(words.view(np.int64)).to(dev)
        d_f = torch.from_numpy(filt).to(dev)
        d_c = torch.zeros(n, dtype=torch.int32, device=dev)
        d_k = torch.zeros(n, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        for _ in range(2):
            s = dd.run_device(d_w.data_ptr(), d_f.data_ptr(), d_c.data_ptr(), d_k.data_ptr(), n, nt, d, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            s = dd.run_device(d_w.data_ptr(), d_f.data_ptr(), d_c.data_ptr(), d_k.data_ptr(), n, nt, d, 0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        cid = d_c.cpu().numpy().view(np.uint32)
        keep = d_k.cpu().numpy()
        ok = (int(keep.sum()) == s["clusters"] == int(cid.max()) and
              bool(np.array_equal(cid == 0, filt == 1)) and not bool(keep[filt == 1].any()))
        print(json.dumps({"config": name, "reads": n, "word_nt": nt, "distance": d, "mode": mode,
                          "ms_per_pass": round(1e3 * dt, 3), "reads_per_s": round(n / dt, 1),
                          "unique": s["unique"], "edges": s["edges"], "clusters": s["clusters"],
                          "count_mode_used": s["count_mode_used"], "properties_ok": ok,
                          "ms": {k: round(v, 3) for k, v in s.items() if k.startswith("ms_")},
                          "gen_s": round(tg, 1)}), flush=True)
        del d_w, d_f, d_c, d_k
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
