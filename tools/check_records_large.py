#!/usr/bin/env python3
"""The record path at N reads (default 50 M: BASELINE configs 3 / 5, where 24-nt words need the 10-bit first level)
against the 12-byte kernels of round 2 on the SAME reads: cluster ids and keep flags bit for bit, the five summary
counts, and the time of both.  usage: check_records_large.py [N] [word_nt] [distance] [mode]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import humid_amd  # noqa: E402
from humid_amd.synth import synth_words  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
    nt = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    d = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    mode = sys.argv[4] if len(sys.argv) > 4 else "umi"
    words, filt = synth_words(n, 1003, nt, mode=mode)
    dev = torch.device("cuda:0")
    d_w = torch.from_numpy(words.view(np.int64)).to(dev)
    d_f = torch.from_numpy(filt).to(dev)
    res = {}
    for label, rec in (("records", 1), ("12-byte pairs", 0)):
        dd = humid_amd.Dedup(device=0)
        dd.set_option("records8", rec)
        d_c = torch.zeros(n, dtype=torch.int32, device=dev)
        d_k = torch.zeros(n, dtype=torch.uint8, device=dev)
        for _ in range(2):
            s = dd.run_device(d_w.data_ptr(), d_f.data_ptr(), d_c.data_ptr(), d_k.data_ptr(), n, nt, d, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            s = dd.run_device(d_w.data_ptr(), d_f.data_ptr(), d_c.data_ptr(), d_k.data_ptr(), n, nt, d, 0)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        res[label] = (d_c.cpu().numpy().copy(), d_k.cpu().numpy().copy(), {k: s[k] for k in ("usable", "unique", "clusters", "edges")},
                      bool(s["records8"]))
        print("%-14s %.3f ms/pass = %.2f G reads/s; records8 %s; %s" % (label, ms, n / ms / 1e6, s["records8"], res[label][2]), flush=True)
        dd.close()
    a, b = res["records"], res["12-byte pairs"]
    same = np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    print("record path taken: %s; identical cluster ids, keep flags and counts: %s" % (a[3] and not b[3], same))
    sys.exit(0 if (same and a[3] and not b[3]) else 1)


if __name__ == "__main__":
    main()
