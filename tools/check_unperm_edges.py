#!/usr/bin/env python3
"""The record path at the read counts where the tiled un-permute changes its form (window of 2^14 / 2^15 reads, bin
tables of 1024 / 1536 / 2048 bins; humid_amd/csrc/pipeline.hip.h unpermute_tiled) and where the record path ends:
prefixes of ONE synthetic read set (67 M + 1 reads, 24 nt), each through the record path and through round 2's
12-byte kernels (option records8 = 0, which un-permute with k_unperm_bins instead): cluster ids and keep flags bit
for bit (compared on the device), the summary counts."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import humid_amd  # noqa: E402
from humid_amd.synth import synth_words  # noqa: E402

W14, W15 = 1 << 14, 1 << 15
SIZES = [1024 * W14 - 1, 1024 * W14, 1024 * W14 + 1, 1536 * W14, 1536 * W14 + 1, 2048 * W14 - 1, 2048 * W14, 2048 * W14 + 1,
         1536 * W15, 1536 * W15 + 1, 2048 * W15 - 1, 2048 * W15, 2048 * W15 + 1]


def main():
    n_max = max(SIZES)
    words, filt = synth_words(n_max, 1077, 24)
    print("synthetic reads: %d" % n_max, flush=True)
    dev = torch.device("cuda:0")
    d_w = torch.from_numpy(words.view(np.int64)).to(dev)
    d_f = torch.from_numpy(filt).to(dev)
    del words, filt
    bad = 0
    for n in SIZES:
        res = {}
        for label, rec in (("records", 1), ("12-byte", 0)):
            dd = humid_amd.Dedup(device=0)
            dd.set_option("records8", rec)
            d_c = torch.full((n + 1,), -7, dtype=torch.int32, device=dev)
            d_k = torch.full((n + 1,), 9, dtype=torch.uint8, device=dev)
            s = dd.run_device(d_w.data_ptr(), d_f.data_ptr(), d_c.data_ptr(), d_k.data_ptr(), n, 24, 1, 0)
            s = dd.run_device(d_w.data_ptr(), d_f.data_ptr(), d_c.data_ptr(), d_k.data_ptr(), n, 24, 1, 0)   # (again: reused buffers)
            torch.cuda.synchronize()
            res[label] = (d_c, d_k, {k: s[k] for k in ("usable", "unique", "clusters", "edges")}, bool(s["records8"]))
            dd.close()
        a, b = res["records"], res["12-byte"]
        same = bool(torch.equal(a[0], b[0])) and bool(torch.equal(a[1], b[1])) and a[2] == b[2]
        guard = int(a[0][n]) == -7 and int(a[1][n]) == 9 and int(b[0][n]) == -7 and int(b[1][n]) == 9
        expect_rec = n <= 2048 * W15
        ok = same and guard and a[3] == expect_rec and not b[3]
        bad += not ok
        print("N %9d  bins %4d x 2^%d  record path %-5s  identical %-5s  nothing written behind the outputs %-5s  %s" % (
            n, -(-n // (W14 if n <= 2048 * W14 else W15)), 14 if n <= 2048 * W14 else 15, a[3], same, guard, a[2]), flush=True)
        del res, a, b
    print("all sizes identical: %s" % (bad == 0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
