#!/usr/bin/env python3
"""One very large single-GPU pass (default 200 M reads = BASELINE config 4 unsharded): size-independent
properties only (one kept read per cluster, ids dense, filtered reads untouched, repeatable)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import humid_amd  # noqa: E402
from humid_amd.synth import synth_words  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
    t0 = time.time()
    parts = [synth_words(n // 8, 2000 + i, 24) for i in range(8)]        # 8 independent shards, one read set
    words = np.concatenate([p[0] for p in parts])
    filt = np.concatenate([p[1] for p in parts])
    del parts
    print("generated %d reads in %.0f s" % (len(words), time.time() - t0), flush=True)
    dev = torch.device("cuda:0")
    d_w = torch.from_numpy(words.view(np.int64)).to(dev)
    d_f = torch.from_numpy(filt).to(dev)
    d_c = torch.zeros(len(words), dtype=torch.int32, device=dev)
    d_k = torch.zeros(len(words), dtype=torch.uint8, device=dev)
    dd = humid_amd.Dedup(device=0)
    res = []
    for rep in range(3):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        s = dd.run_device(d_w.data_ptr(), d_f.data_ptr(), d_c.data_ptr(), d_k.data_ptr(), len(words), 24, 1, 0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        res.append((int(d_c.to(torch.int64).sum().item()), int(d_k.sum().item())))
        print("pass %d: %.1f ms = %.2f G reads/s; unique %d edges %d clusters %d mode %d; GPU mem %.1f GB"
              % (rep, 1e3 * dt, len(words) / dt / 1e9, s["unique"], s["edges"], s["clusters"],
                 s["count_mode_used"], torch.cuda.mem_get_info()[1] / 1e9 - torch.cuda.mem_get_info()[0] / 1e9), flush=True)
    cid = d_c.cpu().numpy().view(np.uint32)
    keep = d_k.cpu().numpy()
    ok = (int(keep.sum()) == s["clusters"] == int(cid.max()) and bool(np.array_equal(cid == 0, filt == 1))
          and not bool(keep[filt == 1].any()) and len(set(res)) == 1)
    kept = cid[keep == 1]
    ok = ok and len(np.unique(kept)) == len(kept)
    print("properties ok: %s" % ok, flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
