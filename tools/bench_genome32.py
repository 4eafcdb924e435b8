"""timing of the one-GPU pass on read-prefix words without a UMI (synth mode "genome": 32-nt words = 16 + 16
nucleotides of the two mates of fragments of a 4 Mbp genome) -- the later combinations' keys are far from
uniform there, the case the short keys of make_plan(short_later) pay for.  usage: python tools/bench_genome32.py"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import humid_amd  # noqa: E402
from humid_amd.synth import synth_words  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    dd = humid_amd.Dedup(device=0)
    for n, d, reads in ((32, 1, 10_000_000), (32, 2, 10_000_000), (28, 1, 10_000_000)):
        w, f = synth_words(reads, 5, n, mode="genome")
        d_w = torch.from_numpy(w.view(np.int64)).to(dev)
        d_f = torch.from_numpy(f).to(dev)
        d_c = torch.zeros(reads, dtype=torch.int32, device=dev)
        d_k = torch.zeros(reads, dtype=torch.uint8, device=dev)
        for _ in range(2):
            s = dd.run_device(d_w.data_ptr(), d_f.data_ptr(), d_c.data_ptr(), d_k.data_ptr(), reads, n, d, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            s = dd.run_device(d_w.data_ptr(), d_f.data_ptr(), d_c.data_ptr(), d_k.data_ptr(), reads, n, d, 0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(json.dumps({"word_nt": n, "distance": d, "reads": reads, "ms_per_pass": round(1e3 * dt, 3),
                          "unique": s["unique"], "edges": s["edges"], "clusters": s["clusters"],
                          "ms_neighbours": round(s["ms_neighbours"], 3)}), flush=True)


if __name__ == "__main__":
    main()
