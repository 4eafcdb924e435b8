import sys, time
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import humid_amd
from humid_amd.synth import synth_words
dev = torch.device("cuda:0")
for n_reads in (1_000_000, 10_000_000):
    w, f = synth_words(n_reads, 1002, 24)
    d_w = torch.from_numpy(w.view(np.int64)).to(dev); d_f = torch.from_numpy(f).to(dev)
    d_c = torch.zeros(n_reads, dtype=torch.int32, device=dev); d_k = torch.zeros(n_reads, dtype=torch.uint8, device=dev)
    dd = humid_amd.Dedup(device=0)
    for d in (2, 3, 4, 5):
        for edit in (0, 1):
            if d >= 4 and (not edit or n_reads > 1_000_000):
                continue          # two insertion/deletion pairs: timed at 1 M reads only
            dd.set_option("edit_distance", edit)
            for _ in range(2):
                s = dd.run_device(d_w.data_ptr(), d_f.data_ptr(), d_c.data_ptr(), d_k.data_ptr(), n_reads, 24, d, 0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3):
                s = dd.run_device(d_w.data_ptr(), d_f.data_ptr(), d_c.data_ptr(), d_k.data_ptr(), n_reads, 24, d, 0)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
            print("%d reads d=%d %s: %.1f ms/pass = %.2f G reads/s; edges %d clusters %d (neighbours %.1f ms)" % (
                n_reads, d, "edit" if edit else "hamming", 1e3 * dt, n_reads / dt / 1e9, s["edges"], s["clusters"], s["ms_neighbours"]), flush=True)
    dd.close()
