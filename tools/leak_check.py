import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import humid_amd
from humid_amd.synth import synth_words, synth_wide_words
w, f = synth_words(300_000, 1, 24, p_sub=5e-3)
ww, wf = synth_wide_words(100_000, 2, 48)
free0 = torch.cuda.mem_get_info()[0]
for it in range(150):
    dd = humid_amd.Dedup()
    dd.run(w, f, word_nt=24, distance=1)
    dd.run(w, f, word_nt=24, distance=2, edit=True)
    dd.run(ww, wf, word_nt=48, distance=1)
    dd.leaves(); dd.adjacency(); dd.clusters(); dd.histograms()
    dd.close()
    if it % 50 == 49:
        print(it + 1, "cycles: device memory in use changed by %.1f MB" % ((free0 - torch.cuda.mem_get_info()[0]) / 1e6), flush=True)
