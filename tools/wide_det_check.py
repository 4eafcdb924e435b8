import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import humid_amd
from humid_amd.synth import synth_wide_words
from oracle import pyoracle as orc
w, f = synth_wide_words(6_000_000, 4242, 48)
dd = humid_amd.Dedup()
ref = None
for d in (1, 2):
    outs = []
    for rep in range(6):
        cid, keep, s = dd.run(w, f, word_nt=48, distance=d)
        outs.append((cid.copy(), keep.copy(), s["edges"], s["clusters"]))
    same = all(np.array_equal(outs[0][0], o[0]) and np.array_equal(outs[0][1], o[1]) and outs[0][2:] == o[2:] for o in outs)
    t0 = time.time()
    ocid, okeep, osum, _ = orc.dedup_run(w, f, 48, d, 0)
    print("d=%d: 6 runs identical: %s; vs oracle (%.0f s): cid %s keep %s clusters %s" % (
        d, same, time.time() - t0, np.array_equal(outs[0][0], ocid), np.array_equal(outs[0][1], okeep), outs[0][3] == osum["clusters"]), flush=True)
