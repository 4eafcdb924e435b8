import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import humid_amd
from humid_amd.synth import synth_words
words, filt = synth_words(10_000_000, 1002, 24)
exp = np.unique(words[filt == 0])
dd = humid_amd.Dedup()
ref = None
for rep in range(10):
    cid, keep, s = dd.run(words, filt)
    lv = dd.leaves()
    ok_words = np.array_equal(lv["word"], exp)
    deg = lv["degree"].copy()
    msg = ""
    if ref is None and s["edges"] == 218883:
        ref = deg
    if ref is not None and s["edges"] != 218883:
        bad = np.nonzero(deg != ref)[0]
        msg = " bad nodes %s" % bad[:12].tolist()
        for b in bad[:12]:
            w = int(exp[b]); msg += "\n   rank %d word %012x deg %d ref %d" % (b, w, deg[b], ref[b])
    print("rep", rep, s["edges"], s["clusters"], "words_sorted_ok", ok_words, msg, flush=True)
